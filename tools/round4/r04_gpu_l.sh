set -u
cd $GRAFT_REPO_ROOT
O=gpurun_out/r04l; mkdir -p $O
timeout -k 10 900 python -m pytest tests -m gpu -x -q > $O/gpu_tests.log 2>&1; echo "pytest rc=$?" | tee -a $O/gpu_tests.log; tail -3 $O/gpu_tests.log
