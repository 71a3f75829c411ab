# GPU box: the whole -m gpu suite, then both parts of the final profile (one call)
set -u
cd $GRAFT_REPO_ROOT
O=gpurun_out/${RUN_TAG:-r04n}; mkdir -p $O
timeout -k 10 700 python -m pytest tests -m gpu -x -q > $O/gpu_tests.log 2>&1; echo "pytest rc=$?" | tee -a $O/gpu_tests.log; tail -3 $O/gpu_tests.log
bash tools/final_profile.sh > $O/final1.log 2>&1; echo "final1 rc=$?"
bash tools/final_profile_2.sh > $O/final2.log 2>&1; echo "final2 rc=$?"
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
