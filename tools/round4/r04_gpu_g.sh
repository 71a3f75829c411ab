# GPU box, round 4, closing call: the whole -m gpu suite on the final build, then the default bench line (traffic record refreshed)
set -u
cd $GRAFT_REPO_ROOT
O=gpurun_out/r04g; mkdir -p $O
timeout -k 10 800 python -m pytest tests -m gpu -x -q > $O/gpu_tests.log 2>&1; echo "pytest rc=$?" | tee -a $O/gpu_tests.log; tail -3 $O/gpu_tests.log
timeout -k 10 300 python bench.py > $O/bench_cfg2.json.log 2>$O/bench_cfg2.err; echo "bench rc=$?"; tail -c 300 $O/bench_cfg2.json.log
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
