# GPU box, round 4, call I: the pair kernel (one wave per sample at N = 512, k = 1): whole -m gpu suite, cfg1 bench line, cfg1 sweep
set -u
cd $GRAFT_REPO_ROOT
O=gpurun_out/r04i; mkdir -p $O
timeout -k 10 800 python -m pytest tests -m gpu -x -q > $O/gpu_tests.log 2>&1; echo "pytest rc=$?" | tee -a $O/gpu_tests.log; tail -3 $O/gpu_tests.log
timeout -k 10 200 python bench.py --workload cfg1 --steps 5 --warmup 2 --no-cpu-baseline > $O/bench_cfg1.json.log 2>&1; echo "cfg1 rc=$?"; tail -c 700 $O/bench_cfg1.json.log
timeout -k 10 300 python bench.py --batch-sweep --sweep-shapes auto,team --sweep-workloads cfg1 --no-cpu-baseline > $O/batch_sweep_cfg1.json.log 2>&1; echo "sweep rc=$?"
