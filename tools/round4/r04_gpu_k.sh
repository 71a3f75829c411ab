# GPU box, round 4, call K: pair kernel with the runtime pair/team choice: parity subset, cfg1 line and sweep
set -u
cd $GRAFT_REPO_ROOT
O=gpurun_out/r04k; mkdir -p $O
timeout -k 10 600 python -m pytest tests/test_gpu_wide.py tests/test_gpu_parity.py tests/test_gpu_golden.py tests/test_gpu_keygen.py -m gpu -x -q > $O/gpu_tests_subset.log 2>&1; echo "pytest rc=$?" | tee -a $O/gpu_tests_subset.log; tail -3 $O/gpu_tests_subset.log
timeout -k 10 200 python bench.py --workload cfg1 --steps 5 --warmup 2 --no-cpu-baseline > $O/bench_cfg1.json.log 2>&1; echo "cfg1 rc=$?"
timeout -k 10 300 python bench.py --batch-sweep --sweep-shapes auto --sweep-workloads cfg1 --sweep-batches 1,64,256,512,1024,1536,2048,4096,16384 --no-cpu-baseline > $O/batch_sweep_cfg1.json.log 2>&1; echo "sweep rc=$?"
TFHE_BR_PAIR_MIN=0 timeout -k 10 300 python bench.py --batch-sweep --sweep-shapes team --sweep-workloads cfg1 --sweep-batches 512,1024,1536,2048,4096 --no-cpu-baseline > $O/batch_sweep_cfg1_pair_always.json.log 2>&1
TFHE_BR_PAIR_MIN=100000000 timeout -k 10 300 python bench.py --batch-sweep --sweep-shapes team --sweep-workloads cfg1 --sweep-batches 512,1024,1536,2048,4096 --no-cpu-baseline > $O/batch_sweep_cfg1_team_always.json.log 2>&1
