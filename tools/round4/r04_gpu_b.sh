# GPU box, round 4, second call: the wide team with the key ring -- parity subset, batch sweep (team vs wide; cfg2, cfg3, cfg1),
# gate-graph bench (fixed timing), cfg1 wide-at-4096 A/B, BMMP in the complex transform A/B.  Output under gpurun_out/r04b/.
set -u
cd $GRAFT_REPO_ROOT
O=gpurun_out/r04b; mkdir -p $O
timeout -k 10 500 python -m pytest tests/test_gpu_wide.py tests/test_gpu_golden.py tests/test_gpu_pool.py tests/test_gpu_gates.py -m gpu -x -q > $O/gpu_tests_subset.log 2>&1; echo "pytest rc=$?" | tee -a $O/gpu_tests_subset.log; tail -3 $O/gpu_tests_subset.log
timeout -k 10 500 python bench.py --batch-sweep --sweep-shapes team,wide --sweep-workloads cfg2,cfg3,cfg1 --no-cpu-baseline > $O/batch_sweep.json.log 2>$O/batch_sweep.err; echo "sweep rc=$?"
GATE_GRAPH_SHAPE=team timeout -k 10 300 python tools/gate_graph_bench.py > $O/gate_graph_team.txt 2>&1; echo "gate graph team rc=$?"; grep -v amdgpu.ids $O/gate_graph_team.txt
GATE_GRAPH_SHAPE=auto timeout -k 10 300 python tools/gate_graph_bench.py > $O/gate_graph_auto.txt 2>&1; echo "gate graph auto rc=$?"; grep -v amdgpu.ids $O/gate_graph_auto.txt
{ DEV_BACKEND=BACKEND_AUTO DEV_SHAPE=team python tools/dev_bench.py cfg1; \
  DEV_BACKEND=BACKEND_AUTO DEV_SHAPE=wide python tools/dev_bench.py cfg1; \
  TFHE_HIP_LIB=$PWD/build/dev/wide9k1_generic_w3.so DEV_BACKEND=BACKEND_AUTO DEV_SHAPE=wide python tools/dev_bench.py cfg1; \
  TFHE_HIP_LIB=$PWD/build/dev/wide9k1_generic_w4.so DEV_BACKEND=BACKEND_AUTO DEV_SHAPE=wide python tools/dev_bench.py cfg1; } 2>&1 | grep -v "^key_switch\|amdgpu.ids" > $O/cfg1_wide_at_4096.txt; cat $O/cfg1_wide_at_4096.txt
{ TFHE_HIP_LIB=$PWD/build/dev/bmmp_fft_k2.so BMMP_BENCH_BACKENDS=fp64-fft python tools/bmmp_bench.py cfg3; \
  TFHE_HIP_LIB=$PWD/build/dev/bmmp_fft_k1.so BMMP_BENCH_BACKENDS=fp64-fft python tools/bmmp_bench.py cfg1; } 2>&1 | grep -v amdgpu.ids > $O/bmmp_fft_ab.txt; cat $O/bmmp_fft_ab.txt
