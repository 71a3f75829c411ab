# GPU box: the compiler's instruction scheduling strategy (-mllvm -amdgpu-sched-strategy=max-ilp, -amdgpu-schedule-metric-bias=0), dev builds
set -u
cd $GRAFT_REPO_ROOT
O=gpurun_out/r04ak; mkdir -p $O
run() { TFHE_HIP_LIB=$PWD/build/dev/$1.so DEV_BACKEND=BACKEND_AUTO python tools/dev_bench.py $2 2>&1 | grep -v "^key_switch\|amdgpu.ids" | tee -a $O/sched_strategy_ab.txt; }
for rep in 1 2; do
for so in base_10_1 max-ilp_10_1 bias0_10_1; do run $so cfg2; DEV_ALIGNED=1 run $so cfg2; done
for so in base_9_2 max-ilp_9_2 bias0_9_2; do run $so cfg3; done
done
