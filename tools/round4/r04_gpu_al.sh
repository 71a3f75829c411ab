# GPU box: inside cfg2's multiply-accumulate, only the key loads (keylow) / only the spectrum reads (speclow) / both at low issue priority (dev builds)
set -u
cd $GRAFT_REPO_ROOT
O=gpurun_out/r04al; mkdir -p $O
run() { TFHE_HIP_LIB=$PWD/build/dev/$1.so DEV_BACKEND=BACKEND_AUTO python tools/dev_bench.py $2 2>&1 | grep -v "^key_switch\|amdgpu.ids" | tee -a $O/mac_memory_ops_priority_ab.txt; }
for rep in 1 2; do for so in base_10_1 keylow_10_1 speclow_10_1 keyspec_10_1; do run $so cfg2; DEV_ALIGNED=1 run $so cfg2; done; done
