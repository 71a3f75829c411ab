# GPU box: three-bit-window transposes in registers (TFHE_SWAP_E8) again, now that the LDS transposes run at low issue priority (dev builds)
set -u
cd $GRAFT_REPO_ROOT
O=gpurun_out/r04ah; mkdir -p $O
run() { TFHE_HIP_LIB=$PWD/build/dev/$1.so DEV_BACKEND=BACKEND_AUTO python tools/dev_bench.py $2 2>&1 | grep -v "^key_switch\|amdgpu.ids" | tee -a $O/swap_e8_under_priorities_ab.txt; }
for rep in 1 2; do
for so in final_10_1 finale8_10_1; do run $so cfg2; DEV_ALIGNED=1 run $so cfg2; done
for so in final_9_1 finale8_9_1; do run $so cfg1; done
done
