set -u
cd $GRAFT_REPO_ROOT
O=gpurun_out/r04m; mkdir -p $O
for so in swap_9_2; do for shape in auto team wide; do TFHE_HIP_LIB=$PWD/build/dev/$so.so DEV_SHAPE=$shape DEV_BACKEND=BACKEND_AUTO python tools/dev_bench.py cfg3 2>&1 | grep -v "^key_switch\|amdgpu.ids" | tee -a $O/swap_ab2.txt; done; done
TFHE_HIP_LIB=$PWD/build/dev/swap_11_2.so DEV_BACKEND=BACKEND_AUTO DEV_BATCH=4096 python tools/dev_bench.py cfg5 2>&1 | grep -v "^key_switch\|amdgpu.ids" | tee -a $O/swap_ab2.txt
for shape in auto team wide; do TFHE_HIP_LIB=$PWD/build/dev/swap_9_1.so DEV_SHAPE=$shape DEV_BACKEND=BACKEND_AUTO python tools/dev_bench.py cfg1 2>&1 | grep -v "^key_switch\|amdgpu.ids" | tee -a $O/swap_ab2.txt; done
