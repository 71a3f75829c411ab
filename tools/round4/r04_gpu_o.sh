set -u
cd $GRAFT_REPO_ROOT
O=gpurun_out/r04o; mkdir -p $O
TFHE_HIP_LIB=$PWD/build/dev/def_9_2.so DEV_BACKEND=BACKEND_AUTO python tools/dev_bench.py cfg3 2>&1 | grep -v "^key_switch\|amdgpu.ids" | tee -a $O/swap_e8_ab.txt
TFHE_HIP_LIB=$PWD/build/dev/def_11_2.so DEV_BACKEND=BACKEND_AUTO DEV_BATCH=4096 python tools/dev_bench.py cfg5 2>&1 | grep -v "^key_switch\|amdgpu.ids" | tee -a $O/swap_e8_ab.txt
for rep in 1 2; do for so in def_10_1 e8_10_1; do
TFHE_HIP_LIB=$PWD/build/dev/$so.so DEV_BACKEND=BACKEND_AUTO python tools/dev_bench.py cfg2 2>&1 | grep -v "^key_switch\|amdgpu.ids" | tee -a $O/swap_e8_ab.txt
TFHE_HIP_LIB=$PWD/build/dev/$so.so DEV_BACKEND=BACKEND_AUTO DEV_ALIGNED=1 python tools/dev_bench.py cfg2 2>&1 | grep -v "^key_switch\|amdgpu.ids" | sed 's/cfg2/cfg2-aligned/' | tee -a $O/swap_e8_ab.txt
done; done
for rep in 1 2; do for so in def_9_1 e8_9_1; do TFHE_HIP_LIB=$PWD/build/dev/$so.so DEV_BACKEND=BACKEND_AUTO python tools/dev_bench.py cfg1 2>&1 | grep -v "^key_switch\|amdgpu.ids" | tee -a $O/swap_e8_ab.txt; done; done
