# GPU box: the cfg2 team kernel at three waves per SIMD (168 VGPRs + 68 spilled, five teams per CU by LDS) against the shipped two
set -u
cd $GRAFT_REPO_ROOT
O=gpurun_out/r04t; mkdir -p $O
for rep in 1 2; do for so in def_10_1 w3_10_1; do
TFHE_HIP_LIB=$PWD/build/dev/$so.so DEV_BACKEND=BACKEND_AUTO python tools/dev_bench.py cfg2 2>&1 | grep -v "^key_switch\|amdgpu.ids" | tee -a $O/cfg2_three_waves_ab.txt
TFHE_HIP_LIB=$PWD/build/dev/$so.so DEV_BACKEND=BACKEND_AUTO DEV_ALIGNED=1 python tools/dev_bench.py cfg2 2>&1 | grep -v "^key_switch\|amdgpu.ids" | sed 's/cfg2/cfg2-aligned/' | tee -a $O/cfg2_three_waves_ab.txt
done; done
