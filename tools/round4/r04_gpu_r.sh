# GPU box: three-bit-window transposes in registers (TFHE_SWAP_E8) under the WIDE team at small batches (dev builds of r04_gpu_o),
# then the pool / C-ABI tests on the relinked library (key loads leave a context keyless until complete)
set -u
cd $GRAFT_REPO_ROOT
O=gpurun_out/r04r; mkdir -p $O
for rep in 1 2; do for so in def_10_1 e8_10_1; do for b in 1 64 256; do
TFHE_HIP_LIB=$PWD/build/dev/$so.so DEV_BACKEND=BACKEND_AUTO DEV_ALIGNED=1 DEV_SHAPE=wide DEV_BATCH=$b python tools/dev_bench.py cfg2 2>&1 | grep -v "^key_switch\|amdgpu.ids" | sed 's/cfg2/cfg2-aligned/' | tee -a $O/swap_e8_wide_ab.txt
done; done; done
for rep in 1 2; do for so in def_9_1 e8_9_1; do for b in 1 64; do
TFHE_HIP_LIB=$PWD/build/dev/$so.so DEV_BACKEND=BACKEND_AUTO DEV_SHAPE=wide DEV_BATCH=$b python tools/dev_bench.py cfg1 2>&1 | grep -v "^key_switch\|amdgpu.ids" | tee -a $O/swap_e8_wide_ab.txt
done; done; done
timeout -k 10 600 python -m pytest tests/test_gpu_pool.py tests/test_gpu_c_abi.py tests/test_gpu_golden.py -x -q 2>&1 | tail -3 | tee $O/pool_tests.txt
