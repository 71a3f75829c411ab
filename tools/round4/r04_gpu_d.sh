# GPU box, round 4, call D: helper waves in the wide team, A/B at batch 1 and 64 (dev builds of one shape each)
set -u
cd $GRAFT_REPO_ROOT
O=gpurun_out/r04d; mkdir -p $O
for so in wide_10_1_nohelpers wide_10_1_helpers; do for b in 1 64 256; do
  TFHE_HIP_LIB=$PWD/build/dev/$so.so DEV_BACKEND=BACKEND_AUTO DEV_SHAPE=wide DEV_BATCH=$b DEV_ALIGNED=1 python tools/dev_bench.py cfg2 2>&1 | grep -v "^key_switch\|amdgpu.ids" | tee -a $O/helpers_ab.txt; done; done
for so in wide_9_2_nohelpers wide_9_2_helpers; do for b in 1 64 256; do
  TFHE_HIP_LIB=$PWD/build/dev/$so.so DEV_BACKEND=BACKEND_AUTO DEV_SHAPE=wide DEV_BATCH=$b python tools/dev_bench.py cfg3 2>&1 | grep -v "^key_switch\|amdgpu.ids" | tee -a $O/helpers_ab.txt; done; done
