# GPU box, round 4, call H: micro-optimisations of the cfg2 team kernel, A/B (dev builds, cfg2 only): first limb without
# carry-in extraction, mask word fetched an iteration ahead
set -u
cd $GRAFT_REPO_ROOT
O=gpurun_out/r04h; mkdir -p $O
for rep in 1 2; do for so in baseline first_limb first_limb_prefetch; do
  TFHE_HIP_LIB=$PWD/build/dev/$so.so DEV_BACKEND=BACKEND_AUTO python tools/dev_bench.py cfg2 2>&1 | grep -v "^key_switch\|amdgpu.ids" | tee -a $O/micro_ab.txt
  TFHE_HIP_LIB=$PWD/build/dev/$so.so DEV_BACKEND=BACKEND_AUTO DEV_ALIGNED=1 python tools/dev_bench.py cfg2 2>&1 | grep -v "^key_switch\|amdgpu.ids" | sed 's/cfg2/cfg2-aligned/' | tee -a $O/micro_ab.txt
done; done
