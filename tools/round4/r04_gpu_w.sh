# GPU box: s_setprio variants (dev builds), cfg2 literal and aligned
set -u
cd $GRAFT_REPO_ROOT
O=gpurun_out/r04w; mkdir -p $O
for rep in 1 2; do for so in $PRIO_BUILDS; do
TFHE_HIP_LIB=$PWD/build/dev/$so.so DEV_BACKEND=BACKEND_AUTO python tools/dev_bench.py $PRIO_CFG 2>&1 | grep -v "^key_switch\|amdgpu.ids" | tee -a $O/setprio_ab_$PRIO_TAG.txt
if [ "$PRIO_CFG" = cfg2 ]; then TFHE_HIP_LIB=$PWD/build/dev/$so.so DEV_BACKEND=BACKEND_AUTO DEV_ALIGNED=1 python tools/dev_bench.py cfg2 2>&1 | grep -v "^key_switch\|amdgpu.ids" | sed 's/cfg2/cfg2-aligned/' | tee -a $O/setprio_ab_$PRIO_TAG.txt; fi
done; done
