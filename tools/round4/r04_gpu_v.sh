# GPU box: s_setprio around the LDS transposes (dev builds), cfg2 literal and aligned
set -u
cd $GRAFT_REPO_ROOT
O=gpurun_out/r04v; mkdir -p $O
for rep in 1 2; do for so in defn_10_1 prio1_10_1 prio3_10_1 prio0a2_10_1; do
TFHE_HIP_LIB=$PWD/build/dev/$so.so DEV_BACKEND=BACKEND_AUTO python tools/dev_bench.py cfg2 2>&1 | grep -v "^key_switch\|amdgpu.ids" | tee -a $O/setprio_ab.txt
TFHE_HIP_LIB=$PWD/build/dev/$so.so DEV_BACKEND=BACKEND_AUTO DEV_ALIGNED=1 python tools/dev_bench.py cfg2 2>&1 | grep -v "^key_switch\|amdgpu.ids" | sed 's/cfg2/cfg2-aligned/' | tee -a $O/setprio_ab.txt
done; done
