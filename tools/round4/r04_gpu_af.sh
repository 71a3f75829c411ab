# GPU box: the unrolled (BMMP) blind rotation before / after the phase priorities reached the prime fields (dev builds)
set -u
cd $GRAFT_REPO_ROOT
O=gpurun_out/r04af; mkdir -p $O
for so in phase_9_2 policy2_9_2; do echo "== $so cfg3" | tee -a $O/bmmp_priority_ab.txt; TFHE_HIP_LIB=$PWD/build/dev/$so.so python tools/bmmp_bench.py cfg3 2>&1 | grep -v amdgpu.ids | tee -a $O/bmmp_priority_ab.txt; done
for so in phase_9_1 policy2_9_1; do echo "== $so cfg1" | tee -a $O/bmmp_priority_ab.txt; TFHE_HIP_LIB=$PWD/build/dev/$so.so python tools/bmmp_bench.py cfg1 2>&1 | grep -v amdgpu.ids | tee -a $O/bmmp_priority_ab.txt; done
