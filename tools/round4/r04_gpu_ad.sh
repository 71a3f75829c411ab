# GPU box: the N = 512, k = 1 TEAM kernel (batches 257 ... 1,536) under the phase priorities, then the whole -m gpu suite and both
# parts of the final profile on the shipped build
set -u
cd $GRAFT_REPO_ROOT
O=gpurun_out/r04ad; mkdir -p $O
run() { TFHE_HIP_LIB=$PWD/build/dev/$1.so DEV_BACKEND=BACKEND_AUTO python tools/dev_bench.py $2 2>&1 | grep -v "^key_switch\|amdgpu.ids" | tee -a $O/phase_priority_cfg1_team.txt; }
for rep in 1 2; do for so in defn_9_1 phase_9_1; do for b in 512 1024 1536; do DEV_SHAPE=team DEV_BATCH=$b run $so cfg1; done; done; done
timeout -k 10 700 python -m pytest tests -m gpu -x -q > $O/gpu_tests.log 2>&1; echo "pytest rc=$?" | tee -a $O/gpu_tests.log; tail -3 $O/gpu_tests.log
bash tools/final_profile.sh > $O/final1.log 2>&1; echo "final1 rc=$?"
bash tools/final_profile_2.sh > $O/final2.log 2>&1; echo "final2 rc=$?"
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
