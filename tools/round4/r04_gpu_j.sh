# GPU box, round 4, call J: launch plan of the pair kernel at cfg1 (segments x streams), product library
set -u
cd $GRAFT_REPO_ROOT
O=gpurun_out/r04j; mkdir -p $O
for st in 1 2; do for seg in 1 2 4 8 16 32; do
  echo -n "segments=$seg streams=$st  " | tee -a $O/pair_plan.txt
  TFHE_BR_SEGMENTS=$seg TFHE_BR_STREAMS=$st DEV_BACKEND=BACKEND_AUTO python tools/dev_bench.py cfg1 2>&1 | grep -v "^key_switch\|amdgpu.ids" | tee -a $O/pair_plan.txt
done; done
for b in 1536 2048 3072 8192 16384; do echo -n "batch=$b default plan  " | tee -a $O/pair_plan.txt; DEV_BATCH=$b DEV_BACKEND=BACKEND_AUTO python tools/dev_bench.py cfg1 2>&1 | grep -v "^key_switch\|amdgpu.ids" | tee -a $O/pair_plan.txt; done
