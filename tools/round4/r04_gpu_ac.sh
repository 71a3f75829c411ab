# GPU box: priority schedule of the 4-element shapes -- low for the LDS transposes and the multiply-accumulate (combo), also for the operand
# read and the final update (combo2); and the wide team of cfg3 under the same builds (dev builds)
set -u
cd $GRAFT_REPO_ROOT
O=gpurun_out/r04ac; mkdir -p $O
run() { TFHE_HIP_LIB=$PWD/build/dev/$1.so DEV_BACKEND=BACKEND_AUTO python tools/dev_bench.py $2 2>&1 | grep -v "^key_switch\|amdgpu.ids" | tee -a $O/priority_schedule_ab.txt; }
for rep in 1 2; do
for so in defq_9_2 combo_9_2 combo2_9_2; do run $so cfg3; done
for so in defq_11_2 combo_11_2 combo2_11_2; do DEV_BATCH=4096 run $so cfg5; done
done
for so in defq_9_2 combo_9_2; do for b in 1 64 256; do DEV_SHAPE=wide DEV_BATCH=$b run $so cfg3; done; done
for so in defq_11_2 combo_11_2; do for b in 1 64 512; do DEV_BATCH=$b run $so cfg5; done; done
