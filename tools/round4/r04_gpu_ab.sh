# GPU box: multiply-accumulate at low priority, with and without low-priority LDS transposes, 4-element shapes (dev builds)
set -u
cd $GRAFT_REPO_ROOT
O=gpurun_out/r04ab; mkdir -p $O
run() { TFHE_HIP_LIB=$PWD/build/dev/$1.so DEV_BACKEND=BACKEND_AUTO python tools/dev_bench.py $2 2>&1 | grep -v "^key_switch\|amdgpu.ids" | tee -a $O/mac_priority_combo_ab.txt; }
for rep in 1 2; do
for so in defq_9_2 maclow_9_2 combo_9_2; do run $so cfg3; done
for so in defq_11_2 maclow_11_2 combo_11_2; do DEV_BATCH=4096 run $so cfg5; done
done
