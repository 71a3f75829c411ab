# GPU box: s_setprio 0 inside the LDS transposes / 2 elsewhere, all shapes (dev builds)
set -u
cd $GRAFT_REPO_ROOT
O=gpurun_out/r04x; mkdir -p $O
run() { TFHE_HIP_LIB=$PWD/build/dev/$1.so DEV_BACKEND=BACKEND_AUTO python tools/dev_bench.py $2 2>&1 | grep -v "^key_switch\|amdgpu.ids" | tee -a $O/setprio_shapes_ab.txt; }
for rep in 1 2; do
for so in defn_10_1 prio0a2_10_1 src0_10_1; do run $so cfg2; DEV_ALIGNED=1 run $so cfg2; done
for so in defn_9_1 prio0a2_9_1; do run $so cfg1; done
for so in defn_9_2 prio0a2_9_2; do run $so cfg3; done
for so in defn_11_2 prio0a2_11_2; do DEV_BATCH=4096 run $so cfg5; done
done
