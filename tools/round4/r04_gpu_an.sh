# GPU box: the phase priorities at the shapes outside BASELINE (N = 1024 with k = 2, N = 2048 with k = 1): on (shipped) against off (dev builds)
set -u
cd $GRAFT_REPO_ROOT
O=gpurun_out/r04an; mkdir -p $O
run() { TFHE_HIP_LIB=$PWD/build/dev/$1.so DEV_BACKEND=BACKEND_AUTO python tools/dev_bench.py $2 2>&1 | grep -v "^key_switch\|amdgpu.ids" | tee -a $O/phase_priority_other_shapes_ab.txt; }
for rep in 1 2; do
for so in noprio_10_2 prio_10_2; do run $so k2n1024; done
for so in noprio_11_1 prio_11_1; do run $so k1n2048; done
done
