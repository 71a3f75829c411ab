# GPU box: more LDS instruction groups at low issue priority -- the twiddle reads in front of a low pass (twlow), the spectrum's
# publishing stores (publow), both -- cfg2 team kernel (dev builds)
set -u
cd $GRAFT_REPO_ROOT
O=gpurun_out/r04ai; mkdir -p $O
run() { TFHE_HIP_LIB=$PWD/build/dev/$1.so DEV_BACKEND=BACKEND_AUTO python tools/dev_bench.py $2 2>&1 | grep -v "^key_switch\|amdgpu.ids" | tee -a $O/lds_groups_priority_ab.txt; }
for rep in 1 2; do for so in base_10_1 pubhigh_10_1 ownhigh_10_1; do run $so cfg2; DEV_ALIGNED=1 run $so cfg2; done; done
