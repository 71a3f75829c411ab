# GPU box: phase priorities in the pair kernel's hand-over / multiply-accumulate, and in the prime-field backends (dev builds)
set -u
cd $GRAFT_REPO_ROOT
O=gpurun_out/r04ae; mkdir -p $O
run() { TFHE_HIP_LIB=$PWD/build/dev/$1.so python tools/dev_bench.py $2 2>&1 | grep -v "^key_switch\|amdgpu.ids" | sed "s/$/ [$DEV_BACKEND]/" | tee -a $O/phase_priority_more_ab.txt; }
for rep in 1 2; do
export DEV_BACKEND=BACKEND_AUTO
for so in phase_9_1 pairmac1_9_1 pairmac2_9_1; do run $so cfg1; done
for be in BACKEND_FP64 BACKEND_GOLDILOCKS; do export DEV_BACKEND=$be; for so in phase_10_1 primet_10_1 primetm_10_1; do run $so cfg2; done; done
for be in BACKEND_FP64_P49 BACKEND_GOLDILOCKS; do export DEV_BACKEND=$be; for so in phase_9_2 primet_9_2 primetm_9_2; do run $so cfg3; done; done
done
