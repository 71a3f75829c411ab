# GPU box: key slices per rotation (TFHE_BR_SEGMENTS) around the default of half an L2 per slice, shipped library
set -u
cd $GRAFT_REPO_ROOT
O=gpurun_out/r04aj; mkdir -p $O
run() { DEV_BACKEND=BACKEND_AUTO python tools/dev_bench.py $1 2>&1 | grep -v "^key_switch\|amdgpu.ids" | sed "s/^default/segments=${TFHE_BR_SEGMENTS:-default}/" | tee -a $O/segments_sweep.txt; }
for seg in default 29 40 80 116; do
if [ $seg = default ]; then unset TFHE_BR_SEGMENTS; else export TFHE_BR_SEGMENTS=$seg; fi
run cfg2; DEV_ALIGNED=1 run cfg2
done
for seg in default 72 100 200 180; do
if [ $seg = default ]; then unset TFHE_BR_SEGMENTS; else export TFHE_BR_SEGMENTS=$seg; fi
run cfg3
done
for seg in default 8 12 24 32; do
if [ $seg = default ]; then unset TFHE_BR_SEGMENTS; else export TFHE_BR_SEGMENTS=$seg; fi
run cfg1
done
