#!/bin/bash
# GPU box: time every dev build under build/dev/*.so on one workload (tools/dev_bench.py); usage: tools/dev_ab.sh cfg2 [names...]
W="$1"; shift
mkdir -p gpurun_out
for so in ${@:-$(ls build/dev/*.so)}; do
  [ -f "$so" ] || so="build/dev/$so.so"
  TFHE_HIP_LIB="$PWD/$so" python tools/dev_bench.py "$W" 2>&1 | grep -v "^key_switch" | tee -a gpurun_out/dev_ab.log
done
