# GPU box, round 4, call E: where does a CMUX of ONE sample spend its time?  SQ counters of the wide team at batch 1 (cfg2, cfg3).
set -u
cd $GRAFT_REPO_ROOT
run() { # tag, bench args
  local OUT=$GRAFT_REPO_ROOT/gpurun_out/r04e/$1; shift
  mkdir -p "$OUT"
  ( cd /tmp && export TMPDIR=/tmp
    for pass in "sq SQ_WAVES SQ_INSTS_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT" \
                "sq2 SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_LDS GRBM_GUI_ACTIVE" \
                "sq3 SQ_WAIT_INST_LDS SQ_INSTS_SMEM SQ_ACTIVE_INST_SCA SQ_INSTS_VALU SQ_INSTS_WAVE32_LDS SQ_WAVE_CYCLES"; do
      set -- $pass; name=$1; shift
      rocprofv3 --pmc "$@" --kernel-include-regex "blind_rotate" --output-format csv -d "$OUT/$name" -- python3 "$GRAFT_REPO_ROOT/bench.py" \
          --steps 1 --warmup 0 --no-cpu-baseline --no-secondary-legs $BARGS > "$OUT/$name.log" 2>&1 || { echo "pass $name failed"; tail -3 "$OUT/$name.log"; }
      find "$OUT/$name" -name "*counter_collection.csv" -exec cp {} "$OUT/$name.csv" \;
    done )
  ls $OUT
}
BARGS="--batch 1" run cfg2_b1
BARGS="--batch 1 --workload cfg3" run cfg3_b1
BARGS="--batch 256" run cfg2_b256
for d in cfg2_b1 cfg3_b1 cfg2_b256; do echo "== $d"; cat gpurun_out/r04e/$d/sq*.csv | awk -F, 'NR==1 || /blind_rotate/ {print $(NF-1), $NF}' | sort | uniq | head -40; done
