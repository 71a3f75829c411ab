#!/bin/bash
# Collects HBM-traffic and SQ counters for the dominant kernel in separate rocprofv3 --pmc passes
# (FETCH_SIZE and WRITE_SIZE do not fit one pass on gfx950: MI355X_MICROARCH.md, rocprofv3 PMC slots).
# Every pass runs ONE step of the workload (no secondary legs): the matching dispatches of a pass are that step's launches.
# Usage (on the GPU box, from the repo root): tools/profile_pmc.sh <tag> [bench args...]
set -u
TAG=${1:-pmc}; shift || true
OUT=$GRAFT_REPO_ROOT/gpurun_out/$TAG
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
run() { # name, counters...
  local name=$1; shift
  rocprofv3 --pmc "$@" --kernel-include-regex "${KERNEL_REGEX:-blind_rotate|external_product}" --output-format csv \
      -d "$OUT/$name" -- python3 "$GRAFT_REPO_ROOT/bench.py" --steps 1 --warmup 0 --no-cpu-baseline --no-secondary-legs ${BENCH_ARGS:-} \
      > "$OUT/$name.log" 2>&1 || { echo "pass $name failed"; tail -5 "$OUT/$name.log"; return 1; }
  find "$OUT/$name" -name "*counter_collection.csv" -exec cp {} "$OUT/$name.csv" \;
}
run fetch FETCH_SIZE && run write WRITE_SIZE && \
run sq SQ_WAVES SQ_INSTS_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT && \
run sq2 SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_LDS GRBM_GUI_ACTIVE
ls "$OUT"
