# GPU box, end-of-round evidence, part 2 of 2: rocprofv3 --kernel-trace --stats of the default bench command (and cfg3, and the
# standalone external product), then the PMC passes for the blind-rotation kernel (cfg2 as shipped and in one launch, cfg5, cfg3)
# and the external product.  Everything lands under gpurun_out/final/ (part 1: tools/final_profile.sh).
set -u
cd $GRAFT_REPO_ROOT
O=gpurun_out/final; mkdir -p $O
(cd /tmp && export TMPDIR=/tmp && rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$O/stats -- python3 $GRAFT_REPO_ROOT/bench.py --steps 5 --warmup 2 --no-cpu-baseline > $GRAFT_REPO_ROOT/$O/stats.log 2>&1) && \
find $O/stats -name "*kernel_stats.csv" -exec cp {} $O/kernel_stats_cfg2.csv \; && head -5 $O/kernel_stats_cfg2.csv && \
(cd /tmp && export TMPDIR=/tmp && rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$O/stats_timed -- python3 $GRAFT_REPO_ROOT/bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-secondary-legs > $GRAFT_REPO_ROOT/$O/stats_timed.log 2>&1) && \
find $O/stats_timed -name "*kernel_stats.csv" -exec cp {} $O/kernel_stats_cfg2_timed_only.csv \; && head -3 $O/kernel_stats_cfg2_timed_only.csv && \
find $O/stats_timed -name "*kernel_trace.csv" -exec cp {} $O/kernel_trace_cfg2_timed_only.csv \; && \
(cd /tmp && export TMPDIR=/tmp && rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$O/stats_cfg3 -- python3 $GRAFT_REPO_ROOT/bench.py --workload cfg3 --steps 5 --warmup 2 --no-cpu-baseline --no-secondary-legs > $GRAFT_REPO_ROOT/$O/stats_cfg3.log 2>&1) && \
find $O/stats_cfg3 -name "*kernel_trace.csv" -exec cp {} $O/kernel_trace_cfg3_timed_only.csv \; && \
find $O/stats_cfg3 -name "*kernel_stats.csv" -exec cp {} $O/kernel_stats_cfg3_timed_only.csv \; && \
(cd /tmp && export TMPDIR=/tmp && rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$O/stats_ep -- python3 $GRAFT_REPO_ROOT/bench.py --kernel external_product --no-cpu-baseline > $GRAFT_REPO_ROOT/$O/stats_ep.log 2>&1) && \
find $O/stats_ep -name "*kernel_stats.csv" -exec cp {} $O/kernel_stats_ep.csv \; && head -4 $O/kernel_stats_ep.csv && \
KERNEL_REGEX=blind_rotate bash tools/profile_pmc.sh final/pmc && \
TFHE_BR_SEGMENTS=1 TFHE_BR_STREAMS=1 KERNEL_REGEX=blind_rotate bash tools/profile_pmc.sh final/pmc_one_launch && \
KERNEL_REGEX=blind_rotate BENCH_ARGS="--workload cfg5" bash tools/profile_pmc.sh final/pmc_cfg5 && \
KERNEL_REGEX=blind_rotate BENCH_ARGS="--workload cfg3" bash tools/profile_pmc.sh final/pmc_cfg3 && \
KERNEL_REGEX=external_product BENCH_ARGS="--kernel external_product" bash tools/profile_pmc.sh final/pmc_ep
