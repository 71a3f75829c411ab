# GPU box, end-of-round evidence (run from the repo root through gpurun): bench lines of every workload, rocprofv3
# kernel stats of the default bench command, PMC passes for the blind-rotation kernel (cfg2, cfg5, cfg3) and the standalone
# external product.  Everything lands under gpurun_out/final/; copy what is to be judged into profiles/.
set -u
cd $GRAFT_REPO_ROOT
O=gpurun_out/final; mkdir -p $O
python bench.py > $O/bench_cfg2.json.log 2>$O/bench_cfg2.err && tail -c 400 $O/bench_cfg2.json.log && \
python bench.py --backend fp64 --no-cpu-baseline > $O/bench_cfg2_fp64_p42.json.log 2>&1 && \
python bench.py --workload cfg4 --steps 3 --warmup 1 --no-cpu-baseline > $O/bench_cfg4.json.log 2>&1 && \
python bench.py --workload cfg3 --gate nand --batch 65536 --steps 2 --warmup 1 --no-cpu-baseline > $O/bench_cfg3_nand.json.log 2>&1 && \
python bench.py --workload cfg3 --steps 5 --warmup 2 --no-cpu-baseline > $O/bench_cfg3.json.log 2>&1 && \
python bench.py --workload cfg5 --steps 2 --warmup 1 --no-cpu-baseline > $O/bench_cfg5.json.log 2>&1 && \
python bench.py --workload cfg1 --steps 5 --warmup 2 --no-cpu-baseline > $O/bench_cfg1.json.log 2>&1 && \
python bench.py --kernel external_product --no-cpu-baseline > $O/bench_ep_shared.json.log 2>&1 && \
python bench.py --kernel external_product --batch 65536 --no-cpu-baseline > $O/bench_ep_shared_64k.json.log 2>&1 && \
python bench.py --kernel external_product --ggsw-per-sample --no-cpu-baseline > $O/bench_ep_streamed.json.log 2>&1 && \
python bench.py --pool-devices 0 --steps 5 --warmup 2 > $O/bench_pool_1member.json.log 2>&1 && \
python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29511 bench.py --gpus 1 --steps 5 --warmup 2 --no-cpu-baseline > $O/bench_cfg2_torchrun1.json.log 2>&1 && \
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
