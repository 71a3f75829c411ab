# GPU box, end-of-round evidence, part 1 of 2 (run from the repo root through gpurun): bench lines of every workload and the batch
# sweep.  Part 2 (tools/final_profile_2.sh): rocprofv3 kernel stats and the PMC passes.  Everything lands under gpurun_out/final/;
# tools/collect_profiles.py copies what is to be judged into profiles/.
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
python bench.py --batch-sweep --sweep-shapes auto --sweep-workloads cfg2,cfg3,cfg1 --no-cpu-baseline > $O/batch_sweep_auto.json.log 2>&1
