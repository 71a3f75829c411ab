# GPU box, round 4, third call: merged gate levels (gate tests + gate-graph bench), bench under torch.distributed.run with one
# RCCL rank, two-rank gloo rehearsal of the N > 1 code path (verified / ranks_seen), then the default bench line.
set -u
cd $GRAFT_REPO_ROOT
O=gpurun_out/r04c; mkdir -p $O
timeout -k 10 400 python -m pytest tests/test_gpu_gates.py tests/test_gpu_wide.py -m gpu -x -q > $O/gpu_tests_gates.log 2>&1; echo "pytest rc=$?" | tee -a $O/gpu_tests_gates.log; tail -3 $O/gpu_tests_gates.log
GATE_GRAPH_SHAPE=auto timeout -k 10 300 python tools/gate_graph_bench.py > $O/gate_graph_auto.txt 2>&1; echo "gate graph auto rc=$?"; grep -v amdgpu.ids $O/gate_graph_auto.txt
GATE_GRAPH_SHAPE=team timeout -k 10 300 python tools/gate_graph_bench.py > $O/gate_graph_team.txt 2>&1; echo "gate graph team rc=$?"; grep -v amdgpu.ids $O/gate_graph_team.txt
timeout -k 10 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29511 bench.py --gpus 1 --steps 5 --warmup 2 > $O/bench_cfg2_torchrun1.json.log 2>$O/bench_cfg2_torchrun1.err; echo "torchrun1 rc=$?"; tail -c 600 $O/bench_cfg2_torchrun1.json.log
TFHE_BENCH_BACKEND=gloo timeout -k 10 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29512 bench.py --gpus 2 --steps 2 --warmup 1 --batch 512 > $O/bench_cfg2_2ranks_gloo_rehearsal.json.log 2>$O/bench_2ranks.err; echo "gloo 2 ranks rc=$?"; tail -c 900 $O/bench_cfg2_2ranks_gloo_rehearsal.json.log
