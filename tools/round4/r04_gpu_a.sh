# GPU box, round 4, first call: environment probe, the -m gpu suite, the default bench line (verified bits, aligned roofline,
# all-cores CPU leg), the LDS microbenchmark, the batch sweep and the gate-graph baseline.  Output under gpurun_out/r04a/.
set -u
cd $GRAFT_REPO_ROOT
O=gpurun_out/r04a; mkdir -p $O
{ echo "nproc: $(nproc)"; echo "cpu.max: $(cat /sys/fs/cgroup/cpu.max 2>/dev/null)"; python -c "import os; print('affinity', len(os.sched_getaffinity(0)), 'cpu_count', os.cpu_count())"; \
  python -c "import torch; print('devices', torch.cuda.device_count())"; free -g | head -2; } > $O/env.txt 2>&1
cat $O/env.txt
timeout -k 10 700 python -m pytest tests -m gpu -x -q > $O/gpu_tests.log 2>&1; echo "pytest rc=$?" | tee -a $O/gpu_tests.log; tail -3 $O/gpu_tests.log
timeout -k 10 300 python bench.py > $O/bench_cfg2.json.log 2>$O/bench_cfg2.err; echo "bench rc=$?"; tail -c 1500 $O/bench_cfg2.json.log
timeout -k 10 120 tools/microbench/lds_rates > $O/lds_rates.txt 2>&1; echo "lds rc=$?"; tail -5 $O/lds_rates.txt
timeout -k 10 400 python bench.py --batch-sweep --sweep-shapes team,wide --no-cpu-baseline > $O/batch_sweep.json.log 2>$O/batch_sweep.err; echo "sweep rc=$?"
GATE_GRAPH_SHAPE=team timeout -k 10 300 python tools/gate_graph_bench.py > $O/gate_graph_team.txt 2>&1; echo "gate graph team rc=$?"; cat $O/gate_graph_team.txt
GATE_GRAPH_SHAPE=auto timeout -k 10 300 python tools/gate_graph_bench.py > $O/gate_graph_auto.txt 2>&1; echo "gate graph auto rc=$?"; cat $O/gate_graph_auto.txt
