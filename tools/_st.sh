run() { python bench.py "$@" --no-cpu-baseline --no-secondary-legs 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(round(d['value']), d['roofline']['frac'])"; }
mkdir -p gpurun_out/t16
python -m pytest tests -m gpu -x -q > gpurun_out/t16/gpu_tests.log 2>&1; tail -3 gpurun_out/t16/gpu_tests.log
for w in cfg2 cfg3 cfg1 cfg5; do echo "$w: $(run --workload $w --steps 5 --warmup 2)"; done
echo "cfg4: $(run --workload cfg4 --steps 2 --warmup 1)"
echo "nand 65536: $(run --workload cfg3 --gate nand --batch 65536 --steps 2 --warmup 1)"
echo "cfg2 p42 default: $(run --workload cfg2 --backend fp64 --steps 3 --warmup 1)"
echo "cfg2 p42 2 streams 2MiB: $(TFHE_BR_STREAMS=2 run --workload cfg2 --backend fp64 --steps 3 --warmup 1)"
echo "cfg2 p49? cfg3 p49 default: $(run --workload cfg3 --backend fp64-p49 --steps 3 --warmup 1)"
echo "cfg3 p49 2 streams: $(TFHE_BR_STREAMS=2 run --workload cfg3 --backend fp64-p49 --steps 3 --warmup 1)"
echo "cfg2 goldilocks default: $(run --workload cfg2 --backend goldilocks --steps 2 --warmup 1)"
echo "cfg2 goldilocks 2 streams: $(TFHE_BR_STREAMS=2 run --workload cfg2 --backend goldilocks --steps 2 --warmup 1)"
