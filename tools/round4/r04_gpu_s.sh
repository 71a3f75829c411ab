# GPU box: BASELINE configs[3] WHOLE (2^20 ciphertexts, cfg2 parameters) on ONE GPU -- eight groups of 131,072 -- with rows of the
# timed batch re-computed by the oracle
set -u
cd $GRAFT_REPO_ROOT
O=gpurun_out/r04s; mkdir -p $O
timeout -k 10 500 python bench.py --workload cfg4 --batch 1048576 --steps 1 --warmup 1 --no-secondary-legs > $O/bench_cfg4_2pow20.json.log 2>$O/bench_cfg4_2pow20.err; echo "rc=$?"
tail -c 1500 $O/bench_cfg4_2pow20.json.log; tail -3 $O/bench_cfg4_2pow20.err
