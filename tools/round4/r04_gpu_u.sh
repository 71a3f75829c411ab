# GPU box: randomised differential run against the oracle (tools/fuzz_gpu.py)
set -u
cd $GRAFT_REPO_ROOT
O=gpurun_out/r04u; mkdir -p $O
timeout -k 10 800 python tools/fuzz_gpu.py ${FUZZ_SECONDS:-420} ${FUZZ_SEED:-1} > $O/fuzz_seed${FUZZ_SEED:-1}.txt 2>&1; echo "fuzz rc=$?"
grep -c "^ok" $O/fuzz_seed${FUZZ_SEED:-1}.txt; grep "MISMATCH\|^#\|Error\|Traceback" $O/fuzz_seed${FUZZ_SEED:-1}.txt | head -20; tail -3 $O/fuzz_seed${FUZZ_SEED:-1}.txt
