# GPU box: effective clock of the literal / aligned cfg2 launches (cycles over wall time), then the default bench line
set -u
cd $GRAFT_REPO_ROOT
O=gpurun_out/r04q; mkdir -p $O gpurun_out/final
cd /tmp && export TMPDIR=/tmp
rocprofv3 --pmc GRBM_GUI_ACTIVE --kernel-include-regex blind_rotate -d $GRAFT_REPO_ROOT/$O/clk -o run --output-format csv -- python3 $GRAFT_REPO_ROOT/bench.py --steps 1 --warmup 0 --no-cpu-baseline > $GRAFT_REPO_ROOT/$O/clk.log 2>&1; echo "rocprof rc=$?"
cd $GRAFT_REPO_ROOT
f=$(find $O/clk -name "*counter_collection.csv" | head -1)
python tools/effective_clock.py $f 116 | tee $O/effective_clock_cfg2.txt
python bench.py > gpurun_out/final/bench_cfg2.json.log 2>gpurun_out/final/bench_cfg2.err && tail -c 600 gpurun_out/final/bench_cfg2.json.log
