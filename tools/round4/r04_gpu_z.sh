# GPU box: the default bench line (with the PMC traffic record of the same kernel sources)
set -u
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/final
python bench.py > gpurun_out/final/bench_cfg2.json.log 2>gpurun_out/final/bench_cfg2.err && tail -c 300 gpurun_out/final/bench_cfg2.json.log
