cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
bash tools/prof_r03.sh > gpurun_out/prof_r03.log 2>&1; tail -40 gpurun_out/prof_r03.log
timeout 900 python bench.py > gpurun_out/bench_default.json 2> gpurun_out/bench_default.err; tail -c 3000 gpurun_out/bench_default.json; tail -5 gpurun_out/bench_default.err
