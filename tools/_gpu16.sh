cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_head_fp16_gpu.py -m gpu -q --tb=short 2>&1 | grep -v "^    \|where\|built-in" | tail -60 > gpurun_out/t1.log; cat gpurun_out/t1.log
timeout 900 python -m pytest tests/test_head_gpu.py -m gpu -x -q --tb=short -k "free_running" 2>&1 | grep -v "where\|built-in" | tail -25 > gpurun_out/t2.log; cat gpurun_out/t2.log
