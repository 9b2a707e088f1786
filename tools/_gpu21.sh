cd $GRAFT_REPO_ROOT
timeout 400 python -m pytest tests/test_dp_gpu.py -m gpu -x -q --tb=short --timeout 180 2>&1 | tail -15
