cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_head_gpu.py tests/test_seq_gpu.py tests/test_pose_fused_gpu.py tests/test_chain_gpu.py tests/test_head_fp16_gpu.py tests/test_dp_gpu.py tests/test_focal_drift.py -m gpu -x -q --tb=short --timeout 300 2>&1 | grep -v "where\|built-in" | tail -12
for rep in 1 2; do
for v in "none" "mlp"; do
  timeout 200 python bench.py --headline-only --steps 200 --warmup 30 --buffer-patches 2000000 --pose-refinement $v 2>&1 | grep metric | python -c "
import sys,json
for l in sys.stdin:
    d=json.loads(l); print('pose=$v', round(d['ms_per_step']*1e3,1),'us median;', {k: round(v,1) for k,v in d['per_class_us_per_step'].items()})
"
done; done 2>&1
