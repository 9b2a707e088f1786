cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_head_fp16_gpu.py -m gpu -q 2>&1 | tail -150 > gpurun_out/t1.log; cat gpurun_out/t1.log
timeout 900 python -m pytest tests/test_head_gpu.py tests/test_seq_gpu.py tests/test_pose_fused_gpu.py -m gpu -x -q 2>&1 | tail -4 > gpurun_out/t2.log; cat gpurun_out/t2.log
for v in "none bf16" "none fp16" "mlp fp16"; do set -- $v
  ACEZ_DTYPE=$2 timeout 200 python bench.py --headline-only --steps 200 --warmup 30 --buffer-patches 2000000 --pose-refinement $1 2>&1 | grep metric | python -c "
import sys,json
for l in sys.stdin:
    d=json.loads(l); print('pose=$1 dtype=$2', round(d['ms_per_step']*1e3,1),'us median;', {k: round(v,1) for k,v in d['per_class_us_per_step'].items()}, 'loss', d['final_loss'])
"
done 2>&1 | tee gpurun_out/ab15.log
