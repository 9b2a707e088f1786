cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
python -c "import torch; print(torch.cuda.get_device_name(0))" 2>&1 | tail -1
timeout 900 python -m pytest tests/test_seq_gpu.py tests/test_pose_fused_gpu.py -m gpu -x -q 2>&1 | tail -25 > gpurun_out/t1.log; cat gpurun_out/t1.log
timeout 900 python -m pytest tests/test_head_gpu.py tests/test_chain_gpu.py tests/test_dp_gpu.py -m gpu -x -q 2>&1 | tail -15 > gpurun_out/t2.log; cat gpurun_out/t2.log
for rep in 1 2; do
for v in "none 1" "mlp 1" "mlp 0"; do set -- $v
  ACEZ_POSE_FUSED=$2 timeout 200 python bench.py --headline-only --steps 200 --warmup 30 --buffer-patches 2000000 --pose-refinement $1 2>&1 | grep metric | python -c "
import sys,json
for l in sys.stdin:
    d=json.loads(l); print('pose=$1 fused=$2', round(d['ms_per_step']*1e3,1),'us median; windows', [round(x*1e3,1) for x in d['window_ms_per_step']], {k: round(v,1) for k,v in d['per_class_us_per_step'].items()})
"
done; done 2>&1 | tee gpurun_out/ab1.log
