cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 1200 python -m pytest tests/test_pose_fused_gpu.py -m gpu -x -q 2>&1 | tail -12 > gpurun_out/t1.log; cat gpurun_out/t1.log
timeout 1200 python -m pytest tests/test_head_gpu.py -m gpu -x -q -k "pose" 2>&1 | tail -5 > gpurun_out/t2.log; cat gpurun_out/t2.log
for rep in 1 2; do
for v in "none 4 4" "mlp 4 4" "mlp 8 8" "mlp 4 8"; do set -- $v
  ACEZ_POSE_TILE=$2 ACEZ_POSE_TILE_FWD=$3 timeout 200 python bench.py --headline-only --steps 200 --warmup 30 --buffer-patches 2000000 --pose-refinement $1 2>&1 | grep metric | python -c "
import sys,json
for l in sys.stdin:
    d=json.loads(l); print('pose=$1 tile=$2 fwd=$3', round(d['ms_per_step']*1e3,1),'us median;', {k: round(v,1) for k,v in d['per_class_us_per_step'].items()})
"
done; done 2>&1 | tee gpurun_out/ab7.log
for cfg in "8 32" "4 32"; do set -- $cfg
ACEZ_POSE_TILE=$1 ACEZ_POSE_WB=$2 rocprofv3 --output-format csv --kernel-trace --stats -d /tmp/tr_$1_$2 -o trace -- python tools/pose_kernels_timing.py > /dev/null 2>&1
echo "tile $1 wb $2"; cut -c1-100 $(find /tmp/tr_$1_$2 -name "*kernel_stats.csv" | head -1) | grep "pose" | head -8
done 2>&1 | tee gpurun_out/pose_kernels2.log
