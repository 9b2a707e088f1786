cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_pose_fused_gpu.py -m gpu -x -q 2>&1 | tail -4
timeout 900 python -m pytest tests/test_head_gpu.py -m gpu -x -q -k pose 2>&1 | tail -2
for rep in 1 2; do
for v in "none " "mlp "; do set -- $v
  ACEZ_LIB=$2 timeout 200 python bench.py --headline-only --steps 200 --warmup 30 --buffer-patches 2000000 --pose-refinement $1 2>&1 | grep metric | python -c "
import sys,json
for l in sys.stdin:
    d=json.loads(l); print('pose=$1 lib=$2'[-40:], round(d['ms_per_step']*1e3,1),'us median;', {k: round(v,1) for k,v in d['per_class_us_per_step'].items()})
"
done; done 2>&1 | tee gpurun_out/ab13.log
ACEZ_POSE_TILE=4 rocprofv3 --output-format csv --kernel-trace --stats -d /tmp/try -o trace -- python tools/pose_kernels_timing.py > /dev/null 2>&1
cut -c1-100 $(find /tmp/try -name "*kernel_stats.csv" | head -1) | grep "pose" | tee gpurun_out/pose13.log
