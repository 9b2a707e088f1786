cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 1200 python -m pytest tests/test_pose_fused_gpu.py tests/test_chain_gpu.py -m gpu -x -q 2>&1 | tail -25 > gpurun_out/t1.log; cat gpurun_out/t1.log
timeout 1200 python -m pytest tests/test_head_gpu.py tests/test_seq_gpu.py -m gpu -x -q -k "pose or seq" 2>&1 | tail -8 > gpurun_out/t2.log; cat gpurun_out/t2.log
for rep in 1 2; do
for v in "none x $PWD/tools/libacez_r02.so" "none 8" "mlp 16" "mlp 8" "mlp 4"; do set -- $v
  ACEZ_POSE_TILE=$2 ACEZ_LIB=$3 timeout 200 python bench.py --headline-only --steps 200 --warmup 30 --buffer-patches 2000000 --pose-refinement $1 2>&1 | grep metric | python -c "
import sys,json
for l in sys.stdin:
    d=json.loads(l); print('pose=$1 tile=$2 lib=$3', round(d['ms_per_step']*1e3,1),'us median;', {k: round(v,1) for k,v in d['per_class_us_per_step'].items()})
"
done; done 2>&1 | tee gpurun_out/ab4.log
for t in 8 4; do
ACEZ_POSE_TILE=$t rocprofv3 --output-format csv --kernel-trace --stats -d /tmp/tr$t -o trace -- python bench.py --headline-only --steps 200 --warmup 30 --buffer-patches 2000000 --pose-refinement mlp > /dev/null 2>&1
echo "tile $t"; cut -c1-110 $(find /tmp/tr$t -name "*kernel_stats.csv" | head -1) | grep acez | head -8
done 2>&1 | tee gpurun_out/prof4.log
