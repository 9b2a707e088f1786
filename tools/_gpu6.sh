cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
for rep in 1 2; do
for v in "none 4 4" "mlp 16 16" "mlp 4 4" "mlp 4 16" "mlp 8 4"; do set -- $v
  ACEZ_POSE_TILE=$2 ACEZ_POSE_TILE_FWD=$3 timeout 200 python bench.py --headline-only --steps 200 --warmup 30 --buffer-patches 2000000 --pose-refinement $1 2>&1 | grep metric | python -c "
import sys,json
for l in sys.stdin:
    d=json.loads(l); print('pose=$1 tile=$2 fwd=$3', round(d['ms_per_step']*1e3,1),'us median;', {k: round(v,1) for k,v in d['per_class_us_per_step'].items()})
"
done; done 2>&1 | tee gpurun_out/ab6.log
