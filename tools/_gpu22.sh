cd $GRAFT_REPO_ROOT
timeout 500 python -m pytest tests/test_head_gpu.py -m gpu -x -q --tb=short -k "announced or fused_step" 2>&1 | grep -v "where\|built-in" | tail -12
for rep in 1 2; do
for v in "none $PWD/tools/libacez_r02.so" "none "; do set -- $v
  ACEZ_LIB=$2 timeout 200 python bench.py --headline-only --steps 200 --warmup 30 --buffer-patches 2000000 --pose-refinement $1 2>&1 | grep metric | python -c "
import sys,json
for l in sys.stdin:
    d=json.loads(l); print('pose=$1 lib=$2'[-40:], round(d['ms_per_step']*1e3,1),'us median;', {k: round(v,1) for k,v in d['per_class_us_per_step'].items()})
"
done; done 2>&1
