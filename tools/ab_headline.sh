#!/bin/bash
# A/B of two builds of the library on ONE box (the pool's boxes differ by several percent): alternates `bench.py --headline-only` between
# the in-tree library and ACEZ_LIB=$1 (default tools/libacez_base.so), N rounds (default 3); prints ms_per_step and the per-class times.
#   bash tools/ab_headline.sh [other.so] [rounds]
OTHER=${1:-tools/libacez_base.so}
N=${2:-3}
for i in $(seq 1 $N); do
  for which in new base; do
    if [ $which = base ]; then export ACEZ_LIB=$OTHER; else unset ACEZ_LIB; fi
    timeout 120 python bench.py --headline-only --steps 200 --warmup 30 --buffer-patches 2000000 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l)
        print('$which', round(d['ms_per_step'] * 1e3, 2), 'us |', ' '.join('%s %.1f' % (k, v) for k, v in d['per_class_us_per_step'].items() if v > 0))"
  done
done
