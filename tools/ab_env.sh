#!/bin/bash
# A/B of one environment variable on ONE box:  bash tools/ab_env.sh VAR VALUE [rounds]   (alternates unset / VAR=VALUE processes)
VAR=$1; VAL=$2; N=${3:-3}
for i in $(seq 1 $N); do
  for which in unset set; do
    if [ $which = set ]; then export $VAR=$VAL; else unset $VAR; fi
    timeout 120 python bench.py --headline-only --steps 200 --warmup 30 --buffer-patches 2000000 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l)
        print('$which', round(d['ms_per_step'] * 1e3, 2), 'us |', ' '.join('%s %.1f' % (k, v) for k, v in d['per_class_us_per_step'].items() if v > 0))"
  done
done
