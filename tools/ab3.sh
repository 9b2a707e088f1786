#!/bin/bash
# A/B/C of library builds on ONE box: alternates `bench.py --headline-only` between the in-tree library ("new") and the ACEZ_LIB builds named
# on the command line, N rounds.   bash tools/ab3.sh "<extra bench args>" rounds lib1.so lib2.so ...
ARGS=$1; N=$2; shift 2
for i in $(seq 1 $N); do
  for which in new "$@"; do
    if [ $which = new ]; then unset ACEZ_LIB; else export ACEZ_LIB=$which; fi
    timeout 120 python bench.py --headline-only --steps 200 --warmup 30 --buffer-patches 2000000 $ARGS 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l)
        print('$which'.ljust(24), round(d['ms_per_step'] * 1e3, 2), 'us |', ' '.join('%s %.1f' % (k, v) for k, v in d['per_class_us_per_step'].items() if v > 0))"
  done
done
