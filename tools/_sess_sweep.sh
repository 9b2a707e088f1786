for cfg in "32 16" "64 16" "64 8" "128 8"; do set -- $cfg
  ACEZ_STUDY_CHUNK=$1 ACEZ_AUG_CHUNK=$1 ACEZ_STUDY_LEVELS=$2 python tools/reconstruct_synth.py 1000 144 2>&1 | tail -1 | python -c "
import sys, json
d = json.loads(sys.stdin.readline())
print('chunk $1 levels $2:', 'total %.2f s' % d['reconstruction_s'], 'registered', d['registered'], {k: round(v, 2) for k, v in d['timings'].items()}, 'rounds', len(d['rounds']))"
done
