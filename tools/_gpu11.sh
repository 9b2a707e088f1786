cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 300 python -m pytest tests/test_seq_gpu.py -m gpu -x -q 2>&1 | tail -3
rocprofv3 --output-format csv --kernel-trace --stats -d /tmp/trm -o trace -- python bench.py --headline-only --steps 200 --warmup 30 --buffer-patches 2000000 --pose-refinement mlp > /dev/null 2>&1
cut -c1-120 $(find /tmp/trm -name "*kernel_stats.csv" | head -1) | grep acez | head -9 | tee gpurun_out/prof11.log
# a window of the raw trace: start/end of consecutive kernels (gaps)
python - <<'PY'
import csv, glob
f = glob.glob('/tmp/trm/**/*kernel_trace.csv', recursive=True)[0]
rows = [r for r in csv.DictReader(open(f))]
rows.sort(key=lambda r: int(r['Start_Timestamp']))
acez = [r for r in rows if 'acez' in r['Kernel_Name']]
mid = len(acez) // 2
prev_end = None
for r in acez[mid:mid + 16]:
    s, e = int(r['Start_Timestamp']), int(r['End_Timestamp'])
    print(r['Kernel_Name'][:50].ljust(50), 'dur %.1f us' % ((e - s) / 1e3), 'gap %.1f us' % ((s - prev_end) / 1e3 if prev_end else 0))
    prev_end = e
PY
