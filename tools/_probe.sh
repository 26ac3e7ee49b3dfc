cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_lsd_gpu.py tests/test_track_gpu.py tests/test_adapters_gpu.py -m gpu -x -q 2>&1 | tail -2
cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof -- python $GRAFT_REPO_ROOT/tools/lsd_bench.py 1024 2>&1 | grep -E "LSD\+LBD" | cut -c1-60
python - << 'PY'
import csv, glob
f = [x for x in glob.glob('/tmp/prof/**/*.csv', recursive=True) if 'kernel_stats' in x][0]
rows = list(csv.DictReader(open(f)))
for r in rows[:12]:
    print(r['Name'][:40], r['Calls'], 'avg ms', round(float(r['AverageNs'])/1e6, 3), 'min', round(float(r['MinNs'])/1e6, 3))
PY
cd $GRAFT_REPO_ROOT
timeout 600 python bench.py --cpu-seconds 0 --latency-reps 0 --pcie-steps 0 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read())
print(d['value'], d['ms_per_step'], d['roofline']['per_kernel']['lsd_sort'])
"
