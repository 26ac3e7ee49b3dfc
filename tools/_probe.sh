cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof -- python $GRAFT_REPO_ROOT/tools/lsd_bench.py 1024 1 2>&1 | grep -E "LSD\+LBD|top-lines" | cut -c1-120
python - << 'PY'
import csv, glob
f = [x for x in glob.glob('/tmp/prof/**/*.csv', recursive=True) if 'kernel_stats' in x][0]
rows = list(csv.DictReader(open(f)))
for r in rows[:16]:
    if 'improve' in r['Name'] or 'accept' in r['Name'] or 'keylines' in r['Name'] or 'detect' in r['Name']:
        print(r['Name'][:44], r['Calls'], 'avg ms', round(float(r['AverageNs'])/1e6, 3), 'min', round(float(r['MinNs'])/1e6, 3), 'max', round(float(r['MaxNs'])/1e6, 3))
PY
