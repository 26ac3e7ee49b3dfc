cd $GRAFT_REPO_ROOT
for m in none lsd peac planepost orb; do PLANAR_TRACK_SKIP=$m timeout 600 python bench.py --cpu-seconds 0 --latency-reps 0 --pcie-steps 0 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read())
s=d['config']['stage_ms_per_step']
print('skip $m', d['value'], d['ms_per_step'], 'peac', s.get('peac(stream 2)'), 'lsd', s.get('lsd_lbd(stream 3)'), 'orb', s.get('orb_extract'))
"; done
