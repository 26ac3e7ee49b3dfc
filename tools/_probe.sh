cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_orb_gpu.py tests/test_track_gpu.py -m gpu -x -q 2>&1 | tail -2
timeout 300 python bench.py --workload orb --cpu-seconds 0 --latency-reps 0 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read())
print('orb only', d['value'], d['ms_per_step'], {k:v['ms_per_step'] for k,v in d['roofline']['kernels'].items() if k.startswith('orb')})
"
