cd $GRAFT_REPO_ROOT
( time timeout 900 python bench.py > /tmp/b.json 2>/tmp/b.err ) 2>&1 | tail -3
tail -3 /tmp/b.err | cut -c1-300
python -c "
import json
d=json.loads(open('/tmp/b.json').read().strip().splitlines()[-1])
print(d['value'], d['ms_per_step'], d['config']['input_generation_s'], d['config'].get('avg_keypoints_per_frame'))
print({k:v for k,v in d['config'].items() if k not in ('workload','stage_ms_per_step','not_yet_in_workload')})
print(d['config']['stage_ms_per_step'])
pk=d['roofline'].get('per_kernel',{})
for k,v in pk.items(): print(k, v)
print(d['roofline'].get('plane_sort_stats'))
"
