cd $GRAFT_REPO_ROOT
for q in 8 16; do GPU_MAX_HW_QUEUES=$q timeout 600 python bench.py 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read())
print('queues', $q, d['value'], d['ms_per_step'])
"; done
