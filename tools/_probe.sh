cd $GRAFT_REPO_ROOT
for d in 4 6; do timeout 600 python bench.py --depth $d 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read())
print('depth', $d, d['value'], d['ms_per_step'])
"; done
