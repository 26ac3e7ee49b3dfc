cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_peac_gpu.py -m gpu -x -q 2>&1 | tail -1
timeout 600 python bench.py --cpu-seconds 0 --latency-reps 0 --pcie-steps 0 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read())
print(d['value'], d['ms_per_step'], d['roofline']['per_kernel']['peac_refine'], d['roofline']['per_kernel']['peac_ahc'])
"
