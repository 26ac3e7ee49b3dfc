cd $GRAFT_REPO_ROOT
export PYTHONPATH=$GRAFT_REPO_ROOT
timeout 240 python -c "import torch; print('warm', torch.cuda.is_available())"
for d in 2 3 4; do timeout 300 python bench.py --depth $d --steps 12 --warmup 4 --cpu-seconds 0 --latency-reps 0 --pcie-steps 0 2>/dev/null | python -c "import sys,json; j=json.loads(sys.stdin.read()); print('depth', $d, j['value'], j['ms_per_step'])"; done
