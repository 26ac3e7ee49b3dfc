cd $GRAFT_REPO_ROOT
export PYTHONPATH=$GRAFT_REPO_ROOT
./tools/micro/heap_time 2000 300 | tail -2
./tools/micro/heap_time 30000 300 | tail -2
./tools/micro/heap_time 100000 300 | tail -2
timeout 240 python -c "import torch; print('warm', torch.cuda.is_available())"
timeout 400 python -m pytest tests/test_lsd_gpu.py tests/test_planepost_gpu.py -q 2>&1 | grep -E "^E|passed|failed" | head -30
timeout 300 python tools/heap_stats.py 256 2>&1 | tail -5 | cut -c1-200
timeout 300 python bench.py --steps 12 --warmup 4 --cpu-seconds 0 --latency-reps 0 --pcie-steps 0 2>/dev/null | python -c "import sys,json; j=json.loads(sys.stdin.read()); print('bench:', j['value'], j['ms_per_step'])"
