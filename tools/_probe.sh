cd $GRAFT_REPO_ROOT
export PYTHONPATH=$GRAFT_REPO_ROOT
./tools/micro/heap_time 2000 300 | tail -2
./tools/micro/heap_time 30000 300 | tail -2
./tools/micro/heap_time 100000 300 8192 | tail -2
timeout 240 python -c "import torch; print('warm', torch.cuda.is_available())"
timeout 400 python -m pytest tests/test_planepost_gpu.py -q 2>&1 | grep -E "^E|passed|failed" | head
P='import sys,json; j=json.loads(sys.stdin.read()); print(j["value"], j["ms_per_step"], j["config"]["stage_ms_per_step"]["wait_lines_planes"], j["roofline"]["kernels"]["plane_clouds(voxels+items+sort+tail)"]["avg_launch_ms"])'
timeout 300 python bench.py --steps 16 --warmup 6 --cpu-seconds 0 --latency-reps 0 --pcie-steps 0 2>/dev/null | python -c "$P"
