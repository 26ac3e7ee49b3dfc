cd $GRAFT_REPO_ROOT
./tools/micro/isort_gtime 195000 1024 0 | tail -1 | cut -c1-330
./tools/micro/isort_gtime 120000 1024 1 | tail -1 | cut -c1-330
timeout 900 python -m pytest tests/test_lsd_gpu.py tests/test_planepost_gpu.py tests/test_track_gpu.py -m gpu -x -q 2>&1 | tail -2
timeout 600 python bench.py --cpu-seconds 0 --latency-reps 0 --pcie-steps 0 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read())
print(d['value'], d['ms_per_step'], d['roofline']['kernels']['plane_clouds(voxels+items+sort+tail)']['alone_launch_ms'], d['roofline']['per_kernel']['lsd_sort'])
"
