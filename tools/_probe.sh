cd $GRAFT_REPO_ROOT
for m in 0 1; do ./tools/micro/isort_time256 5888 300 8192 $m 1 | tail -2 | cut -c1-60; done
timeout 900 python -m pytest tests/test_lsd_gpu.py tests/test_planepost_gpu.py tests/test_track_gpu.py -m gpu -x -q 2>&1 | tail -2
timeout 600 python bench.py --cpu-seconds 0 --latency-reps 0 --pcie-steps 0 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read())
print(d['value'], d['ms_per_step'], d['roofline']['kernels']['plane_clouds(voxels+items+sort+tail)']['alone_launch_ms'], d['roofline']['per_kernel']['lsd_sort'])
"
