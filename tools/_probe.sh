cd $GRAFT_REPO_ROOT
export PYTHONPATH=$GRAFT_REPO_ROOT
timeout 240 python -c "import torch; print('warm', torch.cuda.is_available())"
P='import sys,json; j=json.loads(sys.stdin.read()); print(j["value"], j["ms_per_step"], j["config"]["stage_ms_per_step"]["wait_lines_planes"], j["roofline"]["kernels"]["plane_clouds(voxels+items+sort+tail)"]["avg_launch_ms"])'
F="--steps 16 --warmup 6 --cpu-seconds 0 --latency-reps 0 --pcie-steps 0"
for d in 3 4 6; do echo depth $d; timeout 300 python bench.py --depth $d $F 2>/dev/null | python -c "$P"; done
