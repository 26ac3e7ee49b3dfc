cd $GRAFT_REPO_ROOT
export PYTHONPATH=$GRAFT_REPO_ROOT
timeout 240 python -c "import torch; print('warm', torch.cuda.is_available())"
timeout 120 python bench.py --workload pose --steps 50 --warmup 5 > gpurun_out/r04a_bench_pose.json 2> gpurun_out/r04a_pose.err; cut -c1-900 gpurun_out/r04a_bench_pose.json; tail -3 gpurun_out/r04a_pose.err
timeout 120 python bench.py --workload ba --steps 20 --warmup 3 > gpurun_out/r04a_bench_ba.json 2>> gpurun_out/r04a_pose.err; cut -c1-400 gpurun_out/r04a_bench_ba.json
timeout 400 python bench.py --pcie-steps 0 --latency-reps 3 --cpu-seconds 0 > gpurun_out/r04a_bench.json 2> gpurun_out/r04a_bench.err; cut -c1-300 gpurun_out/r04a_bench.json; tail -3 gpurun_out/r04a_bench.err
