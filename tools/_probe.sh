cd $GRAFT_REPO_ROOT
timeout 300 python bench.py --workload orb --cpu-seconds 0 --latency-reps 0 > gpurun_out/r04_bench_orb.json 2>/dev/null
timeout 300 python bench.py --streams pan --cpu-seconds 0 --latency-reps 0 --pcie-steps 0 > gpurun_out/r04_bench_pan.json 2>/dev/null
cut -c1-220 gpurun_out/r04_bench_orb.json; cut -c1-220 gpurun_out/r04_bench_pan.json
