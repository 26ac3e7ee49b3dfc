#!/bin/bash
# Runs ON THE GPU BOX: the local-BA tests, the BA bench line (BASELINE configs[4]) and its rocprofv3 kernel summary -> gpurun_out/<tag>_ba_*.
tag=${1:-r06}
cd $GRAFT_REPO_ROOT; O=$GRAFT_REPO_ROOT/gpurun_out; mkdir -p $O
timeout 240 python -c "import torch; print(torch.cuda.is_available())"
timeout 300 python -m pytest tests/test_ba_gpu.py tests/test_ba_multirank_gpu.py -m gpu -q -x > $O/${tag}_ba_tests.log 2>&1; grep -a "passed\|failed\|Error" $O/${tag}_ba_tests.log | tail -5
timeout 200 python bench.py --workload ba --steps 20 --warmup 3 > $O/${tag}_bench_ba.json 2> $O/ba.err
cd /tmp && export TMPDIR=/tmp && rm -rf /tmp/ba_prof
timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/ba_prof -- python $GRAFT_REPO_ROOT/bench.py --workload ba --steps 10 --warmup 2 > /dev/null 2>&1
f=$(ls /tmp/ba_prof/*/*kernel_stats.csv | head -1); cp $f $O/${tag}_ba_kernel_stats.csv
t=$(ls /tmp/ba_prof/*/*kernel_trace.csv | head -1); (head -1 $t; tail -400 $t) > $O/${tag}_ba_kernel_trace_tail.csv
cut -c1-300 $O/${tag}_bench_ba.json; head -16 $O/${tag}_ba_kernel_stats.csv | cut -c1-150
