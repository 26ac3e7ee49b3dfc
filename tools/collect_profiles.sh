#!/bin/bash
# Runs ON THE GPU BOX in THREE short gpurun calls (a call that hangs must not eat the round's GPU budget: every command has its own timeout AND each call a
# small gpurun --timeout):   gpurun --timeout 600 -- 'bash tools/collect_profiles.sh r04 bench'   |   ... r04 pmc   |   ... r04 suite
# bench: the bench line, the BA bench, the rocprofv3 kernel summary of the same command.  pmc: HBM traffic and issue counters (separate passes, --kernel-trace only,
# as gpurun requires).  suite: the GPU tests and smoke().  Outputs in gpurun_out/<tag>_*.
# On a box whose image is still paging in, the first `import torch` alone takes 1-2 minutes: every mode starts with an untimed-in-spirit warm-up import (own timeout), so
# that the per-command timeouts below measure the commands and not the cold start (twice this round a call on a cold box ran into every timeout with no output).
# Give the bench mode `gpurun --timeout 600`.
tag=${1:-r06}; what=${2:-bench}
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; mkdir -p $O
cd $R
timeout 240 python -c "import torch, numpy; print('warm', torch.cuda.is_available())" > $O/${tag}_warmup.log 2>&1
FL="--steps 6 --warmup 3 --cpu-seconds 0 --latency-reps 0 --pcie-steps 0 --sub-steps 0"
# counter passes: the textures are synthesised in-process (--gen-procs 1, 16 rooms, 4-frame loops): rocprofv3 --pmc hangs when the profiled process forks workers (profiles/README.md)
PF="--gen-procs 1 --canvases 16 --loop 4 --batch 1024 --seq-cus 0 --steps 3 --warmup 1 --cpu-seconds 0 --latency-reps 0 --pcie-steps 0 --sub-steps 0"   # B = 1024 as in every earlier round's counter files (bench.py scales traffic to its own batch)
if [ $what = bench ]; then
    timeout 400 python bench.py > $O/${tag}_bench.json 2> $O/${tag}_bench.err          # (configs[1] / [3] / [4] ride in its sub_benchmarks since round 5)
    cd /tmp && export TMPDIR=/tmp
    timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d $O/${tag}_trace -- python $R/bench.py $FL > $O/${tag}_bench_under_rocprof.json 2>/dev/null
    cd $R
    f=$(ls $O/${tag}_trace/*/*kernel_stats.csv | head -1); cp $f $O/${tag}_kernel_stats_all.csv
    # the product's kernels only (the at::native rows of the full file are the input renderer, planarslam_amd/synth_se3.py, before the timed region), percentages of their sum
    python - "$O/${tag}_kernel_stats_all.csv" "$O/${tag}_kernel_stats.csv" << 'PY'
import csv, sys
rows = [r for r in csv.DictReader(open(sys.argv[1])) if "planar::" in r["Name"]]
tot = sum(float(r["TotalDurationNs"]) for r in rows) or 1.0
with open(sys.argv[2], "w", newline="") as f:
    w = csv.DictWriter(f, fieldnames=list(rows[0].keys()), quoting=csv.QUOTE_ALL)
    w.writeheader()
    for r in rows:
        r["Percentage"] = f"{100.0 * float(r['TotalDurationNs']) / tot:.2f}"
        w.writerow(r)
PY
    t=$(ls $O/${tag}_trace/*/*kernel_trace.csv | head -1); (head -1 $t; tail -80 $t) > $O/${tag}_kernel_trace_tail.csv
    rm -rf $O/${tag}_trace
    # the same summary at B = 1024 frames per launch: the batch every earlier round's tables are quoted on
    cd /tmp
    timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d $O/${tag}_trace1k -- python $R/bench.py $FL --batch 1024 > $O/${tag}_bench_b1024_under_rocprof.json 2>/dev/null
    cd $R
    f=$(ls $O/${tag}_trace1k/*/*kernel_stats.csv | head -1); python - "$f" "$O/${tag}_kernel_stats_b1024.csv" << 'PY'
import csv, sys
rows = [r for r in csv.DictReader(open(sys.argv[1])) if "planar::" in r["Name"]]
tot = sum(float(r["TotalDurationNs"]) for r in rows) or 1.0
with open(sys.argv[2], "w", newline="") as f:
    w = csv.DictWriter(f, fieldnames=list(rows[0].keys()), quoting=csv.QUOTE_ALL)
    w.writeheader()
    for r in rows:
        r["Percentage"] = f"{100.0 * float(r['TotalDurationNs']) / tot:.2f}"
        w.writerow(r)
PY
    rm -rf $O/${tag}_trace1k
    cut -c1-200 $O/${tag}_bench.json; head -5 $O/${tag}_kernel_stats.csv | cut -c1-160
elif [ $what = pmc ]; then
    cd /tmp && export TMPDIR=/tmp
    timeout 120 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/${tag}_pmc_fetch -- python $R/bench.py $PF > /dev/null 2>&1
    timeout 120 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/${tag}_pmc_write -- python $R/bench.py $PF > /dev/null 2>&1
    python $R/tools/pmc_summary.py $O/${tag}_pmc_fetch $O/${tag}_pmc_write > $O/${tag}_pmc_fetch_write_kb_per_launch.csv 2>> $O/${tag}_bench.err
    # issue counters: instructions per launch, share of a wavefront's resident time with an instruction in flight / waiting
    timeout 120 rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU --kernel-trace --output-format csv -d $O/${tag}_pmc_sq1 -- python $R/bench.py $PF > /dev/null 2>&1
    timeout 120 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_SALU --kernel-trace --output-format csv -d $O/${tag}_pmc_sq2 -- python $R/bench.py $PF > /dev/null 2>&1
    python $R/tools/pmc_counters.py $O/${tag}_pmc_sq1 $O/${tag}_pmc_sq2 > $O/${tag}_pmc_sq_per_launch.csv 2>> $O/${tag}_bench.err
    rm -rf $O/${tag}_pmc_fetch $O/${tag}_pmc_write $O/${tag}_pmc_sq1 $O/${tag}_pmc_sq2
    head -8 $O/${tag}_pmc_fetch_write_kb_per_launch.csv; head -8 $O/${tag}_pmc_sq_per_launch.csv
else
    timeout 600 python -m pytest tests -m gpu -q > $O/${tag}_gpu_tests.log 2>&1
    timeout 60 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $O/${tag}_smoke.log 2>&1
    grep -a "passed\|failed" $O/${tag}_gpu_tests.log; tail -1 $O/${tag}_smoke.log
fi
