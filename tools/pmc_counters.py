#!/usr/bin/env python3
"""Per-kernel averages of arbitrary rocprofv3 counters from one or more `--pmc` passes (each pass its own directory, `--kernel-trace` only).

    python tools/pmc_counters.py <dir> [<dir> ...] > profiles/rNN_pmc_sq_per_launch.csv

Sums a counter over the rows of one dispatch (rocprofv3 writes one row per counter instance / dimension), averages over the dispatches of a kernel without
its first one, and adds two ratios when their inputs are present: wait = SQ_WAIT_INST_ANY / SQ_WAVE_CYCLES (share of its resident time a wavefront waits for
an instruction to complete) and active = SQ_ACTIVE_INST_ANY / SQ_WAVE_CYCLES.

Round 6: the passes' kernel traces (rocprofv3 writes *kernel_trace.csv next to the counters; under --pmc every dispatch runs alone) give each kernel's average duration,
and with it issue_frac = (4 * SQ_INSTS_VALU + SQ_INSTS_SALU) / (duration * 1024 SIMDs * 2.4 GHz): the share of the device's instruction-issue cycles the launch used
(a wave64 VALU instruction occupies its SIMD's 16 lanes for 4 cycles, a scalar one for 1) - the lens that fits a path without a dense contraction and with
L2-resident working sets, where the HBM fraction says little."""
import csv
import glob
import os
import sys
from collections import defaultdict


def main():
    vals = defaultdict(lambda: defaultdict(float))      # (file, dispatch) -> counter -> sum
    name, grid = {}, {}
    for d in sys.argv[1:]:
        for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
            for row in csv.DictReader(open(f)):
                key = (f, int(row["Dispatch_Id"]))
                vals[key][row["Counter_Name"]] += float(row["Counter_Value"])
                name[key] = row["Kernel_Name"].split("(")[0]
                try:
                    grid[key] = int(float(row.get("Grid_Size", 0) or 0))
                except ValueError:
                    grid[key] = 0
    # only the launches of the bench's own batch: the run also launches the extractors on 256-frame chunks (the map stand-ins, track.build_map) - smaller grids
    gmax = defaultdict(int)
    for key, g in grid.items():
        gmax[name[key]] = max(gmax[name[key]], g)
    for key in [k for k in vals if grid.get(k, 0) < gmax[name[k]]]:
        del vals[key]
    dur = defaultdict(list)                              # kernel -> durations (ns) of its dispatches in the passes' kernel traces
    for d in sys.argv[1:]:
        for f in glob.glob(os.path.join(d, "**", "*kernel_trace.csv"), recursive=True):
            rows_t = list(csv.DictReader(open(f)))
            gsz = lambda r: int(float(r.get("Grid_Size_X", r.get("Grid_Size", 0)) or 0)) * max(1, int(float(r.get("Grid_Size_Y", 1) or 1))) * max(1, int(float(r.get("Grid_Size_Z", 1) or 1)))
            gm = defaultdict(int)
            for row in rows_t:
                gm[row["Kernel_Name"].split("(")[0]] = max(gm[row["Kernel_Name"].split("(")[0]], gsz(row))
            for row in rows_t:
                try:
                    k_ = row["Kernel_Name"].split("(")[0]
                    if gsz(row) == gm[k_]:
                        dur[k_].append(float(row["End_Timestamp"]) - float(row["Start_Timestamp"]))
                except (KeyError, ValueError):
                    pass
    per = defaultdict(lambda: defaultdict(list))
    for key in sorted(vals):
        for c, v in vals[key].items():
            per[name[key]][c].append(v)
    counters = sorted({c for k in per for c in per[k]})
    print(",".join(["kernel", "launches_seen"] + counters + ["wait", "active", "avg_duration_us", "issue_frac"]))
    rows = []
    for k, cs in per.items():
        if not k.startswith(("planar::", "void planar::")):
            continue
        avg = {c: (sum(v[1:]) / len(v[1:]) if len(v) > 1 else v[0]) for c, v in cs.items()}
        n = max(len(v) for v in cs.values())
        wc = avg.get("SQ_WAVE_CYCLES", 0.0)
        wait = avg["SQ_WAIT_INST_ANY"] / wc if wc and "SQ_WAIT_INST_ANY" in avg else float("nan")
        act = avg["SQ_ACTIVE_INST_ANY"] / wc if wc and "SQ_ACTIVE_INST_ANY" in avg else float("nan")
        dd = dur.get(k, [])
        dd = dd[1:] if len(dd) > 2 else dd                 # (the first full-batch dispatch of a run includes cold caches)
        d_ns = sum(dd) / len(dd) if dd else float("nan")
        issue = (4.0 * avg.get("SQ_INSTS_VALU", float("nan")) + avg.get("SQ_INSTS_SALU", float("nan"))) / (d_ns * 1e-9 * 1024 * 2.4e9) if dd else float("nan")
        rows.append((avg.get("SQ_WAVE_CYCLES", 0.0), ",".join([('"' + k + '"' if "," in k else k), str(n)] + [f"{avg.get(c, float('nan')):.0f}" for c in counters] + [f"{wait:.3f}", f"{act:.3f}", f"{d_ns / 1e3:.1f}", f"{issue:.4f}"])))
    for _, r in sorted(rows, reverse=True):
        print(r)


if __name__ == "__main__":
    main()
