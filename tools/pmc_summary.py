#!/usr/bin/env python3
"""Per-kernel HBM traffic per launch from two rocprofv3 counter passes (`--pmc FETCH_SIZE` and `--pmc WRITE_SIZE`, separate runs).

    python tools/pmc_summary.py <dir of the FETCH_SIZE pass> <dir of the WRITE_SIZE pass> > profiles/rNN_pmc_fetch_write_kb_per_launch.csv

Sums the counter over the rows of one dispatch (rocprofv3 writes one row per counter instance), then averages over the dispatches of a
kernel, skipping each kernel's first dispatch (warm-up).  Values are the raw counter units (KB)."""
import csv
import glob
import os
import sys
from collections import defaultdict


def per_kernel(d, counter):
    files = glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True)
    if not files:
        raise SystemExit(f"no counter_collection.csv under {d}")
    disp = defaultdict(float)
    name, grid = {}, {}
    for f in files:
        for row in csv.DictReader(open(f)):
            if row.get("Counter_Name") != counter:
                continue
            key = (f, row["Dispatch_Id"])
            disp[key] += float(row["Counter_Value"])
            name[key] = row["Kernel_Name"].split("(")[0]
            try:
                grid[key] = int(float(row.get("Grid_Size", 0) or 0))
            except ValueError:
                grid[key] = 0
    # (round 6) only the launches of the bench's own batch: the run also launches the extractors on 256-frame chunks (track.build_map) - smaller grids
    gmax = defaultdict(int)
    for key, g in grid.items():
        gmax[name[key]] = max(gmax[name[key]], g)
    for key in [k for k in disp if grid.get(k, 0) < gmax[name[k]]]:
        del disp[key]
    by = defaultdict(list)
    for key in sorted(disp, key=lambda k: (k[0], int(k[1]))):
        by[name[key]].append(disp[key])
    return {k: (sum(v[1:]) / len(v[1:]) if len(v) > 1 else v[0], len(v)) for k, v in by.items()}


def main():
    fetch = per_kernel(sys.argv[1], "FETCH_SIZE")
    write = per_kernel(sys.argv[2], "WRITE_SIZE")
    print("kernel,launches_seen,FETCH_SIZE_KB_per_launch_raw,WRITE_SIZE_KB_per_launch_raw")
    for k in sorted(fetch, key=lambda k: -fetch[k][0]):
        if not k.startswith(("planar::", "void planar::")):
            continue
        kq = '"' + k + '"' if "," in k else k          # (a template kernel's name may hold a comma: plane_items_kernel<19, 256>)
        print(f"{kq},{fetch[k][1]},{fetch[k][0]:.2f},{write.get(k, (0.0, 0))[0]:.2f}")


if __name__ == "__main__":
    main()
