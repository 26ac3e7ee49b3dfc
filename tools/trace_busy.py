"""Per-stream busy time and gaps from a rocprofv3 kernel_trace.csv (run on the GPU box right after the trace): for the last `window_ms` of the trace, the union of
kernel intervals per stream, the biggest idle gaps of every stream, and which kernels of OTHER streams were running during those gaps."""
import csv
import sys
from collections import defaultdict

path, window_ms = sys.argv[1], float(sys.argv[2]) if len(sys.argv) > 2 else 350.0
rows = []
for r in csv.DictReader(open(path)):
    rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Stream_Id"], r["Kernel_Name"].split("(")[0].replace("planar::", "").replace("void ", "")[:40]))
t_end = max(r[1] for r in rows)
t0 = t_end - int(window_ms * 1e6)
rows = [r for r in rows if r[1] > t0]
by = defaultdict(list)
for s, e, st, n in rows:
    by[st].append((max(s, t0), e, n))
print("window %.1f ms, %d kernels" % (window_ms, len(rows)))
for st in sorted(by, key=lambda k: -sum(e - s for s, e, _ in by[k])):
    iv = sorted(by[st])
    busy, cur_s, cur_e = 0, None, None
    gaps = []
    for s, e, n in iv:
        if cur_e is None: cur_s, cur_e = s, e
        elif s <= cur_e: cur_e = max(cur_e, e)
        else: gaps.append((s - cur_e, cur_e, s, n)); busy += cur_e - cur_s; cur_s, cur_e = s, e
    busy += cur_e - cur_s
    names = defaultdict(float)
    for s, e, n in iv: names[n] += (e - s) / 1e6
    top = ", ".join("%s %.1f" % (k, v) for k, v in sorted(names.items(), key=lambda kv: -kv[1])[:4])
    print("stream %s: busy %.1f ms (%.0f %%), %d launches | %s" % (st, busy / 1e6, 100 * busy / (window_ms * 1e6), len(iv), top))
    for g, a, b, nxt in sorted(gaps, reverse=True)[:3]:
        if g < 2e6: continue
        during = defaultdict(float)
        for s2, e2, st2, n2 in rows:
            if st2 != st and e2 > a and s2 < b: during[n2] += (min(e2, b) - max(s2, a)) / 1e6
        print("    gap %.1f ms before %s; meanwhile: %s" % (g / 1e6, nxt, ", ".join("%s %.1f" % kv for kv in sorted(during.items(), key=lambda kv: -kv[1])[:5])))
