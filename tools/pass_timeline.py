#!/usr/bin/env python3
"""The dispatches of the LAST pass of a `rocprofv3 --kernel-trace --output-format csv` run of bench.py, in start order, with the gaps in which
no kernel of the process was running (host waits, launch latency): where the device idles inside a pass.
usage: tools/pass_timeline.py <dir with *_kernel_trace.csv> [min gap us]"""
import csv
import glob
import os
import sys

src = sys.argv[1]
min_gap = float(sys.argv[2]) if len(sys.argv) > 2 else 20.0
rows = []
for f in glob.glob(os.path.join(src, "**", "*_kernel_trace.csv"), recursive=True):
    with open(f) as fh:
        rows += list(csv.DictReader(fh))
ev = sorted(((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]) for r in rows))
finds = [i for i, e in enumerate(ev) if "k_bfs_thread" in e[2]]
starts = []
for f in finds:   # a step opens with its K-COV-JOIN launch when bench.py repeats the join (round 5), else with K-BFS
    j = [i for i in range(max(0, f - 40), f) if "k_cov_join<" in ev[i][2]]
    starts.append(j[-1] if j else f)
if len(starts) < 2:
    raise SystemExit("fewer than two passes in the trace")
a, b = starts[-2], starts[-1]
seg = ev[a:b]
t0 = seg[0][0]
busy_end = seg[0][0]
idle = 0
print("pass of %d dispatches, %.3f ms from its first kernel to the next pass's" % (len(seg), (ev[b][0] - t0) / 1e6))
for s, e, n in seg:
    if s > busy_end:
        gap = (s - busy_end) / 1e3
        idle += s - busy_end
        if gap >= min_gap:
            print("      -- idle %7.1f us --" % gap)
    short = n.replace("(anonymous namespace)::", "").replace("pf_call::", "").replace("void ", "").split("(")[0].split("<")[0][:60]
    print("%9.3f ms  %8.1f us  %s" % ((s - t0) / 1e6, (e - s) / 1e3, short))
    busy_end = max(busy_end, e)
tail = ev[b][0] - busy_end
print("idle inside the pass: %.3f ms (+ %.3f ms after its last kernel)" % (idle / 1e6, tail / 1e6))
# how many dispatches are on the device at once, in slices of a quarter of a millisecond: time with none, mean number, the longest
STEP = 250_000
span = ev[b][0] - t0
print("\nslice start ms | share of the slice with no dispatch | mean dispatches running | kernels (share of the slice each is running)")
for k in range(0, (span + STEP - 1) // STEP):
    lo, hi = t0 + k * STEP, min(t0 + (k + 1) * STEP, ev[b][0])
    pts, per = [], {}
    for s, e, n in seg:
        a_, b_ = max(s, lo), min(e, hi)
        if b_ > a_:
            pts += [(a_, 1), (b_, -1)]
            short = n.replace("(anonymous namespace)::", "").replace("pf_call::", "").replace("void ", "").split("(")[0].split("<")[0]
            per[short] = per.get(short, 0) + (b_ - a_)
    pts.sort()
    cur, last, none, area = 0, lo, 0, 0
    for t, d in pts:
        if cur == 0: none += t - last
        area += cur * (t - last)
        cur += d
        last = t
    none += hi - last
    top = sorted(per.items(), key=lambda kv: -kv[1])[:4]
    print("%6.2f | %4.2f | %4.1f | %s" % (k * STEP / 1e6, none / (hi - lo), area / (hi - lo), ", ".join("%s %.1f" % (n_[:22], v / (hi - lo)) for n_, v in top)))
