#!/usr/bin/env python3
"""Condense a `rocprofv3 --kernel-trace --stats --output-format csv` run into the small files kept
under profiles/: the per-kernel stats of this repository's kernels (k_*), and the top rows overall.

Also: the device-busy time per pass from the kernel trace (*_kernel_trace.csv: one row per dispatch with its start and end
timestamps) -- the UNION of the dispatch intervals, since the calling pipeline runs kernels side by side and the per-kernel totals
count overlapped time twice.  Passes are told apart by the K-BFS tier-1 launch that opens each findSuperBubble (k_bfs_thread); the
last `steps` passes of the run are the timed ones.  Written to <out_prefix>_device_busy.json.

usage: tools/summarize_prof.py <dir with *_kernel_stats.csv> <profiles/out_prefix> [steps]
"""
import csv
import glob
import os
import sys


def main(src, out_prefix):
    files = glob.glob(os.path.join(src, "**", "*_kernel_stats.csv"), recursive=True)
    if not files:
        raise SystemExit("no *_kernel_stats.csv under " + src)
    rows = []
    for f in files:
        with open(f) as fh:
            rows += list(csv.DictReader(fh))
    rows.sort(key=lambda r: -float(r["TotalDurationNs"]))
    # this repository's kernels are all named k_*; library kernels (rocprim / hipcub scans and selects, fills) are not
    ours = [r for r in rows if ("k_" in r["Name"]) and "rocprim" not in r["Name"] and "hipcub" not in r["Name"]]
    cols = ["Name", "Calls", "TotalDurationNs", "AverageNs", "Percentage", "MinNs", "MaxNs", "StdDev"]
    with open(out_prefix + "_kernels.csv", "w", newline="") as fh:
        w = csv.DictWriter(fh, fieldnames=cols)
        w.writeheader()
        for r in ours:
            r = dict(r)
            r["Name"] = r["Name"][:120]
            w.writerow({c: r[c] for c in cols})
    with open(out_prefix + "_top20.csv", "w", newline="") as fh:
        w = csv.DictWriter(fh, fieldnames=cols)
        w.writeheader()
        for r in rows[:20]:
            r = dict(r)
            r["Name"] = r["Name"][:120]
            w.writerow({c: r[c] for c in cols})
    for r in ours:
        print("%-40s calls %5s avg %10.1f us  total %8.2f ms" % (r["Name"][:40], r["Calls"], float(r["AverageNs"]) / 1e3,
                                                                float(r["TotalDurationNs"]) / 1e6))


def union_ms(iv):
    iv.sort()
    busy, lo, hi = 0, None, None
    for a, b in iv:
        if lo is None:
            lo, hi = a, b
        elif a <= hi:
            hi = max(hi, b)
        else:
            busy += hi - lo
            lo, hi = a, b
    if lo is not None:
        busy += hi - lo
    return busy / 1e6


def device_busy(src, out_prefix, steps):
    import json
    files = glob.glob(os.path.join(src, "**", "*_kernel_trace.csv"), recursive=True)
    if not files:
        print("no *_kernel_trace.csv under " + src)
        return
    rows = []
    for f in files:
        with open(f) as fh:
            for r in csv.DictReader(fh):
                rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]))
    rows.sort()
    # one per step: the K-COV-JOIN launch that opens it (bench.py's steps since round 5), else the K-BFS launch that opens findSuperBubble
    starts = [a for a, b, n in rows if "k_cov_join<" in n] or [a for a, b, n in rows if "k_bfs_thread" in n]
    n_find = len([1 for a, b, n in rows if "k_bfs_thread" in n])
    if len(starts) > n_find:   # (the join also runs at load and in the look-up comparison: keep the ones that a findSuperBubble follows)
        finds = sorted(a for a, b, n in rows if "k_bfs_thread" in n)
        keep = []
        for f in finds:
            before = [x for x in starts if x <= f]
            if before and (not keep or before[-1] != keep[-1]):
                keep.append(before[-1])
        starts = keep
    if len(starts) < steps + 1:
        print("fewer passes in the trace than steps: no busy figure")
        return
    # the timed passes are the last `steps` (bench.py runs nothing pass-shaped after them when --no-cpu-baseline, except the K-COV
    # probe comparison, which launches no k_bfs_thread); a pass = from its k_bfs_thread to the next one's
    first = starts[-steps]
    after_last = [a for a, b, n in rows if a >= starts[-1] and ("k_call_format" in n or "k_sb_format" in n)]
    end = max(b for a, b, n in rows if a >= starts[-1] and a <= (max(after_last) if after_last else starts[-1]))
    timed = [(a, b) for a, b, n in rows if a >= first and a <= end]
    ours = [(a, b) for a, b, n in rows if a >= first and a <= end and "k_" in n and "rocprim" not in n and "hipcub" not in n]
    out = {"steps": steps, "dispatches": len(timed), "span_ms_per_step": round((end - first) / 1e6 / steps, 3),
           "device_busy_ms_per_step": round(union_ms(timed) / steps, 3),
           "device_busy_ms_per_step_own_kernels_only": round(union_ms(ours) / steps, 3),
           "sum_of_durations_ms_per_step": round(sum(b - a for a, b in timed) / 1e6 / steps, 3)}
    out["device_busy_frac_of_span"] = round(out["device_busy_ms_per_step"] / out["span_ms_per_step"], 4)
    with open(out_prefix + "_device_busy.json", "w") as fh:
        json.dump(out, fh, indent=1)
    print("device busy:", out)


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2])
    device_busy(sys.argv[1], sys.argv[2], int(sys.argv[3]) if len(sys.argv) > 3 else 5)
