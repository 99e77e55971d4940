#!/usr/bin/env python3
"""Condense a `rocprofv3 --kernel-trace --stats --output-format csv` run into the small files kept
under profiles/: the per-kernel stats of this repository's kernels (k_*), and the top rows overall.

usage: tools/summarize_prof.py <dir with *_kernel_stats.csv> <profiles/out_prefix>
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


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2])
