#!/usr/bin/env python3
"""Per-kernel averages of SQ counters from `rocprofv3 --kernel-trace --pmc <SQ_*...>` runs (one or more output dirs):
where a kernel's wave-cycles go (SQ_ACTIVE_INST_ANY / SQ_WAIT_INST_ANY / SQ_WAIT_ANY are disjoint shares of SQ_WAVE_CYCLES,
/opt/skills/guides/MI355X_MICROARCH.md "rocprofv3 PMC slots") and which pipe the issued instructions went to.

usage: tools/pmc_sq.py <out.json> <dir> [<dir> ...] [--bench <bench.json of the profiled command>]
The bench line supplies `_meta` (unitigs, k-mers): bench.py prints an issue-rate roofline only for the workload the counters were
taken on.
"""
import collections
import csv
import glob
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from pmc_traffic import short  # noqa: E402


def main(out, dirs):
    meta = {}
    if "--bench" in dirs:
        i = dirs.index("--bench")
        try:
            with open(dirs[i + 1]) as f:
                b = json.loads(f.read().strip().splitlines()[-1])
            meta = {"unitigs": b["config"]["unitigs_total"], "kmers": b["config"]["kmers_per_gpu"]}
        except (OSError, ValueError, KeyError, IndexError) as e:
            print("pmc_sq: no _meta (%s)" % e, file=sys.stderr)
        dirs = dirs[:i] + dirs[i + 2:]
    acc = collections.defaultdict(lambda: collections.defaultdict(lambda: [0.0, 0]))
    for d in dirs:
        for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
            with open(f) as fh:
                for row in csv.DictReader(fh):
                    k = short(row["Kernel_Name"])
                    if not k:
                        continue
                    a = acc[k][row["Counter_Name"]]
                    a[0] += float(row["Counter_Value"])
                    a[1] += 1
    res = {}
    for k, cs in sorted(acc.items()):
        e = {c: v[0] / max(1, v[1]) for c, v in cs.items()}
        e["launches_sampled"] = max(v[1] for v in cs.values())
        wc = e.get("SQ_WAVE_CYCLES")
        if wc:
            for c in ("SQ_ACTIVE_INST_ANY", "SQ_WAIT_INST_ANY", "SQ_WAIT_ANY", "SQ_WAIT_INST_LDS", "SQ_ACTIVE_INST_VALU", "SQ_ACTIVE_INST_LDS",
                      "SQ_ACTIVE_INST_VMEM", "SQ_ACTIVE_INST_SCA"):
                if c in e:
                    e["share_" + c[3:].lower()] = round(e[c] / wc, 4)
        res[k] = e
    if not res:
        print("pmc_sq: no counter rows found", file=sys.stderr)
        sys.exit(1)
    with open(out, "w") as f:
        json.dump({"_meta": meta, **res}, f, indent=1)
    for k, e in res.items():
        print(k, {x: y for x, y in e.items() if x.startswith("share_") or x.startswith("SQ_INSTS")})


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2:])
