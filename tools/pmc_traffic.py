#!/usr/bin/env python3
"""Turn two rocprofv3 PMC passes (FETCH_SIZE and WRITE_SIZE, collected separately as the TCC slot
budget requires) into profiles/pmc_traffic.json: HBM bytes per launch for each of this repository's
kernels, with the gfx950 corrections of MI355X_MICROARCH.md (HBM section): counters are in KiB;
FETCH_SIZE under-reports wide coalesced reads by 2x and is doubled FOR THE STREAMING KERNELS ONLY (STREAMING below: whole
cache lines consumed by consecutive lanes); for kernels whose reads are random 16/32-byte probes or dependent pointer chases
the raw value is kept (doubling would overstate them) and both figures are recorded.  WRITE_SIZE is taken as is.

usage: tools/pmc_traffic.py <fetch_dir> <write_dir> <out.json> [<bench.json of the profiled command>]
The optional bench line supplies `_meta` (unitigs, k-mers): bench.py emits a traffic figure only for that very workload.
"""
import csv
import glob
import json
import os
import sys
from collections import defaultdict


def per_kernel(d, counter):
    files = glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True)
    acc = defaultdict(lambda: [0.0, 0])
    for f in files:
        with open(f) as fh:
            for row in csv.DictReader(fh):
                if row.get("Counter_Name") != counter:
                    continue
                name = row["Kernel_Name"]
                acc[name][0] += float(row["Counter_Value"])
                acc[name][1] += 1
    return acc


# kernels that read wide coalesced streams: the x2 FETCH_SIZE correction applies to these and only these
STREAMING = {"k_cov", "k_cov_init", "k_cov_colored", "k_kmc_decode"}


def short(name):
    if "k_cov_stream" in name:    # the per-pass K-COV (k_cov_stream4 / k_cov_stream: streams the joined per-k-mer coverage SoA)
        return "k_cov"
    if "k_cov_join_rest" in name:  # K-COV-JOIN, second kernel: the look-ups whose first line was full
        return "k_cov_join_rest"
    if "k_cov_join" in name:       # K-COV-JOIN: every graph k-mer looked up in the count table
        return "k_cov_join"
    if "k_cov(" in name:          # the probing form of K-COV (pf_unitig_cov_probe, databases without canonical counting)
        return "k_cov_probe"
    for k in ("k_call_sides", "k_call_prep", "k_call_paths", "k_call_sites", "k_call_format", "k_call_snp", "k_call_pair", "k_call_resolve", "k_sb_format",
              "k_cc_edges_long", "k_cc_edges", "k_cc_labels", "k_cc_multi", "k_cc_init", "k_bfs_thread",
              "k_replay_small", "k_replay_label", "k_replay_flag_big", "k_replay_merge", "k_gfa_lines", "k_gfa_pack",
              "k_cov_init", "k_cov_colored", "k_cov", "k_bfs_huge", "k_bfs_big", "k_bfs", "k_bubble", "k_strcov_colored", "k_align", "k_strcov", "k_table_build", "k_adj_insert", "k_adj_probe"):
        if k + "(" in name or k + "<" in name:
            if k == "k_bubble" and "Lb0" in name:
                return "k_bubble_big"
            return k
    return None


def main(fetch_dir, write_dir, out, bench_json=None):
    res = {}
    meta = {}
    if bench_json:
        try:
            with open(bench_json) as f:
                b = json.loads(f.read().strip().splitlines()[-1])
            meta = {"unitigs": b["config"]["unitigs_total"], "kmers": b["config"]["kmers_per_gpu"], "workload": b["config"]["workload"]}
        except (OSError, ValueError, KeyError, IndexError) as e:
            print("pmc_traffic: no _meta (%s)" % e, file=sys.stderr)
    fetch = per_kernel(fetch_dir, "FETCH_SIZE")
    write = per_kernel(write_dir, "WRITE_SIZE")
    names = set(filter(None, (short(n) for n in list(fetch) + list(write))))
    for k in sorted(names):
        fs = [(v[0], v[1]) for n, v in fetch.items() if short(n) == k]
        ws = [(v[0], v[1]) for n, v in write.items() if short(n) == k]
        fk = sum(x[0] for x in fs) / max(1, sum(x[1] for x in fs))
        wk = sum(x[0] for x in ws) / max(1, sum(x[1] for x in ws))
        corr = 2 if k in STREAMING else 1
        res[k] = {"fetch_bytes_per_launch": fk * 1024 * corr, "write_bytes_per_launch": wk * 1024,
                  "hbm_bytes_per_launch": fk * 1024 * corr + wk * 1024, "launches_sampled": sum(x[1] for x in fs),
                  "fetch_correction": corr, "raw_FETCH_SIZE_KiB": fk, "raw_WRITE_SIZE_KiB": wk}
    flat = {k: v["hbm_bytes_per_launch"] for k, v in res.items()}
    with open(out, "w") as f:
        json.dump({"_meta": meta, "_detail": res, **flat}, f, indent=1)
    if not res:
        print("pmc_traffic: no counter rows found in %s / %s" % (fetch_dir, write_dir), file=sys.stderr)
        sys.exit(1)
    for k, v in res.items():
        print("%-14s fetch %.1f MB (x%d)  write %.1f MB per launch" % (k, v["fetch_bytes_per_launch"] / 1e6, v["fetch_correction"], v["write_bytes_per_launch"] / 1e6))


if __name__ == "__main__":
    main(*sys.argv[1:5])
