#!/usr/bin/env python3
"""Regenerates the measured tables of DESIGN.md between its `<!-- table:NAME -->` / `<!-- /table:NAME -->` markers from the
committed profiles: one number per cell, no history.
    kernels   profiles/<tag>_bench.json (`kernels`: HIP events on the launch streams inside bench.py), profiles/pmc_traffic.json
              (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes), profiles/pmc_sq.json (SQ counters)
    headline  the bench line's headline fields
usage: python tools/design_tables.py [tag]      (default r14; rewrites DESIGN.md in place)"""
import json
import os
import re
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
TAG = sys.argv[1] if len(sys.argv) > 1 else "r14"

WHAT = {  # kernel -> (what it is, reference function it replaces)
    "k_bfs_thread": ("K-BFS tier 1: thread per candidate entrance, 8-entry LDS tables", "extractSuperBubble_ptr src/CDBG.cpp:253-415"),
    "k_bfs": ("K-BFS tier 2: wavefront per candidate, 128-entry LDS tables", "same"),
    "k_cov_join": ("K-COV-JOIN: every graph k-mer looked up in the minimizer-addressed count table, five rows in flight per wavefront", "CKMCFile::CheckKmer kmc_file.cpp:330-366 under readCov(UnitigMap) src/CDBG.cpp:66-120"),
    "k_cov_join_rest": ("K-COV-JOIN, second kernel: the look-ups whose first line was full", "same"),
    "k_cov": ("K-COV: per-unitig sum / min of the joined per-k-mer coverage SoA (stream)", "readCov(UnitigMap) src/CDBG.cpp:66-120"),
    "k_call_sides": ("K-SCAN: owner / exit / coverage gate / sortSeq_simple per unitig", "ploidyEstimation_ptr src/CDBG.cpp:1146-1222"),
    "k_call_prep": ("K-PREP: bubbles into the alignment tiers' lists", "src/CDBG.cpp:1187-1260"),
    "k_call_snp": ("K-SNP: thread per two-path bubble of one length, single mismatch certified", "SequenceAlignment src/SeqAlign.cpp:550-640"),
    "k_call_pair": ("K-PAIR: thread per two-path bubble, NW in registers, one optimal path", "needlemanWunch / traceback src/SeqAlign.cpp:480-549, 306-478"),
    "k_call_stack": ("K-STACK: thread per bubble of >= 3 paths of one length, diagonal certified", "same"),
    "k_call_paths": ("K-PATHS: wavefront per branching bubble, two-stack walk + sortSeq_branching", "src/CDBG.cpp:1364-1412, 417-480"),
    "k_bubble": ("K-BUBBLE: wavefront per bubble, progressive alignment with every co-optimal traceback", "src/SeqAlign.cpp:8-640"),
    "k_call_sites": ("K-SITES: wavefront per branching bubble, site strings + readCov(string)", "src/CDBG.cpp:1448-1600, 29-60"),
    "k_call_format": ("K-TEXT: thread per bubble, count pass + write pass, ten streams, alignseq packed", "src/CDBG.cpp:1259, 1303-1340, 1552-1652"),
}
ORDER = list(WHAT)


def mb(x):
    x /= 1e6
    return ("%.0f MB" % x) if x >= 100 else ("%.3g MB" % x)


def table_kernels():
    b = json.load(open(os.path.join(ROOT, "profiles", "%s_bench.json" % TAG)))
    tr = json.load(open(os.path.join(ROOT, "profiles", "pmc_traffic.json")))
    sq = json.load(open(os.path.join(ROOT, "profiles", "pmc_sq.json")))
    k = b["kernels"]
    rows = ["| kernel | replaces | launches / pass | ms / pass (sum of launches) | units / pass | algorithmic bytes / launch | achieved | PMC HBM bytes / launch (x algorithmic) | wave-cycles: issuing / waiting |",
            "|---|---|---|---|---|---|---|---|---|"]
    for name in ORDER:
        e = k.get(name)
        if not e:
            continue
        ab = e.get("algorithmic_bytes_per_launch")
        t = tr.get(name)
        s = sq.get(name, {})
        rows.append("| `%s` %s | %s | %g | %s | %s | %s | %s | %s | %s |" % (
            name, WHAT[name][0], WHAT[name][1], e["launches_per_step"], ("%.2f (their union: %.2f)" % (e["ms_per_step"], e["union_ms_per_step"])) if "union_ms_per_step" in e else "%.2f" % e["ms_per_step"],
            ("{:,.0f}".format(e["units_per_step"])) if "units_per_step" in e else "",
            mb(ab) if ab else "",
            ("%.4g GB/s = %.2g %% of 8 TB/s" % (e["achieved_GBps"], e["achieved_GBps"] / 80.0)) if "achieved_GBps" in e else "",
            ("%s (%.2g x)" % (mb(t), t / ab)) if (t and ab) else (mb(t) if t else ""),
            ("%.2f / %.2f" % (s.get("share_active_inst_any", 0), s.get("share_wait_any", 0))) if s else ""))
    return "\n".join(rows)


def table_headline():
    b = json.load(open(os.path.join(ROOT, "profiles", "%s_bench.json" % TAG)))
    c = b["cpu_baseline"] or {}
    rows = ["| field | value |", "|---|---|",
            "| workload | %s |" % b["config"]["workload"].split(",")[0],
            "| `value` (a step = K-COV-JOIN + findSuperBubble + PloidyEstimation) | %.1f M unitigs/s (%.2f ms per step; min %.2f, median %.2f, p95 %.2f) |" % (
                b["value"] / 1e6, b["ms_per_step"], b["ms_per_step_min"], b["ms_per_step_median"], b["ms_per_step_p95"]),
            "| `value_excl_join` (the pass alone, look-ups left to the load-time join: what rounds 1-4 reported) | %.1f M unitigs/s (%.2f ms per pass) |" % (
                b["value_excl_join"] / 1e6, b["ms_per_step_excl_join"]),
            "| `output_check` | %d files, identical_to_reference = %s |" % (b["output_check"]["files"], b["output_check"]["identical_to_reference"]),
            "| device busy | %.2f ms per pass = %.0f %% |" % (b["device_busy_ms_per_step"], 100 * b["device_busy_frac"]),
            "| first pass after the load | %.0f ms (find %.0f + ploidy %.0f) |" % (1e3 * b["first_pass"]["wall_s"], 1e3 * b["first_pass"]["find_total_s"], 1e3 * b["first_pass"]["ploidy_total_s"]),
            "| `load_s` | %.2f s |" % b["load_s"],
            "| `cpu_baseline` (reference `-t 1`, %d-unitig sample) | %.0f unitigs/s |" % (c.get("sample_unitigs", 0), c.get("value", 0)),
            "| reference `-t 1` at the config's size | %s unitigs/s (%s s) |" % (c.get("reference_at_config_size", {}).get("unitigs_per_s"), c.get("reference_at_config_size", {}).get("seconds")),
            "| `roofline` | %s: %.3g GB/s algorithmic = %.2g of the HBM roof |" % (b["roofline"]["kernel"], b["roofline"]["achieved"], b["roofline"]["frac"]),
            "| `roofline_k_cov` | %.0f GB/s = %.2f of the HBM roof |" % (b["roofline_k_cov"]["achieved"], b["roofline_k_cov"]["frac"]),
            "| `roofline_k_cov_join` | %.2f ms, %.0f GB/s algorithmic = %.3f of the HBM roof; PMC traffic %s |" % (
                b["roofline_k_cov_join"]["avg_ms"], b["roofline_k_cov_join"]["achieved"], b["roofline_k_cov_join"]["frac"],
                ("%.2f GB per launch = %.1f x algorithmic" % (b["roofline_k_cov_join"]["traffic"] / 1e9, b["roofline_k_cov_join"]["traffic_over_algorithmic"]))
                if b["roofline_k_cov_join"].get("traffic") else "not in this line"),
            "| `predicted_scaling` | replicated %.1f ms + cut %.1f ms / N: %s x at 2 / 4 / 8 GPUs, ceiling %s x |" % (
                b["predicted_scaling"]["replicated_ms"], b["predicted_scaling"]["cut_ms_at_1_gpu"],
                " / ".join(str(b["predicted_scaling"]["speedup"][n_]) for n_ in ("2", "4", "8")), b["predicted_scaling"]["ceiling"])]
    if b.get("roofline_issue"):
        ri = b["roofline_issue"]
        rows.append("| `roofline_issue` | %s: %.0f G wave-instructions/s = %.2f of the VALU issue roof per launch%s |" % (
            ri["kernel"], ri["achieved"], ri["frac"],
            ("; all launches of a pass over the time any of them runs: %.2f" % ri["frac_all_launches_over_their_union"]) if ri.get("frac_all_launches_over_their_union") else ""))
    return "\n".join(rows)


def main():
    p = os.path.join(ROOT, "DESIGN.md")
    s = open(p).read()
    for name, fn in (("kernels", table_kernels), ("headline", table_headline)):
        pat = re.compile(r"(<!-- table:%s -->\n).*?(\n<!-- /table:%s -->)" % (name, name), re.S)
        if not pat.search(s):
            print("no marker for", name)
            continue
        s = pat.sub(lambda m: m.group(1) + fn() + m.group(2), s)
    open(p, "w").write(s)


if __name__ == "__main__":
    main()
