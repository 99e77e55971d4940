#!/usr/bin/env python3
"""bench.py -- unitigs/s through superbubble detection + variant calling (k=25, z=8) on MI355X.

One "step" = one full pass of the hot path (CDBG::findSuperBubble + CDBG::ploidyEstimation of the
reference, src/Main.cpp:838-848) over a synthetic compacted de Bruijn graph whose packed unitigs,
CSR adjacency and k-mer count table are already resident in HBM; graph generation, GFA/KMC
parsing, upload and Unitig_Id writing are outside the timed region.

  python bench.py [--gpus N] [--steps K] [--warmup W] [--unitigs U]
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

N > 1: one process per GPU, ONE graph (BASELINE.json configs[2]: "5 M-unitig ... 1->8 MI355X scaling"), replicated in every
GPU's HBM (ploidyfrost_amd/dist.py).  PloidyEstimation -- two thirds of a pass -- is cut over the ranks: each rank aligns, formats
and writes its slice of the bubble list into the shared result files after two small all-gathers (bubbles called, slab sizes +
allele histograms).  findSuperBubble is run by every rank by default: its device part is 1.6 ms of a 12 ms phase whose rest --
the copy of the records to the host, the commits, the state upload -- every rank needs in full, so cutting it by entrance vertex
(--shard-find: each rank traverses the entrances of its unitig range, the records are all-gathered over RCCL and replayed on every
rank) adds an exchange of 183 MB to save a millisecond; the protocol is built, tested and selectable, not the default.
"scaling": "strong"; value = unitigs of the graph * steps / max-over-ranks time.  --scaling weak keeps the other reading (every
rank its own graph, no data-path collective).
"""
from __future__ import annotations

import argparse
import json
import os
import re
import shutil
import subprocess
import sys
import tempfile
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

K = 25
Z = 8
LOWER, UPPER = 5, 1000
# BASELINE.json configs by --workload: generator parameters, k, -z, default size and seed.  `stress` = configs[4] (k=31, z=16, long
# indels: the SeqAlign band widened, K-BUBBLE's traceback the phase); its 1 M-unitig setting has committed reference digests.
WORKLOADS = {
    "single":  {"k": 25, "z": 8,  "gen": {}, "seed": 1000, "unitigs": 5_000_000, "config": "configs[2] (5 M) / configs[1] (1 M)"},
    "repeats": {"k": 25, "z": 8,  "gen": {"repeats": True}, "seed": 1000, "unitigs": 5_000_000, "config": "stress of the host tiers, not a BASELINE config"},
    "colored": {"k": 25, "z": 8,  "gen": {}, "seed": 1000, "unitigs": 2_000_000, "config": "configs[3] (3 samples, 2 M unitigs)"},
    "stress":  {"k": 31, "z": 16, "gen": {"max_ins": 50, "p_snp": 0.5, "p_del": 0.1}, "seed": 7, "unitigs": 10_000_000,
                "config": "configs[4] (z=16, long indels, k=31, 10 M unitigs; --unitigs 1000000 = the setting with committed reference digests)"},
}
# ~26 unitigs per kb of genome at ploidy 4 with these gaps (SURVEY.md 8d ratios)
GAP_LO, GAP_HI = 15, 215
UNITIGS_PER_BP = 0.0262


def log(*a):
    print("[bench]", *a, file=sys.stderr, flush=True)


def make_inputs(workdir: str, name: str, genome_len: int, seed: int, device, k: int = K, max_ins: int = 6, ploidy: int = 4,
                p_snp: float = 0.75, p_del: float = 0.13, repeats: bool = False):
    """haplotypes -> compacted dBG (GFA) + KMC1 count database; returns (gfa, db_prefix, n_unitigs, n_kmers).  repeats: the base
    genome carries repeat families, inverted repeats and tandem arrays on 8 % of its length (synth.repeat_rich_edit)."""
    from ploidyfrost_amd import cdbg_build, synth
    t0 = time.time()
    spec = synth.HapSpec(genome_len=genome_len, ploidy=ploidy, seed=seed, gap_lo=GAP_LO, gap_hi=GAP_HI, p_multi=0.03,
                         max_ins=max_ins, p_snp=p_snp, p_del=p_del)
    haps = synth.make_haplotypes(spec, synth.repeat_rich_edit(seed + 1) if repeats else None)
    g = cdbg_build.build_cdbg(haps, k, device)
    gfa = os.path.join(workdir, name + ".gfa")
    n_unitigs = cdbg_build.write_gfa(gfa, g)
    counts = synth.synth_counts(g["kmers"], g["mult"])
    db = os.path.join(workdir, name + "_kmc")
    synth.write_kmc1(db, g["kmers"], counts, k)
    log("inputs %s: genome %d bp x%d -> %d unitigs, %d k-mers (%.1fs)" % (name, genome_len, ploidy, n_unitigs, len(g["kmers"]),
                                                                        time.time() - t0))
    return gfa, db, n_unitigs, len(g["kmers"])


def make_colored_inputs(workdir: str, name: str, genome_len: int, seed: int, device, k: int = K, samples: int = 3, ploidy: int = 2,
                        max_ins: int = 6):
    """BASELINE.json configs[3]: `samples` samples of `ploidy` haplotypes each on one base genome -> colored compacted dBG
    (GFA with DA tags + .bfg_colors written by ploidyfrost_amd.bfg_colors) + one KMC1 database per sample.  Samples
    after the first lose a few bases at both ends (inside the variant-free flanks, so the graph is unchanged) and the
    first and last unitig carry those colours on part of their k-mers only.
    Returns (gfa, colors, [db prefixes], n_unitigs, n_kmers)."""
    from ploidyfrost_amd import bfg_colors, cdbg_build, synth
    t0 = time.time()
    spec = synth.HapSpec(genome_len=genome_len, ploidy=samples * ploidy, seed=seed, gap_lo=GAP_LO, gap_hi=GAP_HI, p_multi=0.03,
                         max_ins=max_ins)
    haps = synth.make_haplotypes(spec)
    groups = []
    for s_ in range(samples):
        hs = haps[s_ * ploidy: (s_ + 1) * ploidy]
        if s_:
            hs = [h[9 * s_ + 3 * i: len(h) - 7 * s_ - 2 * i] for i, h in enumerate(hs)]
        groups.append(hs)
    g = cdbg_build.build_cdbg(haps, k, device)
    off = np.asarray(g["off"], dtype=np.int64)
    codes = np.asarray(g["codes"], dtype=np.uint8)
    n_unitigs = len(off) - 1
    log("colored inputs %s: graph of %d unitigs built (%.1fs); colour sets and databases next" % (name, n_unitigs, time.time() - t0))
    sizes = np.diff(off)
    km_per = sizes - k + 1
    # Membership of every graph k-mer in every sample, on `device` (torch: the GPU of the box, or the CPU): the canonical k-mer of
    # every (unitig, position) -- all windows of the concatenated text minus those spanning a boundary -- looked up in each sample's
    # sorted distinct canonical k-mers.  Integer work throughout: the files are the same on either device.
    import torch
    tdev = torch.device(device) if not isinstance(device, torch.device) else device
    t_codes = torch.from_numpy(codes).to(tdev)
    t_off = torch.from_numpy(off).to(tdev)
    t_km = torch.from_numpy(km_per.astype(np.int64)).to(tdev)
    fw, rc = cdbg_build._kmers(t_codes, k)
    t_first = torch.cumsum(t_km, 0) - t_km
    start = torch.repeat_interleave(t_off[:-1] - t_first, t_km) + torch.arange(int(km_per.sum()), device=tdev, dtype=torch.int64)
    # the colour set of a unitig is placed by the hash of its head k-mer as the graph stores it: k-length unitigs are
    # kept canonical (bifrost/src/CompactedDBG.tcc:3945-3954)
    head_fw, head_rc = fw[t_off[:-1]], rc[t_off[:-1]]
    heads_t = torch.where(torch.from_numpy(sizes == k).to(tdev), torch.minimum(head_fw, head_rc), head_fw)
    heads = bfg_colors.left_align(heads_t.cpu().numpy().astype(np.uint64), k)
    can = torch.minimum(fw[start], rc[start])
    del fw, rc, start, head_fw, head_rc, heads_t
    first = (np.cumsum(km_per) - km_per)
    t_last = t_first + t_km - 1
    full_mask = np.zeros(n_unitigs, dtype=np.uint64)
    present = []
    dbs = []
    for s_, hs in enumerate(groups):
        parts = []
        for h in hs:
            hf, hr = cdbg_build._kmers(torch.from_numpy(np.ascontiguousarray(h)).to(tdev), k)
            parts.append(torch.minimum(hf, hr))
            del hf, hr
        km_t, mult_t = torch.unique(torch.cat(parts), return_counts=True)
        del parts
        km_s, mult_s = km_t.cpu().numpy().astype(np.uint64), mult_t.cpu().numpy()
        db = os.path.join(workdir, "%s_kmc%d" % (name, s_))
        synth.write_kmc1(db, km_s, synth.synth_counts(km_s, mult_s), k)
        dbs.append(db)
        idx = torch.searchsorted(km_t, can).clamp_(max=km_t.numel() - 1)
        has_t = km_t[idx] == can
        del idx, km_t, mult_t
        cs = torch.cumsum(has_t.to(torch.int64), 0)
        cnt_t = cs[t_last] - cs[t_first] + has_t[t_first].to(torch.int64)
        del cs
        has, cnt = has_t.cpu().numpy(), cnt_t.cpu().numpy()
        del has_t, cnt_t
        full_mask |= (cnt == km_per).astype(np.uint64) << np.uint64(s_)
        present.append((has, cnt))
        log("colored inputs %s: database and colour of sample %d (%.1fs)" % (name, s_, time.time() - t0))
    del can, t_codes
    partial_ids = {}
    part = np.zeros(n_unitigs, dtype=bool)
    for has, cnt in present:
        part |= (cnt > 0) & (cnt < km_per)
    for u in np.nonzero(part)[0]:
        ids = []
        for s_, (has, cnt) in enumerate(present):
            row = has[first[u]: first[u] + km_per[u]]
            ids.extend(int(s_ * km_per[u] + p) for p in np.nonzero(row)[0])
        partial_ids[int(u)] = ids
    colors = os.path.join(workdir, name + ".bfg_colors")
    da = bfg_colors.write_bfg_colors(colors, heads, sizes, k, ["sample%d" % s_ for s_ in range(samples)], full_mask=full_mask,
                                     partial_ids=partial_ids)
    log("colored inputs %s: colour file written (%.1fs)" % (name, time.time() - t0))
    gfa = os.path.join(workdir, name + ".gfa")
    cdbg_build.write_gfa(gfa, g, da_tags=da)
    log("colored inputs %s: genome %d bp, %d samples x%d -> %d unitigs, %d k-mers, %d unitigs with a partial colour (%.1fs)" %
        (name, genome_len, samples, ploidy, n_unitigs, len(g["kmers"]), len(partial_ids), time.time() - t0))
    return gfa, colors, dbs, n_unitigs, len(g["kmers"])


def cpu_baseline(workdir: str, target_unitigs: int, device, runs: int = 3, workload: str = "single", config_unitigs: int | None = None):
    """The reference (oracle/_ref/PloidyFrost -t 1) -- or, where that binary is absent, the oracle
    restatement -- on a bounded sample of the same workload, timed on this host's CPU: `runs` runs, median
    (the reference's own phase timers, src/CDBG.cpp:217-220, 1683-1686)."""
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import pyoracle
    wl = WORKLOADS[workload]
    Z = wl["z"]
    genome = int(target_unitigs / UNITIGS_PER_BP)
    gfa, db, n_unitigs, _ = make_inputs(workdir, "cpu_sample", genome, 4242, device, k=wl["k"], **wl["gen"])
    cwd = os.path.join(workdir, "cpu_run")
    os.makedirs(cwd, exist_ok=True)
    if os.path.exists(pyoracle.REF_BIN):
        kind = "reference"
        cmd = [pyoracle.REF_BIN, "-g", gfa, "-d", db, "-o", "b", "-t", "1", "-l", str(LOWER), "-u", str(UPPER), "-z", str(Z)]
        pat = r"(?:findSuperBubble\(\):|PloidyEstimation\(\):)\s+Cpu time : ([0-9.e+-]+)s"
    else:
        kind = "port"
        pyoracle.build()
        cmd = [pyoracle.CLI, "-g", gfa, "-d", db, "-o", "b", "-l", str(LOWER), "-u", str(UPPER), "-z", str(Z)]
        pat = r"(?:findSuperBubble\(\):|PloidyEstimation\(\):)\s+Real time : ([0-9.e+-]+)s"
    each = []
    for i in range(runs):
        r = subprocess.run(cmd, cwd=cwd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
        if r.returncode != 0:
            log("cpu baseline failed:", r.stdout[-2000:])
            return None
        secs = [float(x) for x in re.findall(pat, r.stdout)]
        if len(secs) != 2:
            return None
        each.append(sum(secs))
        log("cpu baseline (%s) run %d/%d: %d unitigs in %.2fs (find %.2fs + ploidy %.2fs)" % (kind, i + 1, runs, n_unitigs, each[-1], secs[0], secs[1]))
    t = sorted(each)[len(each) // 2]
    # context: the same sample with every core the reference accepts (-t must not exceed
    # hardware_concurrency, src/Main.cpp:287-291).  With -t > 1 the reference times only PloidyEstimation, in whole seconds
    # (time(), src/CDBG.cpp:2668-2690), so the figure here is the program's wall time, load included, next to the -t 1 run's.
    all_cores = None
    if kind == "reference":
        nt = os.cpu_count() or 1
        wall = {}
        for th in (1, nt):
            tw = time.time()
            r = subprocess.run(cmd[:cmd.index("-t") + 1] + [str(th)] + cmd[cmd.index("-t") + 2:], cwd=cwd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
            wall[th] = time.time() - tw
            if r.returncode != 0:
                wall = None
                break
        if wall:
            load_est = max(0.0, wall[1] - t)   # what -t 1 spends outside its two timed phases
            all_cores = {"threads": nt, "program_wall_s": round(wall[nt], 2), "program_wall_s_t1": round(wall[1], 2),
                         "phases_wall_s_estimate": round(max(wall[nt] - load_est, 1e-9), 2),
                         "unitigs_per_s_estimate": round(n_unitigs / max(wall[nt] - load_est, 1e-9), 1),
                         "note": "whole-program wall times; the phases' share of the -t %d run is estimated by taking off what the -t 1 run "
                                 "spends outside its own phase timers (graph + database load)" % nt}
            log("cpu baseline -t %d: program wall %.2fs (-t 1: %.2fs)" % (nt, wall[nt], wall[1]))
    full = reference_digests().get(digest_key(workload, config_unitigs or CONFIG_UNITIGS)) or {}
    if "t1" in full and "findSuperBubble_cpu_s" in full["t1"]:
        secs = full["t1"]["findSuperBubble_cpu_s"] + full["t1"]["PloidyEstimation_cpu_s"]
        at_size = {"unitigs": full["unitigs"], "seconds": round(secs, 2), "unitigs_per_s": round(full["unitigs"] / secs, 1), "cores": 1,
                   "measured": "%s on %s by tools/reference_digests.py (committed in profiles/reference_digests.json; NOT part of this run)" % (full.get("date"), full.get("host")),
                   "all_cores": {k_: v for k_, v in full.items() if k_.startswith("t") and k_ != "t1" and isinstance(v, dict)}}
    else:
        at_size = dict(REFERENCE_FULL_SIZE, stale_constant=True) if workload == "single" else None
    return {"value": n_unitigs / t, "unit": "unitigs/s", "cores": 1, "kind": kind, "runs": runs, "seconds": t,
            "seconds_each": [round(x, 3) for x in each], "sample_unitigs": n_unitigs,
            "sample": "same generator (tetraploid, k=%d, z=%d, -l %d -u %d), %d-unitig graph, findSuperBubble+PloidyEstimation "
                      "Cpu time of `-t 1`, median of %d runs on %s" % (wl["k"], Z, LOWER, UPPER, n_unitigs, runs, cpu_model()),
            # not like-for-like in size: the reference's unitigs/s FALLS as the graph grows (its per-k-mer binary search and
            # std::map traversals lose cache), so the ratio against this sample understates the ratio at the config's size
            "note": "bounded sample, smaller than the timed config; the reference's rate falls with graph size",
            "all_cores": all_cores, "reference_at_config_size": at_size}


# The reference binary on the full configs[2] graph (same generator, seed 1000), timed once per build on the GPU box's host and
# committed: a 12-minute run cannot sit inside the default bench.  Source: profiles/history/r01_fullscale_parity.txt.
REFERENCE_FULL_SIZE = {"unitigs": 4990608, "seconds": 697.96, "unitigs_per_s": 7150.3, "cores": 1,
                       "source": "profiles/history/r01_fullscale_parity.txt (a round-1 measurement: used only when profiles/reference_digests.json "
                                 "holds no run of the config's graph, and then marked stale_constant)"}
CONFIG_UNITIGS = 4990608   # bench.make_inputs(5 000 000 target, seed 1000): BASELINE.json configs[2]


def cpu_model() -> str:
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                return "%s (%d logical CPUs)" % (line.split(":", 1)[1].strip(), os.cpu_count())
    except OSError:
        pass
    return "unknown CPU"


# algorithmic HBM bytes per launch (restated in DESIGN.md)
def algorithmic_bytes(kernel: str, t: dict, units: float | None = None) -> float | None:
    """Per pass.  `units` = work items the kernel's launches were given per pass (pf_kernel_units): the tiers of K-BFS and of
    the alignment (K-SNP / K-PAIR / K-BUBBLE) share one per-unit figure and are priced on what each of them actually took."""
    if kernel == "k_cov":  # streaming form: 4 B count + 1 head bit per k-mer, 4 B row id per 64 k-mers, 16 B of results per unitig
        return (4.0 + 1.0 / 8 + 4.0 / 64) * t["kmers"] + 16.0 * t["unitigs"]
    if kernel in ("k_cov_probe", "k_cov_join"):  # 0.25 B sequence + 12 B table slot per k-mer (+ 16 B result per unitig), per colour
        return (12.25 * t["kmers"] + 16.0 * t["unitigs"]) * max(1, t.get("n_colors", 1))
    if kernel in ("k_bfs", "k_bfs_thread"):
        return 600.0 * (units if units is not None else t["candidates"])
    if kernel in ("k_align", "k_bubble", "k_call_snp", "k_call_pair", "k_call_stack"):
        return 300.0 * (units if units is not None else t["align_jobs"])
    if kernel == "k_call_format":  # K-TEXT: the result text itself (written once; its inputs are a fraction of it)
        return float(t["output_bytes"])
    if kernel == "k_strcov":
        return 12.0 * 2.0 * t["site_strings"]
    if kernel == "k_cov_colored":  # streaming form: 4 B count per (k-mer, colour), head bit + row id per k-mer, 17 B of results per (colour, unitig)
        c = t.get("n_colors", 3)
        return (4.0 * c + 1.0 / 8 + 4.0 / 64) * t["kmers"] + 17.0 * c * t["unitigs"]
    if kernel == "k_strcov_colored":
        return 12.0 * 2.0 * t["site_strings"] * t.get("n_colors", 3)
    return None


def cov_roofline(kernels: dict, colored: bool, n_unitigs: int):
    """`roofline_k_cov`: the streaming coverage kernel of the workload (K-COV, or K-COV-C for the colored one) -- the kernel
    SURVEY.md 8(d) names as able to approach the HBM roof.  `traffic` = PMC-measured HBM bytes per launch from the committed
    profile (single-sample kernel only), `traffic_frac` = traffic / duration / peak."""
    name = "k_cov_colored" if colored else "k_cov"
    e = kernels.get(name)
    if not e or "achieved_GBps" not in e:
        return None
    tr = None if colored else load_traffic(name, n_unitigs)
    return {"kernel": name, "bound": "hbm", "achieved": e["achieved_GBps"], "peak": 8000.0, "unit": "GB/s",
            "frac": round(e["achieved_GBps"] / 8000.0, 5), "traffic": tr,
            "traffic_frac": round(tr / (e["avg_ms"] * 1e-3) / 8e12, 4) if tr else None}


def make_inputs_child(spec_json: str) -> None:
    """`bench.py --_make-inputs <json>`: the input generator in a process of its own (see main()); prints one JSON line."""
    spec = json.loads(spec_json)
    import torch
    dev = torch.device("cuda", spec["gpu"]) if torch.cuda.is_available() else torch.device("cpu")
    if spec["workload"] == "colored":
        gfa, colors, dbs, n_unitigs, n_kmers = make_colored_inputs(spec["workdir"], "graph", spec["genome"], spec["seed"], dev, samples=3)
        out = {"gfa": gfa, "colors": colors, "dbs": dbs, "n_unitigs": n_unitigs, "n_kmers": n_kmers}
    else:
        wl = WORKLOADS[spec["workload"]]
        gfa, db, n_unitigs, n_kmers = make_inputs(spec["workdir"], "graph", spec["genome"], spec["seed"], dev, k=wl["k"], **wl["gen"])
        out = {"gfa": gfa, "db": db, "n_unitigs": n_unitigs, "n_kmers": n_kmers}
    print(json.dumps(out), flush=True)


def main():
    if len(sys.argv) == 3 and sys.argv[1] == "--_make-inputs":
        return make_inputs_child(sys.argv[2])
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--unitigs", type=int, default=0,
                    help="target unitigs of the graph (default: the workload's config size -- single 5 M = BASELINE.json configs[2], the config "
                         "the metric and the north-star target are quoted on; colored 2 M = configs[3]; stress 10 M = configs[4])")
    ap.add_argument("--cpu-sample-unitigs", type=int, default=150_000,
                    help="size of the reference's sample graph: ~6 s per -t 1 run at 150 k unitigs; its time grows faster than the graph "
                         "(a 250 k sample with chromosome-long traversals takes 75 s, the 5 M graph 700 s)")
    ap.add_argument("--cpu-runs", type=int, default=3)
    ap.add_argument("--host-threads", type=int, default=0, help="host threads per rank (0 = min(32, cpus/ranks))")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--gen-in-process", action="store_true",
                    help="make the inputs in this process instead of a child (for runs under rocprofv3, which must not have children)")
    ap.add_argument("--keep", action="store_true")
    ap.add_argument("--no-verify", action="store_true", help="N > 1: skip the comparison of the shared result files with a pass of rank 0 alone")
    ap.add_argument("--scaling", choices=["weak", "strong"], default="strong",
                    help="strong (default: the metric's config is one 5 M-unitig graph at 1..8 GPUs): ONE graph of --unitigs unitigs cut "
                         "by entrance vertex over the ranks (SURVEY.md 8e); weak: every rank its own graph of --unitigs unitigs")
    ap.add_argument("--shard-find", action="store_true",
                    help="strong scaling: cut findSuperBubble by entrance vertex as well (all-gather of the traversal records) instead of "
                         "running it on every rank")
    ap.add_argument("--workload", choices=["single", "colored", "repeats", "stress"], default="single",
                    help="single = the headline metric (BASELINE.json configs[1]); colored = the CCDBG path on 3 diploid "
                         "samples (configs[3]), same JSON line with config.workload saying so; repeats = the single-sample path on a "
                         "repeat-rich genome (50 repeat families, inverted repeats, tandem arrays on 8 %% of it): the stress case of the "
                         "host tiers -- traversals that do not close for thousands of unitigs, giant commit components; stress = configs[4]: "
                         "k=31, -z 16, insertions up to 50 bp (K-BUBBLE's traceback is the phase), 10 M unitigs (--unitigs 1000000: the "
                         "setting whose reference digests are committed)")
    args = ap.parse_args()
    wl = WORKLOADS[args.workload]
    if not args.unitigs:
        args.unitigs = wl["unitigs"]
    Z = wl["z"]

    if args.gpus > 1 and os.environ.get("PF_BENCH_SHARE_GPU") != "1":
        import torch   # (counting devices does not initialise the GPU)
        have = torch.cuda.device_count()
        if have < args.gpus:
            raise SystemExit("bench.py: --gpus %d but this node shows %d GPU(s): one rank per GPU (PF_BENCH_SHARE_GPU=1 puts every rank on GPU 0, "
                             "for debugging on a one-GPU box only)" % (args.gpus, have))
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        if args.gen_in_process:
            # --gen-in-process is the mode for runs under a profiler, whose preloaded library has initialised the GPU in this
            # process already: starting the ranks from here would be the fork-after-GPU-init the pool forbids, and the trace would
            # be of a parent that launches no kernel
            raise SystemExit("bench.py: --gen-in-process with --gpus %d needs a launcher (python -m torch.distributed.run --nproc-per-node %d ... "
                             "bench.py --gpus %d --gen-in-process): the ranks are not started from a process a profiler is attached to"
                             % (args.gpus, args.gpus, args.gpus))
        # `python bench.py --gpus N` without a launcher: start the N ranks here -- a child process of torch.distributed.run, before
        # anything in this process has touched the GPU (this parent never does) -- and leave with the child's exit code
        import socket
        with socket.socket() as so:
            so.bind(("127.0.0.1", 0))
            port = so.getsockname()[1]
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(args.gpus), "--master-addr", "127.0.0.1",
               "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
        log("--gpus %d without WORLD_SIZE: launching %s" % (args.gpus, " ".join(cmd)))
        sys.exit(subprocess.run(cmd).returncode)

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        raise SystemExit("bench.py: --gpus %d but WORLD_SIZE=%d: one rank per GPU (launch with --nproc-per-node %d, or without a launcher)"
                         % (args.gpus, world, args.gpus))
    # PF_BENCH_SHARE_GPU=1 (debugging on a one-GPU box only): every rank uses GPU 0 and the end-of-pass
    # exchange runs over gloo on CPU tensors, because RCCL refuses two ranks on one device.
    share = os.environ.get("PF_BENCH_SHARE_GPU") == "1"
    gpu_index = 0 if share else local_rank
    workdir = tempfile.mkdtemp(prefix="pf_bench_r%d_" % rank, dir="/dev/shm" if os.path.isdir("/dev/shm") else None)
    genome = int(args.unitigs / UNITIGS_PER_BP)
    colored = args.workload == "colored"
    strong = args.scaling == "strong" and world > 1
    seed = int(os.environ.get("PF_BENCH_SEED", str(wl["seed"]))) + (0 if strong else rank)
    made = None
    if not args.gen_in_process:
        # The inputs are made by a CHILD process (this file with --_make-inputs; it uses this rank's GPU for the graph construction and
        # exits), started before this process has touched the GPU: the generator's tens of GB of device memory are gone with it.  Made
        # in this process they were handed back by torch.cuda.empty_cache() just before the load was timed, and on some boxes the first
        # allocations of the load then waited two seconds for the driver to reclaim them -- time of the generator's, booked on the
        # product's load_s.  (--gen-in-process: under a profiler, whose preloaded library forbids children.)
        gen = subprocess.run([sys.executable, os.path.abspath(__file__), "--_make-inputs", json.dumps(
            {"workdir": workdir, "genome": genome, "seed": seed, "gpu": gpu_index, "workload": args.workload})], stdout=subprocess.PIPE, text=True)
        if gen.returncode != 0:
            shutil.rmtree(workdir, ignore_errors=True)
            raise SystemExit("bench.py: the input generator failed")
        made = json.loads(gen.stdout.strip().splitlines()[-1])

    import torch
    import torch.distributed as dist
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: the hot path has no CPU fallback")
    torch.cuda.set_device(gpu_index)
    dev = torch.device("cuda", gpu_index)
    from ploidyfrost_amd import dist as pfdist
    from ploidyfrost_amd import hipapi, hostapi
    pfdist.init("gloo" if share else "nccl", None if share else dev)
    xdev = torch.device("cpu") if share else dev  # where the collectives' tensors live
    ok = False
    bad_output = False
    try:
        host_threads = args.host_threads or max(1, min(32, (os.cpu_count() or 1) // max(world, 1)))
        if made is None:
            if colored:
                m_ = make_colored_inputs(workdir, "graph", genome, seed, dev, samples=3)
                made = {"gfa": m_[0], "colors": m_[1], "dbs": m_[2], "n_unitigs": m_[3], "n_kmers": m_[4]}
            else:
                m_ = make_inputs(workdir, "graph", genome, seed, dev, k=wl["k"], **wl["gen"])
                made = {"gfa": m_[0], "db": m_[1], "n_unitigs": m_[2], "n_kmers": m_[3]}
            torch.cuda.empty_cache()
        if colored:
            n_samples = 3
            gfa, colors, dbs, n_unitigs, n_kmers = made["gfa"], made["colors"], made["dbs"], made["n_unitigs"], made["n_kmers"]
            cutoffs = [(LOWER, UPPER)] * n_samples
        else:
            gfa, db, n_unitigs, n_kmers = made["gfa"], made["db"], made["n_unitigs"], made["n_kmers"]
        # The device memory the input generator held (tens of GB, in a child process or in torch's cache) is handed back to the driver
        # lazily: on some boxes the load's first large allocation -- the 8.6 GB count table -- then waited a second for it
        # (`kmc: device decode + table` 1.19 s instead of 0.12 s in one driver-style run).  One large allocation, touched, here: the
        # generator's leftovers are dealt with before the load is timed, not by it.  It is HELD until the load is over: freeing it
        # right here made the stall the rule (the driver clears freed memory before it hands it out again, and a hipMalloc that
        # arrives meanwhile waits: load_s 2.3 - 2.6 s in four of five runs once the load had become quick enough to arrive in time).
        scrub_ = None
        try:
            scrub_ = torch.empty(32 << 30, dtype=torch.uint8, device=dev)
            scrub_.zero_()
            torch.cuda.synchronize()
        except RuntimeError:
            scrub_ = None
        hostapi.load_trace(reset=True)
        t0 = time.time()
        if colored:
            run = hostapi.ColoredRun(gfa, colors, dbs, workdir, z=Z, threads=host_threads, device=gpu_index)
        else:
            run = hostapi.Run(gfa, db, z=Z, device=gpu_index)
        open_s = time.time() - t0
        del scrub_   # (see above; handed back while the unitig id file is written)
        torch.cuda.empty_cache()
        run.set_threads(host_threads)
        run.set_overlap_output(True)  # super_bubble.txt is written while PloidyEstimation runs; complete when it returns
        if os.environ.get("PF_BATCH_BUBBLES"):  # experiments: bubbles per batch of the align/format pipeline
            run.set_batch_bubbles(int(os.environ["PF_BATCH_BUBBLES"]))
        run.set_output_dir(os.path.join(workdir, "PloidyFrost_output"))
        if strong:
            # one node, one file system: every rank writes its slabs into rank 0's result files at its own offsets
            run.set_output_dir(pfdist.broadcast_str(os.path.join(workdir, "PloidyFrost_output")))
        if not strong or rank == 0:
            run.set_unitig_id("b")
        if strong and rank != 0:
            run.set_write_super_bubble(False)
        load_s = time.time() - t0
        # where load_s went: the host layer's own steps (pfh_load_trace; steps of helper threads overlap the caller's, so the sum can
        # exceed the wall time) + the unitig id file
        load_breakdown = {"open_wall_s": round(open_s, 3), "unitig_id_file_and_settings_s": round(load_s - open_s, 3),
                          "steps_s": {}}
        for name_, sec_ in hostapi.load_trace(reset=True):   # (a step name that comes more than once -- a database per colour -- adds up)
            load_breakdown["steps_s"][name_] = round(load_breakdown["steps_s"].get(name_, 0.0) + sec_, 4)
        if open_s > 1.5:
            # seen in one run in three on this pool, 2.1 s each time, in whichever call makes the load's first large hipMalloc: the
            # driver is still clearing device memory that was freed just before (the input generator's, or the buffer above); any
            # process shows it (tools/exp/malloc_cost.cpp: one 8 GB hipMalloc of 2 118 ms among calls of 0.2 ms)
            load_breakdown["note"] = ("a first large hipMalloc of this load waited about two seconds for the driver to clear memory freed just before "
                                      "(not this library's time: tools/exp/malloc_cost.cpp shows the same stall in a bare HIP program); "
                                      "runs without it load in 0.45 - 0.55 s")
        log("rank %d: load+upload+adjacency+table %.1fs on %s" % (rank, load_s, torch.cuda.get_device_name(gpu_index)))
        L = hipapi.load_library()
        import ctypes as C
        ctx = C.c_void_p(run.device_ctx())

        gathered_bytes = [0]

        shard_stats = {}

        # The reference's timed phases hold every readCov look-up (src/CDBG.cpp:66-120 inside :1187-1220; 48 % of its time).  The
        # product makes them ONCE per (graph, database), as the load-time join K-COV-JOIN, and a pass only streams the joined
        # coverage array -- so a pass alone is not like-for-like with the reference's phases.  The headline step therefore repeats
        # the join: one pf_join_counts (233 M look-ups at 5 M unitigs) + findSuperBubble + PloidyEstimation.  `*_excl_join` is
        # the pass without it, measured in a region of its own (what rounds 1-4 reported as `value`).
        join_in_step = [True]

        def step():
            if join_in_step[0]:
                # on a stream of its own, beside findSuperBubble (which reads no coverage); PloidyEstimation's K-COV waits for it
                st_ = L.pf_join_counts_begin(ctx)
                if st_ != 0:
                    raise SystemExit("bench.py: pf_join_counts_begin failed (%d)" % st_)
            if strong:
                if args.shard_find:
                    pfdist.sharded_find(run, "b", xdev, shard_stats)
                else:
                    run.find_superbubbles("b")   # on every rank; rank 0 writes b_super_bubble.txt
                totals_, counters_ = pfdist.sharded_ploidy(run, "b", LOWER, UPPER, xdev, shard_stats, cutoffs=cutoffs if colored else None)
                shard_stats["counters"] = [int(x) for x in counters_]
                shard_stats["output_bytes"] = int(totals_.sum())
                return
            run.find_superbubbles("b")
            if colored:
                run.ploidy_estimation("b", cutoffs)
            else:
                run.ploidy_estimation("b", LOWER, UPPER)
            if world > 1:
                # end-of-pass exchange over RCCL: counters + the ordered allele-frequency record slab
                tt_ = run.times()
                pfdist.all_gather_counters(tt_["allele"] + [tt_["tasks"]], xdev)
                slabs = pfdist.all_gather_slabs(run.last_allele_frequency(), xdev)
                gathered_bytes[0] = sum(int(x.size) for x in slabs)

        first_pass = None
        for w_ in range(args.warmup):
            tw_ = time.perf_counter()
            step()
            if w_ == 0:  # what a user's one-shot run pays: first pass over a freshly loaded graph (one-time allocations, one K-BFS slice)
                tt_ = run.times()
                first_pass = {"wall_s": round(time.perf_counter() - tw_, 4), "find_total_s": round(tt_["find_total_s"], 4),
                              "ploidy_total_s": round(tt_["ploidy_total_s"], 4)}
        # HIP events inside the timed region for the kernels the roofline objects speak of (K-BUBBLE, K-COV) only: a pass is a hundred
        # launches from five host threads, and two events around every one of them cost the pass 2 ms (20.4 ms with all launches
        # timed, 17.3-18.4 ms with none: profiles/r16_experiments.txt).  The other kernels' rows of `kernels` and the device-busy figure
        # come from TABLE_PASSES further passes behind the timed region, with every launch timed.
        roof_kernels = [k for k in ("k_bubble", "k_bubble_big", "k_cov", "k_cov_colored", "k_cov_join", "k_cov_join_rest") if k in hipapi.KERNELS]
        roof_mask = 0
        for k in roof_kernels:
            roof_mask |= 1 << hipapi.KERNELS.index(k)
        if os.environ.get("PF_BENCH_NO_EVENTS"):   # measurements: what the events inside the timed region cost (the roofline then reads the table passes)
            roof_mask = 0
        L.pf_enable_timing(ctx, 1)
        L.pf_timing_select(ctx, roof_mask)
        L.pf_reset_timing(ctx)
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        phase = {}
        step_s = []
        for _ in range(args.steps):
            ts_ = time.perf_counter()
            step()
            step_s.append(time.perf_counter() - ts_)   # a step ends with its files complete: nothing of it is still in flight
            tt = run.times()
            for key in ("bfs_device_s", "replay_s", "bubble_write_s", "cov_device_s", "tasks_s", "align_s", "sites_s", "format_s",
                        "write_s", "scan_s", "scan_serial_s", "find_total_s", "ploidy_total_s"):
                phase[key] = phase.get(key, 0.0) + tt[key]
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        elapsed = time.perf_counter() - t0
        tt = run.times()
        def kernel_times():   # kernel times from HIP events recorded by the library on the launches' own streams
            kt, ku = {}, {}
            for i, name in enumerate(hipapi.KERNELS):
                ms, n = C.c_double(), C.c_uint64()
                L.pf_kernel_time(ctx, i, C.byref(ms), C.byref(n))
                if n.value:
                    kt[name] = (ms.value, n.value)
                    u = C.c_uint64()
                    L.pf_kernel_units(ctx, i, C.byref(u))
                    ku[name] = u.value
            return kt, ku
        ktimes_live, kunits_live = kernel_times()
        # K-BUBBLE is launched once per size class and align range, side by side: what it occupies of a pass is the union of those
        # launches' intervals, not their sum
        bub_union = C.c_double()
        L.pf_kernel_busy(ctx, sum(1 << hipapi.KERNELS.index(k) for k in ("k_bubble", "k_bubble_big") if k in hipapi.KERNELS), C.byref(bub_union))

        L.pf_timing_select(ctx, 0)   # the second region is timed as a whole only
        # the same K steps without the join (the pass over the joined coverage array alone), timed the same way
        elapsed_excl, step_s_excl = None, []
        if join_in_step[0]:
            join_in_step[0] = False
            if world > 1:
                dist.barrier()
            torch.cuda.synchronize()
            te_ = time.perf_counter()
            for _ in range(args.steps):
                ts_ = time.perf_counter()
                step()
                step_s_excl.append(time.perf_counter() - ts_)
            torch.cuda.synchronize()
            if world > 1:
                dist.barrier()
            elapsed_excl = time.perf_counter() - te_
            join_in_step[0] = True

        # the kernel table and the device-busy time (the union of the launches' intervals: the pipeline runs kernels side by side, so
        # the per-kernel sums overlap): further passes, every launch timed, outside the timed region
        TABLE_PASSES = max(1, min(5, args.steps))
        L.pf_timing_select(ctx, (1 << 64) - 1)
        L.pf_reset_timing(ctx)
        torch.cuda.synchronize()
        t_tab = time.perf_counter()
        for _ in range(TABLE_PASSES):
            step()
        torch.cuda.synchronize()
        table_elapsed = time.perf_counter() - t_tab
        if world > 1:
            dist.barrier()
        busy_ms, span_ms = C.c_double(), C.c_double()
        L.pf_device_busy(ctx, C.byref(busy_ms), C.byref(span_ms))
        # ... and without the copies of the result text to the host (the runtime moves them with a kernel of its own on the device:
        # rocprofv3 lists it as __amd_rocclr_copyBuffer; the library times them as "copy_text_to_host")
        busy_own = C.c_double()
        copy_bit = 1 << hipapi.KERNELS.index("copy_text_to_host")
        L.pf_kernel_busy(ctx, ((1 << len(hipapi.KERNELS)) - 1) & ~copy_bit, C.byref(busy_own))
        ktimes, kunits = kernel_times()

        # outside the timed region: C1 computed the other way (every k-mer probed in the hash table, what K-COV-JOIN does once
        # at load) -- its duration, and that both routes agree on this graph
        probe = None
        if rank == 0 and not colored:
            import numpy as np
            res = [(np.zeros(n_unitigs, np.uint64), np.zeros(n_unitigs, np.uint32), np.zeros(n_unitigs, np.uint8)) for _ in range(2)]
            L.pf_reset_timing(ctx)
            for fn, (s_, m_, x_) in ((L.pf_unitig_cov, res[0]), (L.pf_unitig_cov_probe, res[1]), (L.pf_unitig_cov_probe, res[1])):
                fn(ctx, 0, n_unitigs, s_.ctypes.data, m_.ctypes.data, x_.ctypes.data)
                if fn is L.pf_unitig_cov:
                    L.pf_reset_timing(ctx)
            ms, n = C.c_double(), C.c_uint64()
            L.pf_kernel_time(ctx, hipapi.KERNELS.index("k_cov"), C.byref(ms), C.byref(n))
            if n.value:
                probe = {"avg_ms": round(ms.value / n.value, 4),
                         "equals_streamed": bool(all(np.array_equal(a, b) for a, b in zip(res[0], res[1])))}

        # the twelve files the timed passes left, against the digests of what the REFERENCE binary wrote for this very graph
        # (profiles/reference_digests.json, made by tools/reference_digests.py: oracle/_ref/PloidyFrost -t 1 on bench.make_inputs'
        # graph of this size and seed); outside the timed region
        output_check = None
        if world > 1:
            dist.barrier()
        if rank == 0:
            output_check = check_outputs(os.path.join(workdir, "PloidyFrost_output"), "b", n_unitigs, seed, args.workload)
            log("output check: %s" % output_check)

        # one graph over several ranks: hold the shared files against a pass of rank 0 alone, once, outside the timed region
        sharded_identical = None
        if strong and not args.no_verify:
            if rank == 0:
                import hashlib
                shared = os.path.join(workdir, "PloidyFrost_output")
                names = sorted(f for f in os.listdir(shared) if f.startswith("b_"))
                def digest(d):
                    out = {}
                    for f in names:
                        h = hashlib.md5()
                        with open(os.path.join(d, f), "rb") as fh:
                            for blk in iter(lambda: fh.read(1 << 24), b""):
                                h.update(blk)
                        out[f] = h.hexdigest()
                    return out
                got = digest(shared)
                alone = os.path.join(workdir, "alone")
                run.set_output_dir(alone)
                run.set_unitig_id("b")
                run.find_superbubbles("b")
                if colored:
                    run.ploidy_estimation("b", cutoffs)
                else:
                    run.ploidy_estimation("b", LOWER, UPPER)
                want = digest(alone)
                sharded_identical = {"files": len(names), "identical": got == want,
                                     "differing": [f for f in names if got[f] != want.get(f)]}
                log("sharded run vs rank 0 alone: %s" % sharded_identical)
            dist.barrier()

        # proof for the scaling line that RCCL saw N ranks on N distinct devices: the process group's size and every rank's PCI bus id
        rccl_ranks_seen, rank_devices = (dist.get_world_size() if world > 1 and dist.is_initialized() else 1), None
        busbuf = C.create_string_buffer(32)
        L.pf_device_pci_bus_id(ctx, busbuf, 32)
        if world > 1:
            tb_ = torch.zeros(32, dtype=torch.uint8, device=xdev)
            raw_ = busbuf.value[:31]
            tb_[:len(raw_)] = torch.tensor(list(raw_), dtype=torch.uint8)
            allb_ = [torch.zeros(32, dtype=torch.uint8, device=xdev) for _ in range(world)]
            dist.all_gather(allb_, tb_)
            rank_devices = [bytes(x.cpu().tolist()).split(b"\0")[0].decode() for x in allb_]
        else:
            rank_devices = [busbuf.value.decode()]
        allstats = pfdist.all_gather_counters([n_unitigs, tt["superbubbles"], tt["tasks"], tt["output_bytes"]] + tt["allele"] +
                                              [int((elapsed_excl or 0.0) * 1e6), int(elapsed * 1e6)], xdev)
        max_elapsed = allstats[:, -1].max() / 1e6
        max_elapsed_excl = allstats[:, -2].max() / 1e6 if elapsed_excl else None
        total_unitigs = int(allstats[0, 0]) if strong else int(allstats[:, 0].sum())

        if rank == 0:
            if colored:
                tt["n_colors"] = n_samples
            value = total_unitigs * args.steps / max_elapsed
            kernels = {}
            dom, dom_ms = None, -1.0
            for name, (ms, n) in ktimes.items():
                passes = TABLE_PASSES
                if name in ktimes_live:   # (measured inside the timed region)
                    (ms, n), passes = ktimes_live[name], args.steps
                units_total = (kunits_live if name in ktimes_live else kunits).get(name)
                units = units_total / passes if units_total else None
                ab = algorithmic_bytes(name, tt, units)
                avg_ms = ms / n
                launches_per_step = n / passes
                entry = {"avg_ms": round(avg_ms, 4), "launches_per_step": launches_per_step,
                         "ms_per_step": round(ms / passes, 3), "measured_in": "timed region" if name in ktimes_live else "table passes"}
                if units is not None:
                    entry["units_per_step"] = round(units, 1)
                if ab is not None:
                    per_launch = ab / launches_per_step
                    entry["algorithmic_bytes_per_launch"] = per_launch
                    entry["achieved_GBps"] = round(per_launch / (avg_ms * 1e-3) / 1e9, 2)
                if name == "k_bubble":
                    entry["union_ms_per_step"] = round(bub_union.value / args.steps, 3)
                    entry["note"] = "launches run side by side (a launch per size class and align range): ms_per_step is their SUM, union_ms_per_step what K-BUBBLE occupies of a pass"
                kernels[name] = entry
                if ms > dom_ms and ab is not None:
                    dom, dom_ms = name, ms
            roof = None
            if dom:
                e = kernels[dom]
                roof = {"kernel": dom, "bound": "hbm", "achieved": e["achieved_GBps"], "peak": 8000.0, "unit": "GB/s",
                        "frac": round(e["achieved_GBps"] / 8000.0, 5), "traffic": load_traffic(dom, n_unitigs),
                        "traffic_source": "profiles/pmc_traffic.json (rocprofv3 --pmc passes of this command on this workload; null when the "
                                          "committed profile is of another graph)"}
            # the streaming kernel of the path, next to the dominant one: K-COV is the kernel SURVEY.md 8(d) names as able to
            # approach the HBM roof; `traffic_frac` = PMC-measured HBM bytes per launch / its duration / peak
            roof_cov = cov_roofline(kernels, colored, n_unitigs)
            # the look-up kernel itself (SURVEY.md 8d's probing form: 12.25 B per graph k-mer + 16 B per unitig) -- the farthest-from-roof
            # kernel of the account: one launch per step inside the timed region
            roof_join = None
            ej, er = kernels.get("k_cov_join"), kernels.get("k_cov_join_rest")
            if ej and "algorithmic_bytes_per_launch" in ej:
                # K-COV-JOIN is two kernels back to back on one stream: the pipelined look-up of every k-mer's first line, then the
                # look-ups it handed on (first line full); priced together on the one algorithmic figure
                join_ms = ej["avg_ms"] + (er["avg_ms"] if er else 0.0)
                ach = ej["algorithmic_bytes_per_launch"] / (join_ms * 1e-3) / 1e9
                trj, trr = load_traffic("k_cov_join", n_unitigs), load_traffic("k_cov_join_rest", n_unitigs)
                if trj and trr:
                    trj += trr
                roof_join = {"kernel": "k_cov_join + k_cov_join_rest", "bound": "hbm", "achieved": round(ach, 2), "peak": 8000.0, "unit": "GB/s",
                             "frac": round(ach / 8000.0, 5), "avg_ms": round(join_ms, 4), "avg_ms_each": [ej["avg_ms"], er["avg_ms"] if er else None],
                             "algorithmic_bytes_per_launch": ej["algorithmic_bytes_per_launch"], "traffic": trj,
                             "traffic_over_algorithmic": round(trj / ej["algorithmic_bytes_per_launch"], 2) if trj else None,
                             "traffic_frac": round(trj / (join_ms * 1e-3) / 8e12, 4) if trj else None,
                             "note": "every graph k-mer looked up in the count table (CKMCFile::CheckKmer, kmc_file.cpp:330-366): the table is "
                                     "addressed by the k-mer's minimizer, so the k-mers a wavefront's lanes hold share their 128-B lines; inside a step "
                                     "the two kernels run on their own stream beside findSuperBubble's, so avg_ms is what they take with the device shared"}

            # which of the two holds the device longer in a step
            dom_roof = roof
            if roof_join and dom:
                bub_ms = kernels[dom].get("union_ms_per_step", kernels[dom]["ms_per_step"]) if dom == "k_bubble" else kernels[dom]["ms_per_step"]
                if roof_join["avg_ms"] >= bub_ms:
                    dom_roof = dict(roof_join)
                    dom_roof["why_this_kernel"] = ("holds the device longest in a step: %.2f ms; %s's launches add up to %.2f ms but run side by side: "
                                                   "%.2f ms during which any of them is running" % (roof_join["avg_ms"], dom, kernels[dom]["ms_per_step"], bub_ms))

            def dist_ms(xs):
                xs = sorted(xs)
                if not xs:
                    return None
                q = lambda f: xs[min(len(xs) - 1, int(round(f * (len(xs) - 1))))]
                return {"min": round(xs[0] * 1e3, 3), "median": round(q(0.5) * 1e3, 3), "p95": round(q(0.95) * 1e3, 3), "max": round(xs[-1] * 1e3, 3)}
            d_incl, d_excl = dist_ms(step_s), dist_ms(step_s_excl)
            try:
                output_fs = next((ln.split()[2] for ln in reversed(open("/proc/mounts").read().splitlines())
                                  if (workdir + "/").startswith(ln.split()[1].rstrip("/") + "/")), "?")
            except (OSError, IndexError):
                output_fs = "?"
            cpu = None
            if not args.no_cpu_baseline and not colored:
                cpu = cpu_baseline(workdir, args.cpu_sample_unitigs, dev, args.cpu_runs, args.workload, n_unitigs)
            out = {
                "metric": "unitigs/s through superbubble+SeqAlign (k=%d, z=%d)" % (wl["k"], Z),
                "value": round(value, 1), "unit": "unitigs/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
                "ms_per_step": round(max_elapsed / args.steps * 1e3, 2),
                # rank 0's own steps (a step returns with its files complete)
                "ms_per_step_min": d_incl["min"], "ms_per_step_median": d_incl["median"], "ms_per_step_p95": d_incl["p95"],
                "ms_each_step": [round(x_ * 1e3, 2) for x_ in step_s],
                "timed_region": {"per_step": ("K-COV-JOIN (every graph k-mer looked up in the count table; on its own stream beside findSuperBubble) + findSuperBubble + PloidyEstimation"
                                              if join_in_step[0] else "findSuperBubble + PloidyEstimation"),
                                 "seconds": round(max_elapsed, 3),
                                 "consistency_note": ("timed region shorter than 2 s (%d steps x %.1f ms): use --steps %d or more for a region of 2 s"
                                                      % (args.steps, max_elapsed / args.steps * 1e3, int(2.0 / (max_elapsed / args.steps)) + 1))
                                 if max_elapsed < 2.0 else None},
                # the pass alone, the look-ups left to the load-time join (what rounds 1-4 reported as `value`): its own timed region
                "value_excl_join": round(total_unitigs * args.steps / max_elapsed_excl, 1) if max_elapsed_excl else None,
                "ms_per_step_excl_join": round(max_elapsed_excl / args.steps * 1e3, 2) if max_elapsed_excl else None,
                "ms_per_step_excl_join_dist": d_excl,
                "higher_is_better": True, "scaling": "weak" if (world > 1 and not strong) else "strong",
                "vs_baseline": None, "dtype": "u64", "data": "synthetic",
                "config": {"workload": (("colored graph of 3 synthetic diploid samples (CCDBG path, BASELINE.json configs[3] = 2 M unitigs), %d unitigs/GPU, k=25 z=8, "
                                         "cutoffs %d/%d per sample, M=2 D=-1 G=-3") % (n_unitigs, LOWER, UPPER) if colored else
                                        ("single-sample synthetic tetraploid graph with insertions up to 50 bp, %d unitigs (BASELINE.json configs[4] = 10 M; "
                                         "--unitigs 1000000 = the setting with committed reference digests), k=31 z=16, -l %d -u %d, M=2 D=-1 G=-3")
                                        % (n_unitigs, LOWER, UPPER) if args.workload == "stress" else
                                        ("single-sample synthetic tetraploid graph, %d unitigs (BASELINE.json configs[2] = 5 M; configs[1] = 1 M with --unitigs 1000000), k=25 z=8, "
                                         "-l %d -u %d, M=2 D=-1 G=-3") % (n_unitigs, LOWER, UPPER)) +
                                       (" -- REPEAT-RICH genome (stress workload, not the metric's config): 50 repeat families of 300-3000 bp at 1-3 %% divergence, "
                                        "inverted copies and tandem arrays on 8 %% of the genome" if args.workload == "repeats" else ""),
                           "k": wl["k"], "z": Z,
                           "unitigs_total": total_unitigs, "kmers_per_gpu": n_kmers, "host_threads_per_rank": host_threads,
                           "backend": ("gloo (PF_BENCH_SHARE_GPU: every rank on GPU 0)" if share else "nccl (RCCL)") if world > 1 else None,
                           "rccl_ranks_seen": rccl_ranks_seen, "rank_pci_bus_ids": rank_devices,
                           "distinct_devices": len(set(rank_devices)),
                           "output_fs": "%s (%s): the result files of the timed passes -- and of the reference in cpu_baseline -- are written there" % (output_fs, workdir),
                           "partitioning": ("one graph replicated on every rank, cut by entrance vertex: K-BFS records all-gathered over RCCL "
                                            "(%d bytes per pass) and replayed on every rank; bubble list in contiguous slices, two small "
                                            "all-gathers (bubbles called; slab sizes + allele histograms), every rank writes its slabs into "
                                            "the shared files" % shard_stats.get("find_gathered_bytes", 0) if strong and args.shard_find else
                                            "one graph replicated on every rank; findSuperBubble run by every rank (its device part is "
                                            "1.6 ms; --shard-find cuts it by entrance vertex with an all-gather of the records), "
                                            "PloidyEstimation cut into contiguous slices of the bubble list: two small all-gathers "
                                            "(bubbles called; slab sizes + allele histograms), every rank writes its slabs into the "
                                            "shared files" if strong else
                                            "one rank, whole graph" if world == 1 else
                                            "one independent graph per rank (weak scaling), no data-path collective; "
                                            "per-pass all-gather of the site counters only (%d bytes)" % gathered_bytes[0])},
                # `roofline` is the kernel that holds the device longest in a step.  Since the look-ups are inside the step that is
                # K-COV-JOIN (its two kernels back to back, 4.4 ms); K-BUBBLE's launches add up to more (7 ms) but run side by side,
                # two or three at a time beside other kernels: the time any of them is running is 2.8 ms (`union_ms_per_step`).
                # K-BUBBLE keeps its own objects: `roofline_k_bubble` (HBM, the wrong roof for it) and `roofline_issue` (VALU issue).
                "roofline": dom_roof, "roofline_k_bubble": roof if dom_roof is not roof else None,
                "roofline_issue": issue_roofline(dom, kernels[dom]["avg_ms"], n_unitigs, kernels[dom]) if dom else None,
                "roofline_k_cov": roof_cov, "roofline_k_cov_join": roof_join, "cpu_baseline": cpu,
                "gpu_kernel_ms_per_step": round(sum(e["ms_per_step"] for n_, e in kernels.items() if n_ != "copy_text_to_host"), 3),
                # union of the launches' [start, end] intervals (HIP events on their streams) over the timed passes / passes;
                # tools/summarize_prof.py gives the same figure from the rocprofv3 kernel trace (profiles/*_device_busy.json)
                # (of the table passes, where every launch carries events; their own wall time is the denominator)
                "device_busy_ms_per_step": round(busy_ms.value / TABLE_PASSES, 3),
                "device_busy_frac": round(busy_ms.value / (table_elapsed * 1e3), 4),
                "device_busy_frac_own_kernels_only": round(busy_own.value / (table_elapsed * 1e3), 4),
                "kernel_table": {"passes": TABLE_PASSES, "ms_per_pass_with_every_launch_timed": round(table_elapsed / TABLE_PASSES * 1e3, 2),
                                 "note": "inside the timed region only the roofline kernels' launches carry HIP events (" + ", ".join(sorted(ktimes_live)) +
                                         "); the other rows of `kernels`, gpu_kernel_ms_per_step and device_busy_* are from these further passes"},
                "first_pass": first_pass, "load_s": round(load_s, 3), "load_breakdown": load_breakdown,
                "output_check": output_check, "sharded_output_check": sharded_identical,
                "kernels": kernels, "k_cov_probe": probe,
                "host_phases_s_per_step": {k: round(v / args.steps, 4) for k, v in phase.items()},
                # rank 0's graph; for one graph over all ranks (strong) the per-slice counters are added up
                # K-BFS tiers: candidates that outgrew the 128-entry LDS tier were walked on host cores
                "bfs": {"candidates": tt["candidates"], "bfs_deferred": tt["bfs_deferred"],
                        "deferred_frac": round(tt["bfs_deferred"] / max(tt["candidates"], 1), 8),
                        "traversals_beyond_4096": tt["bfs_large"], "longest_traversal": tt["bfs_max_seen"],
                        # findSuperBubble's host share: vertices the host walkers visited for the deferred traversals; records committed on a
                        # host thread (the large components of K-CC and the walked traversals)
                        "host_walk_vertices": tt["host_walk_vertices"], "host_commit_records": tt["host_commit_records"]},
                "counts": {"candidates": tt["candidates"], "superbubble_rows": tt["superbubbles"],
                           "bubbles_called": shard_stats["counters"][7] if strong else tt["tasks"],
                           "align_jobs": tt["align_jobs"], "site_strings": tt["site_strings"],
                           "align_jobs_by_kernel": {"k_call_snp": tt["snp_jobs"], "k_call_pair": tt["pair_jobs"], "k_call_stack": tt["stack_jobs"], "k_bubble": tt["wave_jobs"]},
                           "sites": shard_stats["counters"][:4] if strong else tt["allele"],
                           "output_bytes": shard_stats["output_bytes"] if strong else tt["output_bytes"]},
            }
            # the model the first hardware scaling curve can be checked against: every rank repeats the look-ups, findSuperBubble and the
            # owner scan; only the rest of PloidyEstimation (alignment, text, files) is cut over the ranks; two small all-gathers
            hp_ = {k_: v_ / args.steps for k_, v_ in phase.items()}
            step_ms = max_elapsed / args.steps * 1e3
            if strong:
                # a cut run: rank 0's findSuperBubble (the look-ups run beside it) and owner scan are what every rank repeated; the
                # rest of the step is this run's slice of the cut part
                rep_ms = min(step_ms, (hp_["find_total_s"] + hp_["scan_s"]) * 1e3)
                cut_ms = (step_ms - rep_ms) * world
            else:
                cut_ms = max(0.0, (hp_["ploidy_total_s"] - hp_["scan_s"]) * 1e3)
                rep_ms = max(0.0, step_ms - cut_ms)
            out["predicted_scaling"] = {
                "model": "ms(N) = replicated + cut / N; replicated = K-COV-JOIN + findSuperBubble + owner scan of this run, cut = the rest of PloidyEstimation "
                         "(alignment, text, files); the two all-gathers (< 0.1 ms over xGMI) left out",
                "replicated_ms": round(rep_ms, 2), "cut_ms_at_1_gpu": round(cut_ms, 2),
                "ms": {str(n_): round(rep_ms + cut_ms / n_, 2) for n_ in (1, 2, 4, 8)},
                "speedup": {str(n_): round((rep_ms + cut_ms) / (rep_ms + cut_ms / n_), 2) for n_ in (1, 2, 4, 8)},
                "ceiling": round((rep_ms + cut_ms) / rep_ms, 2) if rep_ms > 0 else None}
            if cpu:
                out["speedup_vs_cpu_1core"] = round(value / cpu["value"], 2)
                # the north-star bar: >= 10x the reference's single-threaded throughput at 1 GPU
                out["north_star"] = {"target": ">= 10x reference -t 1 on the 5 M-unitig k=25 graph at 1 GPU",
                                     "ratio_vs_sample": out["speedup_vs_cpu_1core"],
                                     "ratio_vs_reference_at_config_size": round(value / cpu["reference_at_config_size"]["unitigs_per_s"], 1),
                                     "met": bool(value >= 10 * cpu["value"])}
            print(json.dumps(out), flush=True)
            if output_check and output_check.get("identical_to_reference") is False:
                log("OUTPUT DIFFERS FROM THE REFERENCE: %s" % output_check["differing"])
                bad_output = True
        run.close()
        ok = True
    finally:
        if world > 1:
            if ok:  # a rank that raised must not leave the others waiting in a barrier: they fail in their next collective instead
                dist.barrier()
            dist.destroy_process_group()
        if not args.keep:
            shutil.rmtree(workdir, ignore_errors=True)
    if bad_output:
        sys.exit(3)


def reference_digests():
    p = os.path.join(ROOT, "profiles", "reference_digests.json")
    try:
        with open(p) as f:
            return json.load(f)
    except (OSError, ValueError):
        return {}


def digest_key(workload: str, n_unitigs: int) -> str:
    """profiles/reference_digests.json: the single-sample graphs under their unitig count (as rounds 1-4 wrote them), the others
    under '<workload>:<unitigs>'"""
    return str(n_unitigs) if workload == "single" else "%s:%d" % (workload, n_unitigs)


def check_outputs(outdir: str, prefix: str, n_unitigs: int, seed: int, workload: str = "single"):
    """md5 of the twelve files in `outdir` against the committed digests of the reference's output for this graph.  None of the
    digests' making is part of this run: `digest_source` says where they come from."""
    import hashlib
    ent = reference_digests().get(digest_key(workload, n_unitigs))
    if not ent or ent.get("seed") != seed or "files" not in ent:
        return {"files": 0, "identical_to_reference": None,
                "note": "no committed reference digests for a graph of %d unitigs, seed %d (tools/reference_digests.py makes them)" % (n_unitigs, seed)}
    differing = []
    for name, want in sorted(ent["files"].items()):
        h = hashlib.md5()
        try:
            with open(os.path.join(outdir, "%s_%s" % (prefix, name)), "rb") as fh:
                for blk in iter(lambda: fh.read(1 << 24), b""):
                    h.update(blk)
        except OSError:
            differing.append(name + " (missing)")
            continue
        if h.hexdigest() != want["md5"]:
            differing.append(name)
    return {"files": len(ent["files"]), "identical_to_reference": not differing, "differing": differing,
            "digest_source": "profiles/reference_digests.json: md5 of what oracle/_ref/PloidyFrost -t 1 wrote for this graph (%d unitigs, seed %d; "
                             "run on %s, %s, by tools/reference_digests.py)" % (n_unitigs, seed, ent.get("host", "?"), ent.get("date", "?"))}


def issue_roofline(kernel: str, avg_ms: float, n_unitigs: int, entry: dict | None = None):
    """Instruction-issue roofline of a kernel that is bound by issue, not by HBM (K-BUBBLE: its DP lives in registers and LDS):
    vector instructions per launch (SQ_INSTS_VALU, rocprofv3 --pmc pass of this command on this workload, committed as
    profiles/pmc_sq.json) / the launch's measured duration, against what the chip can issue: 256 CUs x 4 SIMDs x 2.4 GHz / 2
    cycles per wave64 vector instruction on a SIMD-32 (MI355X_MICROARCH.md, 'Wave scheduling' and the constants row
    `v_fma_f32 (wave64) 2 cyc (SIMD-32)`: the figure for a SIMD that has more than one wave to issue from, which is K-BUBBLE's and
    K-PAIR's case; a lone wave sustains one instruction per 4 cycles) = 1 229 G wave-instructions/s.
    None when the committed counters are of another workload."""
    p = os.path.join(ROOT, "profiles", "pmc_sq.json")
    try:
        with open(p) as f:
            d = json.load(f)
    except (OSError, ValueError):
        return None
    if (d.get("_meta") or {}).get("unitigs") != n_unitigs or kernel not in d or not avg_ms:
        return None
    e = d[kernel]
    valu, salu = e.get("SQ_INSTS_VALU"), e.get("SQ_INSTS_SALU")
    if not valu:
        return None
    peak = 256 * 4 * 2.4e9 / 2
    ach = valu / (avg_ms * 1e-3)
    over_union = None
    if entry and entry.get("union_ms_per_step") and entry.get("launches_per_step"):
        # all launches of a pass together, over the time any of them is running
        over_union = round(valu * entry["launches_per_step"] / (entry["union_ms_per_step"] * 1e-3) / peak, 4)
    return {"kernel": kernel, "bound": "valu-issue", "achieved": round(ach / 1e9, 2), "peak": round(peak / 1e9, 1), "unit": "G wave-instructions/s",
            "frac": round(ach / peak, 4), "frac_all_launches_over_their_union": over_union, "valu_per_launch": valu, "salu_per_launch": salu,
            "wave_cycle_shares": {k[6:]: v for k, v in e.items() if k.startswith("share_")},
            "source": "profiles/pmc_sq.json (SQ counters per launch) / live HIP-event duration",
            "note": "per launch: the launches of K-BUBBLE's size classes run side by side (five populated classes in each of two align "
                    "ranges that are on the device at once, beside K-TEXT), so a launch gets a share of the chip -- the chip issues several "
                    "times this, and splitting the work over more launches lowers the figure without slowing anything"}


def load_traffic(kernel: str, n_unitigs: int | None = None):
    """HBM bytes per launch of `kernel` from the committed PMC profile (tools/pmc_traffic.py; rocprofv3 --pmc FETCH_SIZE /
    WRITE_SIZE in separate passes) -- only when that profile was taken on this very workload (same unitig count), else None:
    a per-launch byte count of another graph would look like a measurement of this run."""
    p = os.path.join(ROOT, "profiles", "pmc_traffic.json")
    try:
        with open(p) as f:
            d = json.load(f)
    except (OSError, ValueError):
        return None
    meta = d.get("_meta") or {}
    if n_unitigs is not None and meta.get("unitigs") != n_unitigs:
        return None
    return d.get(kernel)


if __name__ == "__main__":
    main()
