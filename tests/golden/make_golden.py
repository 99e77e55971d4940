#!/usr/bin/env python3
"""Regenerate the golden fixtures under tests/golden/<case>/.

Runs ONLY in the build container: it needs the real reference binaries built by
``make -f oracle/Makefile.ref`` (oracle/_ref/PloidyFrost and oracle/_ref/Bifrost).  For every
case it writes
    graph.gfa            Bifrost 1.0.6 ``build -r`` output for seeded synthetic haplotypes
    db.kmc_pre/.kmc_suf  KMC1-layout count database (ploidyfrost_amd.synth.write_kmc1)
    args.json            the PloidyFrost options of the case
    expected/<prefix>_*  the twelve files the reference wrote with ``-t 1``
The fixtures are data (inputs + expected outputs); no reference source is stored.

Colored (multi-sample, reference src/CCDBG.cpp) cases additionally hold
    graph.bfg_colors     written by the reference's ``Bifrost build -c`` (one colour per sample FASTA)
    colors.txt           oracle/_ref/colors_dump: how the real Bifrost reads that file (k-mer x colour presence,
                         UnitigColors::size) -- data that pins the oracle's and the product's colour semantics
    db<i>.kmc_pre/.suf   one count database per colour
and args.json carries the per-colour cutoffs (the -C file) instead of -l/-u.

usage: python tests/golden/make_golden.py [case ...]
"""
from __future__ import annotations

import json
import os
import shutil
import subprocess
import sys
import tempfile

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
from ploidyfrost_amd import synth  # noqa: E402

REF = os.path.join(ROOT, "oracle", "_ref")


def weird_haplotypes(seed: int, n: int = 12000) -> list[np.ndarray]:
    """Two haplotypes with tandem repeats (cycles), inverted repeats (hairpins), shared
    repeats at distant loci, and haplotype-private ends (tips)."""
    rng = np.random.default_rng(seed)
    base = rng.integers(0, 4, size=n, dtype=np.uint8)
    parts = []
    pos = 0
    while pos < n:
        step = int(rng.integers(200, 600))
        seg = base[pos : pos + step]
        parts.append(seg)
        kind = int(rng.integers(0, 5))
        if len(seg) > 120:
            a = int(rng.integers(0, len(seg) - 100))
            ln = int(rng.integers(30, 100))
            piece = seg[a : a + ln]
            if kind == 0:      # tandem duplication
                parts.append(piece)
                parts.append(piece)
            elif kind == 1:    # inverted repeat
                parts.append((3 - piece)[::-1])
            elif kind == 2:    # hairpin right at the junction
                parts.append((3 - seg[-ln:])[::-1])
            elif kind == 3:    # distant copy
                parts.append(base[a : a + ln])
        pos += step
    h0 = np.concatenate(parts)
    spec = synth.HapSpec(genome_len=len(h0), ploidy=2, seed=seed + 1, gap_lo=10, gap_hi=200)
    # reuse the variant machinery on the repeat-rich genome
    rng2 = np.random.default_rng(spec.seed)
    h1 = h0.copy()
    sites = np.cumsum(rng2.integers(spec.gap_lo, spec.gap_hi, size=len(h0) // spec.gap_lo))
    sites = sites[(sites > 40) & (sites < len(h0) - 40)]
    h1[sites] = (h1[sites] + rng2.integers(1, 4, size=len(sites)).astype(np.uint8)) & 3
    # private ends -> tips
    h1 = np.concatenate([rng2.integers(0, 4, size=60, dtype=np.uint8), h1[30:-45], rng2.integers(0, 4, size=80, dtype=np.uint8)])
    return [h0, h1]


def reads_case(seed=61, genome_len=20000, n_reads=10000, read_len=100, err=0.0006):
    """BASELINE.json configs[0]: a diploid genome sampled by 10 k error-bearing reads -- the graph Bifrost builds from
    *reads* with every k-mer kept (`build -r`) has the tips, error bubbles inside real bubbles and count-1 k-mers that
    haplotype-built graphs lack.  Returns the reads; counts come from the reads themselves."""
    haps = synth.make_haplotypes(synth.HapSpec(genome_len, 2, seed=seed, gap_lo=40, gap_hi=400))
    rng = np.random.default_rng(seed + 1)
    reads = []
    for _ in range(n_reads):
        h = haps[int(rng.integers(0, 2))]
        a = int(rng.integers(0, len(h) - read_len + 1))
        r = h[a: a + read_len].copy()
        hit = rng.random(read_len) < err
        r[hit] = (r[hit] + rng.integers(1, 4, size=int(hit.sum())).astype(np.uint8)) & 3
        if rng.random() < 0.5:
            r = (3 - r)[::-1]
        reads.append(r)
    return reads


CASES = {
    # name: (haplotype factory, k, PloidyFrost args)
    "dip20k": (lambda: synth.make_haplotypes(synth.HapSpec(20000, 2, seed=7)), 25, ["-l", "5", "-u", "1000"]),
    "tet60k": (lambda: synth.make_haplotypes(synth.HapSpec(60000, 4, seed=11, gap_lo=15, gap_hi=300, p_multi=0.08)), 25,
               ["-l", "5", "-u", "1000"]),
    "hex30k": (lambda: synth.make_haplotypes(synth.HapSpec(30000, 6, seed=3, gap_lo=5, gap_hi=60, p_multi=0.1)), 25,
               ["-l", "5", "-u", "1000"]),
    "tri_z5": (lambda: synth.make_haplotypes(synth.HapSpec(30000, 3, seed=9, gap_lo=8, gap_hi=120)), 25,
               ["-l", "5", "-u", "1000", "-z", "5"]),
    "tet_frac": (lambda: synth.make_haplotypes(synth.HapSpec(40000, 4, seed=13, gap_lo=10, gap_hi=200, p_multi=0.08)), 25,
                 ["-l", "5", "-u", "1000", "-M", "1.5", "-D", "-0.5", "-G", "-2.25"]),
    "k31_z16": (lambda: synth.make_haplotypes(synth.HapSpec(30000, 4, seed=17, gap_lo=10, gap_hi=150, max_ins=40, p_snp=0.5,
                                                            p_del=0.1)), 31, ["-l", "5", "-u", "1000", "-z", "16"]),
    "weird12k": (lambda: weird_haplotypes(23), 25, ["-l", "5", "-u", "1000"]),
    "cutoff": (lambda: synth.make_haplotypes(synth.HapSpec(30000, 4, seed=29, gap_lo=10, gap_hi=200)), 25,
               ["-l", "25", "-u", "70"]),
    # the same kind of data with the count database in the KMC2 layout (signature-binned prefix table)
    # graph built from reads, counts = k-mer multiplicities in the reads
    "reads10k": (reads_case, 25, ["-l", "3", "-u", "1000"], "kmc1", "reads"),
    "dip_kmc2": (lambda: synth.make_haplotypes(synth.HapSpec(16000, 2, seed=31)), 25, ["-l", "5", "-u", "1000"], "kmc2"),
    # a 17-mer with a very low minimizer hash planted at 100 places: its bucket in Bifrost's minimizer index is crowded, the
    # k-length unitigs read after the 15th entry become "abundant" k-mers and are numbered last, in hash-table order
    # a 32-bp segment copied from 4 % to 8 % of the genome: the shared unitig has both loci's continuations as successors, each
    # with no other predecessor, so the traversal entering there walks one locus to the end of the chromosome (a tip) before it
    # can come back to the other -- more than 4096 vertices: the third K-BFS tier
    "giant7k": (lambda: synth.make_haplotypes(synth.HapSpec(270000, 4, seed=5, gap_lo=15, gap_hi=300),
                                              lambda b: np.concatenate([b[:21600], b[10800:10832], b[21632:]])), 25,
                ["-l", "5", "-u", "1000"]),
    # the count database written without canonical counting (kmc -b): coverage per orientation, no site-string coverage
    "stranded20k": (lambda: synth.make_haplotypes(synth.HapSpec(24000, 3, seed=43, gap_lo=10, gap_hi=200, p_multi=0.05)), 25,
                    ["-l", "2", "-u", "1000"], "kmc1_stranded"),
    "crowd25": (lambda: synth.make_haplotypes(synth.HapSpec(24000, 4, seed=4, gap_lo=8, gap_hi=150, p_multi=0.05),
                                              lambda b: synth.plant_crowded_minimizer(np.random.default_rng(41), b, 17, 100)), 25,
                ["-l", "5", "-u", "1000"], "kmc1", "haps", "abundant"),
}


def trimmed(haps, cuts):
    """Drop cuts[i] = (head, tail) bases from haplotype i: sequence ends inside shared unitigs -> partial colours."""
    return [h[a: len(h) - b] for h, (a, b) in zip(haps, cuts)]


def col3_dip():
    haps = synth.make_haplotypes(synth.HapSpec(30000, 6, seed=41, gap_lo=15, gap_hi=300, p_multi=0.05))
    rng = np.random.default_rng(41007)
    samples = []
    for s in range(3):
        hs = haps[2 * s: 2 * s + 2]
        if s > 0:   # sample-private sequence: unitigs that do not carry every colour
            hs = [np.concatenate([h, rng.integers(0, 4, size=400, dtype=np.uint8)]) for h in hs]
        samples.append(hs)
    return samples


def col4_mix():
    """Four samples of ploidy 4,2,2,4 on one base genome, longer insertions (branching bubbles), multi-allelic
    sites, and haplotypes that start/stop at different places (k-mer ranges of unitigs without a colour)."""
    haps = synth.make_haplotypes(synth.HapSpec(36000, 12, seed=53, gap_lo=10, gap_hi=150, p_multi=0.1, max_ins=12, p_snp=0.6,
                                               p_del=0.15))
    groups = [haps[0:4], haps[4:6], haps[6:8], haps[8:12]]
    groups[1] = trimmed(groups[1], [(1234, 0), (1234, 977)])
    groups[2] = trimmed(groups[2], [(0, 2100), (311, 2100)])
    groups[3] = trimmed(groups[3], [(0, 0), (40, 0), (0, 55), (5000, 7000)])
    return groups


def col2_weird():
    """Two samples over the repeat-rich genome (cycles, hairpins, tips) of weird_haplotypes."""
    h0, h1 = weird_haplotypes(23)
    rng = np.random.default_rng(77)
    g0 = h0.copy()
    sites = np.cumsum(rng.integers(20, 260, size=len(g0) // 20))
    sites = sites[(sites > 40) & (sites < len(g0) - 40)]
    g0[sites] = (g0[sites] + rng.integers(1, 4, size=len(sites)).astype(np.uint8)) & 3
    return [[h0, h1], [g0, h1[200:-150]]]


def col100():
    """One hundred diploid samples (the reference has no colour limit: src/CCDBG.cpp:2759-2853 loops over getNbColors(); the device's
    resident pipeline keeps a colour set in 64 bits and hands such graphs to the host-threaded one): sample i carries haplotypes i and
    i + 1 (mod 10) of a ten-haplotype genome, loses a few bases at its ends (colours on part of a unitig), and the last ten samples
    have a private tail (unitigs that lack most colours)."""
    haps = synth.make_haplotypes(synth.HapSpec(1300, 10, seed=61, gap_lo=18, gap_hi=110, p_multi=0.05, max_ins=5))
    rng = np.random.default_rng(6100)
    samples = []
    for i in range(100):
        hs = [haps[i % 10], haps[(i + 1) % 10]]
        hs = [h[(3 * i + 5 * j) % 37: len(h) - ((7 * i + 3 * j) % 41)] for j, h in enumerate(hs)]
        if i >= 90:
            hs = [np.concatenate([h, rng.integers(0, 4, size=90, dtype=np.uint8)]) for h in hs]
        samples.append(hs)
    return samples


def col_n(n: int, seed: int):
    """n diploid samples over one ten-haplotype genome, made like col100's: the widths of a colour set around the 64-bit words of
    the device's colour sets -- 64 (one full word), 65 (a second word of one bit), 130 (three words)"""
    def make():
        haps = synth.make_haplotypes(synth.HapSpec(1100, 10, seed=seed, gap_lo=18, gap_hi=100, p_multi=0.06, max_ins=5))
        rng = np.random.default_rng(seed * 100 + n)
        samples = []
        for i in range(n):
            hs = [haps[i % 10], haps[(i + 3) % 10]]
            hs = [h[(5 * i + 3 * j) % 31: len(h) - ((3 * i + 7 * j) % 43)] for j, h in enumerate(hs)]
            if i >= n - 6:   # the last samples' private tails: unitigs that lack most colours, their own colours in the last word
                hs = [np.concatenate([h, rng.integers(0, 4, size=80, dtype=np.uint8)]) for h in hs]
            samples.append(hs)
        return samples
    return make


# name: (sample factory -> list of per-sample haplotype lists, k, PloidyFrost args, per-colour cutoffs)
COLORED_CASES = {
    "col100": (col100, 25, ["-z", "10"], [(5, 1000)] * 100, set(), 1),
    "col64": (col_n(64, 64), 25, ["-z", "10"], [(5, 1000)] * 64, set(), 1),
    "col65": (col_n(65, 65), 25, ["-z", "10"], [(5, 1000)] * 65, set(), 1),
    "col130": (col_n(130, 67), 25, ["-z", "10"], [(5, 1000)] * 130, set(), 1),
    "col3_dip": (col3_dip, 25, [], [(5, 1000)] * 3),
    "col4_mix": (col4_mix, 25, ["-z", "10"], [(5, 1000), (5, 1000), (25, 70), (5, 1000)]),
    "col2_weird": (col2_weird, 31, ["-M", "1.5", "-D", "-0.5", "-G", "-2.25"], [(5, 1000), (10, 400)]),
    # the second sample's database written without canonical counting: that colour is never looked up and counts as coverage 0
    "col3_stranded": (col3_dip, 25, [], [(5, 1000)] * 3, {1}),
}


def run(cmd, **kw):
    r = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, **kw)
    if r.returncode != 0:
        raise RuntimeError("%s failed:\n%s" % (" ".join(cmd), r.stdout))
    return r.stdout


def make_case(name: str) -> None:
    factory, k, args = CASES[name][:3]
    layout = CASES[name][3] if len(CASES[name]) > 3 else "kmc1"
    from_reads = len(CASES[name]) > 4 and CASES[name][4] == "reads"
    out = os.path.join(HERE, name)
    shutil.rmtree(out, ignore_errors=True)
    os.makedirs(os.path.join(out, "expected"))
    haps = factory()
    with tempfile.TemporaryDirectory() as tmp:
        fa = os.path.join(tmp, "haps.fa")
        synth.write_fasta(fa, haps)
        run([os.path.join(REF, "Bifrost"), "build", "-r", fa, "-k", str(k), "-o", os.path.join(tmp, "graph"),
             "-t", "1"])
        shutil.copy(os.path.join(tmp, "graph.gfa"), os.path.join(out, "graph.gfa"))
        km, mult = synth.canonical_counts(haps, k)
        cnt = mult.astype(np.uint32) if from_reads else synth.synth_counts(km, mult)
        if layout == "kmc1_stranded":
            skm, scnt = synth.stranded_counts(km, mult, k)
            synth.write_kmc1(os.path.join(out, "db"), skm, scnt, k, both_strands=False)
        elif layout == "kmc2":
            synth.write_kmc2(os.path.join(out, "db"), km, cnt, k, sig_len=7, n_bins=11)
        else:
            synth.write_kmc1(os.path.join(out, "db"), km, cnt, k)
        log = run([os.path.join(REF, "PloidyFrost"), "-g", os.path.join(out, "graph.gfa"), "-d", os.path.join(out, "db"),
                   "-o", "g", "-t", "1"] + args, cwd=tmp)
        for f in sorted(os.listdir(os.path.join(tmp, "PloidyFrost_output"))):
            shutil.copy(os.path.join(tmp, "PloidyFrost_output", f), os.path.join(out, "expected", f))
    n_unitigs = sum(1 for line in open(os.path.join(out, "graph.gfa")) if line.startswith("S\t"))
    summary = [l for l in log.splitlines() if "SuperBubbles Found" in l or "Alleles in" in l]
    with open(os.path.join(out, "args.json"), "w") as f:
        meta = {"k": k, "args": args, "prefix": "g", "unitigs": n_unitigs, "kmc_layout": layout, "reference_log": summary}
        if len(CASES[name]) > 5 and CASES[name][5] == "abundant":
            # the oracle's loader does not restate Bifrost's abundant-k-mer bookkeeping: this case pins the product to the reference directly
            meta["abundant"] = True
        json.dump(meta, f, indent=1)
    print(name, n_unitigs, "unitigs;", " | ".join(s.strip() for s in summary))


def make_colored_case(name: str) -> None:
    factory, k, args, cutoffs = COLORED_CASES[name][:4]
    stranded = COLORED_CASES[name][4] if len(COLORED_CASES[name]) > 4 else set()
    lut_p = COLORED_CASES[name][5] if len(COLORED_CASES[name]) > 5 else None   # (a short prefix table keeps a hundred databases small)
    out = os.path.join(HERE, name)
    shutil.rmtree(out, ignore_errors=True)
    os.makedirs(os.path.join(out, "expected"))
    samples = factory()
    with tempfile.TemporaryDirectory() as tmp:
        fas = []
        for i, hs in enumerate(samples):
            fa = os.path.join(tmp, "sample%d.fa" % i)
            synth.write_fasta(fa, hs)
            fas.append(fa)
            km, mult = synth.canonical_counts(hs, k)
            if i in stranded:
                skm, scnt = synth.stranded_counts(km, mult, k)
                synth.write_kmc1(os.path.join(out, "db%d" % i), skm, scnt, k, both_strands=False, p=lut_p)
            else:
                synth.write_kmc1(os.path.join(out, "db%d" % i), km, synth.synth_counts(km, mult), k, p=lut_p)
        with open(os.path.join(tmp, "refs.txt"), "w") as f:
            f.write("".join(p + "\n" for p in fas))
        run([os.path.join(REF, "Bifrost"), "build", "-c", "-r", os.path.join(tmp, "refs.txt"), "-k", str(k), "-o",
             os.path.join(tmp, "graph"), "-t", "1"])
        for ext in ("gfa", "bfg_colors"):
            shutil.copy(os.path.join(tmp, "graph." + ext), os.path.join(out, "graph." + ext))
        dump = run([os.path.join(REF, "colors_dump"), os.path.join(out, "graph.gfa"), os.path.join(out, "graph.bfg_colors")])
        # colour names are the temporary FASTA paths: keep the basename only
        dump = "".join(("#name\t" + os.path.basename(l.split("\t")[1]) if l.startswith("#name") else l) + "\n" for l in dump.splitlines())
        with open(os.path.join(out, "colors.txt"), "w") as f:
            f.write(dump)
        with open(os.path.join(tmp, "dbs.txt"), "w") as f:
            f.write("".join(os.path.join(out, "db%d" % i) + "\n" for i in range(len(samples))))
        with open(os.path.join(tmp, "cutoffs.txt"), "w") as f:
            f.write("".join("%d\t%d\n" % c for c in cutoffs))
        log = run([os.path.join(REF, "PloidyFrost"), "-g", os.path.join(out, "graph.gfa"), "-f", os.path.join(out, "graph.bfg_colors"),
                   "-d", os.path.join(tmp, "dbs.txt"), "-C", os.path.join(tmp, "cutoffs.txt"), "-o", "g", "-t", "1"] + args, cwd=tmp)
        for f in sorted(os.listdir(os.path.join(tmp, "PloidyFrost_output"))):
            shutil.copy(os.path.join(tmp, "PloidyFrost_output", f), os.path.join(out, "expected", f))
    n_unitigs = sum(1 for line in open(os.path.join(out, "graph.gfa")) if line.startswith("S\t"))
    summary = [l for l in log.splitlines() if "SuperBubbles Found" in l or "Alleles in" in l]
    with open(os.path.join(out, "args.json"), "w") as f:
        json.dump({"k": k, "args": args, "prefix": "g", "unitigs": n_unitigs, "colored": True, "n_colors": len(samples),
                   "cutoffs": [list(c) for c in cutoffs], "kmc_layout": "kmc1", "reference_log": summary}, f, indent=1)
    print(name, n_unitigs, "unitigs;", " | ".join(s.strip() for s in summary))


if __name__ == "__main__":
    for c in (sys.argv[1:] or list(CASES) + list(COLORED_CASES)):
        make_colored_case(c) if c in COLORED_CASES else make_case(c)
