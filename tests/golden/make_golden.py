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
    "dip_kmc2": (lambda: synth.make_haplotypes(synth.HapSpec(16000, 2, seed=31)), 25, ["-l", "5", "-u", "1000"], "kmc2"),
}


def run(cmd, **kw):
    r = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, **kw)
    if r.returncode != 0:
        raise RuntimeError("%s failed:\n%s" % (" ".join(cmd), r.stdout))
    return r.stdout


def make_case(name: str) -> None:
    factory, k, args = CASES[name][:3]
    layout = CASES[name][3] if len(CASES[name]) > 3 else "kmc1"
    out = os.path.join(HERE, name)
    shutil.rmtree(out, ignore_errors=True)
    os.makedirs(os.path.join(out, "expected"))
    haps = factory()
    with tempfile.TemporaryDirectory() as tmp:
        fa = os.path.join(tmp, "haps.fa")
        synth.write_fasta(fa, haps)
        run([os.path.join(REF, "Bifrost"), "build", "-r", fa, "-k", str(k), "-o", os.path.join(tmp, "graph"), "-t", "1"])
        shutil.copy(os.path.join(tmp, "graph.gfa"), os.path.join(out, "graph.gfa"))
        km, mult = synth.canonical_counts(haps, k)
        cnt = synth.synth_counts(km, mult)
        if layout == "kmc2":
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
        json.dump({"k": k, "args": args, "prefix": "g", "unitigs": n_unitigs, "kmc_layout": layout, "reference_log": summary}, f, indent=1)
    print(name, n_unitigs, "unitigs;", " | ".join(s.strip() for s in summary))


if __name__ == "__main__":
    for c in (sys.argv[1:] or list(CASES)):
        make_case(c)
