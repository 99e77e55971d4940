#!/usr/bin/env python3
"""Manufacture a colored (multi-sample) input set: one FASTA, one KMC database and one cutoff line per
sample, the list files the reference CLI takes (-d <db list>, -C <cutoffs>), and -- when the reference's
``Bifrost`` binary is available under oracle/_ref -- the colored graph (``graph.gfa`` + ``graph.bfg_colors``).

usage: python tools/make_colored.py OUTDIR [--len N] [--samples S] [--ploidy P] [--seed X] [--k K]
"""
from __future__ import annotations

import argparse
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from ploidyfrost_amd import synth  # noqa: E402


def make(outdir: str, genome_len: int, samples: int, ploidy: int, seed: int, k: int, lower: int = 5, upper: int = 1000,
         private_tail: int = 0, **spec_kw) -> dict:
    os.makedirs(outdir, exist_ok=True)
    outdir = os.path.abspath(outdir)
    haps = synth.make_haplotypes(synth.HapSpec(genome_len, samples * ploidy, seed=seed, **spec_kw))
    fas, dbs = [], []
    for s in range(samples):
        hs = haps[ploidy * s: ploidy * (s + 1)]
        if private_tail and s > 0:
            # sample-private sequence: unitigs that do not carry every colour
            import numpy as np
            rng = np.random.default_rng(seed * 1000 + s)
            hs = [np.concatenate([h, rng.integers(0, 4, size=private_tail, dtype=np.uint8)]) for h in hs]
        fa = os.path.join(outdir, "s%d.fa" % s)
        synth.write_fasta(fa, hs)
        km, mult = synth.canonical_counts(hs, k)
        cnt = synth.synth_counts(km, mult)
        db = os.path.join(outdir, "db%d" % s)
        synth.write_kmc1(db, km, cnt, k)
        fas.append(fa)
        dbs.append(db)
    with open(os.path.join(outdir, "refs.txt"), "w") as f:
        f.write("".join(p + "\n" for p in fas))
    with open(os.path.join(outdir, "dbs.txt"), "w") as f:
        f.write("".join(p + "\n" for p in dbs))
    with open(os.path.join(outdir, "cutoffs.txt"), "w") as f:
        f.write(("%d\t%d\n" % (lower, upper)) * samples)
    bifrost = os.path.join(ROOT, "oracle", "_ref", "Bifrost")
    if os.path.exists(bifrost):
        r = subprocess.run([bifrost, "build", "-c", "-r", os.path.join(outdir, "refs.txt"), "-k", str(k), "-o",
                            os.path.join(outdir, "graph"), "-t", "1"], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
        if r.returncode != 0:
            raise RuntimeError(r.stdout)
    return {"fastas": fas, "dbs": dbs, "outdir": outdir}


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("outdir")
    ap.add_argument("--len", type=int, default=30000)
    ap.add_argument("--samples", type=int, default=3)
    ap.add_argument("--ploidy", type=int, default=2)
    ap.add_argument("--seed", type=int, default=41)
    ap.add_argument("--k", type=int, default=25)
    ap.add_argument("--private-tail", type=int, default=0)
    a = ap.parse_args()
    print(make(a.outdir, a.len, a.samples, a.ploidy, a.seed, a.k, private_tail=a.private_tail, gap_lo=15, gap_hi=300, p_multi=0.05))
