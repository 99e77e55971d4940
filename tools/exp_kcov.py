#!/usr/bin/env python3
"""K-COV timing on a bench-style graph (experiments): tools/exp_kcov.py [unitigs]"""
import os, sys, tempfile, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import bench
from ploidyfrost_amd import hipapi, synth
n = int(sys.argv[1]) if len(sys.argv) > 1 else 300000
with tempfile.TemporaryDirectory() as tmp:
    gfa, db, nu, nk = bench.make_inputs(tmp, "g", int(n / bench.UNITIGS_PER_BP), 5, "cuda")
    seqs = [l.split(b"\t")[2].strip() for l in open(gfa, "rb") if l.startswith(b"S")]
    dev = hipapi.Device(0)
    dev.upload_graph(*hipapi.pack_unitigs(seqs), 25)
    km, cnt, meta = synth.read_kmc(db)
    dev.upload_counts(km, cnt, 1, 65535, True)
    dev.enable_timing(True)
    for _ in range(3):
        dev.unitig_cov()
    dev.reset_timing()
    for _ in range(10):
        s, m, x, st = dev.unitig_cov()
    t = dev.kernel_times()
    print("unitigs", nu, "kmers", nk, "k_cov avg ms", t["k_cov"][0] / t["k_cov"][1], "sum", int(s.sum()), "miss", int(x.sum()))
