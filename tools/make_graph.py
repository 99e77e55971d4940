#!/usr/bin/env python3
"""Generate a bench-style synthetic graph + KMC database into a directory.
usage: tools/make_graph.py <outdir> <target_unitigs> [seed] [k] [max_ins] [ploidy]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402

out, target = sys.argv[1], int(sys.argv[2])
seed = int(sys.argv[3]) if len(sys.argv) > 3 else 1000
os.makedirs(out, exist_ok=True)
import torch
dev = "cuda" if torch.cuda.is_available() else "cpu"
k = int(sys.argv[4]) if len(sys.argv) > 4 else bench.K
max_ins = int(sys.argv[5]) if len(sys.argv) > 5 else 6
ploidy = int(sys.argv[6]) if len(sys.argv) > 6 else 4
extra = dict(p_snp=0.5, p_del=0.1) if max_ins > 6 else {}
print(bench.make_inputs(out, "g", int(target / bench.UNITIGS_PER_BP), seed, dev, k=k, max_ins=max_ins, ploidy=ploidy, **extra))
