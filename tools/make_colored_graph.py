#!/usr/bin/env python3
"""Generate a bench-style colored input set (BASELINE.json configs[3]) into a directory:
<out>/g.gfa, g.bfg_colors, g_kmc<i>.kmc_*, dbs.txt, cutoffs.txt.
usage: tools/make_colored_graph.py <outdir> <target_unitigs> [seed] [k] [samples] [ploidy]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402

out, target = sys.argv[1], int(sys.argv[2])
seed = int(sys.argv[3]) if len(sys.argv) > 3 else 1000
k = int(sys.argv[4]) if len(sys.argv) > 4 else bench.K
samples = int(sys.argv[5]) if len(sys.argv) > 5 else 3
ploidy = int(sys.argv[6]) if len(sys.argv) > 6 else 2
os.makedirs(out, exist_ok=True)
import torch
dev = "cuda" if torch.cuda.is_available() else "cpu"
genome = int(target / bench.UNITIGS_PER_BP)  # measured: 3 x 2 haplotypes give the tetraploid bench graph's density
gfa, colors, dbs, n, nk = bench.make_colored_inputs(out, "g", genome, seed, dev, k=k, samples=samples, ploidy=ploidy)
with open(os.path.join(out, "dbs.txt"), "w") as f:
    f.write("".join(d + "\n" for d in dbs))
with open(os.path.join(out, "cutoffs.txt"), "w") as f:
    f.write("5\t1000\n" * samples)
print(gfa, colors, n, nk)
