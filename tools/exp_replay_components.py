#!/usr/bin/env python3
"""How far can the commit replay be spread over host threads?  (GPU box)   usage: tools/exp_replay_components.py [unitigs] [seed]

Two traversal records commute when they touch disjoint state.  A record touches one *side* (plus / minus partner slot and that
side's flag bits) of its entrance and of its exit, and both sides of every interior vertex (poison); partner links only ever
join sides that one accepted record touched together.  So the connected components of the graph {sides} with one clique per
effective record are independent units of work, each replayed in record order.  This prints their size distribution for a
bench-style graph: the records of the largest components are the sequential floor of a parallel replay."""
import os
import sys
import tempfile
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from ploidyfrost_amd import hipapi, hostapi  # noqa: E402

target = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
seed = int(sys.argv[2]) if len(sys.argv) > 2 else 1000
work = tempfile.mkdtemp(prefix="pf_cc.", dir="/dev/shm")
gfa, db, n, nk = bench.make_inputs(work, "g", int(target / bench.UNITIGS_PER_BP), seed, "cuda")
run = hostapi.Run(gfa, db, z=bench.Z, device=0)
run.set_output_dir(os.path.join(work, "out"))
run.set_unitig_id("x")
rec, pool = run.find_shard(0, n)
print("unitigs %d, records %d, pool entries %d" % (n, len(rec), len(pool)))

from scipy.sparse import coo_matrix  # noqa: E402
from scipy.sparse.csgraph import connected_components  # noqa: E402

t0 = time.time()
outcome = rec["outcome"]
eff = np.ones(len(rec), dtype=bool)
eff &= ~((outcome == 0) & (rec["flag_cycle"] == 0))            # no exit, no cycle: nothing is committed
eff &= ~((outcome == 3) & (rec["n_seen"] < 4))                  # accepted, fewer than four vertices: nothing either
names = {0: "none", 1: "cycle_exit", 2: "reject", 3: "accept"}
print("outcomes:", {names.get(int(o), int(o)): int((outcome == o).sum()) for o in np.unique(outcome)})
n_list = rec["n_list"].astype(np.int64)
off = rec["list_off"].astype(np.int64)
idx = np.nonzero(eff & (n_list > 0))[0]
# entries of all effective records, flattened
lens = n_list[idx]
starts = off[idx]
tot = int(lens.sum())
owner = np.repeat(np.arange(len(idx)), lens)
pos = np.arange(tot) - np.repeat(np.cumsum(lens) - lens, lens)
ent = pool[np.repeat(starts, lens) + pos].astype(np.int64)      # oriented vertices 2u + strand
ent_u = ent >> 1
s = rec["entrance"].astype(np.int64)[idx][owner]
t = rec["exit"].astype(np.int64)[idx][owner]
has_exit = (outcome[idx] != 0)[owner]
is_s = ent == s
is_t = has_exit & (ent == t)
all_interior = np.isin(outcome[idx], (0, 1))[owner]              # cycle commits poison every list entry
interior = ~(is_s | is_t) | all_interior
# node ids: side 2u (plus), 2u+1 (minus).  entrance 2u+strand bit 0 == 0 -> plus side; exit -> the opposite side
anchor = (2 * (s >> 1) + (s & 1))                                   # the entrance's side, one per entry (clique centre)
rows, cols = [], []
# interior: both sides joined to the anchor
rows += [2 * ent_u[interior], 2 * ent_u[interior] + 1]
cols += [anchor[interior], anchor[interior]]
# exit: the side entered = opposite of its orientation bit
rows += [(2 * ent_u[is_t] + (1 - (ent[is_t] & 1)))]
cols += [anchor[is_t]]
r = np.concatenate(rows)
c = np.concatenate(cols)
g = coo_matrix((np.ones(len(r), dtype=np.int8), (r, c)), shape=(2 * n, 2 * n))
ncomp, lab = connected_components(g, directed=False)
rec_lab = lab[2 * (rec["entrance"].astype(np.int64)[idx] >> 1) + (rec["entrance"].astype(np.int64)[idx] & 1)]
cnt = np.bincount(rec_lab)
work_per = np.bincount(rec_lab, weights=lens)
order = np.argsort(-work_per)
print("components with records: %d; effective records %d, list entries %d  (%.1fs)" % ((cnt > 0).sum(), len(idx), tot, time.time() - t0))
print("largest components by list entries (records, entries, share of all entries):")
for o in order[:12]:
    print("   %9d %10d  %.4f" % (cnt[o], work_per[o], work_per[o] / tot))
for T in (4, 8, 16, 32):
    # longest-processing-time bound: max(largest component, total / T)
    print("threads %2d: floor = max(%.3f, %.3f) of the sequential work" % (T, work_per[order[0]] / tot, 1.0 / T))
big = rec["n_list"] > 4096
print("records with more than 4096 list entries: %d, their entries %d" % (big.sum(), rec["n_list"][big].sum()))
