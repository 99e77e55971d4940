#!/usr/bin/env python3
"""The parallel commit replay against the sequential one on random graphs, without a device (runs anywhere):
   usage: tools/fuzz_replay.py [n_cases] [first_seed]
Per case (the single-sample inputs of tools/fuzz_parity.py, same environment switches: PF_FUZZ_BIFROST, PF_FUZZ_GIANT ...):
  * traversal records of every candidate entrance from the host walker;
  * pfh_replay_check_footprints: the sequential replay with every state access held against the record's component, the
    components grown in slices of 0 (all at once) / 61 / n/3 records -- any access outside is a failure of the model;
  * the parallel replay with 1..16 threads in 1..4 shards must leave the sequential state."""
import os
import sys
import tempfile

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "oracle"))
sys.path.insert(0, os.path.join(ROOT, "tools"))
import fuzz_parity  # noqa: E402
import pyoracle  # noqa: E402
from ploidyfrost_amd import hostapi  # noqa: E402


def one(seed, tmp):
    c = fuzz_parity.build_case(seed, tmp, "cpu")
    if isinstance(c, str):
        return c
    if c["colored"]:
        return "skipped (colored)"
    o = pyoracle.Oracle(c["gfa"], c["db"])
    succ, pred = o.adjacency()
    n = len(succ) // 2
    rec, pool = hostapi.host_walk_range(succ, pred, 0, n)
    z = c["z"]
    desc = "k=%d ploidy=%d z=%d unitigs=%d records=%d longest=%d" % (c["k"], c["ploidy"], z, n, len(rec), int(rec["n_seen"].max()) if len(rec) else 0)
    for sl in (0, 61, max(1, len(rec) // 3)):
        bad, first = hostapi.check_footprints(rec, pool, n, z, sl)
        if bad:
            r = rec[first]
            return "%s: FOOTPRINT %d accesses outside the component, first at record %d (entrance %d outcome %d n_seen %d), slices of %d" % (
                desc, bad, first, r["entrance"], r["outcome"], r["n_seen"], sl)
    seq = hostapi.Replay(n, z)
    seq.apply(rec, pool)
    want = seq.state()
    rng = np.random.default_rng(seed)
    for threads in (1, 3, 8, 16):
        shards = int(rng.integers(1, 5))
        cuts = sorted(set([0, len(rec)] + [int(x) for x in rng.integers(0, len(rec) + 1, size=shards - 1)]))
        par = hostapi.Replay(n, z)
        for a, b in zip(cuts[:-1], cuts[1:]):
            par.apply(rec[a:b], pool, threads=threads)
        if not all(np.array_equal(x, y) for x, y in zip(par.state(), want)):
            return "%s: STATE differs with %d threads, cuts %s" % (desc, threads, cuts)
    lab = hostapi.side_components(rec, pool, n)
    _, cnt = np.unique(lab, return_counts=True)
    return "%s: identical; %d components, largest %d records" % (desc, len(cnt), int(cnt.max()) if len(cnt) else 0)


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 30
    first = int(sys.argv[2]) if len(sys.argv) > 2 else 1
    pyoracle.build()
    failures = 0
    for seed in range(first, first + n):
        with tempfile.TemporaryDirectory() as tmp:
            msg = one(seed, tmp)
        print("seed %d: %s" % (seed, msg), flush=True)
        failures += "FOOTPRINT" in msg or "STATE" in msg
    print("%d cases, %d failures" % (n, failures))
    return 1 if failures else 0


if __name__ == "__main__":
    sys.exit(main())
