"""The host tier of K-BFS (ploidyfrost_amd/csrc/host/pf_bfs_host.cpp) against the oracle's traversal, candidate by candidate,
without a GPU: outcome, exit, flags, the seen list in first-seen order and the cycle set, on fixtures with cycles, tips, hairpins
and a traversal of 4713 vertices."""
import ctypes as C
import os
import sys

import numpy as np
import pytest

from conftest import ROOT, load_case

sys.path.insert(0, os.path.join(ROOT, "oracle"))
import pyoracle  # noqa: E402

from ploidyfrost_amd import hipapi, hostapi  # noqa: E402


@pytest.mark.parametrize("case", ["weird12k", "reads10k", "giant7k", "hex30k"])
def test_host_walk_matches_the_oracle(case):
    meta = load_case(case)
    o = pyoracle.Oracle(meta["gfa"], meta["db"])
    succ, pred = o.adjacency()
    succ = np.ascontiguousarray(succ, dtype=np.uint32).reshape(-1, 4)
    pred = np.ascontiguousarray(pred, dtype=np.uint32).reshape(-1, 4)
    n = succ.shape[0] // 2
    L = hostapi.load_library()
    L.pfh_host_walk.restype = C.c_int
    L.pfh_host_walk.argtypes = [C.c_void_p, C.c_void_p, C.c_uint32, C.c_uint32, C.c_void_p, C.c_void_p, C.c_uint64]
    cand = [ov for ov in range(2 * n) if (succ[ov] != hipapi.NONE).sum() > 1]
    rec = np.zeros(1, dtype=hipapi.BFS_RECORD)
    buf = np.zeros(2 * n + 8, dtype=np.uint32)
    if len(cand) > 1500:   # every 3rd candidate plus the 50 largest traversals is plenty for the (slow) oracle
        sizes = []
        for s in cand:
            assert L.pfh_host_walk(succ.ctypes.data, pred.ctypes.data, n, s, rec.ctypes.data, buf.ctypes.data, len(buf)) == 0
            sizes.append((int(rec[0]["n_seen"]), s))
        cand = sorted(set(cand[::3]) | {ov for _, ov in sorted(sizes, reverse=True)[:50]})
    biggest = 0
    for s in cand:
        e = o.extract(s)
        assert L.pfh_host_walk(succ.ctypes.data, pred.ctypes.data, n, s, rec.ctypes.data, buf.ctypes.data, len(buf)) == 0
        r = rec[0]
        assert (int(r["outcome"]), int(r["n_seen"]), bool(r["flag_cycle"]), bool(r["flag_tip"])) == \
            (e["outcome"], len(e["seen"]), bool(e["flag_cycle"]), bool(e["flag_tip"])), s
        got = buf[: int(r["n_list"])]
        if e["outcome"] != 0:
            assert int(r["exit"]) == e["exit"] and np.array_equal(got, e["seen"]), s
        else:
            assert int(r["exit"]) == hipapi.NONE
            assert sorted(got.tolist()) == (sorted(set(e["cyc"].tolist())) if e["flag_cycle"] else []), s
        biggest = max(biggest, int(r["n_seen"]))
    if case == "giant7k":
        assert biggest > 4096
