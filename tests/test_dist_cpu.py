"""The N>1 path on CPU: two processes, gloo, 127.0.0.1 -- the exchanges bench.py runs over RCCL.
1. counters and variable-length slabs all-gathered in rank order;
2. findSuperBubble of ONE graph cut by entrance vertex: each rank produces the traversal records of its unitig range (host walker:
   no GPU here), the records and vertex pools are all-gathered, every rank replays all of them in entrance order, and the
   resulting MyUnitig state must equal the oracle's after its own findSuperBubble -- on every rank."""
import os
import subprocess
import sys
import textwrap

import pytest

from conftest import ROOT, load_case

WORKER = textwrap.dedent("""
    import os, sys
    sys.path.insert(0, %r)
    import numpy as np, torch
    from ploidyfrost_amd import dist as pfdist
    rank, local_rank, world = pfdist.env_rank_world()
    pfdist.init("gloo")
    dev = torch.device("cpu")
    # every rank owns a contiguous block of unitigs; its slab is that block's records
    u0, u1 = pfdist.shard_range(1001, rank, world)
    slab = np.frombuffer(("".join("%%d\\n" %% u for u in range(u0, u1))).encode(), dtype=np.uint8)
    counters = pfdist.all_gather_counters([rank, u1 - u0, 7], dev)
    assert counters.shape == (world, 3) and list(counters[:, 0]) == list(range(world)) and counters[:, 1].sum() == 1001
    slabs = pfdist.all_gather_slabs(slab, dev)
    text = b"".join(s.tobytes() for s in slabs).decode()
    assert text == "".join("%%d\\n" %% u for u in range(1001)), "rank-order concatenation must be id order"
    # empty slab on one rank
    slabs = pfdist.all_gather_slabs(slab if rank else np.zeros(0, dtype=np.uint8), dev)
    assert slabs[0].size == 0 and slabs[1].size > 0
    assert pfdist.broadcast_str("dir-of-rank-%%d" %% rank) == "dir-of-rank-0"
    torch.distributed.barrier()
    print("rank", rank, "ok")
""") % ROOT

SHARDED_FIND = textwrap.dedent("""
    import os, sys
    sys.path.insert(0, %r)
    sys.path.insert(0, os.path.join(%r, "oracle"))
    import numpy as np, torch
    import pyoracle
    from ploidyfrost_amd import dist as pfdist, hipapi, hostapi
    rank, local_rank, world = pfdist.env_rank_world()
    pfdist.init("gloo")
    dev = torch.device("cpu")
    gfa, db, z = sys.argv[1], sys.argv[2], int(sys.argv[3])
    o = pyoracle.Oracle(gfa, db)
    succ, pred = o.adjacency()
    n = len(succ) // 2
    # this rank's shard: the candidate entrances on its unitig range
    u0, u1 = pfdist.shard_range(n, rank, world)
    rec, pool = hostapi.host_walk_range(succ, pred, u0, u1)
    assert (np.diff(rec["entrance"].astype(np.int64)) > 0).all() and (len(rec) == 0 or (rec["entrance"][0] >> 1) >= u0)
    recs = pfdist.all_gather_slabs(rec.view(np.uint8).reshape(-1), dev)
    pools = pfdist.all_gather_slabs(pool.view(np.uint8).reshape(-1), dev)
    # the replicated replay: sequentially, and spread over host threads by the components of the records' footprints
    # (csrc/host/pf_replay_par.hpp) -- shard after shard, the components growing with every shard
    rp, rq = hostapi.Replay(n, z), hostapi.Replay(n, z)
    total = 0
    for r, p in zip(recs, pools):
        r = np.ascontiguousarray(r).view(hipapi.BFS_RECORD)
        total += len(r)
        rp.apply(r, np.ascontiguousarray(p).view(np.uint32))
        rq.apply(r, np.ascontiguousarray(p).view(np.uint32), threads=3 + rank)
    f, p, m = rp.state()
    assert all(np.array_equal(a, b) for a, b in zip(rq.state(), (f, p, m))), "parallel replay differs from the sequential one"
    o.find_superbubbles(z=z)
    ef, ep, em = o.state()
    assert total == int(((np.asarray(succ).reshape(-1, 4) != hipapi.NONE).sum(axis=1) > 1).sum())
    assert np.array_equal(f, ef) and np.array_equal(p, ep) and np.array_equal(m, em), "state after the sharded replay differs from the oracle's"
    # and a shard applied out of order is refused
    if world > 1 and len(recs[0]) and len(recs[1]):
        bad = hostapi.Replay(n, z)
        bad.apply(np.ascontiguousarray(recs[1]).view(hipapi.BFS_RECORD), np.ascontiguousarray(pools[1]).view(np.uint32))
        try:
            bad.apply(np.ascontiguousarray(recs[0]).view(hipapi.BFS_RECORD), np.ascontiguousarray(pools[0]).view(np.uint32))
            raise SystemExit("out-of-order shard accepted")
        except RuntimeError:
            pass
    torch.distributed.barrier()
    print("rank", rank, "ok", total, "records,", int((f & 3 != 0).sum()), "open unitigs")
""") % (ROOT, ROOT)


def _run_world(tmp_path, script_text, port, args=(), world=2):
    script = tmp_path / "worker.py"
    script.write_text(script_text)
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), WORLD_SIZE=str(world))
    procs = [subprocess.Popen([sys.executable, str(script)] + [str(a) for a in args], env=dict(env, RANK=str(r), LOCAL_RANK=str(r)),
                              stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True) for r in range(world)]
    outs = [p.communicate(timeout=400)[0] for p in procs]
    for r, (p, o) in enumerate(zip(procs, outs)):
        assert p.returncode == 0, o
        assert "rank %d ok" % r in o
    return outs


def test_world_size_2_gloo(tmp_path):
    _run_world(tmp_path, WORKER, 29613)


@pytest.mark.parametrize("case,world", [("tet60k", 2), ("weird12k", 2), ("giant7k", 3)])
def test_find_superbubbles_sharded_by_entrance_matches_the_oracle(case, world, tmp_path):
    meta = load_case(case)
    _run_world(tmp_path, SHARDED_FIND, 29617 + world, args=(meta["gfa"], meta["db"], int(meta["opts"]["-z"])), world=world)


def test_rank_group_around_rccl_without_rccl(tmp_path):
    """csrc/host/pf_multi.cpp (`ploidyfrost --gpus N`) linked against stand-ins for the device layer's four calls
    (tests/cpp/test_multi.cpp): rank r on device r, rank 0's communicator id reaches every rank across the fork, the gathered words;
    a rank that fails alone -- or is gone without a word -- ends every rank non-zero BEFORE the next collective (RCCL would wait for
    ever), and nothing hangs."""
    import subprocess
    exe = str(tmp_path / "test_multi")
    subprocess.run(["g++", "-O2", "-std=c++17", "-I", os.path.join(ROOT, "include"), "-I", os.path.join(ROOT, "ploidyfrost_amd", "csrc", "host"),
                    os.path.join(ROOT, "tests", "cpp", "test_multi.cpp"), os.path.join(ROOT, "ploidyfrost_amd", "csrc", "host", "pf_multi.cpp"),
                    "-o", exe, "-lpthread"], check=True)

    def run(*a):
        return subprocess.run([exe] + [str(x) for x in a], stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=60)
    for world in (2, 4, 8):
        r = run("ok", world)
        assert r.returncode == 0 and "OK %d ranks, %d gathers entered" % (world, 2 * world) in r.stdout, r.stderr
    for world, who, stage, gathers in ((4, 2, "align", 0), (3, 0, "text", 3), (4, 3, "communicator id", 0), (2, 1, "load", 0), (4, 1, "write", 8)):
        r = run("fail", world, who, stage)
        assert r.returncode != 0, (stage, r.stderr)
        assert "rank %d: " % who in r.stderr and "injected failure at %s" % stage in r.stderr
        assert "gathers entered: %d" % gathers in r.stderr, r.stderr   # nobody went into the collective behind the failed step
        for q in range(world):
            assert "rank %d:" % q in r.stderr   # every rank said why it left
    r = run("die", 4, 1)
    assert r.returncode != 0 and "gathers entered: 0" in r.stderr and "another rank failed" in r.stderr
