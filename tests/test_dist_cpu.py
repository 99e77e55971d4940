"""The N>1 path on CPU: two processes, gloo, 127.0.0.1 -- the same exchange bench.py runs over
RCCL: counters and variable-length record slabs all-gathered in rank order."""
import os
import subprocess
import sys
import textwrap

from conftest import ROOT

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
    torch.distributed.barrier()
    print("rank", rank, "ok")
""") % ROOT


def test_world_size_2_gloo(tmp_path):
    script = tmp_path / "worker.py"
    script.write_text(WORKER)
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT="29613", WORLD_SIZE="2")
    procs = [subprocess.Popen([sys.executable, str(script)], env=dict(env, RANK=str(r), LOCAL_RANK=str(r)),
                              stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True) for r in range(2)]
    outs = [p.communicate(timeout=240)[0] for p in procs]
    for r, (p, o) in enumerate(zip(procs, outs)):
        assert p.returncode == 0, o
        assert "rank %d ok" % r in o
