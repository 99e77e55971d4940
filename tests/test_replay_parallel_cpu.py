"""The commit replay of findSuperBubble spread over host threads (csrc/host/pf_replay_par.hpp), without a device:
1. the model -- a record only ever touches unitig sides of its own component -- is checked access by access while the sequential
   replay runs (pfh_replay_check_footprints), with the components grown slice by slice as the executor sees them;
2. the parallel replay (any thread count, any sharding) leaves exactly the state of the sequential one, which is the oracle's."""
import os
import sys

import numpy as np
import pytest

from conftest import ROOT, colored_cases, golden_cases, load_case

sys.path.insert(0, os.path.join(ROOT, "oracle"))
import pyoracle  # noqa: E402
from ploidyfrost_amd import hostapi  # noqa: E402

_cache = {}


def records_of(case):
    if case not in _cache:
        meta = load_case(case)
        o = pyoracle.Oracle(meta["gfa"], meta["db"])
        succ, pred = o.adjacency()
        n = len(succ) // 2
        rec, pool = hostapi.host_walk_range(succ, pred, 0, n)
        z = int(meta["opts"]["-z"])
        o.find_superbubbles(z=z)
        _cache[case] = (n, z, rec, pool, o.state())
    return _cache[case]


@pytest.mark.parametrize("case", golden_cases())
def test_every_access_stays_inside_the_records_component(case):
    n, z, rec, pool, _ = records_of(case)
    for slice_len in (0, max(1, len(rec) // 4), 97):
        bad, first = hostapi.check_footprints(rec, pool, n, z, slice_len)
        assert bad == 0, "record %s of %s touches a side outside its component (slices of %d)" % (first, case, slice_len)


@pytest.mark.parametrize("case", golden_cases())
def test_parallel_replay_leaves_the_sequential_state(case):
    n, z, rec, pool, want = records_of(case)
    seq = hostapi.Replay(n, z)
    seq.apply(rec, pool)
    for a, b in zip(seq.state(), want):
        assert np.array_equal(a, b)
    for threads, shards in ((1, 1), (4, 1), (8, 3), (16, 5)):
        par = hostapi.Replay(n, z)
        cuts = [len(rec) * i // shards for i in range(shards + 1)]
        for i in range(shards):
            par.apply(rec[cuts[i]:cuts[i + 1]], pool, threads=threads)
        for name, a, b in zip(("flags", "plus", "minus"), par.state(), want):
            assert np.array_equal(a, b), "%s differs after the parallel replay of %s (%d threads, %d shards)" % (name, case, threads, shards)


def test_components_are_small_on_a_bubble_chain():
    """a chain of bubbles shares endpoint unitigs, but an endpoint is touched through one side only: the chain does not collapse
    into one component"""
    n, z, rec, pool, _ = records_of("tet60k")
    lab = hostapi.side_components(rec, pool, n)
    _, counts = np.unique(lab, return_counts=True)
    assert counts.max() < 0.2 * len(rec)
    assert len(counts) > len(rec) // 8


def _random_records(rng, n_unitigs, density, sparse=False):
    """arbitrary traversal records (not those of any graph): every oriented vertex may be an entrance once, ascending; exits,
    outcomes and lists at random -- dense enough that sides are re-linked, released twice and poisoned while linked"""
    from ploidyfrost_amd import hipapi
    ent = np.nonzero(rng.random(2 * n_unitigs) < density)[0].astype(np.uint32)
    rec = np.zeros(len(ent), dtype=hipapi.BFS_RECORD)
    pool = []
    for i, s in enumerate(ent):
        outcome = int(rng.choice([0, 1, 2, 3, 3, 3]))
        t = int(rng.integers(0, 2 * n_unitigs))
        while t >> 1 == s >> 1 and rng.random() < 0.9:   # (now and then a traversal that ends on the unitig it started from)
            t = int(rng.integers(0, 2 * n_unitigs))
        n_inner = int(rng.integers(0, 5)) if not sparse else int(rng.random() < 0.25)
        inner = [int(x) for x in rng.integers(0, 2 * n_unitigs, size=n_inner)]
        if sparse and rng.random() < 0.5:   # exits near the entrance: sides are shared by few records, re-linked often
            t = int((s + rng.integers(2, 12)) % (2 * n_unitigs))
            if t >> 1 == s >> 1 and rng.random() < 0.9:
                t = int((t + 2) % (2 * n_unitigs))
        lst = [int(s)] + inner + ([t] if outcome else [])
        if outcome == 0 and rng.random() < 0.5:
            lst = inner   # the cycle set need not hold the entrance
        rec[i]["entrance"] = s
        rec[i]["exit"] = t if outcome else 0xFFFFFFFF
        rec[i]["outcome"] = outcome
        rec[i]["flag_cycle"] = int(rng.random() < 0.6)
        rec[i]["strict"] = int(rng.random() < 0.5)
        rec[i]["n_seen"] = max(len(lst), int(rng.integers(2, 12)))
        rec[i]["n_list"] = len(lst)
        rec[i]["list_off"] = len(pool)
        pool += lst
    return rec, np.array(pool + [0], dtype=np.uint32)


@pytest.mark.parametrize("seed", range(40))
def test_model_holds_for_arbitrary_records(seed):
    """the footprint model is a property of the commits, not of the graphs that feed them"""
    rng = np.random.default_rng(seed)
    n = int(rng.choice([12, 40, 200, 2000]))
    rec, pool = _random_records(rng, n, float(rng.choice([0.3, 0.7, 1.0])), sparse=seed % 2 == 1)
    for slice_len in (0, 7, max(1, len(rec) // 3)):
        bad, first = hostapi.check_footprints(rec, pool, n, 8, slice_len)
        assert bad == 0, "record %s touches a side outside its component (seed %d, slices of %d): %s" % (first, seed, slice_len, rec[first] if first is not None else "")
    seq = hostapi.Replay(n, 8)
    seq.apply(rec, pool)
    want = seq.state()
    for threads, shards in ((2, 1), (8, 2), (16, 4)):
        par = hostapi.Replay(n, 8)
        cuts = [len(rec) * i // shards for i in range(shards + 1)]
        for i in range(shards):
            par.apply(rec[cuts[i]:cuts[i + 1]], pool, threads=threads)
        for name, a, b in zip(("flags", "plus", "minus"), par.state(), want):
            assert np.array_equal(a, b), "%s differs (seed %d, %d threads, %d shards)" % (name, seed, threads, shards)


@pytest.mark.parametrize("dropped", [2, 4])
def test_the_check_notices_a_missing_rule(dropped):
    """the footprint check itself is held to account: with the two-sides-linked rule (2) or the rejected-exit rule (4) left
    out of the components, some random record sequence must be caught touching a side outside its component"""
    import subprocess
    code = (
        "import sys; sys.path.insert(0, %r); sys.path.insert(0, %r)\n"
        "import numpy as np\n"
        "from test_replay_parallel_cpu import _random_records\n"
        "from ploidyfrost_amd import hostapi\n"
        "hits = 0\n"
        "for seed in range(400):\n"
        "    rng = np.random.default_rng(seed + 10000)\n"
        "    n = int(rng.choice([12, 40, 200, 2000]))\n"
        "    rec, pool = _random_records(rng, n, float(rng.choice([0.3, 0.7, 1.0])), sparse=True)\n"
        "    hits += hostapi.check_footprints(rec, pool, n, 8, 0)[0] > 0\n"
        "print('hits', hits)\n" % (ROOT, os.path.join(ROOT, "tests")))
    out = subprocess.run([sys.executable, "-c", code], env=dict(os.environ, PF_CC_WITHOUT=str(dropped)), capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stderr[-500:]
    assert int(out.stdout.split()[-1]) > 0


@pytest.mark.parametrize("case", colored_cases())
def test_colored_commits_stay_inside_their_components(case):
    """CCDBG's accept commit may mark an endpoint whose colour set is incomplete NON_SUPER -- a write to the whole unitig by a record
    that came in through one side: the components join both sides of such endpoints, and with that the model holds for the
    colored commits too (which is what lets them run on the device, one thread per component)."""
    meta = load_case(case)
    o = pyoracle.Oracle(meta["gfa"], None)
    succ, pred = o.adjacency()
    n = len(succ) // 2
    rec, pool = hostapi.host_walk_range(succ, pred, 0, n)
    col = hostapi.Colors(meta["gfa"], meta["colors"])
    z = int(meta["opts"]["-z"])
    for slice_len in (0, 53, max(1, len(rec) // 3)):
        bad, first = col.check_footprints(succ, rec, pool, z, slice_len)
        assert bad == 0, "record %s of %s touches a side outside its component (slices of %d)" % (first, case, slice_len)


def test_the_check_notices_the_missing_colour_rule():
    """... and without the incomplete-endpoint rule the fixture with partial colours is caught"""
    import subprocess
    code = (
        "import sys; sys.path.insert(0, %r); sys.path.insert(0, %r); sys.path.insert(0, %r)\n"
        "import pyoracle\n"
        "from conftest import load_case\n"
        "from ploidyfrost_amd import hostapi\n"
        "meta = load_case('col4_mix')\n"
        "o = pyoracle.Oracle(meta['gfa'], None)\n"
        "succ, pred = o.adjacency()\n"
        "rec, pool = hostapi.host_walk_range(succ, pred, 0, len(succ) // 2)\n"
        "print('bad', hostapi.Colors(meta['gfa'], meta['colors']).check_footprints(succ, rec, pool, int(meta['opts']['-z']), 0)[0])\n"
        % (ROOT, os.path.join(ROOT, "tests"), os.path.join(ROOT, "oracle")))
    out = subprocess.run([sys.executable, "-c", code], env=dict(os.environ, PF_CC_WITHOUT="8"), capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stderr[-500:]
    assert int(out.stdout.split()[-1]) > 0
