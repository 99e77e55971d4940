"""Parity of every HIP kernel against the CPU oracle, through the C ABI
(include/ploidyfrost_hip.h), on the committed fixtures.  Bit-exact: all of it is integer work."""
import os
import sys

import numpy as np
import pytest

from conftest import ROOT, golden_cases, load_case

sys.path.insert(0, os.path.join(ROOT, "oracle"))
import pyoracle  # noqa: E402

from ploidyfrost_amd import hipapi, synth  # noqa: E402

pytestmark = pytest.mark.gpu

_cache = {}


def setup_case(case):
    """(oracle, device) with graph + counts uploaded; cached per case."""
    if case in _cache:
        return _cache[case]
    meta = load_case(case)
    o = pyoracle.Oracle(meta["gfa"], meta["db"])
    seqs = o.sequences()
    dev = hipapi.Device(0)
    words, off, lens = hipapi.pack_unitigs(seqs)
    dev.upload_graph(words, off, lens, o.k)
    kmers, counts, km = synth.read_kmc(meta["db"])
    dev.upload_counts(kmers, counts, km["min_count"], km["max_count"], km["both_strands"])
    _cache[case] = (meta, o, dev, seqs)
    return _cache[case]


@pytest.mark.parametrize("case", golden_cases())
def test_adjacency_matches_oracle(case):
    meta, o, dev, _ = setup_case(case)
    succ, pred = dev.build_adjacency()
    es, ep = o.adjacency()
    assert np.array_equal(succ, es)
    assert np.array_equal(pred, ep)


@pytest.mark.parametrize("case", golden_cases())
def test_unitig_cov_matches_oracle(case):
    meta, o, dev, _ = setup_case(case)
    s, m, miss, st = dev.unitig_cov()
    es, em, emiss = o.unitig_cov()
    assert st == hipapi.PF_OK and not miss.any() and not emiss.any()
    assert np.array_equal(s, es)
    assert np.array_equal(m, em)
    # sharded call (the multi-GPU partition) gives the same slices
    h = dev.n // 2
    s2, m2, _, _ = dev.unitig_cov(h, dev.n)
    assert np.array_equal(s2, es[h:]) and np.array_equal(m2, em[h:])


@pytest.mark.parametrize("case", golden_cases())
def test_unitig_cov_streamed_equals_probed(case):
    """K-COV streams the per-k-mer coverage SoA joined at load (PF_K_COV_JOIN); pf_unitig_cov_probe looks every k-mer up
    at call time.  Same function (src/CDBG.cpp:66-120), two independent routes: whole range and ragged sub-ranges."""
    meta, o, dev, _ = setup_case(case)
    if not getattr(dev, "both_strands", True):
        pytest.skip("database without canonical counting: neither form applies (pf_unitig_cov_exact)")
    s, m, x, st = dev.unitig_cov()
    ps, pm, px, pst = dev.unitig_cov_probe()
    assert st == pst == hipapi.PF_OK
    assert np.array_equal(s, ps) and np.array_equal(m, pm) and np.array_equal(x, px)
    rng = np.random.default_rng(7)
    for _ in range(6):
        u0 = int(rng.integers(0, dev.n))
        u1 = int(rng.integers(u0 + 1, dev.n + 1))
        a, b = dev.unitig_cov(u0, u1), dev.unitig_cov_probe(u0, u1)
        assert np.array_equal(a[0], s[u0:u1]) and np.array_equal(a[1], m[u0:u1])
        assert np.array_equal(b[0], s[u0:u1]) and np.array_equal(b[1], m[u0:u1])
    for u in (0, dev.n - 1):  # a single unitig at either end of the array
        a = dev.unitig_cov(u, u + 1)
        assert a[0][0] == s[u] and a[1][0] == m[u]


@pytest.mark.parametrize("max_count", [65535, (1 << 25) - 1, (1 << 25), (1 << 32) - 2, (1 << 32) - 1])
def test_unitig_cov_wide_counts_and_reupload(max_count):
    """Row sums are 32-bit only while 64 * max_count fits; a database whose max_count is the SoA's marker value keeps the
    probing form; a second pf_upload_counts replaces the joined SoA.  Counts close to max_count exercise both sum widths."""
    meta, o, dev, seqs = setup_case("dip20k")
    kmers, counts, km = synth.read_kmc(meta["db"])
    big = counts.astype(np.uint32).copy()
    bump = min(1 << 31, max_count - 65536) if max_count > 65535 else 0   # every bumped count stays inside [1, max_count]
    if bump:
        big[::3] += np.uint32(bump)
    d = hipapi.Device(0)
    d.upload_graph(*hipapi.pack_unitigs(seqs), o.k)
    d.upload_counts(kmers, counts, 1, 65535, True)      # first database: joined ...
    first = d.unitig_cov()
    d.upload_counts(kmers, big, 1, max_count, True)      # ... then replaced
    s, m, x, st = d.unitig_cov()
    ps, pm, px, pst = d.unitig_cov_probe()
    assert st == pst == hipapi.PF_OK and not x.any()
    assert np.array_equal(s, ps) and np.array_equal(m, pm)
    es, em, _ = o.unitig_cov()
    assert np.array_equal(first[0], es) and np.array_equal(first[1], em)
    if bump:
        assert int(s.sum()) > int(es.sum()) + bump and (s >= es).all()
    else:
        assert np.array_equal(s, es)
    d.close()


def test_cov_join_is_timed_as_its_own_kernel():
    meta, o, dev, seqs = setup_case("dip20k")
    kmers, counts, km = synth.read_kmc(meta["db"])
    d = hipapi.Device(0)
    d.enable_timing(True)
    d.upload_graph(*hipapi.pack_unitigs(seqs), o.k)
    d.upload_counts(kmers, counts, 1, 65535, True)
    d.unitig_cov()
    t = d.kernel_times()
    assert t["k_cov_join"][1] == 1 and t["k_cov"][1] == 1
    d.upload_graph(*hipapi.pack_unitigs(seqs), o.k)      # a new graph under the same table: joined again
    s, m, x, st = d.unitig_cov()
    es, em, _ = o.unitig_cov()
    assert np.array_equal(s, es) and np.array_equal(m, em)
    assert d.kernel_times()["k_cov_join"][1] == 2
    d.close()


def test_timing_can_be_limited_to_some_kernels_and_reuses_its_events():
    """pf_timing_select: events around the launches of the selected kernels only (bench.py times K-BUBBLE and K-COV inside its
    timed region, everything in further passes); pf_reset_timing hands the events to a pool the next launches draw from."""
    meta, o, dev, seqs = setup_case("dip20k")
    kmers, counts, km = synth.read_kmc(meta["db"])
    d = hipapi.Device(0)
    d.enable_timing(True)
    d.timing_select(["k_cov"])
    d.upload_graph(*hipapi.pack_unitigs(seqs), o.k)
    d.upload_counts(kmers, counts, 1, 65535, True)
    d.unitig_cov()
    t = d.kernel_times()
    assert set(t) == {"k_cov"} and t["k_cov"][1] == 1 and t["k_cov"][0] > 0
    d.timing_select(None)
    for _ in range(3):   # (the second and third round take their events from the pool)
        d.reset_timing()
        d.upload_graph(*hipapi.pack_unitigs(seqs), o.k)
        s, m, x, st = d.unitig_cov()
        t = d.kernel_times()
        assert t["k_cov_join"][1] == 1 and t["k_cov"][1] == 1 and t["k_cov"][0] > 0
    es, em, _ = o.unitig_cov()
    assert np.array_equal(s, es) and np.array_equal(m, em)
    d.close()


def test_missing_kmer_is_reported():
    meta, o, dev, _ = setup_case("dip20k")
    kmers, counts, km = synth.read_kmc(meta["db"])
    dev2 = hipapi.Device(0)
    dev2.upload_graph(*hipapi.pack_unitigs(o.sequences()), o.k)
    dev2.upload_counts(kmers[::2].copy(), counts[::2].copy(), 1, 65535, True)
    s, m, miss, st = dev2.unitig_cov()
    assert st == hipapi.PF_ERR_MISSING_KMER and miss.any()


def test_count_range_filter():
    """records outside [min_count, max_count] are not retrievable (kmc_file.cpp:1459)"""
    meta, o, dev, _ = setup_case("dip20k")
    kmers, counts, km = synth.read_kmc(meta["db"])
    dev2 = hipapi.Device(0)
    dev2.upload_graph(*hipapi.pack_unitigs(o.sequences()), o.k)
    dev2.upload_counts(kmers, counts, 30, 45, True)
    c, f = dev2.lookup(kmers)
    inside = (counts >= 30) & (counts <= 45)
    assert np.array_equal(f.astype(bool), inside)
    assert np.array_equal(c[inside], counts[inside])


@pytest.mark.parametrize("case", [c for c in golden_cases() if load_case(c).get("kmc_layout") != "kmc1_stranded"])
def test_lookup_both_orientations(case):
    meta, o, dev, _ = setup_case(case)
    kmers, counts, km = synth.read_kmc(meta["db"])
    k = o.k
    # reverse complements must find the same counts; random k-mers must miss
    x = kmers.copy()
    rc = np.zeros_like(x)
    for j in range(k):
        rc |= (np.uint64(3) - ((x >> np.uint64(2 * j)) & np.uint64(3))) << np.uint64(2 * (k - 1 - j))
    c, f = dev.lookup(rc)
    assert f.all() and np.array_equal(c, counts)
    rng = np.random.default_rng(5)
    rnd = rng.integers(0, 1 << (2 * k), size=4096, dtype=np.uint64)
    known = set(kmers.tolist())
    c, f = dev.lookup(rnd)
    for q, ff in zip(rnd.tolist(), f.tolist()):
        qq = int(q)
        r = 0
        for j in range(k):
            r |= (3 - ((qq >> (2 * j)) & 3)) << (2 * (k - 1 - j))
        assert bool(ff) == (qq in known or r in known)


@pytest.mark.parametrize("case", golden_cases())
def test_bfs_records_match_oracle(case):
    meta, o, dev, _ = setup_case(case)
    succ, _ = dev.build_adjacency()
    rec, pool = dev.bfs()
    cand = np.nonzero((succ != hipapi.NONE).sum(axis=1) > 1)[0]
    assert np.array_equal(rec["entrance"], cand.astype(np.uint32))
    n_strict = 0
    for r in rec:
        e = o.extract(int(r["entrance"]))
        assert int(r["outcome"]) == e["outcome"], (case, int(r["entrance"]))
        assert int(r["exit"]) == e["exit"]
        assert int(r["n_seen"]) == len(e["seen"])
        assert int(r["flag_cycle"]) == e["flag_cycle"] and int(r["flag_tip"]) == e["flag_tip"]
        lst = pool[int(r["list_off"]) : int(r["list_off"]) + int(r["n_list"])]
        if e["outcome"] != 0:
            assert np.array_equal(lst, e["seen"])
        elif e["flag_cycle"]:
            assert np.array_equal(lst, e["cyc"])
        else:
            assert len(lst) == 0
        n_strict += int(r["strict"])
    assert n_strict > 0
    # sharded traversal = slices of the full run
    h = dev.n // 3
    r2, p2 = dev.bfs(h, dev.n)
    full = rec[rec["entrance"] >= 2 * h]
    assert np.array_equal(r2["entrance"], full["entrance"]) and np.array_equal(r2["outcome"], full["outcome"])
    assert np.array_equal(r2["exit"], full["exit"]) and np.array_equal(r2["strict"], full["strict"])


@pytest.mark.parametrize("case", golden_cases())
def test_side_components_match_the_host_union_find(case):
    """K-CC (pf_cc.hip): the components the parallel commit replay is scheduled by.  Labels = smallest side of the component on
    both sides, so they must agree number for number; the order groups the records by class, ascending inside a class; records
    given in two slices (cumulative) end in the same components; a record pointing outside the graph is refused."""
    from ploidyfrost_amd import hostapi
    meta, o, dev, _ = setup_case(case)
    dev.build_adjacency()
    rec, pool = dev.bfs()
    want = hostapi.side_components(rec, pool, dev.n)
    # the records K-BFS left on the device
    dev.side_components(n_records=len(rec))
    order, off, lab = dev.replay_order(64)
    assert np.array_equal(lab, want)
    assert off[0] == 0 and off[-1] == len(rec) and (np.diff(off.astype(np.int64)) >= 0).all()
    cls = (lab >> 11) % 64
    for c in range(64):
        idx = order[off[c]:off[c + 1]]
        assert (cls[idx] == c).all() and (np.diff(idx.astype(np.int64)) > 0).all()
    assert np.array_equal(np.sort(order), np.arange(len(rec), dtype=np.uint32))
    # the same records from the host, in two slices
    h = len(rec) // 2
    dev.side_components(rec[:h], pool, reset=True)
    dev.side_components(rec[h:], pool, reset=False)
    _, _, lab2 = dev.replay_order(64)
    assert np.array_equal(lab2, want[h:])
    # long lists through the block-per-record path: every record as an "extra" of an empty slice
    dev.side_components(rec[:0], pool, reset=True, extra=rec, extra_pool=pool)
    dev.side_components(rec, pool[:0] if not len(rec) else pool, reset=False)
    _, _, lab3 = dev.replay_order(64)
    assert np.array_equal(lab3, want)
    if len(rec):
        bad = rec.copy()
        bad["entrance"][len(bad) // 2] = 2 * dev.n + 5
        with pytest.raises(hipapi.DeviceError):
            dev.side_components(bad, pool)
        worse = rec.copy()
        eff = np.nonzero(worse["n_list"] > 0)[0]
        if len(eff):
            worse["list_off"][eff[0]] = len(pool) + 7
            with pytest.raises(hipapi.DeviceError):
                dev.side_components(worse, pool)


def test_bfs_tiny_pool_reports_needed_size():
    meta, o, dev, _ = setup_case("tet60k")
    dev.build_adjacency()
    n = dev.count_candidates()
    rec = np.zeros(n, dtype=hipapi.BFS_RECORD)
    pool = np.zeros(8, dtype=np.uint32)
    import ctypes as C
    nr, used = C.c_uint64(), C.c_uint64()
    st = dev.L.pf_bfs_candidates(dev.h, 0, dev.n, rec.ctypes.data, n, pool.ctypes.data, 8, C.byref(nr), C.byref(used))
    assert st == hipapi.PF_ERR_OVERFLOW and used.value > 8
    rec2, pool2 = dev.bfs()
    assert len(pool2) == used.value


@pytest.mark.parametrize("case", ["tet60k", "k31_z16"])
def test_string_cov_matches_oracle(case):
    meta, o, dev, seqs = setup_case(case)
    k = o.k
    rng = np.random.default_rng(11)
    strings = []
    for s in seqs:
        if len(s) >= k + 6:
            a = int(rng.integers(0, len(s) - k - 5))
            strings.append(s[a : a + k + int(rng.integers(0, 6))])
    strings = strings[:2000]
    for low, up in [(5, 1000), (25, 70), (0, 10001)]:
        s, ok, miss = dev.string_cov(strings, low, up)
        for i, t in enumerate(strings):
            es, eok, emiss = o.string_cov(t, low, up)
            assert (int(s[i]), int(ok[i]), int(miss[i])) == (es, eok, emiss)
    # a string with a k-mer that is not in the database
    bogus = b"A" * (k + 2)
    s, ok, miss = dev.string_cov([bogus], 5, 1000)
    assert int(miss[0]) == o.string_cov(bogus, 5, 1000)[2] == 1


def test_two_strand_database_keeps_forward_first_order():
    """A database holding BOTH orientations of a k-mer (not what canonical counting produces) must
    still answer as the reference does: the forward form wins (src/CDBG.cpp:38-56).  The upload-time
    check then disables the canonical-first probe order."""
    meta, o, dev, seqs = setup_case("dip20k")
    kmers, counts, km = synth.read_kmc(meta["db"])
    k = o.k

    def rc(x):
        r = 0
        for j in range(k):
            r |= (3 - ((x >> (2 * j)) & 3)) << (2 * (k - 1 - j))
        return r

    extra = np.array([rc(int(x)) for x in kmers[:500]], dtype=np.uint64)
    both_k = np.concatenate([kmers, extra])
    both_c = np.concatenate([counts, counts[:500] + 1000]).astype(np.uint32)
    d2 = hipapi.Device(0)
    d2.upload_graph(*hipapi.pack_unitigs(seqs), k)
    d2.upload_counts(both_k, both_c, 1, 65535, True)
    c, f = d2.lookup(kmers[:500])
    assert f.all() and np.array_equal(c, counts[:500])            # forward form present -> its own count
    c, f = d2.lookup(extra)
    assert f.all() and np.array_equal(c, counts[:500] + 1000)     # the reverse form queried forward -> its own count
    c, f = d2.lookup(kmers[500:1000])
    assert f.all() and np.array_equal(c, counts[500:1000])


def test_count_table_with_a_minimizer_shared_by_thousands_of_kmers():
    """The count table is addressed by a key's MINIMIZER (pf_device_common.hpp: the minimizer's line of ten keys, its buddy, a
    second pair of lines, then the lines from mix64 of the whole key on).  Keys that share one minimizer by the thousand -- a repeat
    family -- fill those four lines and must still all be found, in both orientations, next to keys that are absent; k below the
    minimizer length + 2 falls back to hashing the canonical k-mer."""
    K_MUL, M = 0x9E3779B1, 16
    inv = pow(K_MUL, -1, 1 << 32)

    def rcn(x, n):
        r = 0
        for j in range(n):
            r |= (3 - ((x >> (2 * j)) & 3)) << (2 * (n - 1 - j))
        return r

    # the canonical 16-mer whose hash c * K_MUL mod 2^32 is smallest: no other 16-mer of a key can beat it, so every key that
    # holds it has it as its minimizer
    mmer = next(c for c in ((h * inv) & 0xFFFFFFFF for h in range(1, 1 << 20)) if c <= rcn(c, M))
    rng = np.random.default_rng(11)
    for k in (25, 31, 18, 17):
        left = (k - M) // 2
        right = k - M - left
        if k >= M + 2:
            flank = rng.integers(0, 1 << (2 * (k - M)), size=6000, dtype=np.uint64)
            keys = ((flank >> np.uint64(2 * right)) << np.uint64(2 * (M + right))) | (np.uint64(mmer) << np.uint64(2 * right)) | (flank & np.uint64((1 << (2 * right)) - 1))
        else:
            keys = rng.integers(0, 1 << (2 * k), size=6000, dtype=np.uint64)
        keys = np.unique(np.concatenate([keys, rng.integers(0, 1 << (2 * k), size=20000, dtype=np.uint64)]))
        can = np.array([min(int(x), rcn(int(x), k)) for x in keys], dtype=np.uint64)
        can = np.unique(can)
        counts = (np.arange(len(can), dtype=np.uint32) * 7 + 3) % 60000 + 1
        d = hipapi.Device(0)
        d.upload_counts(can, counts, 1, 65535, True, k=k)      # before any graph: k comes with the call
        c, f = d.lookup(can)
        assert f.all() and np.array_equal(c, counts), k
        other = np.array([rcn(int(x), k) for x in can[:3000]], dtype=np.uint64)   # queried in the other orientation
        c, f = d.lookup(other)
        assert f.all() and np.array_equal(c, counts[:3000]), k
        absent = np.setdiff1d(rng.integers(0, 1 << (2 * k), size=5000, dtype=np.uint64), np.concatenate([can, np.array([rcn(int(x), k) for x in can], dtype=np.uint64)]))
        c, f = d.lookup(absent)
        assert not f.any(), k
        # a graph of another k is refused, and so is a table of another k beside a graph
        with pytest.raises(RuntimeError):
            d.upload_graph(*hipapi.pack_unitigs([b"ACGT" * 20]), k + 2 if k < 30 else k - 2)


def lattice_gfa(path, k=25, depth=7, seed=3):
    """A superbubble wider than the LDS tables of K-BFS: a binary tree of `depth` levels fanning out
    from one entrance and its mirror image collapsing into one exit (2^depth unitigs in the middle)."""
    rng = np.random.default_rng(seed)

    def rnd(n):
        return bytes(rng.choice(list(b"ACGT"), size=n).tolist())

    segs = [rnd(60)]
    level = [0]  # indices into segs
    # expanding half: node -> two children that start with the node's last k-1 bases + a distinct base
    for _ in range(depth):
        nxt = []
        for i in level:
            for b in (b"A", b"C"):
                segs.append(segs[i][-(k - 1):] + b + rnd(30))
                nxt.append(len(segs) - 1)
        level = nxt
    # collapsing half: two parents are extended so that both end with the same k-1 bases after distinct bases
    while len(level) > 1:
        nxt = []
        for i in range(0, len(level), 2):
            join = rnd(k - 1)
            segs[level[i]] += b"G" + join
            segs[level[i + 1]] += b"T" + join
            segs.append(join + rnd(30))
            nxt.append(len(segs) - 1)
        level = nxt
    with open(path, "wb") as f:
        f.write(b"H\tVN:Z:1.0\tKL:Z:%d\tML:Z:17\n" % k)
        for i, s in enumerate(segs):
            f.write(b"S\t%d\t%s\n" % (i + 1, s))
    return len(segs)


@pytest.mark.parametrize("depth,limit", [(7, 128), (12, 4096)])
def test_bfs_big_tier_matches_oracle(tmp_path, depth, limit):
    """traversals that outgrow the 128-entry LDS tables rerun over global scratch (k_bfs_big, linear
    tables up to 4096 entries), and beyond that over direct-indexed state (k_bfs_huge)"""
    gfa = str(tmp_path / "lattice.gfa")
    n = lattice_gfa(gfa, depth=depth)
    o = pyoracle.Oracle(gfa, None)
    assert o.n == n
    dev = hipapi.Device(0)
    dev.upload_graph(*hipapi.pack_unitigs(o.sequences()), o.k)
    succ, pred = dev.build_adjacency()
    es, ep = o.adjacency()
    assert np.array_equal(succ, es) and np.array_equal(pred, ep)
    rec, pool = dev.bfs()
    big = 0
    for r in rec:
        e = o.extract(int(r["entrance"]))
        assert (int(r["outcome"]), int(r["exit"]), int(r["n_seen"])) == (e["outcome"], e["exit"], len(e["seen"]))
        lst = pool[int(r["list_off"]) : int(r["list_off"]) + int(r["n_list"])]
        if e["outcome"] != 0:
            assert np.array_equal(lst, e["seen"])
        big += len(e["seen"]) > limit
    assert big >= 2  # the whole lattice from either end


@pytest.mark.parametrize("k", [5, 9, 25, 31])
def test_count_table_low_complexity_and_both_orientations(k):
    """The count table is laid out by strand-symmetric minimizers and the minimizer's position in the key: k-mers
    in which the minimizer m-mer repeats (homopolymers, tandem repeats, palindromic halves) must be found from
    either orientation, and every k-mer of a heavy minimizer (thousands of k-mers sharing one bucket) too."""
    rng = np.random.default_rng(k)
    base = rng.integers(0, 4, size=3000, dtype=np.uint8)
    parts = [base]
    for unit in ([0], [0, 1], [0, 1, 2], [3, 0], [0, 0, 3, 3], [1, 2]):
        parts.append(np.array((unit * 64)[:64], dtype=np.uint8))
        parts.append(rng.integers(0, 4, size=40, dtype=np.uint8))
    half = rng.integers(0, 4, size=20, dtype=np.uint8)
    parts.append(np.concatenate([half, (3 - half)[::-1]]))            # reverse-complement palindrome
    # one m-mer followed by many different continuations: a crowded bucket that spills into its neighbours
    core = rng.integers(0, 4, size=13, dtype=np.uint8)
    for _ in range(300):
        parts.append(np.concatenate([rng.integers(0, 4, size=20, dtype=np.uint8), core, rng.integers(0, 4, size=20, dtype=np.uint8)]))
    seq = np.concatenate(parts)
    fw, rc = synth.kmers_u64(seq, k)
    can, idx = np.unique(np.minimum(fw, rc), return_index=True)
    counts = (np.arange(len(can), dtype=np.uint32) % 60000) + 1
    dev = hipapi.Device(0)
    dev.upload_graph(*hipapi.pack_unitigs([synth.BASES[seq].tobytes()]), k)
    dev.upload_counts(can, counts, 1, 65535, True)
    want = counts[np.searchsorted(can, np.minimum(fw, rc))]
    for q in (fw, rc):
        c, f = dev.lookup(q)
        assert f.all() and np.array_equal(c, want)
    absent = np.setdiff1d(rng.integers(0, 1 << (2 * k), size=2000, dtype=np.uint64), np.concatenate([fw, rc]))
    c, f = dev.lookup(absent)
    assert not f.any()
    s, m, miss, st = dev.unitig_cov()
    assert st == 0 and int(s[0]) == int(want.astype(np.uint64).sum()) and int(m[0]) == int(want.min())
    dev.close()


@pytest.mark.parametrize("k,p,cs", [(25, 5, 2), (31, 7, 4), (21, 1, 1), (9, 1, 3)])
def test_kmc_records_are_decoded_on_the_device(k, p, cs, tmp_path):
    """K-KMC against the writer's own arrays: records of a KMC1-layout file, bin-wise tables (KMC2 style), empty entries."""
    rng = np.random.default_rng(k)
    n = 50_000 if k > 9 else 3000
    km = np.unique(rng.integers(0, 1 << (2 * k), size=n, dtype=np.uint64))
    ct = rng.integers(1, min(1 << (8 * cs), 1 << 32), size=len(km), dtype=np.uint64).astype(np.uint32)
    synth.write_kmc1(str(tmp_path / "db"), km, ct, k, counter_size=cs, max_count=(1 << (8 * cs)) - 1, p=p)
    raw = np.fromfile(str(tmp_path / "db.kmc_suf"), dtype=np.uint8)[4:-4]
    sb = (k - p) // 4
    pre = (km >> np.uint64(2 * (k - p))).astype(np.int64)
    lut = np.append(np.searchsorted(pre, np.arange(4 ** p, dtype=np.int64), side="left"), len(km)).astype(np.uint64)
    dev = hipapi.Device()
    got_k, got_c = dev.kmc_decode(raw, len(km), sb, cs, lut, p, k)
    assert np.array_equal(got_k, km) and np.array_equal(got_c, ct)
    # two "bins": the same records split in the middle, each half with its own prefix table (the KMC2 arrangement)
    h = len(km) // 2
    lut2 = np.concatenate([np.minimum(lut[:-1], h), np.maximum(lut[:-1], h), [len(km)]]).astype(np.uint64)
    got_k, got_c = dev.kmc_decode(raw, len(km), sb, cs, lut2, p, k)
    assert np.array_equal(got_k, km) and np.array_equal(got_c, ct)
    # nothing to decode / a table that does not cover the records
    e_k, e_c = dev.kmc_decode(raw[:0], 0, sb, cs, np.zeros(4 ** p + 1, dtype=np.uint64), p, k)
    assert len(e_k) == 0 and len(e_c) == 0
    with pytest.raises(RuntimeError):
        dev.kmc_decode(raw, len(km), sb, cs, lut[:-1], p, k)


def _minz_tables(gfa):
    """(device K-MINZ counters, host counters of pf_host_minz.cpp) of one GFA file"""
    from ploidyfrost_amd import hostapi
    L = hostapi.load_library()
    slots = L.pfh_gfa_minimizer_counts(gfa.encode(), None, 0)
    host = np.zeros(slots, dtype=np.uint8)
    assert L.pfh_gfa_minimizer_counts(gfa.encode(), host.ctypes.data, slots) == slots
    import ctypes as C
    out = os.path.join(os.path.dirname(gfa), "ids_for_minz.txt")
    assert L.pfh_gfa_write_unitig_ids(gfa.encode(), out.encode()) == 0
    seqs = [line.split(b"\t")[1].strip() for line in open(out, "rb")]
    k = int([f for f in open(gfa, "rb").readline().split(b"\t") if f.startswith(b"KL:Z:")][0][5:])
    g = int([f for f in open(gfa, "rb").readline().split(b"\t") if f.startswith(b"ML:Z:")][0][5:])
    dev = hipapi.Device()
    dev.upload_graph(*hipapi.pack_unitigs(seqs), k)
    mx, crowded, table = dev.minimizer_crowding(g, 15, want_table=True)
    if slots == 0:   # no k-length unitig: the host pass has nothing to decide and counts nothing
        assert not any(len(s) == k for s in seqs)
        host = np.zeros(len(table), dtype=np.uint8)
    assert len(table) == len(host)
    return mx, crowded, table, host, C


@pytest.mark.parametrize("case", ["dip20k", "weird12k", "reads10k", "k31_z16", "crowd25"])
def test_minimizer_census_bounds_the_host_counters(case):
    """K-MINZ counts every position the reference's minimizer iterator reports (and, with tied hashes, possibly more): slot by
    slot >= the host pass, equal where no window has tied minima; a crowded graph is seen as crowded."""
    meta = load_case(case)
    mx, crowded, table, host, _ = _minz_tables(meta["gfa"])
    assert np.all(table >= host)
    if host.any():
        # the surplus comes from windows with tied minima (repeats, palindromes): small, and never a deficit
        assert int(table.sum()) - int(host.sum()) <= max(4, int(host.sum()) // 20), (int(table.sum()), int(host.sum()))
    assert mx == table.max() and crowded == int((table >= 15).sum())
    assert (mx >= 15) == bool(meta.get("abundant"))


def test_minimizer_census_on_graphs_built_to_crowd_buckets(tmp_path):
    from test_host_logic_cpu import _crowded_graphs, _write_gfa
    for k, g in ((25, 17), (31, 23), (15, 8)):
        for name, seqs in _crowded_graphs(k, k, g).items():
            gfa = str(tmp_path / ("%s_%d.gfa" % (name, k)))
            _write_gfa(gfa, seqs, k, g)
            mx, crowded, table, host, _ = _minz_tables(gfa)
            assert np.all(table >= host), name
            assert mx >= 15 or host.max() < 15, name


def test_database_without_canonical_counting_is_read_per_orientation():
    """stranded20k: every k-mer stored in both orientations with different counts (kmc -b).  readCov(UnitigMap) looks the
    mapped sequence up as it reads (src/CDBG.cpp:94-117), readCov(string) returns (0, true) without a lookup (:34, :59)."""
    import ctypes as C
    meta, o, dev, seqs = setup_case("stranded20k")
    assert dev.both_strands is False
    o.L.pfo_unitig_cov_oriented.argtypes = [C.c_void_p, C.c_uint32, C.POINTER(C.c_uint64), C.POINTER(C.c_uint32)]
    for rev in (False, True):
        s, m, miss, st = dev.unitig_cov_exact(rev)
        assert st == hipapi.PF_OK and not miss.any()
        for u in range(dev.n):
            es, em = C.c_uint64(), C.c_uint32()
            assert o.L.pfo_unitig_cov_oriented(o.h, 2 * u + int(rev), C.byref(es), C.byref(em)) == 0
            assert (int(s[u]), min(int(m[u]), 10000)) == (es.value, min(em.value, 10000)), (u, rev)
    fwd, rev = dev.unitig_cov_exact(False)[0], dev.unitig_cov_exact(True)[0]
    assert (fwd != rev).any()
    with pytest.raises(hipapi.DeviceError):   # the composite lookup of canonical databases does not apply to this table
        s0, m0, x0 = np.zeros(dev.n, dtype=np.uint64), np.zeros(dev.n, dtype=np.uint32), np.zeros(dev.n, dtype=np.uint8)
        dev._check(dev.L.pf_unitig_cov(dev.h, 0, dev.n, s0.ctypes.data, m0.ctypes.data, x0.ctypes.data))
    ssum, ok, smiss = dev.string_cov([seqs[0][:40], seqs[1][:30]], 5, 1000)
    assert list(ssum) == [0, 0] and list(ok) == [1, 1] and not smiss.any()


def test_gather_over_rccl_with_one_rank():
    """pf_comm_unique_id / pf_comm_init / pf_gather (csrc/pf_gather.hip): the exchange step of a graph cut over GPUs -- an all-gather of
    a few 64-bit words over RCCL, librccl loaded on first use.  One rank is all a one-GPU box allows (RCCL refuses two ranks on a
    device; the N-rank protocol itself is run by test_cli_cuts_one_graph_over_ranks over socket pairs and by tests/test_dist_cpu.py
    over gloo): the communicator is made, the collective runs on the context's stream, the words come back."""
    import ctypes as C
    L = hipapi.load_library()
    dev = hipapi.Device(0)
    ident = (C.c_ubyte * 128)()
    assert L.pf_comm_unique_id(ident) == 0
    assert any(ident)
    assert L.pf_comm_init(dev.h, ident, 0, 1) == 0, L.pf_last_error(dev.h)
    mine = np.arange(18, dtype=np.uint64) * np.uint64(0x0123456789) + np.uint64(7)
    got = np.zeros(18, dtype=np.uint64)
    for _ in range(3):
        assert L.pf_gather(dev.h, mine.ctypes.data, 18, got.ctypes.data) == 0, L.pf_last_error(dev.h)
        assert np.array_equal(got, mine)
    assert L.pf_gather(dev.h, mine.ctypes.data, 65, got.ctypes.data) != 0   # at most 64 words per call
    L.pf_comm_destroy(dev.h)
    assert L.pf_gather(dev.h, mine.ctypes.data, 18, got.ctypes.data) != 0   # no communicator any more
    dev.close()


@pytest.mark.parametrize("n", [1, 63, 2047, 2048, 2049, 100_000, 2048 * 1024, 2048 * 1024 + 1, 5_000_011, 2048 * 16384 + 5])
def test_prefix_sums_and_flag_selection_against_host_arithmetic(n):
    """csrc/pf_scan.hip (tile sums, their scan by one block, the tiles again): one element, a tile to the element, more tiles than
    the offsets kernel takes in one round, the sizes of a pass, more tiles than a block adds up by itself -- the four scans and both selections, each against a sequential
    loop on the host (pf_selftest_scan)"""
    d = hipapi.Device(0)
    try:
        for seed in (1, 2):
            d._check(d.L.pf_selftest_scan(d.h, n, seed))
    finally:
        d.close()
