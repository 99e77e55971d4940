"""Host-side logic that needs no GPU: input generators, the packed layout handed to
pf_upload_graph, the KMC1 writer/reader pair, and the graph builder used by bench.py (checked
against the Bifrost-built fixture graphs)."""
import os
import sys

import numpy as np
import pytest

from conftest import ROOT, golden_cases, load_case

sys.path.insert(0, os.path.join(ROOT, "oracle"))
import pyoracle  # noqa: E402

from ploidyfrost_amd import cdbg_build, hipapi, synth  # noqa: E402


def rc(s: bytes) -> bytes:
    return s.translate(bytes.maketrans(b"ACGT", b"TGCA"))[::-1]


def test_pack_unitigs_layout():
    seqs = [b"ACGT" * 10 + b"A", b"T" * 33, b"G" * 25]
    words, off, lens = hipapi.pack_unitigs(seqs)
    assert list(lens) == [41, 33, 25] and list(off) == [0, 2, 4, 5]
    for u, s in enumerate(seqs):
        for j, ch in enumerate(s):
            w = int(words[int(off[u]) + j // 32])
            assert (w >> (62 - 2 * (j % 32))) & 3 == b"ACGT".index(ch)
    # padding bits are zero
    assert int(words[1]) & ((1 << (64 - 2 * 9)) - 1) == 0


def test_kmc1_roundtrip_and_oracle_reader(tmp_path):
    rng = np.random.default_rng(3)
    k = 25
    kmers = np.unique(rng.integers(0, 1 << 50, size=5000, dtype=np.uint64))
    counts = rng.integers(1, 60000, size=len(kmers)).astype(np.uint32)
    synth.write_kmc1(str(tmp_path / "db"), kmers, counts, k)
    k2, c2, meta = synth.read_kmc1(str(tmp_path / "db"))
    assert np.array_equal(kmers, k2) and np.array_equal(counts, c2) and meta["k"] == k and meta["both_strands"]
    # the oracle's reader (a restatement of the reference's CheckKmer) finds every record
    seq = "ACGT" * 10
    (tmp_path / "g.gfa").write_text("H\tVN:Z:1.0\tKL:Z:25\tML:Z:17\nS\t1\t%s\n" % seq)
    o = pyoracle.Oracle(str(tmp_path / "g.gfa"), str(tmp_path / "db"))
    import ctypes as C
    for x, c in list(zip(kmers.tolist(), counts.tolist()))[:300]:
        s = bytes(b"ACGT"[(x >> (2 * (k - 1 - j))) & 3] for j in range(k))
        got = C.c_uint32()
        assert o.L.pfo_kmer_count(o.h, s, C.byref(got)) == 1 and got.value == c
        assert o.L.pfo_kmer_count(o.h, rc(s), C.byref(got)) == 1 and got.value == c


@pytest.mark.parametrize("case", ["dip20k", "tet60k", "k31_z16"])
def test_count_database_covers_every_graph_kmer(case):
    meta = load_case(case)
    o = pyoracle.Oracle(meta["gfa"], meta["db"])
    s, m, miss = o.unitig_cov()
    assert not miss.any() and (m >= 1).all() and o.n_kmers == len(synth.read_kmc(meta["db"])[0])


@pytest.mark.parametrize("spec,k", [(synth.HapSpec(60000, 4, seed=5, gap_lo=15, gap_hi=215), 25),
                                    (synth.HapSpec(30000, 6, seed=8, gap_lo=5, gap_hi=60), 25),
                                    (synth.HapSpec(30000, 3, seed=9, gap_lo=10, gap_hi=200, max_ins=40), 31)])
def test_graph_builder_is_a_valid_compacted_dbg(spec, k, tmp_path):
    haps = synth.make_haplotypes(spec)
    g = cdbg_build.build_cdbg(haps, k, "cpu")
    units = cdbg_build.unitig_strings(g)
    km, mult = synth.canonical_counts(haps, k)
    assert np.array_equal(g["kmers"], km) and np.array_equal(g["mult"], mult)
    # every canonical k-mer exactly once over all unitigs
    seen = []
    for s in units:
        codes = np.frombuffer(s.translate(bytes.maketrans(b"ACGT", bytes([0, 1, 2, 3]))), dtype=np.uint8)
        fw, rcv = synth.kmers_u64(codes, k)
        seen.append(np.minimum(fw, rcv))
    seen = np.sort(np.concatenate(seen))
    assert np.array_equal(seen, km)
    # maximal: through the oracle's adjacency no unitig end can be glued to a unique neighbour
    gfa = str(tmp_path / "g.gfa")
    cdbg_build.write_gfa(gfa, g)
    o = pyoracle.Oracle(gfa, None)
    succ, pred = o.adjacency()
    NONE = 0xFFFFFFFF
    outdeg = (succ != NONE).sum(axis=1)
    indeg = (pred != NONE).sum(axis=1)
    for ov in np.nonzero(outdeg == 1)[0]:
        w = int(succ[ov][succ[ov] != NONE][0])
        assert indeg[w] != 1 or (w >> 1) == (ov >> 1), "unitigs %d and %d should have been merged" % (ov >> 1, w >> 1)


def test_shard_ranges_partition_in_order():
    from ploidyfrost_amd import dist as pfdist
    w = np.random.default_rng(1).integers(1, 100, size=1000)
    for world in (1, 2, 3, 8):
        for weights in (None, w):
            cuts = [pfdist.shard_range(1000, r, world, weights) for r in range(world)]
            assert cuts[0][0] == 0 and cuts[-1][1] == 1000
            assert all(a[1] == b[0] for a, b in zip(cuts, cuts[1:]))
        if world > 1:
            loads = [w[a:b].sum() for a, b in [pfdist.shard_range(1000, r, world, w) for r in range(world)]]
            assert max(loads) < 1.3 * w.sum() / world


def test_kmc2_layout_roundtrip_and_signature_lookup(tmp_path):
    """KMC2 layout (what kmc >= 2 writes): the writer's signature binning, the enumerating reader,
    and the oracle's signature -> bin -> LUT -> binary-search lookup (kmc_file.cpp:330-366) agree."""
    import ctypes as C
    rng = np.random.default_rng(5)
    k = 25
    kmers = np.unique(rng.integers(0, 1 << 50, size=4000, dtype=np.uint64))
    counts = rng.integers(1, 60000, size=len(kmers)).astype(np.uint32)
    for sig_len, n_bins in ((7, 11), (9, 37)):
        pre = str(tmp_path / ("db%d" % sig_len))
        synth.write_kmc2(pre, kmers, counts, k, sig_len=sig_len, n_bins=n_bins)
        k2, c2, meta = synth.read_kmc(pre)
        order = np.argsort(k2)
        assert meta["layout"] == "kmc2" and np.array_equal(k2[order], kmers) and np.array_equal(c2[order], counts)
        (tmp_path / "g.gfa").write_text("H\tVN:Z:1.0\tKL:Z:25\tML:Z:17\nS\t1\t%s\n" % ("ACGT" * 10))
        o = pyoracle.Oracle(str(tmp_path / "g.gfa"), pre)
        for x, c in list(zip(kmers.tolist(), counts.tolist()))[:400]:
            s = bytes(b"ACGT"[(x >> (2 * (k - 1 - j))) & 3] for j in range(k))
            got = C.c_uint32()
            assert o.L.pfo_kmer_count(o.h, s, C.byref(got)) == 1 and got.value == c
            assert o.L.pfo_kmer_count(o.h, rc(s), C.byref(got)) == 1 and got.value == c
        assert o.L.pfo_kmer_count(o.h, b"A" * k, C.byref(got)) == 0


# ---- the one known limit of the reproduced unitig numbering: Bifrost's "abundant" k-mers ----------------------
def _rephash(s: bytes) -> int:
    """Bifrost's minimizer hash (bifrost/src/RepHash.hpp) restated for building the test graph."""
    M = (1 << 64) - 1
    hv = [2053695854357871005, 5073395517033431291, 10060236952204337488, 7783083932390163561]
    g, h, ht = len(s), 0, 0
    for i in range(g):
        h = (((h << 1) | (h >> 63)) & M) ^ hv[(s[i] & 6) >> 1]
        ht = (((ht << 1) | (ht >> 63)) & M) ^ hv[((s[g - 1 - i] ^ 4) & 6) >> 1]
    lo, hi = min(h, ht), max(h, ht)
    a = ((lo & 0xFFFFFFFF) << 32) | (hi & 0xFFFFFFFF)
    b = ((hi >> 32) << 32) | (lo >> 32)

    def mix(x, y):
        r = (x & M) * (y & M)
        return (r & M) ^ (r >> 64)
    return mix(0xE7037ED1A0B428DB ^ 16, mix(a ^ 0xE7037ED1A0B428DB, b ^ 0xA0761D6478BD642F))


def _graph_with_a_crowded_minimizer(tmp_path, n_sharing):
    k, g = 25, 17
    rng = np.random.default_rng(2)
    rnd = lambda n: synth.BASES[rng.integers(0, 4, size=n, dtype=np.uint8)].tobytes()  # noqa: E731
    core = min((rnd(g) for _ in range(4000)), key=_rephash)      # a g-mer that wins the minimizer race in its k-mers
    seqs = [rnd(60) for _ in range(5)]
    for i in range(n_sharing):
        off = 1 + i % 7                                          # minimizers may not start at offset 0 or k-g
        seqs.append(rnd(off) + core + rnd(k - g - off))
    seqs += [rnd(25) for _ in range(10)]
    gfa = str(tmp_path / "g.gfa")
    with open(gfa, "wb") as f:
        f.write(b"H\tVN:Z:1.0\tBV:Z:1.0.6\tKL:Z:%d\tML:Z:%d\n" % (k, g))
        for i, s in enumerate(seqs):
            f.write(b"S\t%d\t%s\n" % (i + 1, s))
    return gfa, seqs, k


def test_abundant_kmer_suspects_are_counted(tmp_path):
    from ploidyfrost_amd import hostapi
    L = hostapi.load_library()
    gfa, _, _ = _graph_with_a_crowded_minimizer(tmp_path, 60)
    assert L.pfh_gfa_abundant_suspects(gfa.encode()) == 60 - 15
    (tmp_path / "few").mkdir()
    gfa2, _, _ = _graph_with_a_crowded_minimizer(tmp_path / "few", 15)
    assert L.pfh_gfa_abundant_suspects(gfa2.encode()) == 0
    for case in ("dip20k", "weird12k", "col4_mix"):
        assert L.pfh_gfa_abundant_suspects(load_case(case)["gfa"].encode()) == 0


def test_abundant_kmer_count_agrees_with_the_reference(tmp_path):
    """Where the reference binary exists: it numbers exactly the first 15 sharers in file order and moves the others
    to the end -- the count reported by the product is the number of displaced unitigs."""
    import subprocess
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import pyoracle
    if not os.path.exists(pyoracle.REF_BIN):
        pytest.skip("reference binary (oracle/_ref) not built here")
    from ploidyfrost_amd import hostapi
    gfa, seqs, k = _graph_with_a_crowded_minimizer(tmp_path, 40)
    inv = {65: 0, 67: 1, 71: 2, 84: 3}
    km, mult = synth.canonical_counts([np.array([inv[c] for c in s], dtype=np.uint8) for s in seqs], k)
    synth.write_kmc1(str(tmp_path / "db"), km, synth.synth_counts(km, mult), k)
    # (the reference may die after setUnitigId on this bubble-free graph -- it divides by the number of sites, 0 here)
    subprocess.run([pyoracle.REF_BIN, "-g", gfa, "-d", str(tmp_path / "db"), "-o", "x", "-t", "1"], cwd=tmp_path,
                   stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
    ref = [line.split("\t")[1].strip() for line in open(tmp_path / "PloidyFrost_output" / "x_Unitig_Id.txt")]
    mine = [s.decode() for s in pyoracle.Oracle(gfa, str(tmp_path / "db")).sequences()]
    assert sorted(ref) == sorted(mine)
    moved = hostapi.load_library().pfh_gfa_abundant_suspects(gfa.encode())
    assert moved == 25
    assert ref[: 5 + 15] == mine[: 5 + 15]                       # long unitigs and the first 15 sharers keep their rank
    assert set(ref[-moved:]) == set(mine[5 + 15: 5 + 40])       # the others are numbered last (in hash order)


def test_work_pool(tmp_path):
    """The host layer's persistent fork-join pool, on its own (no GPU, no library): tests/cpp/test_workpool.cpp."""
    import shutil
    import subprocess
    if not shutil.which("g++"):
        pytest.skip("no g++")
    exe = str(tmp_path / "test_workpool")
    subprocess.run(["g++", "-O2", "-std=c++17", "-pthread", "-I", os.path.join(ROOT, "ploidyfrost_amd", "csrc", "host"),
                    os.path.join(ROOT, "tests", "cpp", "test_workpool.cpp"), "-o", exe], check=True)
    r = subprocess.run([exe], stdout=subprocess.PIPE, text=True, timeout=120)
    assert r.returncode == 0 and r.stdout.strip() == "ok", r.stdout
