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


# ---- unitig numbering: Bifrost's "abundant" k-mers are numbered last, in its hash table's slot order ---------------
_rephash = synth.bifrost_minimizer_hash
_B = b"ACGT"


def _write_gfa(path, seqs, k, g):
    with open(path, "wb") as f:
        f.write(b"H\tVN:Z:1.0\tBV:Z:1.0.6\tKL:Z:%d\tML:Z:%d\n" % (k, g))
        for i, s in enumerate(seqs):
            f.write(b"S\t%d\t%s\n" % (i + 1, s))


def _crowded_graphs(seed, k, g):
    """S-line sets that crowd minimizer buckets in different ways (not valid de Bruijn graphs -- numbering only)."""
    rng = np.random.default_rng(seed)
    rnd = lambda n: bytes(_B[i] for i in rng.integers(0, 4, size=n))  # noqa: E731
    cores = sorted((rnd(g) for _ in range(3000)), key=_rephash)[:6]   # g-mers that win the minimizer race in their k-mers
    core = cores[0]
    out = {}
    # k-length unitigs only: the first 15 sharers keep their rank, the others move to the end
    out["shorts"] = [rnd(60) for _ in range(5)] + [rnd(1 + i % (k - g - 1)) + core + rnd(k - g - 1 - i % (k - g - 1)) for i in range(50)] \
        + [rnd(k) for _ in range(10)]
    # long unitigs through the same minimizer, interleaved: they are redirected to their next-best minimizer
    seqs = []
    for i in range(80):
        if i % 3 == 0:
            seqs.append(rnd(int(rng.integers(5, 40))) + core + rnd(int(rng.integers(5, 40))))
        else:
            o = 1 + int(rng.integers(0, k - g - 1))
            seqs.append(rnd(o) + core + rnd(k - g - o))
        if i % 7 == 0:
            seqs.append(rnd(int(rng.integers(k, 90))))
    out["mixed"] = seqs
    # several crowded minimizers per unitig: redirect chains
    seqs = []
    for i in range(300):
        parts = [rnd(int(rng.integers(1, 8)))]
        for _ in range(int(rng.integers(1, 5))):
            parts += [cores[int(rng.integers(0, len(cores)))], rnd(int(rng.integers(0, 9)))]
        s = b"".join(parts)
        seqs.append(s + rnd(max(0, k - len(s))))
    out["chains"] = list(dict.fromkeys(seqs))
    # more abundant k-mers than the hash table's first 1024 slots take: it is rebuilt twice
    seqs = []
    for i in range(2600):
        c = cores[int(rng.integers(0, len(cores)))]
        o = 1 + int(rng.integers(0, k - g - 1))
        seqs.append(rnd(o) + c + rnd(k - g - o))
    out["many"] = list(dict.fromkeys(seqs))
    # low complexity: the same g-mer several times inside one k-mer (tied minimizer positions)
    seqs = []
    for i in range(150):
        unit = rnd(int(rng.integers(1, 4)))
        s = bytearray((unit * 40)[: int(rng.integers(k, k + 30))])
        for _ in range(int(rng.integers(0, 3))):
            s[int(rng.integers(0, len(s)))] = _B[int(rng.integers(0, 4))]
        seqs.append(bytes(s))
    out["lowcomplexity"] = list(dict.fromkeys(seqs))
    return out


def _ids_written_by_the_loader(L, gfa, tmp_path):
    out = str(tmp_path / "ids.txt")
    assert L.pfh_gfa_write_unitig_ids(gfa.encode(), out.encode()) == 0
    with open(out, "rb") as f:
        return f.read()


def test_unitig_numbering_of_every_fixture(tmp_path):
    """Unitig_Id.txt of the reference, from the GFA file alone; the crowded fixture has abundant k-mers, the others none."""
    from conftest import abundant_cases, colored_cases
    from ploidyfrost_amd import hostapi
    L = hostapi.load_library()
    for case in golden_cases() + colored_cases() + abundant_cases():
        meta = load_case(case)
        with open(os.path.join(meta["dir"], "expected", "g_Unitig_Id.txt"), "rb") as f:
            assert _ids_written_by_the_loader(L, meta["gfa"], tmp_path) == f.read(), case
        n = L.pfh_gfa_abundant_kmers(meta["gfa"].encode())
        if meta.get("abundant"):
            assert n > 0 and L.pfh_gfa_numbering_replays(meta["gfa"].encode()) >= 1
        else:
            assert n == 0
    assert abundant_cases()


def test_host_loader_reads_gfa_dialects_like_the_reference(tmp_path):
    """Unitig_Id.txt as the REFERENCE wrote it for every dialect fixture (tests/golden/dialects, made by make_dialect_golden.py):
    GFA 1 / 2, tags, lower case, interleaved lines, no final line feed, and CRLF, whose '\\r' after a sequence is that sequence's
    last base (A in a segment longer than k, T in a k-length one; bifrost/src/GFA_Parser.cpp:497-520)."""
    from conftest import dialect_cases, load_dialect
    from ploidyfrost_amd import hostapi
    L = hostapi.load_library()
    assert len(dialect_cases()) >= 14
    for case in dialect_cases():
        meta = load_dialect(case)
        with open(os.path.join(meta["dir"], "expected", "g_Unitig_Id.txt"), "rb") as f:
            assert _ids_written_by_the_loader(L, meta["gfa"], tmp_path) == f.read(), case


def test_abundant_kmers_move_to_the_end(tmp_path):
    from ploidyfrost_amd import hostapi
    L = hostapi.load_library()
    k, g = 25, 17
    seqs = _crowded_graphs(2, k, g)["shorts"]
    gfa = str(tmp_path / "g.gfa")
    _write_gfa(gfa, seqs, k, g)
    assert L.pfh_gfa_abundant_kmers(gfa.encode()) == 50 - 15
    ids = [line.split(b"\t")[1] for line in _ids_written_by_the_loader(L, gfa, tmp_path).splitlines()]
    canon = lambda s: min(s, s.translate(bytes.maketrans(b"ACGT", b"TGCA"))[::-1])  # noqa: E731
    assert ids[:5] == seqs[:5]
    kept = [canon(s) for s in seqs[5: 5 + 15] + seqs[55:]]        # the first 15 sharers and the unrelated k-mers keep file order
    assert ids[5: 5 + 25] == kept
    assert sorted(ids[30:]) == sorted(canon(s) for s in seqs[20:55])
    _write_gfa(gfa, seqs[:5 + 15] + seqs[55:], k, g)               # 15 sharers only: nothing is crowded yet
    assert L.pfh_gfa_abundant_kmers(gfa.encode()) == 0


@pytest.mark.parametrize("k,g", [(25, 17), (31, 23), (21, 13), (15, 8)])
def test_abundant_numbering_agrees_with_the_reference(tmp_path, k, g):
    """Where the reference binary exists (this container): its Unitig_Id.txt on graphs built to crowd minimizer buckets."""
    import subprocess
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import pyoracle
    if not os.path.exists(pyoracle.REF_BIN):
        pytest.skip("reference binary (oracle/_ref) not built here")
    from ploidyfrost_amd import hostapi
    L = hostapi.load_library()
    inv = {65: 0, 67: 1, 71: 2, 84: 3}
    replays = 0
    for name, seqs in _crowded_graphs(k, k, g).items():
        d = tmp_path / name
        d.mkdir()
        gfa = str(d / "g.gfa")
        _write_gfa(gfa, seqs, k, g)
        km, mult = synth.canonical_counts([np.array([inv[c] for c in s], dtype=np.uint8) for s in seqs], k)
        synth.write_kmc1(str(d / "db"), km, synth.synth_counts(km, mult), k)
        # (the reference may die after setUnitigId on these bubble-free graphs -- it divides by the number of sites, 0 here)
        subprocess.run([pyoracle.REF_BIN, "-g", gfa, "-d", str(d / "db"), "-o", "x", "-t", "1"], cwd=d,
                       stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
        with open(d / "PloidyFrost_output" / "x_Unitig_Id.txt", "rb") as f:
            assert _ids_written_by_the_loader(L, gfa, d) == f.read(), name
        replays = max(replays, L.pfh_gfa_numbering_replays(gfa.encode()))
    assert replays >= 2   # redirects into buckets the first replay had not followed


@pytest.mark.parametrize("k,g", [(25, 17), (31, 23), (15, 8)])
def test_numbering_replay_with_its_inputs_handed_in(tmp_path, k, g):
    """On a GPU box the replay's two passes over every unitig come from K-MINZ (pf_minimizer_replay_inputs): a counter table that bounds
    the host's from above, and flags that cover the unitigs the host would flag.  Here the host's own counters, raised by 0 / 1 / 3
    (saturating), with every unitig flagged: the numbering must not change -- upper bounds only make the replay follow more buckets."""
    from ploidyfrost_amd import hostapi
    L = hostapi.load_library()
    for name, seqs in _crowded_graphs(k + 1, k, g).items():
        gfa = str(tmp_path / ("%s.gfa" % name))
        _write_gfa(gfa, seqs, k, g)
        want = str(tmp_path / "want.txt")
        assert L.pfh_gfa_write_unitig_ids(gfa.encode(), want.encode()) == 0
        for bump in (0, 1, 3):
            got = str(tmp_path / "got.txt")
            assert L.pfh_gfa_write_unitig_ids_given_inputs(gfa.encode(), got.encode(), bump) == 0
            assert open(got, "rb").read() == open(want, "rb").read(), (name, bump)


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


def test_packed_alignseq_round_trip(tmp_path):
    """alignseq.txt leaves the device packed (csrc/pf_alnpack.hpp) and becomes text in the host's writer: the host half on its own,
    tests/cpp/test_alnpack.cpp (the device half is held to the reference's files by every end-to-end test, both ways: PF_ALIGNSEQ_ASCII)."""
    import shutil
    import subprocess
    if not shutil.which("g++"):
        pytest.skip("no g++")
    exe = str(tmp_path / "test_alnpack")
    subprocess.run(["g++", "-O2", "-std=c++17", "-I", os.path.join(ROOT, "ploidyfrost_amd", "csrc"), os.path.join(ROOT, "tests", "cpp", "test_alnpack.cpp"), "-o", exe],
                   check=True)
    r = subprocess.run([exe], stdout=subprocess.PIPE, text=True, timeout=120)
    assert r.returncode == 0 and r.stdout.strip() == "ok", r.stdout


def test_packed_numeric_streams_round_trip(tmp_path):
    """the nine numeric streams leave the device at four bits a character (csrc/pf_nibble.hpp, K-NIB) and become text in the host's
    writer: the host half on its own, tests/cpp/test_nibble.cpp (the device half is held to the reference's files by every
    end-to-end test, both ways: PF_NUMERIC_ASCII)."""
    import shutil
    import subprocess
    if not shutil.which("g++"):
        pytest.skip("no g++")
    exe = str(tmp_path / "test_nibble")
    subprocess.run(["g++", "-O2", "-std=c++17", "-I", os.path.join(ROOT, "ploidyfrost_amd", "csrc"), os.path.join(ROOT, "tests", "cpp", "test_nibble.cpp"), "-o", exe],
                   check=True)
    r = subprocess.run([exe], stdout=subprocess.PIPE, text=True, timeout=120)
    assert r.returncode == 0 and r.stdout.strip() == "ok", r.stdout


def test_bench_cov_roofline_picks_the_streaming_kernel_of_the_workload():
    """bench.py `roofline_k_cov`: K-COV for the single-sample workload (with the committed PMC traffic), K-COV-C for the colored."""
    import bench
    k = {"k_cov": {"avg_ms": 0.1, "achieved_GBps": 2000.0}, "k_cov_colored": {"avg_ms": 0.3, "achieved_GBps": 2400.0}, "k_bfs": {"avg_ms": 1.0}}
    a, b = bench.cov_roofline(k, False, 123), bench.cov_roofline(k, True, 123)
    assert a["kernel"] == "k_cov" and a["frac"] == 0.25 and a["bound"] == "hbm" and a["peak"] == 8000.0
    assert a["traffic"] is None   # the committed PMC profile is of another graph: no figure rather than a stale one
    assert b["kernel"] == "k_cov_colored" and b["frac"] == 0.3 and b["traffic"] is None
    assert bench.cov_roofline({"k_bfs": {"avg_ms": 1.0}}, False, 123) is None
    assert bench.algorithmic_bytes("k_cov", {"kmers": 64, "unitigs": 2}) == 64 * (4 + 1 / 8 + 1 / 16) + 32
