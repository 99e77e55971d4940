"""The product's own .bfg_colors reader (ploidyfrost_amd/csrc/host/pf_host_colors.cpp) against the real
Bifrost's reading of the same files (tests/golden/<case>/colors.txt, written by oracle/_ref/colors_dump)."""
import os

import numpy as np
import pytest

from conftest import colored_cases, load_case
from ploidyfrost_amd import hostapi


def read_dump(path):
    rows, names, n_colors = [], [], 0
    with open(path) as f:
        for line in f:
            p = line.rstrip("\n").split("\t")
            if p[0] == "#colors":
                n_colors = int(p[1])
            elif p[0] == "#name":
                names.append(p[1])
            else:
                km = int(p[2])
                bits = []
                for tok in p[5].split(","):
                    bits.append(np.ones(km, np.uint8) if tok == "F" else np.zeros(km, np.uint8) if tok == "E"
                                else np.frombuffer(tok.encode(), np.uint8) - ord("0"))
                rows.append((km, int(p[3]), int(p[4]), np.stack(bits)))
    return n_colors, names, rows


@pytest.mark.parametrize("case", colored_cases())
@pytest.mark.parametrize("threads", [1, 3])
def test_reader_matches_bifrost_dump(case, threads):
    meta = load_case(case)
    n_colors, names, rows = read_dump(meta["colors_dump"])
    c = hostapi.Colors(meta["gfa"], meta["colors"], threads)
    assert c.n_colors == n_colors == meta["n_colors"] and c.n == len(rows)
    assert [os.path.basename(n) for n in c.names] == names
    for u, (km, size, n_full, bits) in enumerate(rows):
        got, gsize, gfull = c.unitig(u)
        assert got.shape == (n_colors, km)
        assert np.array_equal(got, bits), "unitig %d" % (u + 1)
        assert gsize == size and gfull == n_full


def test_missing_colour_file_is_an_error(tmp_path):
    meta = load_case(colored_cases()[0])
    with pytest.raises(RuntimeError, match="Could not open"):
        hostapi.Colors(meta["gfa"], str(tmp_path / "nope.bfg_colors"))
    # a graph without DataAccessor tags cannot be joined to colour sets (ColoredCDBG.tcc:511-527)
    plain = load_case("dip20k")
    with pytest.raises(RuntimeError, match="DataAccessor"):
        hostapi.Colors(plain["gfa"], meta["colors"])


# ---- every encoding UnitigColors::write has, through this repository's writer ---------------------------
def _synthetic_colored_graph(tmp_path, k=25, n_colors=5, seed=3, shared=False, shared_refs=None):
    """Isolated random unitigs of many lengths with assorted colour patterns, one encoding each.  shared: the file also holds a
    SharedUnitigColors section (three sets of different encodings, with reference counts); shared_refs: unitigs written as
    references to a shared set (the flag word 0x5 alone)."""
    from ploidyfrost_amd import bfg_colors, synth
    rng = np.random.default_rng(seed)
    lengths = [k] * 6 + [k + 1, k + 5, 40, 60, 100, 133, 700, 3000, 3001, 9000, 30000] * 2 + [k] * 3
    lengths.sort(key=lambda L: L == k)  # Bifrost numbers long unitigs first
    seqs = [rng.integers(0, 4, size=L, dtype=np.uint8) for L in lengths]
    want, sets, heads = [], [], []
    encs = list(bfg_colors.ENCODINGS)
    for u, s in enumerate(seqs):
        km = len(s) - k + 1
        pres = np.zeros((n_colors, km), dtype=np.uint8)
        for c in range(n_colors):
            mode = int(rng.integers(0, 5))
            if mode == 0:
                pres[c] = 1
            elif mode == 1:
                pass
            elif mode == 2:      # a prefix / suffix, the shape a sample ending inside the unitig leaves
                a = int(rng.integers(0, km + 1))
                pres[c, :a] = 1
            elif mode == 3:
                a = int(rng.integers(0, km + 1))
                pres[c, a:] = 1
            else:                # scattered
                pres[c] = rng.random(km) < 0.5
        if len(s) == k:          # a k-mer long unitig is stored canonical
            fw, rc = synth.kmers_u64(s, k)
            if rc[0] < fw[0]:
                seqs[u] = s = (3 - s)[::-1]
        want.append(pres)
        ids = [int(c * km + p) for c in range(n_colors) for p in np.nonzero(pres[c])[0]]
        sets.append(bfg_colors.encode_set(ids, encs[u % len(encs)], n_kmers=km, n_colors=n_colors))
        heads.append(synth.kmers_u64(s, k)[0][0])
    heads = bfg_colors.left_align(np.array(heads, dtype=np.uint64), k)
    sizes = np.array([len(s) for s in seqs])
    colors = str(tmp_path / "g.bfg_colors")
    extra = {}
    if shared:
        extra["shared_sets"] = [(bfg_colors.encode_set([0, 3, 4], "bitvector"), 2), (bfg_colors.encode_set(list(range(0, 4000, 3)), "roaring_array"), 1),
                                (bfg_colors.encode_set([7], "single"), 0)]
        extra["shared_refs"] = set(shared_refs or ())
    da = bfg_colors.write_bfg_colors(colors, heads, sizes, k, ["c%d" % i for i in range(n_colors)], sets, overflow_every=7,
                                     slack=0.8, **extra)
    gfa = str(tmp_path / "g.gfa")
    with open(gfa, "wb") as f:
        f.write(b"H\tVN:Z:1.0\tBV:Z:1.0.6\tKL:Z:%d\tML:Z:17\n" % k)
        for u, s in enumerate(seqs):
            f.write(b"S\t%d\t%s\tDA:Z:%d\n" % (u + 1, synth.BASES[s].tobytes(), da[u]))
    assert (da == 0).any() and (da > 1).any()  # overflow table and several hash seeds in play
    return gfa, colors, want, [encs[u % len(encs)] for u in range(len(seqs))]


def test_reader_understands_every_encoding(tmp_path):
    gfa, colors, want, encs = _synthetic_colored_graph(tmp_path)
    c = hostapi.Colors(gfa, colors, 2)
    assert c.n == len(want)
    for u, pres in enumerate(want):
        got, size, n_full = c.unitig(u)
        assert np.array_equal(got, pres), "unitig %d (%s)" % (u + 1, encs[u])
        assert size == int(pres.sum())
        km = pres.shape[1]
        assert n_full == (int((pres.sum(axis=1) == km).sum()) if encs[u] == "pair" else 0)


def test_shared_colour_sets_are_read_like_the_reference_reads_them(tmp_path):
    """A file with a SharedUnitigColors section (sz_shared_cs > 0; DataStorage.tcc:811-820, 832-917): the shared sets are parsed and,
    as in the reference, linked to nothing -- every unitig's colours are what they are without the section.  A unitig whose own set
    is a REFERENCE to a shared set (flag 5) has no colours anybody could tell: UnitigColors::write stores the flag alone and
    UnitigColors::read has no case for it (ColorSet.cpp:1190-1194, 1228-1283) -- an error that says so, where the reference
    dereferences a null pointer."""
    gfa, colors, want, encs = _synthetic_colored_graph(tmp_path, shared=True)
    c = hostapi.Colors(gfa, colors, 2)
    assert c.n == len(want)
    for u, pres in enumerate(want):
        got, size, n_full = c.unitig(u)
        assert np.array_equal(got, pres), "unitig %d (%s)" % (u + 1, encs[u])
        assert size == int(pres.sum())
    sub = tmp_path / "refs"
    sub.mkdir()
    gfa2, colors2, _, _ = _synthetic_colored_graph(sub, shared=True, shared_refs=[4])
    with pytest.raises(RuntimeError, match="reference to a shared colour set"):
        hostapi.Colors(gfa2, colors2, 2)
    # the real Bifrost (where built) reads the first file the same way
    import subprocess
    import sys
    from conftest import ROOT
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import pyoracle
    if os.path.exists(pyoracle.REF_COLORS_DUMP):
        dump = tmp_path / "dump.txt"
        with open(dump, "w") as f:
            subprocess.run([pyoracle.REF_COLORS_DUMP, gfa, colors], check=True, stdout=f)
        _, _, rows = read_dump(str(dump))
        assert len(rows) == len(want)
        for u, (km, size, n_full, bits) in enumerate(rows):
            assert np.array_equal(bits, want[u]) and size == int(want[u].sum()), "unitig %d (%s)" % (u + 1, encs[u])
        # ... and does not survive the file with a reference
        r = subprocess.run([pyoracle.REF_COLORS_DUMP, gfa2, colors2], stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
        assert r.returncode != 0


def test_writer_is_read_the_same_way_by_the_real_bifrost(tmp_path):
    """Only where oracle/_ref exists (the build container): the real library's view of this repository's files."""
    import subprocess
    import sys
    from conftest import ROOT
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import pyoracle
    if not os.path.exists(pyoracle.REF_COLORS_DUMP):
        pytest.skip("reference Bifrost (oracle/_ref/colors_dump) not built here")
    gfa, colors, want, encs = _synthetic_colored_graph(tmp_path)
    dump = tmp_path / "dump.txt"
    with open(dump, "w") as f:
        subprocess.run([pyoracle.REF_COLORS_DUMP, gfa, colors], check=True, stdout=f)
    n_colors, names, rows = read_dump(str(dump))
    assert len(rows) == len(want)
    for u, (km, size, n_full, bits) in enumerate(rows):
        assert np.array_equal(bits, want[u]), "unitig %d (%s)" % (u + 1, encs[u])
        assert size == int(want[u].sum())
        assert n_full == (int((want[u].sum(axis=1) == km).sum()) if encs[u] == "pair" and km > 1 else 0)


def test_kmer_hash_matches_the_python_restatement():
    from ploidyfrost_amd import bfg_colors
    L = hostapi.load_library()
    rng = np.random.default_rng(1)
    xs = rng.integers(0, 1 << 63, size=64, dtype=np.uint64) << np.uint64(1)
    for seed in (0, 1, 0xDEADBEEFCAFEF00D):
        v = bfg_colors.kmer_hash_np(xs, seed)
        for x, h in zip(xs, v):
            assert L.pfh_bifrost_kmer_hash(int(x), seed) == int(h) == bfg_colors.kmer_hash(int(x), seed)
