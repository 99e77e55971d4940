"""K-BUBBLE (pf_align_bubbles): the complete SeqAlign::SequenceAlignment on the device against the
oracle -- aligned rows, variant columns, allele groups, indel flags and indel lengths, for 2..7
paths per bubble, text paths and paths decoded from the packed graph.  Bit-exact."""
import os
import sys

import numpy as np
import pytest

from conftest import ROOT, load_case

sys.path.insert(0, os.path.join(ROOT, "oracle"))
import pyoracle  # noqa: E402

from ploidyfrost_amd import hipapi  # noqa: E402
from test_gpu_align import mutate  # noqa: E402

pytestmark = pytest.mark.gpu


def expected(strs, M, D, G):
    e = pyoracle.seq_align(strs, M, D, G)
    if not e["rows"]:
        return None
    part = e["partition"]
    R = len(e["rows"])
    sites = []
    indel_pos = set(e["indel_pos"].tolist())
    for col in range(part.shape[0]):
        if part[col][R - 1] > 0:
            g = part[col].tolist()
            sites.append((col, 1 if col in indel_pos else 0, max(g), g))
    return dict(rows=e["rows"], sites=sites, indel_len=e["indel_len"].tolist())


def make_bubbles(seed, n, lo, hi, max_paths, low_complexity=0.2):
    rng = np.random.default_rng(seed)
    out = []
    for _ in range(n):
        L = int(rng.integers(lo, hi))
        if rng.random() < low_complexity:
            base = bytes(rng.choice(list(b"AC"), size=L, p=[0.75, 0.25]).tolist())
        else:
            base = bytes(rng.choice(list(b"ACGT"), size=L).tolist())
        npaths = int(rng.integers(2, max_paths + 1))
        paths = set()
        tries = 0
        while len(paths) < npaths and tries < 50:
            paths.add(mutate(rng, base, int(rng.integers(0, 3)), int(rng.integers(0, 3)), 4))
            tries += 1
        paths = sorted(paths, key=lambda s: (-len(s), s), reverse=False)
        paths.sort(key=lambda s: (len(s), s), reverse=True)  # sortSeq_branching order
        if len(paths) >= 2:
            out.append(paths)
    return out


def check(dev, bubbles, M=2.0, D=-1.0, G=-3.0):
    got = dev.align_bubbles(bubbles, M, D, G)
    n_multi = 0
    for b, g in zip(bubbles, got):
        e = expected(b, M, D, G)
        if e is None:
            assert g is None, b
            continue
        assert g is not None, b
        assert g["rows"] == e["rows"], (b, g["rows"], e["rows"])
        assert g["sites"] == e["sites"], (b, g["sites"], e["sites"])
        assert g["indel_len"] == e["indel_len"], b
        n_multi += len(b) > 2
    return n_multi


@pytest.fixture(scope="module")
def dev():
    d = hipapi.Device(0)
    meta = load_case("tet60k")
    o = pyoracle.Oracle(meta["gfa"], meta["db"])
    d.upload_graph(*hipapi.pack_unitigs(o.sequences()), o.k)
    d._oracle = o
    return d


def test_two_paths(dev):
    check(dev, make_bubbles(1, 500, 30, 70, 2))


def test_progressive_rounds(dev):
    assert check(dev, make_bubbles(2, 400, 30, 90, 6)) > 100


def test_many_paths_and_long(dev):
    check(dev, make_bubbles(3, 60, 100, 260, 7) + make_bubbles(4, 6, 300, 600, 4))


def test_fractional_scores(dev):
    b = make_bubbles(5, 300, 30, 80, 5)
    check(dev, b, 1.5, -0.5, -2.25)
    check(dev, b, 3.0, -2.0, -1.0)


def test_paths_decoded_from_the_graph(dev):
    """strict bubbles of the fixture: the inner unitigs, both strands, straight from the 2-bit graph"""
    o = dev._oracle
    o.find_superbubbles()
    flags, plus, minus = o.state()
    succ, _ = o.adjacency()
    seqs = o.sequences()
    comp = bytes.maketrans(b"ACGT", b"TGCA")
    bubbles_ov, bubbles_txt = [], []
    for u in range(o.n):
        for strand, bit in ((0, 0x10), (1, 0x08)):
            if flags[u] & bit:
                inner = [int(x) for x in succ[2 * u + strand] if x != hipapi.NONE]
                bubbles_ov.append(inner)
                bubbles_txt.append([seqs[w >> 1] if (w & 1) == 0 else seqs[w >> 1].translate(comp)[::-1] for w in inner])
    assert len(bubbles_ov) > 100
    got = dev.align_bubbles(bubbles_ov)
    for b, g in zip(bubbles_txt, got):
        e = expected(b, 2.0, -1.0, -3.0)
        assert (g is None) == (e is None)
        if e:
            assert g["rows"] == e["rows"] and g["sites"] == e["sites"] and g["indel_len"] == e["indel_len"]


def test_no_alignment_survives(dev):
    rng = np.random.default_rng(9)
    bubbles = []
    for _ in range(30):
        base = bytes(rng.choice(list(b"ACGT"), size=150).tolist())
        bubbles.append([base, mutate(rng, base, 0, 9, 3)])
    got = dev.align_bubbles(bubbles)
    exp = [expected(b, 2.0, -1.0, -3.0) for b in bubbles]
    assert [g is None for g in got] == [e is None for e in exp] and any(e is None for e in exp)
    check(dev, bubbles)


@pytest.mark.parametrize("scores", [(2, -1, -3), (1, -1, -1), (3, -2, -4), (5, 4, -1), (1, -3, -2), (2, 0, -2), (2, -1, -1), (4, -1, -2),
                                    (2, 2, -3), (7, -5, -9)])
def test_single_snp_pairs_under_many_scorings(dev, scores):
    """Equal-length paths differing in one base skip the dynamic programming when the scores make the diagonal the
    strict optimum (the shortcut in round 0 of K-BUBBLE); with other scores they take the full path.  Both must give
    the oracle's answer -- including SNPs at the very first and last base, in homopolymers and in tandem repeats."""
    M, D, G = (float(x) for x in scores)
    rng = np.random.default_rng(abs(hash(scores)) % (1 << 31))
    bubbles = []
    for i in range(120):
        L = int(rng.integers(3, 90))
        kind = i % 4
        if kind == 0:
            base = bytes(rng.choice(list(b"ACGT"), size=L).tolist())
        elif kind == 1:
            base = bytes(rng.choice(list(b"AC"), size=L, p=[0.85, 0.15]).tolist())
        elif kind == 2:
            unit = bytes(rng.choice(list(b"ACGT"), size=int(rng.integers(1, 4))).tolist())
            base = (unit * L)[:L]
        else:
            base = bytes([b"ACGT"[int(rng.integers(0, 4))]]) * L
        pos = 0 if i % 7 == 0 else (L - 1 if i % 7 == 1 else int(rng.integers(0, L)))
        alt = bytes([rng.choice([c for c in b"ACGT" if c != base[pos]])])
        other = base[:pos] + alt + base[pos + 1:]
        pair = sorted([base, other], reverse=True)
        bubbles.append(pair)
        if i % 5 == 0:  # a third path makes the progressive rounds start from the shortcut's rows
            third = mutate(rng, base, 1, 1, 3)
            if third not in pair:
                trio = sorted(pair + [third], key=lambda s: (len(s), s), reverse=True)
                bubbles.append(trio)
    check(dev, bubbles, M, D, G)
