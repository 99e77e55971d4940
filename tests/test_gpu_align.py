"""K-ALN (pf_align_batch) against the oracle's needlemanWunch + traceback: same kept alignments,
same order, same scores and gap positions -- including many-co-optimal, gap-budget-exhausted,
profile-row ('-' in A) and fractional-score cases.  Bit-exact."""
import os
import sys

import numpy as np
import pytest

from conftest import ROOT

sys.path.insert(0, os.path.join(ROOT, "oracle"))
import pyoracle  # noqa: E402

from ploidyfrost_amd import hipapi  # noqa: E402

pytestmark = pytest.mark.gpu


def mutate(rng, s: bytes, n_snp, n_indel, max_indel=6):
    a = bytearray(s)
    for _ in range(n_snp):
        i = int(rng.integers(0, len(a)))
        a[i] = b"ACGT"[(b"ACGT".index(a[i]) + int(rng.integers(1, 4))) % 4]
    for _ in range(n_indel):
        i = int(rng.integers(1, len(a) - 1))
        ln = int(rng.integers(1, max_indel + 1))
        if rng.random() < 0.5:
            del a[i : i + ln]
        else:
            a[i:i] = bytes(rng.choice(list(b"ACGT"), size=ln).tolist())
    return bytes(a)


def make_pairs(seed, n, lo, hi, profile=False):
    rng = np.random.default_rng(seed)
    pairs = []
    for _ in range(n):
        L = int(rng.integers(lo, hi))
        kind = rng.random()
        if kind < 0.2:  # low complexity: many co-optimal placements
            base = bytes(rng.choice(list(b"AC"), size=L, p=[0.8, 0.2]).tolist())
        elif kind < 0.3:  # tandem repeat
            unit = bytes(rng.choice(list(b"ACGT"), size=int(rng.integers(1, 5))).tolist())
            base = (unit * (L // len(unit) + 1))[:L]
        else:
            base = bytes(rng.choice(list(b"ACGT"), size=L).tolist())
        a = mutate(rng, base, int(rng.integers(0, 3)), int(rng.integers(0, 3)))
        b = mutate(rng, base, int(rng.integers(0, 4)), int(rng.integers(0, 4)))
        if profile:  # A is a row of an earlier alignment: sprinkle gap runs
            aa = bytearray(a)
            for _ in range(int(rng.integers(1, 4))):
                i = int(rng.integers(1, len(aa)))
                aa[i:i] = b"-" * int(rng.integers(1, 5))
            a = bytes(aa)
        if len(a) and len(b):
            pairs.append((a, b))
    return pairs


def check(dev, pairs, M=2.0, D=-1.0, G=-3.0):
    got = dev.align_batch(pairs, M, D, G)
    assert len(got) == len(pairs)
    n_multi = 0
    for (a, b), g in zip(pairs, got):
        e = pyoracle.pairwise(a, b, M, D, G)
        assert len(g) == len(e), (a, b, len(g), len(e))
        for (ga, gb, gg, gs, gn, gi), (ea, eb, eg, es, en, ei) in zip(g, e):
            assert (ga, gb, list(gg), gs, gn, gi) == (ea, eb, eg, es, en, ei), (a, b)
        n_multi += len(e) > 1
    return n_multi


@pytest.fixture(scope="module")
def dev():
    return hipapi.Device(0)


def test_small_jobs_lds_tier(dev):
    pairs = make_pairs(1, 600, 20, 60)
    assert check(dev, pairs) >= 1  # some jobs keep several co-optimal alignments


def test_medium_and_large_jobs(dev):
    pairs = make_pairs(2, 60, 100, 250) + make_pairs(3, 8, 300, 700)
    check(dev, pairs)


def test_profile_rows(dev):
    check(dev, make_pairs(4, 300, 20, 80, profile=True))


def test_fractional_scores(dev):
    pairs = make_pairs(5, 300, 20, 70) + make_pairs(6, 100, 20, 70, profile=True)
    check(dev, pairs, 1.5, -0.5, -2.25)
    check(dev, pairs, 3.0, -2.0, -1.0)


def test_gap_budget_exhausted(dev):
    rng = np.random.default_rng(7)
    pairs = []
    for _ in range(40):
        base = bytes(rng.choice(list(b"ACGT"), size=120).tolist())
        pairs.append((base, mutate(rng, base, 0, 8, 3)))  # > 5 gap opens: usually no alignment survives
    got = dev.align_batch(pairs)
    exp = [pyoracle.pairwise(a, b) for a, b in pairs]
    assert [len(g) for g in got] == [len(e) for e in exp]
    assert any(len(e) == 0 for e in exp)
    check(dev, pairs)


def test_degenerate_lengths(dev):
    check(dev, [(b"A", b"A"), (b"A", b"C"), (b"ACGT", b"A"), (b"A", b"ACGTT"), (b"AAAAAAAA", b"AAAA"), (b"-A-", b"A")])
