"""K-STACK's certificates (ploidyfrost_amd/csrc/pf_stack_dev.hpp) as a host-side model (tests/models/cert_model.py, the device code
line by line) against the full needlemanWunch fill of the reference (src/SeqAlign.cpp:480-549: flags on ties, +1 for continuing a
direction): whenever the certificate says yes, the full matrix must have exactly the claimed path -- down the diagonal, one run of
gaps, down the diagonal -- as its ONE optimal path.  Random pairs, low-complexity pairs, homopolymer and tandem-repeat indels (where
it must say no), several scorings.  (That it says yes often enough is measured, not asserted: tests/test_gpu_call.py counts what
K-STACK takes.)"""
import os
import random
import sys

import pytest

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "models"))
from cert_model import certify2, claimed_path, indel_certify, indel_place, scores_ok, unique_path  # noqa: E402


def _pairs(seed, n_cases):
    rng = random.Random(seed)

    def rnd(n, alphabet="ACGT"):
        return "".join(rng.choice(alphabet) for _ in range(n))

    def other(c):
        return rng.choice([x for x in "ACGT" if x != c])

    for _ in range(n_cases):
        X, Y = rnd(rng.randint(8, 26)), rnd(rng.randint(8, 26))
        kind = rng.randrange(10)
        if kind == 0:      # substitutions
            mid = rnd(rng.randint(1, 12))
            alt = "".join(other(c) if rng.random() < 0.4 else c for c in mid)
            yield X + mid + Y, X + alt + Y
        elif kind == 1:    # one deleted base
            yield X + rnd(1) + Y, X + Y
        elif kind == 2:    # an insertion
            yield X + rnd(rng.randint(1, 9)) + Y, X + Y
        elif kind == 3:    # a long insertion
            yield X + rnd(rng.randint(15, 40)) + Y, X + Y
        elif kind == 4:    # homopolymer: several co-optimal places
            b, k = rng.choice("ACGT"), rng.randint(2, 7)
            yield X + b * k + Y, X + b * (k - 1) + Y
        elif kind == 5:    # tandem repeat
            u, k = rnd(rng.randint(2, 3)), rng.randint(2, 4)
            yield X + u * k + Y, X + u * (k - 1) + Y
        elif kind == 6:    # deletion beside a substitution
            r3 = rnd(3)
            yield X + r3 + Y, X + other(r3[0]) + r3[2] + Y
        elif kind == 7:    # low complexity
            lc = rnd(rng.randint(30, 60), "AC")
            p = rng.randint(3, len(lc) - 4)
            yield lc, lc[:p] + lc[p + 1:]
        elif kind == 8:    # low complexity, substitution
            lc = rnd(rng.randint(30, 60), "AAC")
            p = rng.randint(3, len(lc) - 4)
            yield lc, lc[:p] + other(lc[p]) + lc[p + 1:]
        else:              # unrelated ends: the gap run at the very start / end
            yield rnd(rng.randint(1, 5)) + X + Y, X + Y


@pytest.mark.parametrize("scores", [(2, -1, -3), (1, -1, -1), (3, -2, -4), (2, -1, -2), (5, 4, -1), (2, 2, -3)])
def test_certificates_never_claim_what_the_full_matrix_does_not_have(scores):
    M, D, G = scores
    assert scores_ok(M, D, G)
    said_yes = 0
    for A, B in _pairs(hash(scores) & 0xFFFF, 350):
        if len(A) < len(B):
            A, B = B, A
        d = len(A) - len(B)
        a = indel_place(A, B, d) if d else len(B)
        if a is None:
            continue
        for cert in (indel_certify, certify2):
            if cert(A, B, a, M, D, G):
                said_yes += cert is indel_certify
                assert unique_path(A, B, M, D, G) == claimed_path(len(A), len(B), a), (cert.__name__, A, B, a, scores)
    assert said_yes >= 30, said_yes


def test_transposed_pair_has_the_transposed_path():
    """the device certifies a pair whose SECOND path is the longer with the roles swapped: the recurrence is symmetric in its two
    strings, so the one optimal path of (B, A) is that of (A, B) with UP and LEFT exchanged"""
    rng = random.Random(3)
    for _ in range(150):
        X, Y = "".join(rng.choice("ACGT") for _ in range(20)), "".join(rng.choice("ACGT") for _ in range(20))
        A, B = X + "".join(rng.choice("ACGT") for _ in range(rng.randint(1, 6))) + Y, X + Y
        p, q = unique_path(A, B, 2, -1, -3), unique_path(B, A, 2, -1, -3)
        assert (p is None) == (q is None)
        if p:
            assert q == p.replace("U", "l").replace("L", "U").replace("l", "L")
