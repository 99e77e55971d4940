"""Host-side model of K-STACK's certificates (ploidyfrost_amd/csrc/pf_stack_dev.hpp: stack_certify, indel_place, indel_certify) next
to the full needlemanWunch fill of the reference (src/SeqAlign.cpp:480-549, flags on ties, +1 for continuing a direction): a
certificate may only ever say yes when the full matrix has exactly the claimed path as its one optimal path.  Used by
tests/test_cert_model_cpu.py (soundness on random and adversarial pairs) -- the device code follows this model line by line."""
import random

W = 3
NEG = -(1 << 28)
UP, DIAG, LEFT = 1, 2, 4


def nw_flags(A, B, M, D, G):
    m, n = len(A), len(B)
    S = [[0] * (n + 1) for _ in range(m + 1)]
    F = [[0] * (n + 1) for _ in range(m + 1)]
    for i in range(1, m + 1):
        S[i][0] = G * i
        F[i][0] = UP
    for j in range(1, n + 1):
        S[0][j] = G * j
        F[0][j] = LEFT
    for i in range(1, m + 1):
        for j in range(1, n + 1):
            up = S[i - 1][j] + G + (1 if F[i - 1][j] & UP else 0)
            dg = S[i - 1][j - 1] + (M if A[i - 1] == B[j - 1] else D) + (1 if F[i - 1][j - 1] & DIAG else 0)
            lf = S[i][j - 1] + G + (1 if F[i][j - 1] & LEFT else 0)
            best = max(up, dg, lf)
            S[i][j] = best
            F[i][j] = (UP if up == best else 0) | (DIAG if dg == best else 0) | (LEFT if lf == best else 0)
    return S, F


def unique_path(A, B, M, D, G):
    """the one optimal path as a move string ('D', 'U', 'L' from (0,0) to (m,n)), or None when some cell on the way back carries several flags"""
    _, F = nw_flags(A, B, M, D, G)
    i, j = len(A), len(B)
    moves = []
    while i > 0 or j > 0:
        f = F[i][j]
        if f == DIAG:
            moves.append("D"); i -= 1; j -= 1
        elif f == UP:
            moves.append("U"); i -= 1
        elif f == LEFT:
            moves.append("L"); j -= 1
        else:
            return None
    return "".join(reversed(moves))


def ugen(p, q, n, M, G):
    if p < 0 or q < 0 or q > n:
        return NEG
    if q == 0:
        return G * p
    if p == 0:
        return G * q
    return min(p, q) * (M + 1) + abs(p - q) * (G + 1)


def indel_place(A, B, d):
    n = len(B)
    cur = sum(1 for i in range(n) if A[i + d] == B[i])
    best, best_a, tie = cur, 0, False
    for a in range(1, n + 1):
        i = a - 1
        cur += (1 if A[i] == B[i] else 0) - (1 if A[i + d] == B[i] else 0)
        if cur > best:
            best, best_a, tie = cur, a, False
        elif cur == best:
            tie = True
    return None if tie else best_a


def indel_certify(A, B, a, M, D, G):
    """A the longer (m), B (n), d = m - n >= 0 (d = 0: the diagonal); path: a diagonal moves, d UP moves, the rest diagonal"""
    m, n = len(A), len(B)
    d = m - n
    V = [(G * q if 0 <= q <= n else NEG) for q in range(-W, W + 1)]
    fprev = 0
    for r in range(1, m + 1):
        diag_row = r <= a or r > a + d
        sg = 1 if diag_row else 0
        c = r if r <= a else (a if r <= a + d else r - d)
        N = [NEG] * (2 * W + 1)
        for q in range(-W, W + 1):
            col = c + q
            ui, di, li = q + sg, q + sg - 1, q - 1
            upv = V[ui + W] if -W <= ui <= W else ugen(r - 1, col, n, M, G)
            dgv = V[di + W] if -W <= di <= W else ugen(r - 1, col - 1, n, M, G)
            lfv = N[li + W] if li >= -W else ugen(r, col - 1, n, M, G)
            s = M if (1 <= col <= n and A[r - 1] == B[col - 1]) else D
            cu = upv + G + ((1 if fprev == 2 else 0) if ui == 0 else 1)
            cd = dgv + s + ((1 if fprev == 1 else 0) if di == 0 else 1)
            cl = lfv + G + (0 if li == 0 else 1)
            if q == 0:
                if diag_row:
                    if not (cd > cu and cd > cl) and col > 0:
                        return False
                    v = cd
                else:
                    if not (cu > cd and cu > cl) and col > 0:
                        return False
                    v = cu
            else:
                v = max(cu, cd, cl)
            if col < 0 or col > n:
                v = NEG
            elif col == 0:
                v = G * r
            N[q + W] = v
        V = N
        fprev = 0 if c == 0 else (1 if diag_row else 2)
    return True


def scores_ok(M, D, G):
    return M >= D and M + 1 >= 2 * (G + 1)


def claimed_path(m, n, a):
    d = m - n
    return "D" * a + "U" * d + "D" * (n - a)


# ---- second form: lower bounds beside the upper ones, so that the bonus is credited only where the flag it continues can be set ----
def ugen2(p, q, n, M, G):
    """(lower, upper) of a cell beyond the band"""
    if p < 0 or q < 0 or q > n:
        return NEG, NEG
    if q == 0:
        return G * p, G * p
    if p == 0:
        return G * q, G * q
    return NEG, min(p, q) * (M + 1) + abs(p - q) * (G + 1)


def certify2(A, B, a, M, D, G):
    """as indel_certify; every band cell carries (L, U, P): lower and upper bound of its value and the set of flags it may carry.
    A candidate's bonus is credited in the upper bound when the neighbour MAY carry the continued flag, in the lower bound when it
    MUST (its only possible flag).  A flag may be set at a cell when that candidate's upper bound reaches the cell's lower bound."""
    m, n = len(A), len(B)
    d = m - n
    def border_row0(q):
        if q < 0 or q > n:
            return (NEG, NEG, 0)
        return (G * q, G * q, LEFT if q > 0 else 0)
    V = [border_row0(q) for q in range(-W, W + 1)]
    for r in range(1, m + 1):
        diag_row = r <= a or r > a + d
        sg = 1 if diag_row else 0
        c = r if r <= a else (a if r <= a + d else r - d)
        N = [None] * (2 * W + 1)
        for q in range(-W, W + 1):
            col = c + q
            if col < 0 or col > n:
                N[q + W] = (NEG, NEG, 0)
                continue
            if col == 0:
                N[q + W] = (G * r, G * r, UP)
                continue
            ui, di, li = q + sg, q + sg - 1, q - 1
            def nb(idx, row, column, cur):
                if -W <= idx <= W:
                    return (N if cur else V)[idx + W]
                lo, hi = ugen2(row, column, n, M, G)
                return (lo, hi, UP | DIAG | LEFT if hi > NEG else 0)
            nu, nd, nl = nb(ui, r - 1, col, False), nb(di, r - 1, col - 1, False), (nb(li, r, col - 1, True) if li >= -W else nb(-W - 1, r, col - 1, True))
            s = M if A[r - 1] == B[col - 1] else D
            def cand(nbr, sc, flag):
                lo, hi, pf = nbr
                may = 1 if pf & flag else 0
                must = 1 if pf == flag else 0
                return (lo + sc + must if lo > NEG else NEG, hi + sc + may if hi > NEG else NEG)
            (ul, uu), (dl, du), (ll, lu) = cand(nu, G, UP), cand(nd, s, DIAG), cand(nl, G, LEFT)
            Lc, Uc = max(ul, dl, ll), max(uu, du, lu)
            pf = (UP if uu >= Lc and uu > NEG else 0) | (DIAG if du >= Lc and du > NEG else 0) | (LEFT if lu >= Lc and lu > NEG else 0)
            if q == 0:
                want = DIAG if diag_row else UP
                if pf != want:
                    return False
            N[q + W] = (Lc, Uc, pf)
        V = N
    return True
