"""The resident calling pipeline (pf_call_*, reference CDBG::ploidyEstimation_ptr src/CDBG.cpp:1101-1705) through the C ABI,
kernel by kernel: what pf_call_align leaves on the device for every bubble (pf_call_peek) against the oracle --

  K-SCAN / K-PREP   the inner unitigs of a strict bubble, their mean coverages and their order (sortSeq_simple, src/CDBG.cpp:482-551)
  K-PATHS           the path strings of a branching bubble: the reference's two-stack walk (src/CDBG.cpp:1364-1412, restated below
                    in Python from the oracle's adjacency) and sortSeq_branching's order (:417-480)
  K-SNP / K-PAIR / K-BUBBLE   aligned rows, variant columns, allele groups, indel lengths of every bubble against
                    pfo_seq_align = SeqAlign::SequenceAlignment (src/SeqAlign.cpp:550-640), with the routing counters showing that
                    all three kernels took bubbles
  K-SITES           per site of a branching bubble the group coverages: the site strings of src/CDBG.cpp:1448-1600 (restated below)
                    through the oracle's readCov(string) (:29-60)
  K-TEXT            the ten streams against the oracle's files

on designed random graphs: >= 2 000 two-path bubbles (single mismatches, several mismatches, indels up to 8 bp, indels in homopolymers
and short tandem repeats -- co-optimal alignments --, insertions of 20-45 bp: paths beyond 64 bases), >= 500 bubbles of 3-8 paths
(multi-allelic sites: strict bubbles of three and four paths; clusters of variants: branching bubbles), integral and fractional
scores; plus the fixture hex30k and braids (every unitig of a layer followed by both of the next: where the reference's walk loses paths)."""
import os
import struct
import sys

import numpy as np
import pytest

from conftest import GOLDEN, ROOT, compare_outputs, load_case

sys.path.insert(0, os.path.join(ROOT, "oracle"))
import pyoracle  # noqa: E402

pytestmark = pytest.mark.gpu
K = 25
COMP = bytes.maketrans(b"ACGT", b"TGCA")


def test_device_formats_doubles_like_printf():
    """O1: `ostream << double` == printf("%g"), byte for byte -- random bit patterns, the path's magnitudes, rounding ties."""
    from ploidyfrost_amd import hipapi
    rng = np.random.default_rng(7)
    bits = rng.integers(0, 2 ** 64, size=200_000, dtype=np.uint64)
    v = [bits.view(np.float64)]
    e = rng.uniform(-8, 10, size=200_000)
    v.append(10.0 ** e * (1 + rng.random(200_000)))
    v.append(rng.integers(0, 100000, 200_000) / rng.integers(1, 5000, 200_000))   # sum / length
    n6 = rng.integers(100000, 1000000, 20_000).astype(np.float64) + 0.5           # 6 digits and a trailing 5
    for p in range(-12, 13):
        t = n6 * 10.0 ** p
        v += [t, np.nextafter(t, 0), np.nextafter(t, 1e300)]
    v.append(np.array([0.0, -0.0, 1.0, 0.5, 999999.5, 1e6, 1e-4, 9.99995e-5, 1e-5, 5e-324, 1.7976931348623157e308, 2.0 ** 53, 1e22, 1e23,
                       np.inf, -np.inf, 100.0 / 3, 2.0 / 3]))
    vals = np.concatenate(v)
    dev = hipapi.Device(0)
    got = dev.format_doubles(vals)
    dev.close()
    bad = 0
    for x, g in zip(vals.tolist(), got):
        want = b"-nan" if x != x else ("%g" % x).encode()
        if g != want:
            bad += 1
            assert bad < 5, (x, struct.pack("<d", x).hex(), want, g)
    assert bad == 0


# ---- designed graphs -------------------------------------------------------------------------------------------------------
def _rnd(rng, n):
    return rng.integers(0, 4, size=n, dtype=np.uint8)


def _other(rng, b):
    return np.uint8((int(b) + int(rng.integers(1, 4))) & 3)


def two_path_haplotypes(seed, n_var):
    """Two haplotypes that differ at n_var sites at least k + 3 apart: every site is a bubble of two paths.  Returns (haps, kinds)."""
    rng = np.random.default_rng(seed)
    kinds = ["snp", "snp", "snp", "snp2", "mnp", "del1", "ins_small", "ins_small", "tie_hp", "tie_str", "ins_long", "del_snp"]
    ref, alt, used = [_rnd(rng, 80)], [], []
    alt.append(ref[0].copy())
    for i in range(n_var):
        kind = kinds[int(rng.integers(0, len(kinds)))]
        used.append(kind)
        if kind == "snp":
            r = _rnd(rng, 1)
            a = np.array([_other(rng, r[0])], dtype=np.uint8)
        elif kind == "snp2":
            r = _rnd(rng, int(rng.integers(3, 22)))
            a = r.copy()
            a[0], a[-1] = _other(rng, r[0]), _other(rng, r[-1])
        elif kind == "mnp":
            r = _rnd(rng, int(rng.integers(2, 7)))
            a = np.array([_other(rng, x) for x in r], dtype=np.uint8)
        elif kind == "del1":
            r, a = _rnd(rng, 1), np.zeros(0, dtype=np.uint8)
        elif kind == "ins_small":
            r, a = np.zeros(0, dtype=np.uint8), _rnd(rng, int(rng.integers(1, 9)))
        elif kind == "tie_hp":      # one base less in a homopolymer run: as many co-optimal gap places as the run is long
            n = int(rng.integers(3, 8))
            b = _rnd(rng, 1)[0]
            r, a = np.full(n, b, dtype=np.uint8), np.full(n - 1, b, dtype=np.uint8)
        elif kind == "tie_str":     # one unit less in a short tandem repeat
            unit = _rnd(rng, int(rng.integers(2, 4)))
            n = int(rng.integers(3, 5))
            r, a = np.tile(unit, n), np.tile(unit, n - 1)
        elif kind == "ins_long":    # paths of more than 64 bases
            r, a = np.zeros(0, dtype=np.uint8), _rnd(rng, int(rng.integers(20, 46)))
        else:                       # del_snp: a deletion right beside a substitution
            r = _rnd(rng, 3)
            a = np.array([_other(rng, r[0]), r[2]], dtype=np.uint8)
        if rng.random() < 0.5:
            r, a = a, r
        flank = _rnd(rng, int(rng.integers(K + 3, K + 26)))
        ref += [r, flank]
        alt += [a, flank]
    tail = _rnd(rng, 80)
    return [np.concatenate(ref + [tail]), np.concatenate(alt + [tail])], used


def multi_path_haplotypes(seed, ploidy, n_clusters):
    """`ploidy` haplotypes; clusters of one to three sites less than k - 1 apart, each site carried by its own subset of the
    haplotypes, some sites with up to four alleles: bubbles of 3 .. min(ploidy, 8) paths, strict (one multi-allelic site) and
    branching (several sites)."""
    rng = np.random.default_rng(seed)
    haps = [[_rnd(rng, 80)] for _ in range(ploidy)]
    for h in haps[1:]:
        h[0] = haps[0][0].copy()
    for c in range(n_clusters):
        n_sites = int(rng.choice([1, 2, 2, 3]))
        for s in range(n_sites):
            kind = "multi" if n_sites == 1 or rng.random() < 0.3 else ["snp", "snp", "del1", "ins"][int(rng.integers(0, 4))]
            if kind == "multi":
                base = _rnd(rng, 1)[0]
                n_alleles = int(rng.integers(3, 5))
                alleles = [np.array([(int(base) + j) & 3], dtype=np.uint8) for j in range(n_alleles)]
                if rng.random() < 0.25:
                    alleles[-1] = np.zeros(0, dtype=np.uint8)   # one allele is a deletion
                pick = [int(rng.integers(0, n_alleles)) for _ in range(ploidy)]
                for j in range(min(n_alleles, ploidy)):
                    pick[j] = j   # every allele occurs
                for h in range(ploidy):
                    haps[h].append(alleles[pick[h]])
            else:
                if kind == "snp":
                    r = _rnd(rng, 1)
                    a = np.array([_other(rng, r[0])], dtype=np.uint8)
                elif kind == "del1":
                    r, a = _rnd(rng, 1), np.zeros(0, dtype=np.uint8)
                else:
                    r, a = np.zeros(0, dtype=np.uint8), _rnd(rng, int(rng.integers(1, 5)))
                carriers = int(rng.integers(1, (1 << ploidy) - 1))
                for h in range(ploidy):
                    haps[h].append(a if (carriers >> h) & 1 else r)
            if s + 1 < n_sites:
                gap = _rnd(rng, int(rng.integers(2, 16)))
                for h in range(ploidy):
                    haps[h].append(gap)
        flank = _rnd(rng, int(rng.integers(K + 3, K + 26)))
        for h in range(ploidy):
            haps[h].append(flank)
    tail = _rnd(rng, 80)
    return [np.concatenate(h + [tail]) for h in haps]


def write_inputs(tmp, haps, name="g"):
    from ploidyfrost_amd import cdbg_build, synth
    g = cdbg_build.build_cdbg(haps, K, "cuda")
    gfa = os.path.join(tmp, name + ".gfa")
    n = cdbg_build.write_gfa(gfa, g)
    db = os.path.join(tmp, name + "_kmc")
    synth.write_kmc1(db, g["kmers"], synth.synth_counts(g["kmers"], g["mult"]), K)
    return gfa, db, n


# ---- the reference's per-bubble steps, restated for the check (small cases, pure Python) -----------------------------------------
class Graph:
    def __init__(self, o):
        self.seqs = o.sequences()
        self.succ, _ = o.adjacency()
        s, _, _ = o.unitig_cov()
        self.mean = [int(s[u]) / (len(self.seqs[u]) - K + 1) for u in range(o.n)]   # readCov(UnitigMap), src/CDBG.cpp:119

    def mapped(self, ov):
        s = self.seqs[ov >> 1]
        return s if (ov & 1) == 0 else s.translate(COMP)[::-1]

    def len_km(self, ov):
        return len(self.seqs[ov >> 1]) - K + 1

    def successors(self, ov):
        return [int(x) for x in self.succ[ov] if x != 0xFFFFFFFF]


def reference_paths(G, s_ov, t_ov):
    """the two-stack walk of src/CDBG.cpp:1364-1412, statement for statement"""
    minor, major, bstr, out = [s_ov], [], b"", []
    ulen = G.len_km(s_ov)
    while minor:
        umi = minor.pop()
        major.append(umi)
        st = G.mapped(umi)
        L = G.len_km(umi)
        bstr += st[:L]
        if (umi >> 1) == (t_ov >> 1):
            bstr += st[L:]
            out.append(bstr[ulen - 1: ulen - 1 + (len(bstr) - ulen + 1 - L + 1)])
            bstr = bstr[: len(bstr) - len(st)]
            major.pop()
            while major and minor:
                if minor[-1] in G.successors(major[-1]):
                    break
                bstr = bstr[: len(bstr) - G.len_km(major[-1])]
                major.pop()
        else:
            minor.extend(G.successors(umi))
    return out


class OutOfRange(Exception):
    """std::out_of_range from std::string::substr: the reference terminates"""


def substr(s, pos, count=None):
    """std::string::substr(pos, count): the empty string at pos == size(), a throw beyond, a negative count (as size_t) = npos"""
    if pos < 0 or pos > len(s):
        raise OutOfRange()
    return s[pos:] if count is None or count < 0 else s[pos: pos + count]


def site_strings(rows, sites):
    """per site the string of every path, src/CDBG.cpp:1471-1525 (indel sites) and :1559-1596 (others), statement for statement;
    SURVEY.md B.1.  A row that ends in gaps runs out inside the indel loop: substr(size(), 1) is empty, nothing is appended to that
    row and '\\0' (c[0] of the empty string) joins the set of characters."""
    R = len(rows)
    out, indel = [], 0
    for col, is_indel, maxnum, groups, _ok in sites:
        ks = [b""] * R
        if is_indel:
            pos = [col] * R
            app = [b""] * R
            while True:
                last = []
                for p in range(R):
                    c = substr(rows[p], pos[p], 1)
                    while c == b"-":
                        pos[p] += 1
                        c = substr(rows[p], pos[p], 1)
                    pos[p] += 1
                    app[p] += c
                    last.append(c)
                if len(set(last)) != 1:
                    break
            for p in range(R):
                n = len(app[p])
                if indel == 0:
                    ks[p] = substr(rows[p], col - K + n, K - n) + app[p]
                else:
                    tmp = rows[p][:col].replace(b"-", b"")
                    if K - n < 0 or len(tmp) < K - n:     # (size_t < int: a negative right-hand side is huge)
                        s = tmp + app[p]
                        q = pos[p]
                        while len(s) < K:
                            c = substr(rows[p], q, 1)
                            if c != b"-":
                                s += c
                            q += 1
                        ks[p] = s
                    else:
                        ks[p] = tmp[len(tmp) - (K - n):] + app[p] if K - n > 0 else app[p]
            indel += 1
        else:
            for p in range(R):
                if indel == 0:
                    ks[p] = substr(rows[p], col - K + 1, K)
                else:
                    tmp = rows[p][: col + 1].replace(b"-", b"")
                    if len(tmp) < K:
                        q = col + 1
                        while len(tmp) < K:
                            c = substr(rows[p], q, 1)
                            if c != b"-":
                                tmp += c
                            q += 1
                        ks[p] = tmp
                    else:
                        ks[p] = tmp[-K:]
        out.append(ks)
    return out


def expected_alignment(strs, M, D, G):
    e = pyoracle.seq_align(strs, M, D, G)
    if not e["rows"]:
        return None
    part = e["partition"]
    R = len(e["rows"])
    indel_pos = set(e["indel_pos"].tolist())
    sites = [(col, 1 if col in indel_pos else 0, int(max(part[col])), [int(x) for x in part[col]]) for col in range(part.shape[0]) if part[col][R - 1] > 0]
    return dict(rows=e["rows"], sites=sites, indel_len=e["indel_len"].tolist())


def check_pipeline(tmp, gfa, db, z=8, lower=5, upper=1000, scores=(2.0, -1.0, -3.0), max_dfs_paths=300):
    """one graph through oracle and product; returns statistics of what was checked"""
    from ploidyfrost_amd import hipapi, hostapi
    M, D, Gp = scores
    o = pyoracle.Oracle(gfa, db)
    want_dir = os.path.join(tmp, "oracle_%g_%g_%g" % scores)
    os.makedirs(want_dir, exist_ok=True)
    o.run(want_dir, "g", z=z, lower=lower, upper=upper, M=M, D=D, G=Gp)
    G = Graph(o)
    run = hostapi.Run(gfa, db, z=z, M=M, D=D, G=Gp)
    got_dir = os.path.join(tmp, "gpu_%g_%g_%g" % scores)
    os.makedirs(tmp, exist_ok=True)
    run.set_output_dir(got_dir)
    run.set_unitig_id("g")
    run.find_superbubbles("g")
    nb = run.ploidy_select(lower, upper)
    called = run.ploidy_align(0, nb)
    bubbles = hipapi.call_peek(run.device_ctx(), 0)
    assert len(bubbles) == nb
    stats = dict(bubbles=nb, called=called, two_path=0, multi_path=0, strict_multi=0, branching=0, equal_len=0, le64=0, gt64=0, no_alignment=0,
                 site_checks=0, dropped_sites=0, indel_sites=0, max_paths=0)
    for b in bubbles:
        s_ov, t_ov = b["entrance_ov"], b["exit_ov"]
        if b["strict"]:
            # K-SCAN: inner unitigs = the successors of the entrance, their coverages, sortSeq_simple's order
            inner = b["inner"]
            assert sorted(inner) == sorted(G.successors(s_ov)), (s_ov, inner)
            assert b["cov"] == [G.mean[w >> 1] for w in inner], (s_ov, b["cov"])
            key = [(-G.mean[w >> 1], G.seqs[w >> 1]) for w in inner]
            assert all(key[i][0] < key[i + 1][0] or (key[i][0] == key[i + 1][0] and key[i][1] >= key[i + 1][1]) for i in range(len(key) - 1)), (s_ov, key)
            assert b["cov_sum"] == sum(b["cov"]) or abs(b["cov_sum"] - sum(b["cov"])) < 1e-9
            paths = [G.mapped(w) for w in inner]
        else:
            # K-PATHS: the reference's walk, then sortSeq_branching (descending length, ties descending strcmp)
            paths = reference_paths(G, s_ov, t_ov)
            assert len(paths) <= max_dfs_paths
            paths.sort(key=lambda s: (len(s), s), reverse=True)
        e = expected_alignment(paths, M, D, Gp)
        if e is None:
            assert b["rows"] is None, (s_ov, paths)
            stats["no_alignment"] += 1
            continue
        assert b["rows"] is not None, (s_ov, paths)
        assert [r.replace(b"-", b"") for r in b["rows"]] == paths, (s_ov, b["rows"], paths)
        assert b["rows"] == e["rows"], (s_ov, paths, b["rows"], e["rows"])
        assert [x[:4] for x in b["sites"]] == e["sites"], (s_ov, paths, b["sites"], e["sites"])
        assert b["indel_len"] == e["indel_len"], (s_ov, paths)
        n = len(paths)
        stats["max_paths"] = max(stats["max_paths"], n)
        if n == 2:
            stats["two_path"] += 1
            la, lb = len(paths[0]), len(paths[1])
            stats["equal_len"] += la == lb
            stats["le64" if max(la, lb) <= 64 else "gt64"] += 1
        else:
            stats["multi_path"] += 1
            stats["strict_multi"] += bool(b["strict"])
        if not b["strict"]:
            stats["branching"] += 1
            # K-SITES: group coverages of every site from the site strings through readCov(string, lower, upper)
            for (col, is_indel, maxnum, groups, ok), ks, (gcov, total) in zip(b["sites"], site_strings(b["rows"], b["sites"]), b["site_cov"]):
                sets = [set() for _ in range(maxnum)]
                for p, g in enumerate(groups):
                    sets[g - 1].add(ks[p])
                want_ok, want = True, []
                for st in sets:
                    tc = 0.0
                    for s in sorted(st):
                        assert len(s) >= K, (s_ov, col, s)
                        sm, in_range, miss = o.string_cov(s, lower, upper)
                        assert not miss
                        if not in_range:
                            want_ok = False
                        tc += sm / (len(s) - K + 1)
                    want.append(tc)
                stats["indel_sites"] += is_indel
                assert bool(ok) == want_ok, (s_ov, col)
                if want_ok:
                    assert gcov == want and total == sum_in_order(want), (s_ov, col, gcov, want)
                    stats["site_checks"] += 1
                else:
                    stats["dropped_sites"] += 1
    # K-TEXT (and everything before it once more): the ten streams as files
    sizes, counters = run.ploidy_text(0)
    run.ploidy_write("g", np.zeros(10, dtype=np.uint64), sizes, truncate=True)
    bad = compare_outputs(want_dir, got_dir)
    assert not bad, bad
    t = run.times()
    stats.update(snp_jobs=t["snp_jobs"], pair_jobs=t["pair_jobs"], wave_jobs=t["wave_jobs"], stack_jobs=t["stack_jobs"])
    run.close()
    o.close()
    return stats


def sum_in_order(v):
    t = 0.0
    for x in v:
        t += x
    return t


# ---- the tests ------------------------------------------------------------------------------------------------------------------
@pytest.fixture(scope="module")
def two_path_graph(tmp_path_factory):
    tmp = str(tmp_path_factory.mktemp("two"))
    haps, kinds = two_path_haplotypes(31, 2600)
    gfa, db, n = write_inputs(tmp, haps)
    return tmp, gfa, db, kinds


@pytest.fixture(scope="module")
def multi_path_graphs(tmp_path_factory):
    tmp = str(tmp_path_factory.mktemp("multi"))
    out = []
    for ploidy, n_clusters, seed in ((3, 220, 41), (4, 220, 42), (6, 220, 43), (8, 220, 44)):
        sub = os.path.join(tmp, "p%d" % ploidy)
        os.makedirs(sub)
        out.append((sub,) + write_inputs(sub, multi_path_haplotypes(seed, ploidy, n_clusters))[:2])
    return out


@pytest.fixture
def second_pair_tier(monkeypatch):
    """K-PAIR's tier for paths of 65 .. 128 bases (or an indel longer than the first tier's band follows) runs from a few thousand
    bubbles on -- fewer join K-BUBBLE's queues; the tests run it whatever the number"""
    monkeypatch.setenv("PF_PAIR2_MIN", "1")


@pytest.fixture
def stack_everything(monkeypatch):
    """K-STACK is given only bubbles of three and more paths of one length by default (what pays at BASELINE.json's configs[2]); level 3
    also hands it the paths shorter than the first and the two-path bubbles (the one-gap-run certificate): exact as well, and held to
    the oracle here"""
    monkeypatch.setenv("PF_STACK_LEVEL", "3")


def test_two_path_bubbles_through_the_certificates(two_path_graph, second_pair_tier, stack_everything):
    tmp, gfa, db, kinds = two_path_graph
    s = check_pipeline(tmp + "/level3", gfa, db)
    assert s["two_path"] >= 2000 and s["stack_jobs"] >= 400 and s["pair_jobs"] >= 300, s


def test_multi_path_bubbles_through_the_certificates(multi_path_graphs, stack_everything):
    tot = dict(multi_path=0, stack_jobs=0, wave_jobs=0, indel_sites=0)
    for sub, gfa, db in multi_path_graphs:
        s = check_pipeline(sub + "/level3", gfa, db, z=16)
        for k_ in tot:
            tot[k_] += s[k_]
    assert tot["multi_path"] >= 500 and tot["stack_jobs"] >= 400 and tot["indel_sites"] >= 100, tot


def test_two_path_bubbles_kernel_by_kernel(two_path_graph, second_pair_tier):
    tmp, gfa, db, kinds = two_path_graph
    s = check_pipeline(tmp, gfa, db)
    assert s["two_path"] >= 2000, s
    assert s["equal_len"] >= 500 and s["le64"] >= 1500 and s["gt64"] >= 100, s
    # all three alignment kernels took their share: single mismatches (K-SNP), short pairs with one best alignment (K-PAIR),
    # co-optimal ties and long paths (K-BUBBLE)
    print(s)
    assert s["snp_jobs"] >= 300 and s["pair_jobs"] >= 300 and s["wave_jobs"] >= 100, s
    assert s["snp_jobs"] + s["pair_jobs"] + s["wave_jobs"] + s["stack_jobs"] == s["bubbles"], s
    # the long insertions are aligned by K-PAIR's second tier, not by K-BUBBLE: what is left there are the ties
    assert s["wave_jobs"] < s["gt64"] + s["two_path"] // 4 and s["pair_jobs"] + s["stack_jobs"] >= s["gt64"] // 2, s


def test_second_pair_tier_off_gives_the_same(two_path_graph, monkeypatch):
    """... and with that tier's list handed to K-BUBBLE (the default for a short list) nothing changes but the counters"""
    monkeypatch.delenv("PF_PAIR2_MIN", raising=False)
    tmp, gfa, db, kinds = two_path_graph
    s = check_pipeline(tmp + "/off", gfa, db)
    print(s)
    assert s["two_path"] >= 2000, s


@pytest.mark.parametrize("scores", [(1.5, -0.5, -2.25), (1.0, -1.0, -1.0), (3.0, -2.0, -4.0)])
def test_two_path_bubbles_under_other_scores(two_path_graph, scores, second_pair_tier):
    """fractional scores (the NW fill in doubles truncated to int, src/SeqAlign.cpp:480-549; no single-mismatch shortcut) and
    scores under which gaps are cheap"""
    tmp, gfa, db, kinds = two_path_graph
    s = check_pipeline(tmp, gfa, db, scores=scores)
    print(s)
    assert s["two_path"] >= 2000 and s["pair_jobs"] + s["snp_jobs"] + s["stack_jobs"] >= 300 and s["wave_jobs"] >= 100, s


def test_multi_path_bubbles_kernel_by_kernel(multi_path_graphs):
    tot = dict(multi_path=0, strict_multi=0, branching=0, site_checks=0, indel_sites=0, max_paths=0, stack_jobs=0, wave_jobs=0)
    for sub, gfa, db in multi_path_graphs:
        for scores in ((2.0, -1.0, -3.0), (1.5, -0.5, -2.25)):
            s = check_pipeline(sub, gfa, db, z=16, scores=scores)
            for k_ in tot:
                tot[k_] = max(tot[k_], s[k_]) if k_ == "max_paths" else tot[k_] + s[k_]
    assert tot["multi_path"] >= 500 and tot["strict_multi"] >= 100 and tot["branching"] >= 400, tot
    assert tot["site_checks"] >= 500 and tot["indel_sites"] >= 100 and tot["max_paths"] >= 6, tot
    # K-STACK (paths of one length, alignment certified to be the paths stacked) and K-BUBBLE (the rest) both took bubbles
    assert tot["stack_jobs"] >= 300 and tot["wave_jobs"] >= 50, tot


def test_paths_and_sites_of_hex30k(tmp_path):
    """the dense hexaploid fixture: branching bubbles as Bifrost built them"""
    meta = load_case("hex30k")
    op = meta["opts"]
    s = check_pipeline(str(tmp_path), meta["gfa"], meta["db"], z=int(op["-z"]), lower=int(op["-l"]), upper=int(op["-u"]),
                       scores=(float(op["-M"]), float(op["-D"]), float(op["-G"])))
    assert s["branching"] >= 50 and s["site_checks"] >= 100, s
    assert not compare_outputs(os.path.join(meta["dir"], "expected"), os.path.join(str(tmp_path), "gpu_%g_%g_%g" % (float(op["-M"]), float(op["-D"]), float(op["-G"]))))


@pytest.mark.parametrize("layers,seed", [(3, 11), (5, 11), (6, 12), (7, 11), (8, 11)])
def test_paths_of_a_braid(layers, seed, tmp_path):
    """2^layers paths through one bubble, every unitig of a layer followed by both unitigs of the next (tests/test_gpu_end_to_end.py:
    _braid): K-PATHS' walk against the reference's two-stack walk -- which, entered from the far end, leaves paths out (8 layers,
    seed 11: 32 of 256) -- and the rows, sites and group coverages of what it found"""
    from test_gpu_end_to_end import _braid
    gfa, db, n = _braid(tmp_path, K, layers, seed=seed)
    s = check_pipeline(str(tmp_path), gfa, db, z=40, lower=1, upper=1000)
    assert s["branching"] == 1 and s["max_paths"] >= 4, s


@pytest.mark.parametrize("case", ["multi", "hex30k", "braid"])
def test_paths_walked_in_scratch_give_the_same(case, multi_path_graphs, tmp_path, monkeypatch):
    """K-PATHS keeps its two stacks in registers and repeats a bubble with them in global scratch when one outgrows the registers
    (64 / 256 entries); PF_PATHS_SCRATCH=1 sends every bubble down that second road"""
    monkeypatch.setenv("PF_PATHS_SCRATCH", "1")
    if case == "multi":
        sub, gfa, db = multi_path_graphs[0]
        s = check_pipeline(sub + "/scratch", gfa, db, z=16)
        assert s["branching"] >= 100, s
    elif case == "hex30k":
        meta = load_case("hex30k")
        op = meta["opts"]
        sc = (float(op["-M"]), float(op["-D"]), float(op["-G"]))
        s = check_pipeline(str(tmp_path), meta["gfa"], meta["db"], z=int(op["-z"]), lower=int(op["-l"]), upper=int(op["-u"]), scores=sc)
        assert s["branching"] >= 50, s
        assert not compare_outputs(os.path.join(meta["dir"], "expected"), os.path.join(str(tmp_path), "gpu_%g_%g_%g" % sc))
    else:
        from test_gpu_end_to_end import _braid
        gfa, db, n = _braid(tmp_path, K, 7, seed=11)
        s = check_pipeline(str(tmp_path), gfa, db, z=40, lower=1, upper=1000)
        assert s["branching"] == 1 and s["max_paths"] >= 4, s


GAP_FRIENDLY = [(2.0, -1.0, 1.0), (0.0, 0.0, 0.0), (2.0, -1.0, 0.0), (1.0, -1.0, 0.5), (2.0, 2.0, -3.0), (1e5, -1e5, -3.0), (3.0, 1.0, 2.0),
                (-1.0, -7.0, -1.0)]


@pytest.mark.parametrize("scores", GAP_FRIENDLY)
def test_two_path_bubbles_under_gap_friendly_scores(two_path_graph, scores, second_pair_tier):
    """the rest of the region the reference accepts (D <= M, G <= M and nothing else, src/Main.cpp:470-479): a gap that scores
    better than a mismatch, all-zero scores, D = M, a positive gap, magnitudes of 1e5, a negative match.  Rows end in gaps; the last
    indel run is open at the last column; the score-dependent gates of K-SNP / K-STACK / K-PAIR must send such bubbles on."""
    tmp, gfa, db, kinds = two_path_graph
    s = check_pipeline(tmp, gfa, db, scores=scores)
    print(s)
    assert s["two_path"] >= 2000, s


@pytest.mark.parametrize("scores", [(2.0, -1.0, 1.0), (0.0, 0.0, 0.0), (2.0, -1.0, -0.5), (1.0, 0.5, -1.0)])
def test_multi_path_bubbles_under_gap_friendly_scores(multi_path_graphs, scores):
    """the same for branching bubbles: site strings that run to the end of a row (K-SITES), k-mers holding a '-'"""
    tot = dict(multi_path=0, branching=0, site_checks=0, indel_sites=0)
    for sub, gfa, db in multi_path_graphs[:3]:
        s = check_pipeline(sub, gfa, db, z=16, scores=scores)
        for k_ in tot:
            tot[k_] += s[k_]
    assert tot["multi_path"] >= 300 and tot["branching"] >= 200, tot
