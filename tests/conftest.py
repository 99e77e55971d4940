import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
GOLDEN = os.path.join(ROOT, "tests", "golden")
OUTPUT_SUFFIXES = ["Unitig_Id", "super_bubble", "alignseq", "allele_frequency", "bicov", "bifre", "tricov", "trifre",
                   "tetracov", "tetrafre", "pentacov", "pentafre"]


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def _cases(colored, abundant=False):
    out = []
    for d in sorted(os.listdir(GOLDEN)):
        a = os.path.join(GOLDEN, d, "args.json")
        if os.path.isfile(a):
            with open(a) as f:
                meta = json.load(f)
            if bool(meta.get("colored")) == colored and bool(meta.get("abundant")) == abundant:
                out.append(d)
    return out


def golden_cases():
    """single-sample cases (reference src/CDBG.cpp)"""
    return _cases(False)


def abundant_cases():
    """single-sample cases in which Bifrost numbers some k-length unitigs last ("abundant" k-mers); the expected files
    come from the reference, the oracle does not restate that part of the numbering"""
    return _cases(False, True)


def colored_cases():
    """multi-sample cases (reference src/CCDBG.cpp)"""
    return _cases(True)


def dialect_cases():
    """tests/golden/dialects/<base>__<dialect>: a base fixture's graph in another GFA dialect, with what the REFERENCE binary
    wrote for it (tests/golden/make_dialect_golden.py)"""
    d = os.path.join(GOLDEN, "dialects")
    return sorted(x for x in os.listdir(d) if os.path.isfile(os.path.join(d, x, "meta.json"))) if os.path.isdir(d) else []


def load_dialect(name):
    d = os.path.join(GOLDEN, "dialects", name)
    with open(os.path.join(d, "meta.json")) as f:
        dm = json.load(f)
    meta = load_case(dm["base"])
    meta["gfa"] = os.path.join(d, "graph.gfa")
    meta["dir"] = d
    meta["dialect"] = dm
    return meta


def load_case(name):
    d = os.path.join(GOLDEN, name)
    with open(os.path.join(d, "args.json")) as f:
        meta = json.load(f)
    meta["dir"] = d
    meta["gfa"] = os.path.join(d, "graph.gfa")
    meta["db"] = os.path.join(d, "db")
    if meta.get("colored"):
        meta["colors"] = os.path.join(d, "graph.bfg_colors")
        meta["colors_dump"] = os.path.join(d, "colors.txt")
        meta["dbs"] = [os.path.join(d, "db%d" % i) for i in range(meta["n_colors"])]
    opts = {"-l": "10", "-u": "1000", "-z": "8", "-M": "2", "-D": "-1", "-G": "-3"}
    a = meta["args"]
    for i in range(0, len(a), 2):
        opts[a[i]] = a[i + 1]
    meta["opts"] = opts
    return meta


def compare_outputs(expected_dir, got_dir, prefix_expected="g", prefix_got="g"):
    bad = []
    for suf in OUTPUT_SUFFIXES:
        e = os.path.join(expected_dir, "%s_%s.txt" % (prefix_expected, suf))
        g = os.path.join(got_dir, "%s_%s.txt" % (prefix_got, suf))
        if not os.path.exists(g):
            bad.append(suf + " (missing)")
            continue
        with open(e, "rb") as fe, open(g, "rb") as fg:
            if fe.read() != fg.read():
                bad.append(suf)
    return bad


# ---- tests/golden/scores/<case>__<tag>: the reference under scores outside the default fixtures' region --------------------

def score_cases():
    d = os.path.join(GOLDEN, "scores")
    return sorted(x for x in os.listdir(d) if os.path.isfile(os.path.join(d, x, "meta.json"))) if os.path.isdir(d) else []


def load_score_case(name):
    """the base case's inputs + what the reference made of them under meta['extra_args'] (tests/golden/make_score_golden.py)"""
    d = os.path.join(GOLDEN, "scores", name)
    with open(os.path.join(d, "meta.json")) as f:
        sm = json.load(f)
    meta = load_case(sm["case"])
    a = sm["extra_args"]
    for i in range(0, len(a), 2):
        meta["opts"][a[i]] = a[i + 1]
    meta["score"] = sm
    meta["score_dir"] = d
    meta["ub_cells"] = {k: set(v) for k, v in sm["ub_cells"].items()}
    return meta


def mask_ub(data, lines, colored):
    """a *cov.txt file with the indel-length field of the given (1-based) lines replaced by '?': the cells whose value the
    reference leaves undefined (oracle/pf_oracle_align.hpp: indel_len_at).  Rows end in a tab; counted from the end the field is
    ... strict, LEN, var_count, sites, dist, '' (single-sample) or ... strict, LEN, var_count, sites, Cramer V, dist, ''."""
    rows = data.split(b"\n")
    at = -6 if colored else -5
    for ln in lines:
        f = rows[ln - 1].split(b"\t")
        f[at] = b"?"
        rows[ln - 1] = b"\t".join(f)
    return b"\n".join(rows)


def compare_score_outputs(meta, got_dir, mask=True):
    """the twelve files of a score case against the reference's (files equal to the base case's are taken from there), the
    undefined cells masked on both sides; with mask=False every byte counts (product against oracle: both define those cells)."""
    sm = meta["score"]
    bad = []
    for suf in OUTPUT_SUFFIXES:
        f = "g_%s.txt" % suf
        src = os.path.join(meta["dir"], "expected", f) if suf in sm["same_as_base_case"] else os.path.join(meta["score_dir"], "expected", f)
        e = open(src, "rb").read()
        g = open(os.path.join(got_dir, f), "rb").read()
        if mask and suf in meta["ub_cells"]:
            if e.count(b"\n") != g.count(b"\n"):
                bad.append(suf + " (rows)")
                continue
            e, g = mask_ub(e, meta["ub_cells"][suf], sm["colored"]), mask_ub(g, meta["ub_cells"][suf], sm["colored"])
        if e != g:
            bad.append(suf)
    return bad
