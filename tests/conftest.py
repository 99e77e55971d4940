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
