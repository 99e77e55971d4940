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


def golden_cases():
    return sorted(d for d in os.listdir(GOLDEN) if os.path.isfile(os.path.join(GOLDEN, d, "args.json")))


def load_case(name):
    d = os.path.join(GOLDEN, name)
    with open(os.path.join(d, "args.json")) as f:
        meta = json.load(f)
    meta["dir"] = d
    meta["gfa"] = os.path.join(d, "graph.gfa")
    meta["db"] = os.path.join(d, "db")
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
