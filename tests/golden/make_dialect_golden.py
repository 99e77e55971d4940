#!/usr/bin/env python3
"""Regenerate tests/golden/dialects/<case>__<dialect>/: what the REFERENCE binary writes for the GFA dialects of a fixture's graph.

Runs ONLY in the build container (needs oracle/_ref/PloidyFrost, built by `make -f oracle/Makefile.ref`).  For every base case and
dialect (tests/gfa_dialects.py) it stores
    graph.gfa      the dialect file, byte for byte as the reference was given it
    meta.json      base case (its count database and options are used), the reference's exit status and last lines
    expected/      the twelve files `PloidyFrost -t 1` wrote
The fixtures are data (inputs + the reference's outputs).  tests/test_gpu_gfa.py compares the product with them.

usage: python tests/golden/make_dialect_golden.py
"""
import json
import os
import shutil
import subprocess
import sys
import tempfile

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, os.path.join(ROOT, "tests"))
from conftest import load_case  # noqa: E402
from gfa_dialects import variants  # noqa: E402

REF = os.path.join(ROOT, "oracle", "_ref", "PloidyFrost")
BASES = ["dip20k", "weird12k"]


def main():
    out_root = os.path.join(HERE, "dialects")
    shutil.rmtree(out_root, ignore_errors=True)
    for base in BASES:
        meta = load_case(base)
        with open(meta["gfa"], newline="") as f:
            text = f.read()
        for name, body in variants(text, meta["k"]).items():
            d = os.path.join(out_root, "%s__%s" % (base, name))
            os.makedirs(os.path.join(d, "expected"))
            gfa = os.path.join(d, "graph.gfa")
            with open(gfa, "w", newline="") as f:
                f.write(body)
            with tempfile.TemporaryDirectory() as tmp:
                r = subprocess.run([REF, "-g", gfa, "-d", meta["db"], "-o", "g", "-t", "1"] + meta["args"], cwd=tmp,
                                   stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
                outdir = os.path.join(tmp, "PloidyFrost_output")
                files = sorted(os.listdir(outdir)) if os.path.isdir(outdir) else []
                for fn in files:
                    shutil.copy(os.path.join(outdir, fn), os.path.join(d, "expected", fn))
            same = all(open(os.path.join(d, "expected", fn), "rb").read() == open(os.path.join(meta["dir"], "expected", fn), "rb").read()
                       for fn in files) and len(files) == 12
            with open(os.path.join(d, "meta.json"), "w") as f:
                json.dump({"base": base, "dialect": name, "reference_returncode": r.returncode, "files": len(files),
                           "same_as_base_expected": same, "reference_log_tail": r.stdout.strip().split("\n")[-3:]}, f, indent=1)
            print("%-28s rc %4d  files %2d  same as %s/expected: %s" % (os.path.basename(d), r.returncode, len(files), base, same))


if __name__ == "__main__":
    main()
