#!/usr/bin/env python3
"""Fixtures for the reference's `-t N` (N > 1) output format: runs oracle/_ref/PloidyFrost -t 2 on golden cases (in this
container, where the reference tree exists) and commits a CANONICAL form of its files per case as tests/golden/<case>/expected_t2.json.
The threaded reference writes rows in an order that depends on thread timing and numbers bubbles in that order, so the files are
not comparable byte for byte between two of its own runs; the canonical form removes exactly that freedom and nothing else:

  super_bubble.txt      rows without their BubbleId, sorted; the ids must be 0 .. n-1
  alignseq.txt          rows grouped by var_count (a bubble's rows are written in one piece, path order kept), the var_count
                        replaced by the bubble's "entrance:exit" unitig ids; groups sorted; the var_counts must be 0 .. n-1
  *cov.txt              rows grouped by var_count in file order, var_count replaced the same way; groups sorted
  *fre.txt, allele_frequency.txt   sorted lines (no bubble id in these files)

Every case is run several times; a case whose canonical form is not the same in every run (the commit order of findSuperBubble
can change the bubble set itself) is not used.   usage: tests/golden/make_t2_golden.py [case ...]"""
import hashlib
import json
import os
import subprocess
import sys
import tempfile

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, os.path.join(ROOT, "oracle"))
sys.path.insert(0, os.path.join(ROOT, "tests"))
import pyoracle  # noqa: E402
from t2_canonical import canonical  # noqa: E402

CASES = ["dip20k", "tet60k", "hex30k", "weird12k", "k31_z16", "frac_scores"]


def main():
    cases = sys.argv[1:] or CASES
    for case in cases:
        d = os.path.join(HERE, case)
        if not os.path.isdir(d):
            print(case, "absent")
            continue
        meta = json.load(open(os.path.join(d, "args.json")))
        forms = []
        for rep in range(5):
            with tempfile.TemporaryDirectory() as tmp:
                cmd = [pyoracle.REF_BIN, "-g", os.path.join(d, "graph.gfa"), "-d", os.path.join(d, "db"), "-o", "g", "-t", "2"] + meta["args"]
                r = subprocess.run(cmd, cwd=tmp, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
                if r.returncode != 0:
                    print(case, "reference failed:", r.stdout[-500:])
                    forms = None
                    break
                forms.append(canonical(os.path.join(tmp, "PloidyFrost_output"), "g"))
        if not forms:
            continue
        if any(f != forms[0] for f in forms[1:]):
            print(case, "unstable between runs of the threaded reference: not used")
            continue
        out = {k: {"lines": v.count("\n"), "sha256": hashlib.sha256(v.encode()).hexdigest()} for k, v in forms[0].items()}
        json.dump(out, open(os.path.join(d, "expected_t2.json"), "w"), indent=1, sort_keys=True)
        print(case, "ok", {k: v["lines"] for k, v in out.items()})


if __name__ == "__main__":
    main()
