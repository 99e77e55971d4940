#!/usr/bin/env python3
"""Goldens for the part of the score region the default fixtures do not reach: gap-friendly and degenerate scores the reference
accepts (src/Main.cpp:470-479 asks only D <= M and G <= M).  Runs ONLY in the build container (needs oracle/_ref/PloidyFrost).

For every (case, tag) of SCORE_CASES the reference binary is run `-t 1` on the case's committed inputs with the extra options and
    tests/golden/scores/<case>__<tag>/expected/   the files the reference wrote that differ from <case>/expected/
    tests/golden/scores/<case>__<tag>/meta.json   extra options, return code, the files equal to the base case's, and `ub_cells`
are written.  `ub_cells` = the cells of the *cov.txt files whose value the reference leaves UNDEFINED (it reads `indel_len` one
element past its end for an indel run still open at the last column: heap garbage that differs from run to run): listed by the
oracle (PFO_UB_LOG), and checked here two ways -- a second run of the reference differs from the first in no other byte, and the
oracle equals the reference in every other byte.  A run in which the reference dies (exit(1) on a k-mer it cannot find, SIGSEGV
on that same read when the vector is empty) keeps its return code and last message; its partly flushed files are not fixtures.

usage: python tests/golden/make_score_golden.py [case__tag ...]"""
import json
import os
import shutil
import subprocess
import sys
import tempfile

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, os.path.join(ROOT, "oracle"))
import pyoracle  # noqa: E402

TAGS = {
    "G1": ["-G", "1"],                       # a gap scores better than a mismatch: rows end in gaps
    "Z0": ["-M", "0", "-D", "0", "-G", "0"],  # every alignment scores 0
    "DeqM": ["-D", "2"],                     # a mismatch scores as a match
    "GeqM": ["-G", "2"],                     # a gap scores as a match
    "G05": ["-G", "0.5"],
    "big": ["-M", "100000", "-D", "-100000", "-G", "1"],
}
CASES = ["hex30k", "k31_z16", "weird12k", "col4_mix"]
SCORE_CASES = [(c, t) for c in CASES for t in ("G1", "Z0", "DeqM", "GeqM")] + [("tet60k", "G05"), ("hex30k", "big")]


def reference_cmd(case, meta, tmp):
    d = os.path.join(HERE, case)
    if meta.get("colored"):
        lst, cut = os.path.join(tmp, "dbs.txt"), os.path.join(tmp, "cut.txt")
        open(lst, "w").write("".join(os.path.join(d, "db%d" % i) + "\n" for i in range(meta["n_colors"])))
        open(cut, "w").write("".join("%d\t%d\n" % tuple(c) for c in meta["cutoffs"]))
        ref = ["-g", os.path.join(d, "graph.gfa"), "-f", os.path.join(d, "graph.bfg_colors"), "-d", lst, "-C", cut]
        ora = ["-g", os.path.join(d, "graph.gfa"), "-f", os.path.join(d, "colors.txt"), "-d", lst, "-C", cut]
    else:
        ref = ora = ["-g", os.path.join(d, "graph.gfa"), "-d", os.path.join(d, "db")]
    return ref, ora


def make(case, tag):
    meta = json.load(open(os.path.join(HERE, case, "args.json")))
    extra = TAGS[tag]
    out = os.path.join(HERE, "scores", "%s__%s" % (case, tag))
    shutil.rmtree(out, ignore_errors=True)
    os.makedirs(out)
    with tempfile.TemporaryDirectory() as tmp:
        ref, ora = reference_cmd(case, meta, tmp)
        runs = []
        for i in range(2):
            w = os.path.join(tmp, "ref%d" % i)
            os.makedirs(w)
            r = subprocess.run([pyoracle.REF_BIN] + ref + ["-o", "g", "-t", "1"] + meta["args"] + extra, cwd=w, capture_output=True, text=True)
            runs.append((w, r))
        w0, r0 = runs[0]
        info = {"case": case, "extra_args": extra, "returncode": r0.returncode, "colored": bool(meta.get("colored"))}
        ub = os.path.join(tmp, "ub.txt")
        wo = os.path.join(tmp, "ora")
        os.makedirs(wo)
        ro = subprocess.run([pyoracle.CLI] + ora + ["-o", "g", "-O", os.path.join(wo, "PloidyFrost_output")] + meta["args"] + extra, cwd=wo,
                            capture_output=True, text=True, env=dict(os.environ, PFO_UB_LOG=ub))
        cells = pyoracle.read_ub_log(ub)
        info["ub_cells"] = {k: sorted(v) for k, v in cells.items()}
        info["oracle_returncode"] = ro.returncode
        if r0.returncode != 0:
            last = [x for x in r0.stdout.splitlines() if x.strip()]
            info["reference_last_line"] = last[-1] if last else ""
            info["note"] = ("the reference died of its own out-of-bounds read of indel_len (empty vector)" if r0.returncode == -11 and cells
                            else "the reference ended the run itself")
        else:
            d0 = os.path.join(w0, "PloidyFrost_output")
            bad2 = pyoracle.compare_outputs(d0, os.path.join(runs[1][0], "PloidyFrost_output"), "g", cells, info["colored"])
            assert not bad2, "two reference runs differ outside the undefined cells: %s" % bad2
            bado = pyoracle.compare_outputs(d0, os.path.join(wo, "PloidyFrost_output"), "g", cells, info["colored"])
            assert ro.returncode == 0 and not bado, "oracle differs from the reference: rc %d %s" % (ro.returncode, bado)
            os.makedirs(os.path.join(out, "expected"))
            same = []
            for suf in pyoracle.OUTPUT_SUFFIXES:
                f = "g_%s.txt" % suf
                a = open(os.path.join(d0, f), "rb").read()
                if a == open(os.path.join(HERE, case, "expected", f), "rb").read():
                    same.append(suf)
                else:
                    open(os.path.join(out, "expected", f), "wb").write(a)
            info["same_as_base_case"] = same
            info["reference_log"] = [x for x in r0.stdout.splitlines() if "SuperBubbles Found" in x or "Alleles in" in x]
        json.dump(info, open(os.path.join(out, "meta.json"), "w"), indent=1)
        print("%s__%s: reference rc %d, oracle rc %d, %d undefined cells%s" % (case, tag, r0.returncode, ro.returncode,
                                                                              sum(len(v) for v in cells.values()),
                                                                              "" if r0.returncode else ", oracle identical elsewhere"))


if __name__ == "__main__":
    todo = [tuple(a.split("__")) for a in sys.argv[1:]] or SCORE_CASES
    for c, t in todo:
        make(c, t)
