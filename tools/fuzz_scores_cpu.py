#!/usr/bin/env python3
"""CPU-only differential fuzzing of the ORACLE against the REFERENCE BINARY over the whole score region the reference accepts
(src/Main.cpp:470-479 asks only D <= M and G <= M): zero, positive and fractional gap scores, D = M, G = M, magnitudes to 1e5.
Runs in the build container (needs oracle/_ref).  Cells the reference leaves undefined (pfo::indel_len_at) are masked by the
oracle's PFO_UB_LOG; a reference that dies of its own undefined read (SIGSEGV / abort) is reported and the files it had completed
before are not compared.      usage: tools/fuzz_scores_cpu.py [n_cases] [first_seed] [--colored]"""
import os
import subprocess
import sys
import tempfile

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "oracle"))
sys.path.insert(0, os.path.join(ROOT, "tools"))
import pyoracle  # noqa: E402
import fuzz_parity  # noqa: E402


def main():
    args = [a for a in sys.argv[1:] if not a.startswith("--")]
    n = int(args[0]) if args else 40
    first = int(args[1]) if len(args) > 1 else 1
    colored = "--colored" in sys.argv
    pyoracle.build()
    os.environ["PF_FUZZ_BIFROST"] = "1"     # graphs by the reference's own Bifrost: no device needed
    fails = died = 0
    for seed in range(first, first + n):
        with tempfile.TemporaryDirectory() as tmp:
            c = fuzz_parity.build_case(seed, tmp, "cpu", force_colored=colored, wide_scores=True)
            if isinstance(c, str):
                print("seed %d: %s" % (seed, c), flush=True)
                continue
            M, D, G = c["scores"]
            common = ["-o", "x", "-z", str(c["z"]), "-M", repr(M), "-D", repr(D), "-G", repr(G)]
            og, rg = os.path.join(tmp, "oracle"), os.path.join(tmp, "ref")
            os.makedirs(og), os.makedirs(rg)
            ub = os.path.join(tmp, "ub.txt")
            env = dict(os.environ, PFO_UB_LOG=ub)
            if c["colored"]:
                dump = os.path.join(tmp, "colors.txt")
                with open(dump, "w") as f:
                    subprocess.run([pyoracle.REF_COLORS_DUMP, c["gfa"], c["colors"]], check=True, stdout=f)
                lst, cut = os.path.join(tmp, "dbs.txt"), os.path.join(tmp, "cut.txt")
                open(lst, "w").write("".join(d + "\n" for d in c["dbs"]))
                open(cut, "w").write(("%d\t%d\n" % (c["lower"], c["upper"])) * len(c["dbs"]))
                rr = subprocess.run([pyoracle.REF_BIN, "-g", c["gfa"], "-f", c["colors"], "-d", lst, "-C", cut, "-t", "1"] + common, cwd=rg,
                                    capture_output=True, text=True, timeout=900)
                ro = subprocess.run([pyoracle.CLI, "-g", c["gfa"], "-f", dump, "-d", lst, "-C", cut, "-O", os.path.join(og, "PloidyFrost_output")]
                                    + common, cwd=og, capture_output=True, text=True, env=env, timeout=900)
            else:
                cut = ["-l", str(c["lower"]), "-u", str(c["upper"])]
                rr = subprocess.run([pyoracle.REF_BIN, "-g", c["gfa"], "-d", c["db"], "-t", "1"] + cut + common, cwd=rg, capture_output=True,
                                    text=True, timeout=900)
                ro = subprocess.run([pyoracle.CLI, "-g", c["gfa"], "-d", c["db"], "-O", os.path.join(og, "PloidyFrost_output")] + cut + common,
                                    cwd=og, capture_output=True, text=True, env=env, timeout=900)
            desc = "k=%d ploidy=%d z=%d scores=%s cut=%d/%d %s unitigs=%d" % (c["k"], c["ploidy"], c["z"], (M, D, G), c["lower"], c["upper"],
                                                                             "colored" if c["colored"] else "single", c["n_unitigs"])
            cells = pyoracle.read_ub_log(ub)
            n_ub = sum(len(v) for v in cells.values())
            if rr.returncode in (-11, -6) and n_ub and ro.returncode == 0:
                died += 1
                print("seed %d: %s: the reference died (rc %d) where %d cells are undefined; not compared" % (seed, desc, rr.returncode, n_ub),
                      flush=True)
                continue
            if rr.returncode == -8 and ro.returncode == 0:
                print("seed %d: %s: skipped (the reference's division by zero: no site)" % (seed, desc), flush=True)
                continue
            if rr.returncode != 0 or ro.returncode != 0:
                same = (rr.returncode != 0) == (ro.returncode != 0)
                if rr.returncode == 1 and ro.returncode == 1 and not c["colored"]:   # both name the k-mer that was not found
                    want = [x for x in rr.stdout.splitlines() if "kmer can not found" in x]
                    same = bool(want) and want[-1].strip() in ro.stderr
                fails += not same
                print("seed %d: %s: reference rc %d, oracle rc %d%s" % (seed, desc, rr.returncode, ro.returncode,
                                                                        "" if same else "  MISMATCH " + ro.stderr[-200:]), flush=True)
                continue
            bad = pyoracle.compare_outputs(os.path.join(rg, "PloidyFrost_output"), os.path.join(og, "PloidyFrost_output"), "x", cells, c["colored"])
            fails += bool(bad)
            print("seed %d: %s: %s%s" % (seed, desc, "identical" if not bad else "DIFFERENT " + ",".join(bad),
                                         " (%d undefined cells masked)" % n_ub if n_ub else ""), flush=True)
            if bad and os.environ.get("PF_FUZZ_KEEP"):
                import shutil
                shutil.copytree(tmp, os.path.join(os.environ["PF_FUZZ_KEEP"], "seed%d" % seed))
    print("%d cases, %d failures, %d reference deaths" % (n, fails, died))
    return 1 if fails else 0


if __name__ == "__main__":
    sys.exit(main())
