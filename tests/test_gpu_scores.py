"""The product under the scores the default fixtures do not reach -- gap-friendly, all-zero, D = M, G = M (the reference accepts any
D <= M, G <= M: src/Main.cpp:470-479) -- against what the REFERENCE binary wrote (tests/golden/scores, made by
tests/golden/make_score_golden.py) and against the oracle.  Rows that end in gaps, site strings that run to the end of a row
(`substr(size(), 1)` is the empty string: src/CDBG.cpp:1478-1490), k-mers that hold a '-' (looked up as the k-mer the CKmerAPI object
held before: KMC/kmc_api/kmer_api.h:502-510), indel runs open at the last column (the one value the reference leaves undefined)."""
import os
import subprocess
import sys

import pytest

from conftest import ROOT, compare_outputs, compare_score_outputs, load_score_case, score_cases

sys.path.insert(0, os.path.join(ROOT, "oracle"))
import pyoracle  # noqa: E402

pytestmark = pytest.mark.gpu
CLI = os.path.join(ROOT, "ploidyfrost_amd", "csrc", "ploidyfrost")


def _oracle(meta, out):
    op = meta["opts"]
    if meta["score"]["colored"]:
        o = pyoracle.ColoredOracle(meta["gfa"], meta["colors_dump"], meta["dbs"], os.path.dirname(out))
        o.run(out, "g", meta["cutoffs"], z=int(op["-z"]), M=float(op["-M"]), D=float(op["-D"]), G=float(op["-G"]))
    else:
        o = pyoracle.Oracle(meta["gfa"], meta["db"])
        o.run(out, "g", z=int(op["-z"]), lower=int(op["-l"]), upper=int(op["-u"]), M=float(op["-M"]), D=float(op["-D"]), G=float(op["-G"]))


def _product(meta, tmp_path, env=None):
    sm = meta["score"]
    if sm["colored"]:
        (tmp_path / "dbs.txt").write_text("".join(p + "\n" for p in meta["dbs"]))
        (tmp_path / "cutoffs.txt").write_text("".join("%d\t%d\n" % tuple(c) for c in meta["cutoffs"]))
        cmd = [CLI, "-g", meta["gfa"], "-f", meta["colors"], "-d", str(tmp_path / "dbs.txt"), "-C", str(tmp_path / "cutoffs.txt")]
    else:
        cmd = [CLI, "-g", meta["gfa"], "-d", meta["db"]]
    return subprocess.run(cmd + ["-o", "g", "-t", "1"] + meta["args"] + sm["extra_args"], cwd=tmp_path, stdout=subprocess.PIPE,
                          stderr=subprocess.STDOUT, text=True, env=dict(os.environ, **(env or {})))


@pytest.mark.parametrize("pipeline", ["resident", "host"])
@pytest.mark.parametrize("case", score_cases())
def test_cli_matches_the_reference_under_gap_friendly_scores(case, pipeline, tmp_path):
    meta = load_score_case(case)
    sm = meta["score"]
    r = _product(meta, tmp_path, {"PF_CALL": "host"} if pipeline == "host" else None)
    got = os.path.join(tmp_path, "PloidyFrost_output")
    if sm["returncode"] != 0 and sm["oracle_returncode"] != 0:
        # the reference ends the run itself (a k-mer of a site string it cannot find; a site string that is on no unitig): so does the product
        assert r.returncode != 0, r.stdout
        if sm["returncode"] == 1:
            assert "can not found" in r.stdout and "can not found" in sm["reference_last_line"]
        return
    assert r.returncode == 0, r.stdout
    want = str(tmp_path / "oracle")
    _oracle(meta, want)
    bad = compare_outputs(want, got)   # every byte, the cells the reference leaves undefined included: one definition for both
    assert not bad, "files differ from the oracle: %s\n%s" % (bad, r.stdout)
    if sm["returncode"] == 0:
        bad = compare_score_outputs(meta, got)
        assert not bad, "files differ from the reference: %s\n%s" % (bad, r.stdout)
        for line in sm["reference_log"]:
            assert line.strip() in r.stdout.replace("\r", ""), line
