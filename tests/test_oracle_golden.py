"""The oracle (oracle/pf_oracle*.cpp) against the committed outputs of the real reference
binary (tests/golden/<case>/expected, produced by tests/golden/make_golden.py with -t 1).
This is the pin that lets the oracle stand in for the reference on the GPU box."""
import os
import sys

import pytest

from conftest import ROOT, colored_cases, compare_outputs, dialect_cases, golden_cases, load_case, load_dialect

sys.path.insert(0, os.path.join(ROOT, "oracle"))
import pyoracle  # noqa: E402


@pytest.mark.parametrize("case", golden_cases())
def test_oracle_matches_reference_outputs(case, tmp_path):
    meta = load_case(case)
    o = pyoracle.Oracle(meta["gfa"], meta["db"])
    assert o.k == meta["k"] and o.n == meta["unitigs"]
    op = meta["opts"]
    o.run(str(tmp_path), "g", z=int(op["-z"]), lower=int(op["-l"]), upper=int(op["-u"]), M=float(op["-M"]),
          D=float(op["-D"]), G=float(op["-G"]))
    bad = compare_outputs(os.path.join(meta["dir"], "expected"), str(tmp_path))
    assert not bad, "files differ from the reference: %s" % bad


@pytest.mark.parametrize("case", dialect_cases())
def test_oracle_reads_gfa_dialects_like_the_reference(case, tmp_path):
    """GFA 1 / 2, tags, lower case, interleaved lines, a last line without a line feed, and CRLF -- where the reference takes the
    '\\r' that ends a segment line's sequence as that sequence's last base (bifrost/src/GFA_Parser.cpp:497-520): the files the
    reference binary wrote for each dialect (tests/golden/make_dialect_golden.py)."""
    meta = load_dialect(case)
    o = pyoracle.Oracle(meta["gfa"], meta["db"])
    op = meta["opts"]
    o.run(str(tmp_path), "g", z=int(op["-z"]), lower=int(op["-l"]), upper=int(op["-u"]), M=float(op["-M"]),
          D=float(op["-D"]), G=float(op["-G"]))
    bad = compare_outputs(os.path.join(meta["dir"], "expected"), str(tmp_path))
    assert not bad, "files differ from the reference: %s" % bad


def test_dialect_fixtures_cover_what_they_claim():
    names = {c.split("__")[1] for c in dialect_cases()}
    assert {"crlf", "crlf_exact", "lowercase", "tags", "gfa2", "interleaved", "no_final_newline"} <= names
    m = load_dialect("weird12k__crlf_exact")
    assert m["dialect"]["same_as_base_expected"] and m["dialect"]["reference_returncode"] == 0
    with open(m["gfa"], newline="") as f:
        rows = [ln.split("\t") for ln in f.read().split("\n") if ln.startswith("S\t")]
    # k-length segments written as k - 1 bases + '\\r' (read back as ...T) and longer ones that lost their final A to it
    assert sum(1 for f in rows if len(f) == 3 and len(f[2]) == m["k"] and f[2].endswith("\r")) >= 3
    assert sum(1 for f in rows if len(f) == 3 and len(f[2]) > m["k"] and f[2].endswith("\r")) >= 20
    assert load_dialect("dip20k__crlf")["dialect"]["reference_returncode"] == -8


@pytest.mark.parametrize("case", colored_cases())
def test_colored_oracle_matches_reference_outputs(case, tmp_path):
    """CCDBG twin: oracle fed with the real Bifrost's colour dump against the reference's twelve files."""
    meta = load_case(case)
    o = pyoracle.ColoredOracle(meta["gfa"], meta["colors_dump"], meta["dbs"], str(tmp_path))
    assert o.k == meta["k"] and o.n == meta["unitigs"] and o.n_colors == meta["n_colors"]
    op = meta["opts"]
    out = tmp_path / "out"
    o.run(str(out), "g", meta["cutoffs"], z=int(op["-z"]), M=float(op["-M"]), D=float(op["-D"]), G=float(op["-G"]))
    bad = compare_outputs(os.path.join(meta["dir"], "expected"), str(out))
    assert not bad, "files differ from the reference: %s" % bad


@pytest.mark.skipif(not os.path.exists(pyoracle.REF_BIN), reason="reference binary (oracle/_ref) not built here")
def test_reference_binary_reproduces_golden(tmp_path):
    """The committed expectations really are what the reference emits (guards against stale fixtures)."""
    import subprocess
    meta = load_case("dip20k")
    subprocess.run([pyoracle.REF_BIN, "-g", meta["gfa"], "-d", meta["db"], "-o", "g", "-t", "1"] + meta["args"],
                   cwd=tmp_path, check=True, stdout=subprocess.DEVNULL)
    assert not compare_outputs(os.path.join(meta["dir"], "expected"), os.path.join(tmp_path, "PloidyFrost_output"))


# ---- scores outside the default fixtures' region (tests/golden/scores, made by the reference binary) ----------------------

from conftest import compare_score_outputs, load_score_case, score_cases  # noqa: E402


def _oracle_run_scores(meta, out, ub_log=None):
    op = meta["opts"]
    old = os.environ.get("PFO_UB_LOG")
    if ub_log:
        os.environ["PFO_UB_LOG"] = ub_log
    try:
        if meta["score"]["colored"]:
            o = pyoracle.ColoredOracle(meta["gfa"], meta["colors_dump"], meta["dbs"], os.path.dirname(out))
            o.run(out, "g", meta["cutoffs"], z=int(op["-z"]), M=float(op["-M"]), D=float(op["-D"]), G=float(op["-G"]))
        else:
            o = pyoracle.Oracle(meta["gfa"], meta["db"])
            o.run(out, "g", z=int(op["-z"]), lower=int(op["-l"]), upper=int(op["-u"]), M=float(op["-M"]), D=float(op["-D"]), G=float(op["-G"]))
    finally:
        if ub_log:
            if old is None:
                del os.environ["PFO_UB_LOG"]
            else:
                os.environ["PFO_UB_LOG"] = old


@pytest.mark.parametrize("case", score_cases())
def test_oracle_matches_reference_under_gap_friendly_scores(case, tmp_path):
    """-G 1, all-zero scores, D = M, G = M, ... (src/Main.cpp:470-479 accepts any D <= M, G <= M): rows that end in gaps, site
    strings that run to the end of a row (substr at size() is the empty string), k-mers holding '-' (looked up as the k-mer the
    CKmerAPI object held before).  Every byte of the reference's files except the cells it leaves undefined; those the oracle
    lists itself, and the list is the committed one."""
    meta = load_score_case(case)
    sm = meta["score"]
    out, ub = str(tmp_path / "out"), str(tmp_path / "ub.txt")
    if sm["returncode"] != 0:
        if sm["oracle_returncode"] != 0:   # the reference ended the run itself: so does the oracle
            with pytest.raises(RuntimeError):
                _oracle_run_scores(meta, out)
        else:                              # the reference died of its own undefined read; the oracle defines the cell and goes on
            _oracle_run_scores(meta, out, ub)
            assert sm["returncode"] == -11 and pyoracle.read_ub_log(ub)
        return
    _oracle_run_scores(meta, out, ub)
    assert pyoracle.read_ub_log(ub) == meta["ub_cells"]
    bad = compare_score_outputs(meta, out)
    assert not bad, "files differ from the reference: %s" % bad


def test_score_fixtures_cover_what_they_claim():
    names = set(score_cases())
    assert {"hex30k__G1", "hex30k__Z0", "hex30k__DeqM", "hex30k__GeqM", "k31_z16__G1", "weird12k__Z0", "col4_mix__G1", "col4_mix__Z0"} <= names
    m = load_score_case("hex30k__G1")
    assert m["score"]["returncode"] == 0 and len(m["ub_cells"]["bicov"]) > 100     # rows ending in gaps: open indel runs
    assert load_score_case("hex30k__GeqM")["score"]["returncode"] == 1 and "kmer can not found" in load_score_case("hex30k__GeqM")["score"]["reference_last_line"]
    assert load_score_case("hex30k__DeqM")["score"]["returncode"] == -11
    assert load_score_case("weird12k__DeqM")["score"]["returncode"] == 0
