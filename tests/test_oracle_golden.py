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
