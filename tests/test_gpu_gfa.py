"""K-GFA (csrc/pf_gfa.hip): the S-lines of a GFA file parsed and 2-bit packed on the device -- and the host loader
(csrc/host/pf_host_graph.cpp, PF_GFA=host) -- against the files the REFERENCE binary wrote for the same input on the dialects the
reference's GFA_Parser meets (tests/golden/dialects): GFA 1 and 2, CRLF line ends, lower-case bases, extra tags, link lines
between the segments, a last line without a newline; and the three refusals.  (That the twelve files of every fixture match the
reference through this ingest is what tests/test_gpu_end_to_end.py checks: pfh_open and the CLI use it by default.)"""
import os
import subprocess

import numpy as np
import pytest

from conftest import ROOT, compare_outputs, dialect_cases, load_case, load_dialect
from ploidyfrost_amd import hipapi

CLI = os.path.join(ROOT, "ploidyfrost_amd", "csrc", "ploidyfrost")
pytestmark = pytest.mark.gpu


def _lines(meta):
    with open(meta["gfa"]) as f:
        return f.read().split("\n")[:-1]


def _run(gfa, meta, tmp, host):
    env = dict(os.environ)
    if host:
        env["PF_GFA"] = "host"
    else:
        env.pop("PF_GFA", None)
    os.makedirs(tmp, exist_ok=True)
    r = subprocess.run([CLI, "-g", gfa, "-d", meta["db"], "-o", "g", "-t", "2"] + meta["args"], cwd=tmp, capture_output=True, text=True, env=env)
    return r


@pytest.mark.parametrize("case", dialect_cases())
def test_dialects_give_the_references_files(case, tmp_path):
    """Both loaders (K-GFA on the device, the host loader behind PF_GFA=host) against what the REFERENCE binary wrote for the
    dialect (tests/golden/dialects/<base>__<dialect>/expected, made by tests/golden/make_dialect_golden.py).  `crlf`: the
    reference reads the '\\r' that ends a segment's sequence as a base (bifrost/src/GFA_Parser.cpp:497-520), finds no superbubble
    in the graph that gives and dies of its division by zero after the files are complete (src/CDBG.cpp:1703; return code -8 in
    meta.json); the product writes the same files and returns.  `crlf_exact` places the '\\r' where it restores the base graph."""
    meta = load_dialect(case)
    for host in (False, True):
        out = str(tmp_path / ("host" if host else "dev"))
        r = _run(meta["gfa"], meta, out, host)
        assert r.returncode == 0, (case, host, r.stdout[-300:], r.stderr[-300:])
        bad = compare_outputs(os.path.join(meta["dir"], "expected"), os.path.join(out, "PloidyFrost_output"))
        assert not bad, (case, "host loader" if host else "K-GFA", bad)


@pytest.mark.parametrize("kind,message", [("base", "non-ACGT base in a segment"), ("short", "segment shorter than k"),
                                          ("fields", "missing fields in a segment line")])
def test_refusals_are_the_host_loaders(kind, message, tmp_path):
    meta = load_case("dip20k")
    lines = _lines(meta)
    seg = [i for i, ln in enumerate(lines) if ln.startswith("S\t")]
    i = seg[len(seg) // 2]
    f = lines[i].split("\t")
    if kind == "base":
        f[2] = f[2][:7] + "N" + f[2][8:]
    elif kind == "short":
        f[2] = f[2][:5]
    else:
        f = f[:2]
    lines[i] = "\t".join(f)
    gfa = str(tmp_path / "bad.gfa")
    with open(gfa, "w") as fh:
        fh.write("\n".join(lines) + "\n")
    for host in (True, False):
        r = _run(gfa, meta, str(tmp_path / ("h" if host else "d")), host)
        assert r.returncode != 0
        assert "CompactedDBG::read(): Graph could not be loaded! Exit. (%s)" % message in r.stdout + r.stderr, (host, r.stdout[-300:], r.stderr[-300:])


def test_segment_table(tmp_path):
    """pf_gfa_ingest / pf_gfa_segments directly: order (long segments in file order, then the k-length ones), lengths, offsets of
    the sequence fields, file ranks, DA tags, canonical storage of the k-length ones"""
    k = 5
    segs = ["ACGTACGTAC", "TTTTT", "ACGTT", "GGGCCCAAAT", "CCCCC", "acgtaCGTTTA"]
    body = "".join("S\t%d\t%s%s\n" % (i + 1, s, "\tDA:Z:%d" % i if i % 2 == 0 else "") + ("L\t1\t+\t2\t-\t4M\n" if i == 1 else "") for i, s in enumerate(segs))
    body += "S\tlast\tAAAAAAAA"   # unterminated: dropped
    dev = hipapi.Device()
    raw = np.frombuffer(body.encode(), dtype=np.uint8).copy()
    n, ns = hipapi.C.c_uint32(), hipapi.C.c_uint32()
    dev._check(dev.L.pf_gfa_ingest(dev.h, raw.ctypes.data, len(raw), 1, k, hipapi.C.byref(n), hipapi.C.byref(ns)))
    assert (n.value, ns.value) == (6, 3)
    ln = np.zeros(6, np.uint32)
    off = np.zeros(6, np.uint64)
    rank = np.zeros(6, np.uint32)
    da = np.zeros(6, np.int16)
    rc = np.zeros(6, np.uint8)
    any_da = hipapi.C.c_int()
    dev._check(dev.L.pf_gfa_segments(dev.h, ln.ctypes.data, off.ctypes.data, rank.ctypes.data, da.ctypes.data, rc.ctypes.data, hipapi.C.byref(any_da)))
    order = [0, 3, 5, 1, 2, 4]
    assert list(rank) == order and list(ln) == [len(segs[i]) for i in order]
    for u, i in enumerate(order):
        assert body[int(off[u]):int(off[u]) + int(ln[u])] == segs[i]
    assert list(da) == [0, -1, -1, -1, 2, 4] and any_da.value == 1
    # TTTTT -> AAAAA (reverse complement is smaller), ACGTT -> AACGT, CCCCC stays
    assert list(rc) == [0, 0, 0, 1, 1, 0]
    dev.close()
