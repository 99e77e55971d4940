"""`PloidyFrost model` (reference src/GmmModel.cpp, src/Main.cpp:636-692) without a GPU: the oracle's restatement against the
result files the reference binary wrote (tests/golden/model, made by tests/golden/make_model_golden.py), and the product's
readers (host code, no device) against the oracle's."""
import json
import os
import sys

import numpy as np
import pytest

from conftest import GOLDEN, ROOT

sys.path.insert(0, os.path.join(ROOT, "oracle"))
import pyoracle  # noqa: E402

MODEL = os.path.join(GOLDEN, "model")
with open(os.path.join(MODEL, "cases.json")) as _f:
    CASES = json.load(_f)


def model_options(opts):
    o = {"-l": "1", "-u": "9", "-q": "0", "-m": "5", "-n": "2", "-k": "1000", "-a": "0.01"}
    for i in range(0, len(opts), 2):
        o[opts[i]] = opts[i + 1]
    return dict(lo=int(o["-l"]), hi=int(o["-u"]), m_thre=float(o["-m"]), n_thre=float(o["-n"]), max_iter=int(o["-k"]),
                max_delta=float(o["-a"])), float(o["-q"])


def load_into(model, case):
    kw, q = model_options(case["options"])
    path = os.path.join(GOLDEN, case["input"])
    (model.read_cov if case["kind"] == "cov" else model.read_fre)(path, q)
    return kw


@pytest.mark.parametrize("name", sorted(CASES))
def test_oracle_writes_the_reference_result_file(name, tmp_path):
    o = pyoracle.GmmOracle()
    kw = load_into(o, CASES[name])
    o.run(str(tmp_path / "x"), **kw)
    with open(tmp_path / "x_model_result.txt") as got, open(os.path.join(MODEL, name + "_expected.txt")) as exp:
        assert got.read() == exp.read()


@pytest.mark.parametrize("name", sorted(CASES))
def test_product_readers_match_the_oracle(name):
    from ploidyfrost_amd import hostapi
    o, m = pyoracle.GmmOracle(), hostapi.Gmm()
    load_into(o, CASES[name])
    load_into(m, CASES[name])
    assert np.array_equal(o.values(), m.values())


def test_reader_quirks(tmp_path):
    """What the reference's parsers really do (src/GmmModel.cpp:21-257), on both implementations."""
    from ploidyfrost_amd import hostapi
    p = str(tmp_path / "q")
    with open(p + "_bicov.txt", "w") as f:
        f.write("30.9\t10.2\t1\t0\t1\t1\t48\t\n"      # atoi: 30 and 10
                "7\t0\t1\t0\t2\t1\t48\t\n"             # one allele holds everything: the integer ratio is 1
                "9000\t2000\t1\t\n"                    # sum >= 10000: dropped
                "12\n")                                # too few fields: skipped
    with open(p + "_tricov.txt", "w") as f:
        f.write("5\t9\t7\t0\t\n")                      # "min" walks neighbour pairs: 7 < 9 wins over 5
    with open(p + "_tetracov.txt", "w") as f:
        f.write("4\t4\t4\t4\t0\t\n")
    with open(p + "_pentacov.txt", "w") as f:
        f.write("1\t1\t1\t1\t1\t0\t\n")                # never read (the stream is closed before its loop)
    for q, want in ((0.0, [30 / 40, 10 / 40, 5 / 21, 9 / 21, 7 / 21, .25, .25, .25, .25]), (0.1, [])):
        o, m = pyoracle.GmmOracle(), hostapi.Gmm()
        o.read_cov(p, q)
        m.read_cov(p, q)
        # q = 0: the integer ratio 0 passes 0 <= r <= 1, the ratio 1 of the second row fails r <= 1 - 0?  no: 1 <= 1 holds
        if q == 0.0:
            want = [30 / 40, 10 / 40, 1.0, 0.0] + want[2:]
        assert np.allclose(o.values(), want, rtol=0, atol=1e-15) and np.array_equal(o.values(), m.values())
    fre = str(tmp_path / "f.txt")
    with open(fre, "w") as f:
        f.write("0.25\n1.5\n0.5 0.75\n")                # trailing newline: the read that fails at the end leaves the last value in place
    o, m = pyoracle.GmmOracle(), hostapi.Gmm()
    o.read_fre(fre, 0.0)
    m.read_fre(fre, 0.0)
    assert list(o.values()) == [0.25, 0.5, 0.75, 0.75] and np.array_equal(o.values(), m.values())
    with open(fre, "w") as f:
        f.write("0.25\nabc\n")
    with pytest.raises(RuntimeError):
        hostapi.Gmm().read_fre(fre, 0.0)
    with pytest.raises(RuntimeError):
        hostapi.Gmm().read_cov(str(tmp_path / "missing"), 0.0)


def test_fit_needs_the_gpu():
    """no CPU fallback: without a device the fit fails loudly"""
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    from ploidyfrost_amd import hostapi
    m = hostapi.Gmm()
    m.set_values(np.linspace(0.1, 0.9, 50))
    with pytest.raises(RuntimeError):
        m.fit(3)
