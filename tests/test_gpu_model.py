"""`PloidyFrost model` on the device (K-GMM, ploidyfrost_amd/csrc/pf_gmm.hip) against the CPU oracle and the reference's
result files.  fp64 throughout; the device sums are tree-shaped and the reference's sequential, so parameters agree to
rounding -- tolerance 1e-9 relative (stated here) -- and the result files, printed with six significant digits, are
compared as text."""
import json
import os
import subprocess
import sys

import numpy as np
import pytest

from conftest import GOLDEN, ROOT

sys.path.insert(0, os.path.join(ROOT, "oracle"))
import pyoracle  # noqa: E402

from ploidyfrost_amd import hostapi  # noqa: E402
from test_model_cpu import CASES, MODEL, load_into  # noqa: E402

pytestmark = pytest.mark.gpu
CLI = os.path.join(ROOT, "ploidyfrost_amd", "csrc", "ploidyfrost")
RTOL = 1e-9


@pytest.mark.parametrize("name", sorted(CASES))
def test_result_file_matches_the_reference(name, tmp_path):
    m = hostapi.Gmm()
    kw = load_into(m, CASES[name])
    m.run(str(tmp_path / "x"), **kw)
    with open(tmp_path / "x_model_result.txt") as got, open(os.path.join(MODEL, name + "_expected.txt")) as exp:
        assert got.read() == exp.read()


@pytest.mark.parametrize("name", ["tetra", "hexa_q", "tri_thresholds", "fixture_cov"])
def test_fit_parameters_match_the_oracle(name):
    o, m = pyoracle.GmmOracle(), hostapi.Gmm()
    kw = load_into(o, CASES[name])
    load_into(m, CASES[name])
    for g in range(kw["lo"], kw["hi"] + 1):
        args = dict(m_thre=kw["m_thre"], n_thre=kw["n_thre"], max_iter=kw["max_iter"], max_delta=kw["max_delta"])
        a, b = o.fit(g, **args), m.fit(g, **args)
        assert a["iterations"] == b["iterations"], (g, a["iterations"], b["iterations"])
        for key in ("weights", "means", "vars"):
            assert np.allclose(a[key], b[key], rtol=RTOL, atol=0), (g, key)
        assert abs(a["loglik"] - b["loglik"]) <= RTOL * abs(a["loglik"]) and abs(a["aic"] - b["aic"]) <= RTOL * abs(a["aic"])


def test_fit_is_reproducible_and_handles_sizes():
    rng = np.random.default_rng(5)
    for n in (1, 2, 63, 64, 65, 1023, 1024 * 4 + 1, 300_000):
        x = np.clip(rng.normal(rng.choice([0.25, 0.5, 0.75], size=n), 0.04), 0.001, 0.999)
        o, m = pyoracle.GmmOracle(), hostapi.Gmm()
        o.set_values(x)
        m.set_values(x)
        a, b, c = o.fit(3, max_iter=40), m.fit(3, max_iter=40), m.fit(3, max_iter=40)
        assert b["loglik"] == c["loglik"] and np.array_equal(b["vars"], c["vars"]) and np.array_equal(b["weights"], c["weights"])
        assert a["iterations"] == b["iterations"]
        assert np.allclose(a["vars"], b["vars"], rtol=RTOL, atol=0) and np.allclose(a["weights"], b["weights"], rtol=RTOL, atol=0)
        assert abs(a["loglik"] - b["loglik"]) <= RTOL * max(1.0, abs(a["loglik"]))
    m = hostapi.Gmm()
    m.set_values(x)
    z = m.fit(4, max_iter=0)                     # no iteration: the initial parameters and their likelihood
    assert z["iterations"] == 0 and np.allclose(z["vars"], 0.01) and np.allclose(z["weights"], 0.25)
    with pytest.raises(RuntimeError):
        m.fit(17)                                # PF_GMM_MAX_GAUSS


@pytest.mark.parametrize("name", ["tetra", "fixture_cov", "fixture_cov_q"])
def test_cli_model_sub_command(name, tmp_path):
    case = CASES[name]
    arg = ["-f" if case["kind"] == "cov" else "-g", os.path.join(GOLDEN, case["input"])]
    r = subprocess.run([CLI, "model"] + arg + ["-o", "x"] + case["options"], cwd=tmp_path, stdout=subprocess.PIPE,
                       stderr=subprocess.STDOUT, text=True)
    assert r.returncode == 0, r.stdout
    with open(tmp_path / "x_model_result.txt") as got, open(os.path.join(MODEL, name + "_expected.txt")) as exp:
        assert got.read() == exp.read()
    # option errors print the usage and return 0, as the reference does (src/Main.cpp:692-719)
    r = subprocess.run([CLI, "model", "-o", "x"], cwd=tmp_path, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
    assert r.returncode == 0 and "Usage: PloidyFrost model" in r.stdout
