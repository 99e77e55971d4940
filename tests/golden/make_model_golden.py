#!/usr/bin/env python3
"""Golden vectors for `PloidyFrost model` (reference src/GmmModel.cpp): inputs drawn here, expected result files written by
the reference binary (oracle/_ref/PloidyFrost model ...) in this container.  Usage: python tests/golden/make_model_golden.py
Cases live in tests/golden/model/cases.json: name -> {input kind, input path (relative to tests/golden), options}."""
import json
import os
import shutil
import subprocess
import tempfile

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
REF = os.path.join(ROOT, "oracle", "_ref", "PloidyFrost")
OUT = os.path.join(HERE, "model")


def mixture(seed, n, ploidy, sd, noise=0.05):
    """allele frequencies of a `ploidy`-ploid sample: peaks at i/ploidy, a little uniform noise"""
    rng = np.random.default_rng(seed)
    peaks = np.arange(1, ploidy) / ploidy
    w = rng.dirichlet(np.full(len(peaks), 4.0))
    which = rng.choice(len(peaks), size=n, p=w)
    x = rng.normal(peaks[which], sd)
    u = rng.random(n) < noise
    x[u] = rng.random(int(u.sum()))
    return np.clip(x, 0.001, 0.999)


CASES = {
    # name: (input kind, generator or fixture path, extra options)
    "tetra": ("fre", lambda: mixture(1, 6000, 4, 0.03), []),
    "hexa_q": ("fre", lambda: mixture(2, 9000, 6, 0.02), ["-q", "0.05"]),
    "diploid_tight": ("fre", lambda: mixture(3, 3000, 2, 0.04, noise=0.0), ["-l", "1", "-u", "5", "-a", "0.0001", "-k", "60"]),
    "tri_thresholds": ("fre", lambda: mixture(4, 5000, 3, 0.05, noise=0.2), ["-m", "2", "-n", "1.2", "-u", "7"]),
    "fixture_fre": ("fre_file", "tet60k/expected/g_allele_frequency.txt", []),
    "fixture_cov": ("cov", "hex30k/expected/g", ["-u", "6"]),
    "fixture_cov_q": ("cov", "tet60k/expected/g", ["-q", "0.2", "-u", "4"]),
}


def main():
    os.makedirs(OUT, exist_ok=True)
    meta = {}
    for name, (kind, src, opts) in CASES.items():
        if kind == "fre":
            path = os.path.join(OUT, name + "_fre.txt")
            with open(path, "w") as f:
                for v in src():
                    f.write("%.6g\n" % v)
            rel = os.path.relpath(path, HERE)
            arg = ["-g", path]
        elif kind == "fre_file":
            rel = src
            arg = ["-g", os.path.join(HERE, src)]
        else:
            rel = src
            arg = ["-f", os.path.join(HERE, src)]
        with tempfile.TemporaryDirectory() as tmp:
            subprocess.run([REF, "model"] + arg + ["-o", "x"] + opts, cwd=tmp, check=True, stdout=subprocess.DEVNULL)
            shutil.copy(os.path.join(tmp, "x_model_result.txt"), os.path.join(OUT, name + "_expected.txt"))
        meta[name] = {"kind": "cov" if kind == "cov" else "fre", "input": rel, "options": opts}
        print(name, open(os.path.join(OUT, name + "_expected.txt")).read().splitlines()[-1])
    with open(os.path.join(OUT, "cases.json"), "w") as f:
        json.dump(meta, f, indent=1)


if __name__ == "__main__":
    main()
