#!/usr/bin/env python3
"""Measurement of the `model` row (K-GMM): `PloidyFrost model -l 1 -u 9` on N synthetic allele frequencies (hexaploid
mixture), device fit vs the CPU oracle on a bounded sample.  One JSON line.   usage: tools/bench_model.py [n_values]"""
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "oracle"))
sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
import pyoracle  # noqa: E402
from make_model_golden import mixture  # noqa: E402
from ploidyfrost_amd import hostapi  # noqa: E402


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 2_000_000
    x = mixture(11, n, 6, 0.02)
    m = hostapi.Gmm()
    m.set_values(x)
    m.fit(2, max_iter=3)            # context, upload, first launches
    m.enable_timing(True)
    t0 = time.perf_counter()
    its, res = 0, {}
    for g in range(1, 10):
        res[g] = m.fit(g)
        its += res[g]["iterations"] + 1          # passes over the values
    wall = time.perf_counter() - t0
    ms, launches = m.kernel_time()
    best = min(res, key=lambda g: res[g]["aic"]) + 1
    # CPU oracle: one model (5 Gaussians) on a bounded sample, scaled per value and pass
    ns = min(n, 200_000)
    o = pyoracle.GmmOracle()
    o.set_values(x[:ns])
    c0 = time.perf_counter()
    ro = o.fit(5, max_iter=20)
    cpu = time.perf_counter() - c0
    cpu_per_value_pass = cpu / (ns * (2 * ro["iterations"] + 1))   # the reference makes two passes per iteration
    gpu_per_value_pass = wall / (n * its)
    print(json.dumps({
        "metric": "allele frequencies x EM passes per second (PloidyFrost model -l 1 -u 9)", "value": n * its / wall, "unit": "values*passes/s",
        "n_values": n, "models": 9, "passes": its, "wall_s": round(wall, 4), "estimated_ploidy": best,
        "kernel": {"name": "k_gmm (hipGraph of 16 pass+update pairs)", "graph_launches": launches, "device_ms": round(ms, 3),
                   "ms_per_graph_launch": round(ms / max(1, launches), 4)},
        "roofline": {"bound": "hbm", "achieved": round(8.0 * n * its / (ms * 1e-3) / 1e9, 2), "peak": 8000.0, "unit": "GB/s",
                     "frac": round(8.0 * n * its / (ms * 1e-3) / 1e9 / 8000.0, 5), "traffic": None,
                     "note": "algorithmic bytes = 8 B per value and pass; finished fits leave idle pairs inside the last graph launch"},
        "cpu_baseline": {"value": 1.0 / cpu_per_value_pass, "unit": "values*passes/s", "cores": 1, "kind": "port",
                         "sample": "oracle fit of 5 Gaussians, %d values, %d iterations (two passes each)" % (ns, ro["iterations"])},
        "speedup_vs_cpu_1core": round(cpu_per_value_pass / gpu_per_value_pass, 1)}))


if __name__ == "__main__":
    main()
