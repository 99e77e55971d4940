"""The resident calling pipeline (pf_call_*, reference CDBG::ploidyEstimation_ptr src/CDBG.cpp:1101-1705) through the C ABI."""
import os
import struct

import numpy as np
import pytest

from conftest import GOLDEN, ROOT

pytestmark = pytest.mark.gpu


def test_device_formats_doubles_like_printf():
    """O1: `ostream << double` == printf("%g"), byte for byte -- random bit patterns, the path's magnitudes, rounding ties."""
    from ploidyfrost_amd import hipapi
    rng = np.random.default_rng(7)
    bits = rng.integers(0, 2 ** 64, size=200_000, dtype=np.uint64)
    v = [bits.view(np.float64)]
    e = rng.uniform(-8, 10, size=200_000)
    v.append(10.0 ** e * (1 + rng.random(200_000)))
    v.append(rng.integers(0, 100000, 200_000) / rng.integers(1, 5000, 200_000))   # sum / length
    n6 = rng.integers(100000, 1000000, 20_000).astype(np.float64) + 0.5           # 6 digits and a trailing 5
    for p in range(-12, 13):
        t = n6 * 10.0 ** p
        v += [t, np.nextafter(t, 0), np.nextafter(t, 1e300)]
    v.append(np.array([0.0, -0.0, 1.0, 0.5, 999999.5, 1e6, 1e-4, 9.99995e-5, 1e-5, 5e-324, 1.7976931348623157e308, 2.0 ** 53, 1e22, 1e23,
                       np.inf, -np.inf, 100.0 / 3, 2.0 / 3]))
    vals = np.concatenate(v)
    dev = hipapi.Device(0)
    got = dev.format_doubles(vals)
    dev.close()
    bad = 0
    for x, g in zip(vals.tolist(), got):
        want = b"-nan" if x != x else ("%g" % x).encode()
        if g != want:
            bad += 1
            assert bad < 5, (x, struct.pack("<d", x).hex(), want, g)
    assert bad == 0
