import sys; sys.path.insert(0,'.')
from ploidyfrost_amd import hipapi
d=hipapi.Device(0)
for n in (1_400_000, 5_000_000, 10_000_000):
    d._check(d.L.pf_selftest_scan(d.h, n, 1))
d.close()
