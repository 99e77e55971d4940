"""K-COV-JOIN alone, for profiling: one graph loaded, pf_join_counts repeated.
usage (GPU box): python3 tools/exp/join_exp.py [unitigs] [reps] [workload]   (under rocprofv3: the generator runs in this process)"""
import ctypes as C
import os
import sys
import tempfile
import time

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import torch  # noqa: E402

import bench  # noqa: E402
from ploidyfrost_amd import hipapi, hostapi  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 5
wl = bench.WORKLOADS[sys.argv[3] if len(sys.argv) > 3 else "single"]
dev = torch.device("cuda", 0)
work = tempfile.mkdtemp(prefix="pf_join_", dir="/dev/shm" if os.path.isdir("/dev/shm") else None)
gfa, db, n_unitigs, n_kmers = bench.make_inputs(work, "g", int(n / bench.UNITIGS_PER_BP), wl["seed"], dev, k=wl["k"], **wl["gen"])
torch.cuda.empty_cache()
t0 = time.perf_counter()
run = hostapi.Run(gfa, db, z=wl["z"], device=0)
print("load %.3f s" % (time.perf_counter() - t0), flush=True)
L = hipapi.load_library()
ctx = C.c_void_p(run.device_ctx())
L.pf_enable_timing(ctx, 1)
L.pf_reset_timing(ctx)
for i in range(reps):
    st = L.pf_join_counts(ctx)
    assert st == 0, st
ms, cnt = C.c_double(), C.c_uint64()
L.pf_kernel_time(ctx, hipapi.KERNELS.index("k_cov_join"), C.byref(ms), C.byref(cnt))
alg = 12.25 * n_kmers + 16.0 * n_unitigs
avg = ms.value / max(cnt.value, 1)
ms2, cnt2 = C.c_double(), C.c_uint64()
L.pf_kernel_time(ctx, hipapi.KERNELS.index("k_cov_join_rest"), C.byref(ms2), C.byref(cnt2))
print("k_cov_join_rest: %.3f ms each" % (ms2.value / max(cnt2.value, 1)))
avg += ms2.value / max(cnt2.value, 1)
print("k_cov_join: %d unitigs, %d k-mers, %d launches, %.3f ms each, algorithmic %.1f MB -> %.1f GB/s = %.4f of 8 TB/s" %
      (n_unitigs, n_kmers, cnt.value, avg, alg / 1e6, alg / (avg * 1e-3) / 1e9, alg / (avg * 1e-3) / 8e12), flush=True)
run.close()
import shutil
shutil.rmtree(work, ignore_errors=True)
