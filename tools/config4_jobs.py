"""BASELINE.json configs[4] through the library: which kernel finished how many bubbles, and the kernels' times.
usage (GPU box): python tools/config4_jobs.py [unitigs]"""
import ctypes as C
import os
import sys
import tempfile
import time

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch  # noqa: E402

import bench  # noqa: E402
from ploidyfrost_amd import hipapi, hostapi  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 10_000_000
dev = torch.device("cuda", 0)
work = tempfile.mkdtemp(prefix="pf_c4_", dir="/dev/shm" if os.path.isdir("/dev/shm") else None)
gfa, db, n_unitigs, _ = bench.make_inputs(work, "g", int(n / bench.UNITIGS_PER_BP), 7, dev, k=31, max_ins=50, ploidy=4, p_snp=0.5, p_del=0.1)
torch.cuda.empty_cache()
run = hostapi.Run(gfa, db, z=16, device=0)
run.set_threads(32)
run.set_overlap_output(True)
run.set_output_dir(os.path.join(work, "out"))
run.set_unitig_id("b")
L = hipapi.load_library()
ctx = C.c_void_p(run.device_ctx())
for i in range(3):
    if i == 2:
        L.pf_enable_timing(ctx, 1)
        L.pf_reset_timing(ctx)
    t0 = time.perf_counter()
    run.find_superbubbles("b")
    run.ploidy_estimation("b", 5, 1000)
    t = run.times()
    print("pass %d: %.1f ms (find %.1f, ploidy %.1f); bubbles %d: snp %d pair %d stack %d wave %d; site strings %d" %
          (i, (time.perf_counter() - t0) * 1e3, t["find_total_s"] * 1e3, t["ploidy_total_s"] * 1e3, t["tasks"], t["snp_jobs"], t["pair_jobs"],
           t["stack_jobs"], t["wave_jobs"], t["site_strings"]), flush=True)
for kid, name in enumerate(hipapi.KERNELS):
    ms, cnt = C.c_double(), C.c_uint64()
    L.pf_kernel_time(ctx, kid, C.byref(ms), C.byref(cnt))
    if cnt.value:
        print("  %-22s %3d launches  %8.3f ms total  %8.3f ms each" % (name, cnt.value, ms.value, ms.value / cnt.value))
run.close()
