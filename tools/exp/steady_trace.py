"""Host-side timeline of ONE steady pass (the fourth) over the 5 M-unitig bench graph: PF_TRACE_FIND / PF_TRACE_PLOIDY lines."""
import os, sys, tempfile, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import torch
import bench
from ploidyfrost_amd import hostapi
n = int(sys.argv[1]) if len(sys.argv) > 1 else 5_000_000
dev = torch.device("cuda", 0)
work = tempfile.mkdtemp(prefix="pf_st_", dir="/dev/shm")
gfa, db, n_unitigs, _ = bench.make_inputs(work, "graph", int(n / bench.UNITIGS_PER_BP), 1000, dev)
torch.cuda.empty_cache()
run = hostapi.Run(gfa, db, z=bench.Z, device=0)
run.set_threads(32); run.set_overlap_output(True)
run.set_output_dir(os.path.join(work, "PloidyFrost_output")); run.set_unitig_id("b")
for i in range(5):
    if i == 4:
        os.environ["PF_TRACE_PLOIDY"] = "1"; os.environ["PF_TRACE_FIND"] = "1"
    t = time.perf_counter()
    run.find_superbubbles("b"); tf = time.perf_counter()
    run.ploidy_estimation("b", bench.LOWER, bench.UPPER)
    print("pass %d: find %.2f ms, ploidy %.2f ms" % (i, (tf - t) * 1e3, (time.perf_counter() - tf) * 1e3), file=sys.stderr, flush=True)
run.close()
import shutil; shutil.rmtree(work, ignore_errors=True)
