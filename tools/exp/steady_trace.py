"""Host-side timeline of steady passes over the 5 M-unitig bench graph: PF_TRACE_FIND / PF_TRACE_PLOIDY lines.
usage: steady_trace.py [n_unitigs] ["ENV=V ENV2=V" ...]   one traced pass per configuration ("-" = defaults); the environment
variables are the ones the pipeline reads per pass (tools/ab_pass.py)"""
import os, sys, tempfile, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import torch
import bench
from ploidyfrost_amd import hostapi
n = int(sys.argv[1]) if len(sys.argv) > 1 else 5_000_000
configs = sys.argv[2:] or ["-"]
dev = torch.device("cuda", 0)
work = tempfile.mkdtemp(prefix="pf_st_", dir="/dev/shm")
gfa, db, n_unitigs, _ = bench.make_inputs(work, "graph", int(n / bench.UNITIGS_PER_BP), 1000, dev)
torch.cuda.empty_cache()
run = hostapi.Run(gfa, db, z=bench.Z, device=0)
run.set_threads(32); run.set_overlap_output(True)
run.set_output_dir(os.path.join(work, "PloidyFrost_output")); run.set_unitig_id("b")
def one(tag):
    t = time.perf_counter()
    run.find_superbubbles("b"); tf = time.perf_counter()
    run.ploidy_estimation("b", bench.LOWER, bench.UPPER)
    print("pass %s: find %.2f ms, ploidy %.2f ms" % (tag, (tf - t) * 1e3, (time.perf_counter() - tf) * 1e3), file=sys.stderr, flush=True)
for i in range(4): one(str(i))
for c in configs:
    sets = [] if c == "-" else [kv.split("=", 1) for kv in c.split()]
    for k, v in sets: os.environ[k] = v
    for i in range(2): one("%s warm" % c)
    os.environ["PF_TRACE_PLOIDY"] = "1"
    if os.environ.get("TRACE_FIND"): os.environ["PF_TRACE_FIND"] = "1"
    print("==== %s" % c, file=sys.stderr, flush=True)
    one(c)
    os.environ.pop("PF_TRACE_PLOIDY", None); os.environ.pop("PF_TRACE_FIND", None)
    for k, v in sets: os.environ.pop(k, None)
run.close()
import shutil; shutil.rmtree(work, ignore_errors=True)
