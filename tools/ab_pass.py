"""A/B measurement of measurement knobs on ONE loaded graph in ONE process: the passes of the configurations alternate, so that
box-to-box and run-to-run spread (a few ms on a 30 ms pass: host threads, page placement) cancels out.
  python tools/ab_pass.py [--unitigs N] [--rounds R] "NAME=VALUE ..." "NAME=VALUE ..." [...]
Each argument is one configuration: a space-separated list of environment settings the library reads per pass
(PF_ALIGN_RANGES, PF_ALIGN_THREADS, PF_BATCH_BUBBLES: the knobs that are left -- INTEGRATION.md lists them), or BATCH=n for Run.set_batch_bubbles(n); "-" is the default configuration.  Prints per configuration the median,
the minimum and the mean pass time over the rounds."""
import argparse
import os
import statistics
import sys
import tempfile
import time

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch  # noqa: E402

import bench  # noqa: E402
from ploidyfrost_amd import hostapi  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--unitigs", type=int, default=5_000_000)
    ap.add_argument("--rounds", type=int, default=12)
    ap.add_argument("configs", nargs="+")
    a = ap.parse_args()
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(0)
    work = tempfile.mkdtemp(prefix="pf_ab_", dir="/dev/shm" if os.path.isdir("/dev/shm") else None)
    gfa, db, n_unitigs, _ = bench.make_inputs(work, "graph", int(a.unitigs / bench.UNITIGS_PER_BP), 1000, dev)
    torch.cuda.empty_cache()
    run = hostapi.Run(gfa, db, z=bench.Z, device=0)
    run.set_threads(max(1, min(32, os.cpu_count() or 1)))
    run.set_overlap_output(True)
    run.set_output_dir(os.path.join(work, "PloidyFrost_output"))
    run.set_unitig_id("b")
    configs = [dict(kv.split("=", 1) for kv in c.split()) if c != "-" else {} for c in a.configs]
    names = sorted({k for c in configs for k in c})

    def one(cfg):
        for k in names:
            os.environ.pop(k, None)
        os.environ.update({k: v for k, v in cfg.items() if k != "BATCH"})
        run.set_batch_bubbles(int(cfg.get("BATCH", 32768)))   # (BATCH=n: bubbles per batch, four batches to a text piece)
        t = time.perf_counter()
        run.find_superbubbles("b")
        run.ploidy_estimation("b", bench.LOWER, bench.UPPER)
        return (time.perf_counter() - t) * 1e3

    for cfg in configs:   # every configuration sizes its pools once
        one(cfg), one(cfg)
    times = [[] for _ in configs]
    for _ in range(a.rounds):
        for i, cfg in enumerate(configs):
            times[i].append(one(cfg))
    for c, t in zip(a.configs, times):
        print("%-44s median %6.2f  min %6.2f  mean %6.2f ms  (%d passes, %d unitigs)" % (c, statistics.median(t), min(t), statistics.mean(t), len(t), n_unitigs), flush=True)
    run.close()
    import shutil
    shutil.rmtree(work, ignore_errors=True)


if __name__ == "__main__":
    main()
