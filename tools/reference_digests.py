#!/usr/bin/env python3
"""Digests and timings of the REFERENCE binary (oracle/_ref/PloidyFrost) on bench.py's own graphs, committed as
profiles/reference_digests.json and read by bench.py: `output_check` holds the twelve files of the timed passes against these md5s,
`cpu_baseline.reference_at_config_size` quotes these timings.  Runs on a GPU box or on any host: the graph builder takes the device
when there is one and torch's CPU otherwise, and makes the same files either way (checked: the 996 064-unitig digests made on a
GPU box and in a container without one are equal); the reference itself runs on the host's cores.
    usage: tools/reference_digests.py <out.json> <unitigs> [seed] [--workload single|stress|colored] [--threads N ...] [--skip-t1]
--workload: bench.py's workloads (bench.WORKLOADS: k, -z, generator parameters, default seed); colored = 3 diploid samples through
the CCDBG path (-f <bfg_colors> -d <list> -C <cutoffs>).  Entries are keyed as bench.digest_key says.
For every size: `-t 1` (the order-defining run: md5 of its twelve files; the reference's own Cpu / Real time lines) and, for each
--threads N, a `-t N` run for context (its rows are unordered: timings only).  Entries are merged into <out.json> by unitig count."""
import hashlib
import json
import os
import re
import shutil
import subprocess
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "oracle"))
import bench  # noqa: E402
import pyoracle  # noqa: E402

PAT = r"(findSuperBubble|PloidyEstimation)\(\):\s+(?:Finding superbubbles )?(Cpu|Real) time : ([0-9.e+-]+)s"


def md5_dir(d, prefix):
    out = {}
    for f in sorted(os.listdir(d)):
        h = hashlib.md5()
        with open(os.path.join(d, f), "rb") as fh:
            for blk in iter(lambda: fh.read(1 << 24), b""):
                h.update(blk)
        out[f[len(prefix) + 1:]] = {"md5": h.hexdigest(), "bytes": os.path.getsize(os.path.join(d, f))}
    return out


def timed(cmd, cwd):
    t0 = time.time()
    r = subprocess.run(cmd, cwd=cwd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
    wall = time.time() - t0
    t = {"%s_%s_s" % (a, b.lower()): float(c) for a, b, c in re.findall(PAT, r.stdout)}
    t["program_wall_s"] = round(wall, 2)
    t["returncode"] = r.returncode
    t["summary"] = [x.strip() for x in r.stdout.splitlines() if "SuperBubbles Found" in x or "Alleles in" in x]
    return t


def main():
    argv = sys.argv[1:]
    workload = "single"
    if "--workload" in argv:
        i = argv.index("--workload")
        workload = argv[i + 1]
        del argv[i:i + 2]
    threads = []
    while "--threads" in argv:
        i = argv.index("--threads")
        threads.append(int(argv[i + 1]))
        del argv[i:i + 2]
    args = [a for a in argv if not a.startswith("--")]
    out_json, target = args[0], int(args[1])
    wl = bench.WORKLOADS[workload]
    seed = int(args[2]) if len(args) > 2 else wl["seed"]
    import torch
    dev = "cuda" if torch.cuda.is_available() else "cpu"
    work = tempfile.mkdtemp(prefix="pf_refdig_", dir=os.environ.get("PF_REFDIG_TMP") or ("/dev/shm" if os.path.isdir("/dev/shm") else None))
    try:
        genome = int(target / bench.UNITIGS_PER_BP)
        if workload == "colored":
            gfa, colors, dbs, n_unitigs, n_kmers = bench.make_colored_inputs(work, "graph", genome, seed, dev, samples=3)
            with open(os.path.join(work, "dbs.txt"), "w") as f:
                f.write("".join(d + "\n" for d in dbs))
            with open(os.path.join(work, "cutoffs.txt"), "w") as f:
                f.write(("%d\t%d\n" % (bench.LOWER, bench.UPPER)) * len(dbs))
            base = [pyoracle.REF_BIN, "-g", gfa, "-f", colors, "-d", os.path.join(work, "dbs.txt"), "-C", os.path.join(work, "cutoffs.txt"),
                    "-o", "b", "-z", str(wl["z"])]
            gen = "bench.make_colored_inputs (3 diploid samples, k=25)"
        else:
            gfa, db, n_unitigs, n_kmers = bench.make_inputs(work, "graph", genome, seed, dev, k=wl["k"], **wl["gen"])
            base = [pyoracle.REF_BIN, "-g", gfa, "-d", db, "-o", "b", "-l", str(bench.LOWER), "-u", str(bench.UPPER), "-z", str(wl["z"])]
            gen = "bench.make_inputs (tetraploid, k=%d%s)" % (wl["k"], "".join(", %s=%s" % kv for kv in sorted(wl["gen"].items())))
        entry = {"unitigs": n_unitigs, "kmers": n_kmers, "seed": seed, "workload": workload, "generator": gen, "host": bench.cpu_model(),
                 "options": "-l %d -u %d -z %d (M=2 D=-1 G=-3)" % (bench.LOWER, bench.UPPER, wl["z"]), "date": time.strftime("%Y-%m-%d")}
        if "--skip-t1" not in sys.argv:
            cwd = os.path.join(work, "t1")
            os.makedirs(cwd)
            t = timed(base + ["-t", "1"], cwd)
            print("[refdig] -t 1: %s" % t, flush=True)
            entry["t1"] = t
            entry["files"] = md5_dir(os.path.join(cwd, "PloidyFrost_output"), "b")
            shutil.rmtree(cwd)
        for n in threads:
            cwd = os.path.join(work, "t%d" % n)
            os.makedirs(cwd)
            t = timed(base + ["-t", str(n)], cwd)
            print("[refdig] -t %d: %s" % (n, t), flush=True)
            entry["t%d" % n] = t
            shutil.rmtree(cwd)
        doc = {}
        if os.path.exists(out_json):
            doc = json.load(open(out_json))
        key = bench.digest_key(workload, n_unitigs)
        old = doc.get(key, {})
        old.update(entry)
        doc[key] = old
        doc["_meta"] = {"what": "oracle/_ref/PloidyFrost (the reference, built by oracle/Makefile.ref) on bench.py's graphs; made by tools/reference_digests.py "
                                "(each entry's `host` says where)"}
        os.makedirs(os.path.dirname(os.path.abspath(out_json)), exist_ok=True)
        json.dump(doc, open(out_json, "w"), indent=1, sort_keys=True)
    finally:
        shutil.rmtree(work, ignore_errors=True)


if __name__ == "__main__":
    main()
