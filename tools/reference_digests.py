#!/usr/bin/env python3
"""Digests and timings of the REFERENCE binary (oracle/_ref/PloidyFrost) on bench.py's own graphs -- run on a GPU box (the graph
builder uses the device; the reference itself runs on the host's cores), committed as profiles/reference_digests.json and read by
bench.py: `output_check` holds the twelve files of the timed passes against these md5s, `cpu_baseline.reference_at_config_size`
quotes these timings.
    usage: tools/reference_digests.py <out.json> <unitigs> [seed] [--threads N ...] [--skip-t1]
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

PAT = r"(findSuperBubble|PloidyEstimation)\(\):\s+(Cpu|Real) time : ([0-9.e+-]+)s"


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
    args = [a for a in sys.argv[1:] if not a.startswith("--")]
    out_json, target = args[0], int(args[1])
    seed = int(args[2]) if len(args) > 2 else 1000
    threads = [int(sys.argv[i + 1]) for i, a in enumerate(sys.argv) if a == "--threads"]
    import torch
    dev = "cuda" if torch.cuda.is_available() else "cpu"
    work = tempfile.mkdtemp(prefix="pf_refdig_", dir="/dev/shm" if os.path.isdir("/dev/shm") else None)
    try:
        gfa, db, n_unitigs, n_kmers = bench.make_inputs(work, "graph", int(target / bench.UNITIGS_PER_BP), seed, dev)
        entry = {"unitigs": n_unitigs, "kmers": n_kmers, "seed": seed, "generator": "bench.make_inputs (tetraploid, k=25)", "host": bench.cpu_model(),
                 "options": "-l %d -u %d -z %d (M=2 D=-1 G=-3)" % (bench.LOWER, bench.UPPER, bench.Z), "date": time.strftime("%Y-%m-%d")}
        base = [pyoracle.REF_BIN, "-g", gfa, "-d", db, "-o", "b", "-l", str(bench.LOWER), "-u", str(bench.UPPER), "-z", str(bench.Z)]
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
        old = doc.get(str(n_unitigs), {})
        old.update(entry)
        doc[str(n_unitigs)] = old
        doc["_meta"] = {"what": "oracle/_ref/PloidyFrost (the reference, built by oracle/Makefile.ref) on bench.py's graphs; made by tools/reference_digests.py on a GPU box"}
        os.makedirs(os.path.dirname(os.path.abspath(out_json)), exist_ok=True)
        json.dump(doc, open(out_json, "w"), indent=1, sort_keys=True)
    finally:
        shutil.rmtree(work, ignore_errors=True)


if __name__ == "__main__":
    main()
