"""Scale inside the GPU suite: bench-style graphs two orders of magnitude above the fixtures, all twelve files against the
reference binary itself (oracle/_ref/PloidyFrost -t 1, built from /root/reference by oracle/Makefile.ref and shipped as a
binary) -- or, where that binary is absent, the pinned CPU restatement.  Batch sizes are set so that the run crosses what the
fixtures never reach: several pipeline batches, K-BFS slices (second pass over the graph), growth of the path / text / site
pools between batches, the retry after an undersized pool."""
import os
import subprocess
import sys

import numpy as np
import pytest

from conftest import OUTPUT_SUFFIXES, ROOT

sys.path.insert(0, os.path.join(ROOT, "oracle"))
import pyoracle  # noqa: E402

from ploidyfrost_amd import hostapi  # noqa: E402

pytestmark = pytest.mark.gpu


def _checker(cmd_ref, cmd_oracle, cwd):
    os.makedirs(cwd, exist_ok=True)
    if os.path.exists(pyoracle.REF_BIN):
        r = subprocess.run([pyoracle.REF_BIN] + cmd_ref + ["-t", "1"], cwd=cwd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
        kind = "reference"
    else:
        pyoracle.build()
        r = subprocess.run([pyoracle.CLI] + cmd_oracle + ["-O", os.path.join(cwd, "PloidyFrost_output")], cwd=cwd, stdout=subprocess.PIPE,
                           stderr=subprocess.STDOUT, text=True)
        kind = "restatement"
    assert r.returncode == 0, r.stdout[-2000:]
    return kind, os.path.join(cwd, "PloidyFrost_output")


def _same(a_dir, b_dir, prefix):
    bad = []
    for suf in OUTPUT_SUFFIXES:
        with open(os.path.join(a_dir, "%s_%s.txt" % (prefix, suf)), "rb") as fa, open(os.path.join(b_dir, "%s_%s.txt" % (prefix, suf)), "rb") as fb:
            if fa.read() != fb.read():
                bad.append(suf)
    return bad


def test_200k_unitig_tetraploid_graph_matches_the_reference(tmp_path):
    import torch
    import bench
    work = str(tmp_path)
    gfa, db, n_unitigs, n_kmers = bench.make_inputs(work, "g", int(200_000 / bench.UNITIGS_PER_BP), 4711, torch.device("cuda", 0))
    assert n_unitigs > 150_000
    common = ["-g", gfa, "-d", db, "-o", "x", "-l", "5", "-u", "1000", "-z", "8"]
    kind, want = _checker(common, common, os.path.join(work, "cpu"))
    run = hostapi.Run(gfa, db, z=8)
    run.set_threads(8)
    run.set_batch_bubbles(3000)          # 12 000 bubbles per text piece ...
    run.set_align_pieces(2)              # ... and two pieces per alignment launch: several launches, pools sized by the first
    run.set_output_dir(os.path.join(work, "gpu"))
    run.set_unitig_id("x")
    for rep in range(2):                 # the second pass runs K-BFS in slices behind the replay and reuses every pool
        run.find_superbubbles("x")
        run.ploidy_estimation("x", 5, 1000)
        assert not _same(want, os.path.join(work, "gpu"), "x"), (kind, rep)
    t = run.times()
    assert t["tasks"] > 40_000 and t["candidates"] > 100_000, t
    # the same graph cut over three ranks (ploidyfrost_amd/dist.py played in-process)
    from ploidyfrost_amd import dist as pfdist
    shards = [run.find_shard(*pfdist.shard_range(n_unitigs, r, 3)) for r in range(3)]
    run.find_replay("x", [s[0] for s in shards], [s[1] for s in shards], write_file=True)
    nb = run.ploidy_select(5, 1000)
    sizes, called = [], 0
    out3 = os.path.join(work, "gpu3")
    os.makedirs(out3)
    for f in ("x_Unitig_Id.txt", "x_super_bubble.txt"):
        os.link(os.path.join(work, "gpu", f), os.path.join(out3, f))
    run.set_output_dir(out3)
    slabs = []
    for r in range(3):   # one context plays the ranks in turn: text and write of a rank before the next rank's align
        c = run.ploidy_align(*pfdist.shard_range(nb, r, 3))
        sz, _ = run.ploidy_text(called)
        called += c
        slabs.append(sz)
        # (one context plays the ranks in turn, so the final lengths are not known when the first slabs go out: no truncation
        # here -- the directory is fresh -- the real exchange of ploidyfrost_amd/dist.py knows the totals before anyone writes)
        run.ploidy_write("x", np.sum(slabs[:-1], axis=0).astype(np.uint64) if r else np.zeros(10, np.uint64), np.sum(slabs, axis=0).astype(np.uint64),
                         truncate=False)
    assert not _same(want, out3, "x")
    run.close()


def test_100k_unitig_three_colour_graph_matches_the_reference(tmp_path):
    import torch
    import bench
    work = str(tmp_path)
    gfa, colors, dbs, n_unitigs, _ = bench.make_colored_inputs(work, "g", int(100_000 / bench.UNITIGS_PER_BP), 4712, torch.device("cuda", 0))
    assert n_unitigs > 70_000
    lst, cut = os.path.join(work, "dbs.txt"), os.path.join(work, "cut.txt")
    open(lst, "w").write("".join(d + "\n" for d in dbs))
    open(cut, "w").write("5\t1000\n" * len(dbs))
    tail = ["-d", lst, "-C", cut, "-o", "x", "-z", "8"]
    dump = os.path.join(work, "colors.txt")
    if not os.path.exists(pyoracle.REF_BIN):
        if not os.path.exists(pyoracle.REF_COLORS_DUMP):
            pytest.skip("neither the reference binary nor the Bifrost colour dump is available for the colored checker")
        with open(dump, "w") as f:
            subprocess.run([pyoracle.REF_COLORS_DUMP, gfa, colors], check=True, stdout=f)
    kind, want = _checker(["-g", gfa, "-f", colors] + tail, ["-g", gfa, "-f", dump] + tail, os.path.join(work, "cpu"))
    run = hostapi.ColoredRun(gfa, colors, dbs, work, z=8, threads=8)
    run.set_batch_bubbles(5000)
    run.set_output_dir(os.path.join(work, "gpu"))
    run.set_unitig_id("x")
    for rep in range(2):
        run.find_superbubbles("x")
        run.ploidy_estimation("x", [(5, 1000)] * len(dbs))
        assert not _same(want, os.path.join(work, "gpu"), "x"), (kind, rep)
    run.close()


def test_150k_unitig_repeat_rich_graph_matches_the_reference(tmp_path):
    """bench.py --workload repeats at test size: 50 repeat families, inverted copies and tandem arrays on 8 % of the genome -- hundreds
    of traversals that outgrow the device tiers and are walked on host cores, commit components of thousands of records, cycles and
    hairpins at every scale -- against the reference binary; device commits and host commits both."""
    import torch
    import bench
    work = str(tmp_path)
    gfa, db, n_unitigs, n_kmers = bench.make_inputs(work, "g", int(150_000 / bench.UNITIGS_PER_BP), 2024, torch.device("cuda", 0), repeats=True)
    assert n_unitigs > 100_000
    common = ["-g", gfa, "-d", db, "-o", "x", "-l", "5", "-u", "1000", "-z", "8"]
    kind, want = _checker(common, common, os.path.join(work, "cpu"))
    run = hostapi.Run(gfa, db, z=8)
    run.set_threads(8)
    run.set_output_dir(os.path.join(work, "gpu"))
    run.set_unitig_id("x")
    for rep in range(2):
        run.find_superbubbles("x")
        run.ploidy_estimation("x", 5, 1000)
        assert not _same(want, os.path.join(work, "gpu"), "x"), (kind, rep)
    t = run.times()
    # the host tiers had work: traversals the device gave up on, records committed on a host thread
    assert t["bfs_deferred"] >= 50 and t["host_walk_vertices"] > 10 * t["bfs_deferred"] and t["host_commit_records"] >= t["bfs_deferred"], t
    run.close()
    # the same through the CLI with every long traversal on the device (k_bfs_big / k_bfs_huge) instead of the host walkers
    out = os.path.join(work, "cli")
    os.makedirs(out)
    cli = os.path.join(ROOT, "ploidyfrost_amd", "csrc", "ploidyfrost")
    r = subprocess.run([cli] + common + ["-t", "8"], cwd=out, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True,
                       env=dict(os.environ, PF_BFS_HUGE_ON_DEVICE="1"))
    assert r.returncode == 0, r.stdout
    assert not _same(want, os.path.join(out, "PloidyFrost_output"), "x")
