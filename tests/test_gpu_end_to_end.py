"""The whole path through the product (C++ host layer + HIP kernels) against the committed
outputs of the real reference binary: all twelve files byte-identical, for every fixture."""
import ctypes as C
import os
import subprocess
import sys

import numpy as np
import pytest

from conftest import ROOT, abundant_cases, compare_outputs, golden_cases, load_case

sys.path.insert(0, os.path.join(ROOT, "oracle"))
import pyoracle  # noqa: E402

from ploidyfrost_amd import hostapi  # noqa: E402

pytestmark = pytest.mark.gpu
CLI = os.path.join(ROOT, "ploidyfrost_amd", "csrc", "ploidyfrost")


@pytest.mark.parametrize("case", golden_cases() + abundant_cases())
def test_cli_outputs_match_reference(case, tmp_path):
    meta = load_case(case)
    r = subprocess.run([CLI, "-g", meta["gfa"], "-d", meta["db"], "-o", "g", "-t", "1"] + meta["args"], cwd=tmp_path,
                       stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
    assert r.returncode == 0, r.stdout
    bad = compare_outputs(os.path.join(meta["dir"], "expected"), os.path.join(tmp_path, "PloidyFrost_output"))
    assert not bad, "files differ from the reference: %s\n%s" % (bad, r.stdout)
    # the summary lines the reference prints
    for line in meta["reference_log"]:
        assert line.strip() in r.stdout.replace("\r", ""), line


@pytest.mark.parametrize("case", ["hex30k", "k31_z16", "giant7k"])
def test_alignseq_written_as_text_on_the_device_gives_the_same(case, tmp_path):
    """alignseq.txt leaves the device packed by default (csrc/pf_alnpack.hpp: a header per bubble, rows at 3 bits per character)
    and becomes text in the host's writer; PF_ALIGNSEQ_ASCII=1 keeps the device writing the rows itself.  Small pieces too, so that
    pieces end inside index groups."""
    meta = load_case(case)
    for env in ({"PF_ALIGNSEQ_ASCII": "1"}, {"PF_BATCH_BUBBLES": "97"}, {"PF_BATCH_BUBBLES": "64", "PF_ALIGNSEQ_ASCII": "1"}):
        out = tmp_path / "_".join(env)
        out.mkdir()
        r = subprocess.run([CLI, "-g", meta["gfa"], "-d", meta["db"], "-o", "g", "-t", "4"] + meta["args"], cwd=out, stdout=subprocess.PIPE, stderr=subprocess.STDOUT,
                           text=True, env=dict(os.environ, **env))
        assert r.returncode == 0, r.stdout
        bad = compare_outputs(os.path.join(meta["dir"], "expected"), os.path.join(out, "PloidyFrost_output"))
        assert not bad, "%s: files differ from the reference: %s" % (env, bad)


@pytest.mark.parametrize("case", ["tet60k", "stranded20k", "hex30k"])
def test_numeric_streams_at_four_bits_a_character_and_as_text_give_the_same(case, tmp_path):
    """the nine numeric streams leave the device at four bits a character (K-NIB) and become text in the host's writer; a stream
    with other characters than the sixteen -- stranded20k: "nan" / "inf" in its frequencies -- is fetched as text instead;
    PF_NUMERIC_ASCII=1 sends all of them as text.  Odd piece sizes too (pieces that end on an odd character)."""
    meta = load_case(case)
    if case == "stranded20k":   # (the fixture must keep exercising the way out)
        exp = open(os.path.join(meta["dir"], "expected", "g_allele_frequency.txt")).read()
        assert "nan" in exp or "inf" in exp
    for env in ({}, {"PF_NUMERIC_ASCII": "1"}, {"PF_BATCH_BUBBLES": "97"}, {"PF_BATCH_BUBBLES": "33", "PF_ALIGNSEQ_ASCII": "1"}):
        out = tmp_path / ("default" if not env else "_".join(env))
        out.mkdir()
        r = subprocess.run([CLI, "-g", meta["gfa"], "-d", meta["db"], "-o", "g", "-t", "4"] + meta["args"], cwd=out, stdout=subprocess.PIPE, stderr=subprocess.STDOUT,
                           text=True, env=dict(os.environ, **env))
        assert r.returncode == 0, r.stdout
        bad = compare_outputs(os.path.join(meta["dir"], "expected"), os.path.join(out, "PloidyFrost_output"))
        assert not bad, "%s: files differ from the reference: %s" % (env, bad)


@pytest.mark.parametrize("case,world", [("tet60k", 2), ("hex30k", 3), ("giant7k", 2), ("stranded20k", 4)])
def test_cli_cuts_one_graph_over_ranks(case, world, tmp_path):
    """`ploidyfrost --gpus N` (csrc/host/pf_multi.hpp): N processes forked before anything touches the GPU, every rank the whole
    graph, rank r its slice of the bubble list, two small all-gathers, every rank's slabs straight into the shared files.  On this
    one-GPU box the ranks share the device (PF_SHARE_GPU=1: the words then travel over the socket pairs instead of RCCL, which refuses
    two ranks on one device); the files must be the reference's, the summary lines rank 0's."""
    meta = load_case(case)
    r = subprocess.run([CLI, "-g", meta["gfa"], "-d", meta["db"], "-o", "g", "-t", "2", "--gpus", str(world), "-v"] + meta["args"], cwd=tmp_path,
                       stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, env=dict(os.environ, PF_SHARE_GPU="1"), timeout=600)
    assert r.returncode == 0, r.stdout
    bad = compare_outputs(os.path.join(meta["dir"], "expected"), os.path.join(tmp_path, "PloidyFrost_output"))
    assert not bad, "files differ from the reference: %s\n%s" % (bad, r.stdout)
    for line in meta["reference_log"]:
        assert line.strip() in r.stdout.replace("\r", ""), line
    assert "[ranks]  %d ranks" % world in r.stdout


def test_cli_ranks_end_together_when_one_fails(tmp_path):
    """a k-mer of the graph that is in no database ends the reference's run (src/CDBG.cpp:92-96): every rank of a cut run leaves, rank 0
    with a non-zero status"""
    from ploidyfrost_amd import synth
    meta = load_case("dip20k")
    kmers, counts, km = synth.read_kmc(meta["db"])
    keep = np.ones(len(kmers), dtype=bool)
    keep[::3] = False
    synth.write_kmc1(str(tmp_path / "holes"), kmers[keep], counts[keep], km["k"])
    r = subprocess.run([CLI, "-g", meta["gfa"], "-d", str(tmp_path / "holes"), "-o", "g", "-l", "5", "--gpus", "2"], cwd=tmp_path,
                       stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, env=dict(os.environ, PF_SHARE_GPU="1"), timeout=300)
    assert r.returncode != 0 and "can not found" in r.stdout


@pytest.mark.parametrize("who,stage", [(1, "align"), (0, "text"), (2, "load"), (1, "findSuperBubble")])
def test_cli_one_rank_failing_alone_ends_every_rank(who, stage, tmp_path):
    """A rank that fails ALONE (PF_FAIL_RANK, the test seam of RankGroup::agree: over RCCL the others would wait for it for ever in
    ncclCommInitRank / the all-gather): every rank says why it leaves, rank 0 ends non-zero, nothing hangs."""
    meta = load_case("dip20k")
    r = subprocess.run([CLI, "-g", meta["gfa"], "-d", meta["db"], "-o", "g", "-t", "2", "--gpus", "3"] + meta["args"], cwd=tmp_path,
                       stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, env=dict(os.environ, PF_SHARE_GPU="1", PF_FAIL_RANK="%d:%s" % (who, stage)),
                       timeout=120)
    assert r.returncode != 0, r.stdout
    assert "rank %d: injected failure at %s" % (who, stage) in r.stdout or "rank %d: leaving" % who in r.stdout, r.stdout
    assert "another rank failed before %s" % stage in r.stdout, r.stdout


@pytest.mark.parametrize("case", ["tet60k", "weird12k", "hex30k"])
def test_state_after_find_superbubbles_matches_oracle(case, tmp_path):
    meta = load_case(case)
    op = meta["opts"]
    run = hostapi.Run(meta["gfa"], meta["db"], z=int(op["-z"]))
    run.set_output_dir(str(tmp_path))
    run.set_unitig_id("g")
    run.find_superbubbles("g")
    f, p, m = run.state()
    o = pyoracle.Oracle(meta["gfa"], meta["db"])
    nb = o.find_superbubbles(z=int(op["-z"]))
    ef, ep, em = o.state()
    assert np.array_equal(f, ef) and np.array_equal(p, ep) and np.array_equal(m, em)
    assert run.times()["superbubbles"] == nb


def test_facade_full_run_and_no_disk_mode(tmp_path):
    meta = load_case("tet60k")
    run = hostapi.Run(meta["gfa"], meta["db"])
    run.set_output_dir(str(tmp_path))
    run.set_unitig_id("g")
    run.find_superbubbles("g")
    run.ploidy_estimation("g", 5, 1000)
    assert not compare_outputs(os.path.join(meta["dir"], "expected"), str(tmp_path))
    t = run.times()
    assert t["tasks"] > 0 and t["align_jobs"] >= t["tasks"] and t["output_bytes"] > 0
    # format-only mode (bench): same bytes produced, nothing written
    run2 = hostapi.Run(meta["gfa"], meta["db"])
    d2 = tmp_path / "none"
    run2.set_output_dir(str(d2))
    run2.set_write_files(False)
    run2.set_unitig_id("g")
    run2.find_superbubbles("g")
    run2.ploidy_estimation("g", 5, 1000)
    assert run2.times()["output_bytes"] == t["output_bytes"] and not d2.exists()
    # the rows of allele_frequency.txt as the caller's ranks exchange them: read back from the file after a pass that wrote it,
    # kept in memory by a pass that wrote nothing
    want = open(os.path.join(meta["dir"], "expected", "g_allele_frequency.txt"), "rb").read()
    assert run.last_allele_frequency().tobytes() == want
    assert run2.last_allele_frequency().tobytes() == want
    run.find_superbubbles("g")
    run.ploidy_estimation("g", 5, 1000)   # (a second pass over the same files)
    assert run.last_allele_frequency().tobytes() == want


@pytest.mark.parametrize("case", ["hex30k", "k31_z16", "tet_frac"])
def test_host_threads_do_not_change_a_byte(case, tmp_path):
    """-t in the reference reorders rows; here threads only speed up the per-bubble phases"""
    meta = load_case(case)
    op = meta["opts"]
    run = hostapi.Run(meta["gfa"], meta["db"], z=int(op["-z"]), M=float(op["-M"]), D=float(op["-D"]), G=float(op["-G"]))
    run.set_threads(7)
    run.set_output_dir(str(tmp_path))
    run.set_unitig_id("g")
    run.find_superbubbles("g")
    run.ploidy_estimation("g", int(op["-l"]), int(op["-u"]))
    assert not compare_outputs(os.path.join(meta["dir"], "expected"), str(tmp_path))


def test_missing_database_kmer_fails_like_the_reference(tmp_path):
    """reference: "kmer can not found" + exit(EXIT_FAILURE) (src/CDBG.cpp:52-56, 92-96)"""
    from ploidyfrost_amd import synth
    meta = load_case("dip20k")
    kmers, counts, km = synth.read_kmc(meta["db"])
    keep = np.ones(len(kmers), dtype=bool)
    keep[::3] = False
    synth.write_kmc1(str(tmp_path / "holes"), kmers[keep], counts[keep], km["k"])
    r = subprocess.run([CLI, "-g", meta["gfa"], "-d", str(tmp_path / "holes"), "-o", "g", "-l", "5"], cwd=tmp_path,
                       stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
    assert r.returncode != 0 and "can not found" in r.stdout


@pytest.mark.parametrize("case,batch", [("tet60k", 64), ("hex30k", 1), ("k31_z16", 500), ("weird12k", 37)])
def test_pipeline_batches_do_not_change_a_byte(case, batch, tmp_path):
    """PloidyEstimation runs its bubbles through a two-stage pipeline in batches; tiny batches force many hand-overs."""
    meta = load_case(case)
    op = meta["opts"]
    run = hostapi.Run(meta["gfa"], meta["db"], z=int(op["-z"]), M=float(op["-M"]), D=float(op["-D"]), G=float(op["-G"]))
    run.set_threads(5)
    run.set_batch_bubbles(batch)
    run.set_output_dir(str(tmp_path))
    run.set_unitig_id("g")
    for _ in range(2):  # the second pass reuses the exchange buffers
        run.find_superbubbles("g")
        run.ploidy_estimation("g", int(op["-l"]), int(op["-u"]))
        assert not compare_outputs(os.path.join(meta["dir"], "expected"), str(tmp_path))


@pytest.mark.parametrize("ranges,aligners", [("2", "1"), ("3", "1"), ("5", "1"), ("6", "2"), ("9", "4"), ("13", "3")])
def test_align_ranges_and_text_lanes_soak(ranges, aligners, tmp_path, monkeypatch):
    """The calling pipeline is four threads and more -- aligners (PF_ALIGN_THREADS of them, each pf_call_align_lane on a lane of
    its own) | format range r | fetch | write -- over PF_CALL_LANES result lanes and four device slabs.  Tiny pieces (4 bubbles)
    and forced range counts make every hand-over happen dozens of times per pass; forty passes must give the reference's bytes
    every time."""
    monkeypatch.setenv("PF_ALIGN_RANGES", ranges)
    monkeypatch.setenv("PF_ALIGN_THREADS", aligners)
    meta = load_case("tet60k")
    op = meta["opts"]
    run = hostapi.Run(meta["gfa"], meta["db"], z=int(op["-z"]), M=float(op["-M"]), D=float(op["-D"]), G=float(op["-G"]))
    run.set_threads(6)
    run.set_batch_bubbles(1)
    run.set_overlap_output(True)
    run.set_output_dir(str(tmp_path))
    run.set_unitig_id("g")
    for _ in range(40):
        run.find_superbubbles("g")
        run.ploidy_estimation("g", int(op["-l"]), int(op["-u"]))
        assert not compare_outputs(os.path.join(meta["dir"], "expected"), str(tmp_path))
    run.close()


def test_overlapped_output_is_complete_when_ploidy_returns(tmp_path):
    meta = load_case("tet60k")
    run = hostapi.Run(meta["gfa"], meta["db"])
    run.set_overlap_output(True)
    run.set_batch_bubbles(200)
    run.set_output_dir(str(tmp_path))
    run.set_unitig_id("g")
    for _ in range(3):
        run.find_superbubbles("g")
        run.ploidy_estimation("g", 5, 1000)
        assert not compare_outputs(os.path.join(meta["dir"], "expected"), str(tmp_path))
    run.find_superbubbles("g")   # pending write joined by close()
    run.close()
    assert open(os.path.join(meta["dir"], "expected", "g_super_bubble.txt"), "rb").read() == open(tmp_path / "g_super_bubble.txt", "rb").read()


def test_third_tier_on_host_and_on_device_agree_with_the_oracle(tmp_path):
    """giant7k: a repeat links two loci, the traversals entering it visit > 4096 vertices.  Those run on host cores by default
    (pf_bfs_candidates_split + pf_bfs_host.cpp) or on the device (k_bfs_huge); state, counters and files must not differ."""
    meta = load_case("giant7k")
    op = meta["opts"]
    o = pyoracle.Oracle(meta["gfa"], meta["db"])
    nb = o.find_superbubbles(z=int(op["-z"]))
    ef, ep, em = o.state()
    for on_host in (True, False):
        d = tmp_path / ("host" if on_host else "device")
        run = hostapi.Run(meta["gfa"], meta["db"], z=int(op["-z"]))
        run.set_third_tier_on_host(on_host)
        run.set_threads(4)
        run.set_output_dir(str(d))
        run.set_unitig_id("g")
        run.find_superbubbles("g")
        f, p, m = run.state()
        assert np.array_equal(f, ef) and np.array_equal(p, ep) and np.array_equal(m, em), on_host
        t = run.times()
        assert t["superbubbles"] == nb and t["bfs_large"] >= 1, t
        run.ploidy_estimation("g", int(op["-l"]), int(op["-u"]))
        assert not compare_outputs(os.path.join(meta["dir"], "expected"), str(d)), on_host


@pytest.mark.parametrize("case,world", [("tet60k", 2), ("hex30k", 3), ("weird12k", 5), ("k31_z16", 4), ("giant7k", 2), ("stranded20k", 3)])
def test_one_graph_partitioned_over_ranks(case, world, tmp_path):
    """SURVEY.md 8e, the protocol of ploidyfrost_amd/dist.py with the ranks played one after the other on this GPU: every rank
    traverses the entrances of its unitig range (K-BFS + host walkers), the records of all ranks are replayed on every rank,
    each rank aligns its slice of the bubble list, learns how many bubbles the ranks before it called, formats, and writes its
    slabs at its offsets of the shared files.  Result: the reference's twelve files, and counters that add up."""
    from ploidyfrost_amd import dist as pfdist
    meta = load_case(case)
    op = meta["opts"]
    out = str(tmp_path / "shared")
    runs = []
    for rank in range(world):
        run = hostapi.Run(meta["gfa"], meta["db"], z=int(op["-z"]), M=float(op["-M"]), D=float(op["-D"]), G=float(op["-G"]))
        run.set_threads(3)
        run.set_output_dir(out)
        if case in ("hex30k", "k31_z16"):
            run.set_batch_bubbles(4)   # (a slice is written in pieces of 4 x this many bubbles: dozens of pieces here, one elsewhere)
        runs.append(run)
    runs[0].set_unitig_id("g")
    n = runs[0].times()["unitigs"]
    shards = [runs[r].find_shard(*pfdist.shard_range(n, r, world)) for r in range(world)]
    n_shard_records = sum(len(s[0]) for s in shards)
    for r in range(world):   # all-gather: every rank sees every shard
        runs[r].find_replay("g", [s[0] for s in shards], [s[1] for s in shards], write_file=r == 0)
    states = [run.state() for run in runs]
    for st in states[1:]:
        assert all(np.array_equal(a, b) for a, b in zip(states[0], st))
    nb = [run.ploidy_select(int(op["-l"]), int(op["-u"])) for run in runs]
    assert len(set(nb)) == 1
    slices = [pfdist.shard_range(nb[0], r, world) for r in range(world)]
    called = [runs[r].ploidy_align(*slices[r]) for r in range(world)]
    texts = [runs[r].ploidy_text(int(sum(called[:r]))) for r in range(world)]
    sizes = np.array([t[0] for t in texts], dtype=np.uint64)
    totals = sizes.sum(axis=0)
    for r in reversed(range(world)):   # any order: every slab has its own place
        runs[r].ploidy_write("g", sizes[:r].sum(axis=0), totals, truncate=True)
    assert not compare_outputs(os.path.join(meta["dir"], "expected"), out)
    counters = np.array([t[1] for t in texts], dtype=np.int64).sum(axis=0)
    whole = hostapi.Run(meta["gfa"], meta["db"], z=int(op["-z"]), M=float(op["-M"]), D=float(op["-D"]), G=float(op["-G"]))
    whole.set_output_dir(str(tmp_path / "whole"))
    whole.set_unitig_id("g")
    whole.find_superbubbles("g")
    whole.ploidy_estimation("g", int(op["-l"]), int(op["-u"]))
    tw = whole.times()
    assert list(counters[:4]) == tw["allele"] and int(counters[7]) == tw["tasks"] == nb[0]
    assert n_shard_records == tw["candidates"]   # every candidate entrance belongs to exactly one rank's unitig range
    assert all(np.array_equal(a, b) for a, b in zip(states[0], _state_after_find(meta, op)))
    for run in runs + [whole]:
        run.close()


@pytest.mark.parametrize("case,world", [("tet60k", 2), ("giant7k", 3), ("weird12k", 4)])
def test_shards_replayed_from_device_memory(case, world, tmp_path):
    """The branch of dist.sharded_find an RCCL run takes (ploidyfrost_amd/dist.py: `on_gpu and world > 1`): after the all-gather the
    shards of all ranks lie in DEVICE memory only, and pfh_find_replay is given their device addresses and sizes -- no host
    copy (records=None).  Here the gathered buffers are torch.cuda tensors filled from each rank's shard; the per-unitig state
    afterwards is the oracle's (reference CDBG::findSuperBubble_ptr, src/CDBG.cpp:178-252 with the commits of :552-846) and
    super_bubble.txt the reference's."""
    import torch
    from ploidyfrost_amd import dist as pfdist, hipapi
    meta = load_case(case)
    op = meta["opts"]
    o = pyoracle.Oracle(meta["gfa"], meta["db"])
    nb = o.find_superbubbles(z=int(op["-z"]))
    want = o.state()
    run = hostapi.Run(meta["gfa"], meta["db"], z=int(op["-z"]))
    run.set_threads(3)
    run.set_output_dir(str(tmp_path))
    run.set_unitig_id("g")
    n = run.times()["unitigs"]
    dev = torch.device("cuda:0")
    d_recs, d_pools, sizes = [], [], []
    for r in range(world):
        rec, pool = run.find_shard(*pfdist.shard_range(n, r, world))
        # (an all-gather pads every shard to the longest one: the sizes say how much of each buffer counts)
        tr = torch.zeros(rec.nbytes + 64 * r + 8, dtype=torch.uint8, device=dev)
        tp = torch.zeros(pool.nbytes + 32 * r + 8, dtype=torch.uint8, device=dev)
        if rec.nbytes:
            tr[:rec.nbytes] = torch.from_numpy(rec.view(np.uint8).reshape(-1).copy()).to(dev)
        if pool.nbytes:
            tp[:pool.nbytes] = torch.from_numpy(pool.view(np.uint8).reshape(-1).copy()).to(dev)
        d_recs.append(tr)
        d_pools.append(tp)
        sizes.append((len(rec), len(pool)))
    torch.cuda.synchronize()
    run.find_replay("g", None, sizes, write_file=True, dev_records=[t.data_ptr() for t in d_recs], dev_pools=[t.data_ptr() for t in d_pools])
    got = run.state()
    assert all(np.array_equal(a, b) for a, b in zip(got, want)), case
    assert run.times()["superbubbles"] == nb
    with open(os.path.join(meta["dir"], "expected", "g_super_bubble.txt"), "rb") as f, open(str(tmp_path / "g_super_bubble.txt"), "rb") as g:
        assert f.read() == g.read()
    # and the calling phase on that state gives the reference's files
    run.ploidy_estimation("g", int(op["-l"]), int(op["-u"]))
    assert not compare_outputs(os.path.join(meta["dir"], "expected"), str(tmp_path))
    run.close()


def test_a_rank_with_an_empty_slice_reports_nothing(tmp_path):
    """fewer bubbles than ranks (dist.sharded_ploidy): the rank whose slice is empty must report zero called bubbles, zero sizes and
    zero counters -- not the figures of the pass before (pf_call_align_lane left them in the lane)"""
    meta = load_case("dip20k")
    op = meta["opts"]
    run = hostapi.Run(meta["gfa"], meta["db"], z=int(op["-z"]))
    run.set_output_dir(str(tmp_path))
    run.set_unitig_id("g")
    run.find_superbubbles("g")
    nb = run.ploidy_select(int(op["-l"]), int(op["-u"]))
    assert run.ploidy_align(0, nb) > 0
    sizes, counters = run.ploidy_text(0)
    assert sizes.sum() > 0 and counters[6] > 0
    for t in (0, nb // 2, nb):
        assert run.ploidy_align(t, t) == 0
        sizes, counters = run.ploidy_text(7)
        assert sizes.sum() == 0 and not counters[:7].any(), (t, sizes, counters)
    run.close()


def _state_after_find(meta, op):
    run = hostapi.Run(meta["gfa"], meta["db"], z=int(op["-z"]))
    run.set_write_files(False)
    run.find_superbubbles("g")
    st = run.state()
    run.close()
    return st


T2_CASES = [c for c in ["dip20k", "tet60k", "hex30k", "weird12k", "k31_z16"] if os.path.exists(os.path.join(ROOT, "tests", "golden", c, "expected_t2.json"))]


@pytest.mark.parametrize("case", T2_CASES)
def test_reference_threads_format_matches_the_threaded_reference(case, tmp_path):
    """the text format of the reference's `-t N` functions (N > 1) -- BubbleId / var_count from 0, the
    allele_frequency rows of a bubble grouped by arity, rows of arity > 5 absent, penta rows absent for branching bubbles.  The
    threaded reference numbers unitigs and bubbles and orders rows by thread timing, so the comparison is on the canonical form
    of tests/t2_canonical.py against fixtures made from `oracle/_ref/PloidyFrost -t 2` (tests/golden/make_t2_golden.py)."""
    import hashlib
    import json
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from t2_canonical import canonical
    meta = load_case(case)
    op = meta["opts"]
    want = json.load(open(os.path.join(meta["dir"], "expected_t2.json")))
    out = tmp_path / "cli"
    out.mkdir()
    cmd = [CLI, "-g", meta["gfa"], "-d", meta["db"], "-o", "g", "-t", "4", "--ref-threads", "2"] + meta["args"]
    r = subprocess.run(cmd, cwd=str(out), stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
    assert r.returncode == 0, r.stdout[-2000:]
    assert "CDBG::findSuperBubble(): Finding superbubbles Cpu time" in r.stdout and "CDBG::PloidyEstimation(): Cpu time" in r.stdout
    got = canonical(str(out / "PloidyFrost_output"), "g")
    bad = [k for k, v in want.items() if hashlib.sha256(got[k].encode()).hexdigest() != v["sha256"]]
    assert not bad, {k: (got[k].count("\n"), want[k]["lines"]) for k in bad}
    # and the same through the facade, against this run's own `-t 1` files: ids shifted by one, nothing else in the id-bearing files
    run = hostapi.Run(meta["gfa"], meta["db"], z=int(op["-z"]), M=float(op["-M"]), D=float(op["-D"]), G=float(op["-G"]))
    run.set_reference_threads(3)
    run.set_output_dir(str(tmp_path / "t3"))
    run.set_unitig_id("g")
    run.find_superbubbles("g")
    run.ploidy_estimation("g", int(op["-l"]), int(op["-u"]))
    exp = os.path.join(meta["dir"], "expected")
    rows1 = open(os.path.join(exp, "g_super_bubble.txt")).read().splitlines()[1:]
    rows3 = open(tmp_path / "t3" / "g_super_bubble.txt").read().splitlines()[1:]
    assert [r.split("\t", 1)[1] for r in rows1] == [r.split("\t", 1)[1] for r in rows3]
    assert [int(r.split("\t", 1)[0]) for r in rows3] == list(range(len(rows3)))
    a1 = open(os.path.join(exp, "g_alignseq.txt")).read().splitlines()
    a3 = open(tmp_path / "t3" / "g_alignseq.txt").read().splitlines()
    assert [r.split("\t", 1)[1] for r in a1] == [r.split("\t", 1)[1] for r in a3]
    assert [int(r.split("\t", 1)[0]) - 1 for r in a1] == [int(r.split("\t", 1)[0]) for r in a3]
    for name in ("bifre", "trifre", "tetrafre", "pentafre"):
        assert open(os.path.join(exp, "g_%s.txt" % name), "rb").read() == open(tmp_path / "t3" / ("g_%s.txt" % name), "rb").read()
    run.close()


def _braid(tmp_path, k, layers, seed=11):
    """A superbubble of 2^layers paths over 2 * layers + 2 unitigs: every unitig of a layer is followed by both unitigs of the
    next one (they share their last k-1 bases), so no inner pair of layers closes into a bubble of its own."""
    from ploidyfrost_amd import synth
    rng = np.random.default_rng(seed)
    code = {ord("A"): 0, ord("C"): 1, ord("G"): 2, ord("T"): 3}

    def rnd(n):
        return bytes(rng.choice(list(b"ACGT"), size=n).tolist())

    junction = [rnd(k - 1) for _ in range(layers + 1)]
    segs = [rnd(40) + junction[0]]
    for i in range(layers):
        segs.append(junction[i] + b"A" + rnd(6) + junction[i + 1])
        segs.append(junction[i] + b"C" + rnd(6) + junction[i + 1])
    segs.append(junction[layers] + b"G" + rnd(40))
    gfa = str(tmp_path / "braid.gfa")
    with open(gfa, "wb") as f:
        f.write(b"H\tVN:Z:1.0\tKL:Z:%d\tML:Z:%d\n" % (k, k - 8))
        for i, s in enumerate(segs):
            f.write(b"S\t%d\t%s\n" % (i + 1, s))
    # the database holds the k-mers of four walks that use every edge between the layers
    walks = []
    for pick in (lambda i: 0, lambda i: 1, lambda i: i % 2, lambda i: 1 - i % 2):
        w = segs[0]
        for i in range(layers):
            w += segs[1 + 2 * i + pick(i)][k - 1:]
        w += segs[-1][k - 1:]
        walks.append(np.array([code[c] for c in w], dtype=np.uint8))
    km, mult = synth.canonical_counts(walks, k)
    db = str(tmp_path / "braid_db")
    synth.write_kmc1(db, km, np.full(len(km), 20, dtype=np.uint32), k)
    return gfa, db, len(segs)


def test_bubbles_of_more_than_255_paths(tmp_path):
    """K-PATHS keeps the tables of a bubble's walks in LDS (255 of them); a bubble with more leaves its first launch for a list and
    is walked by a second launch with global tables (65 535), K-SITES sizes its row tables by the largest bubble of the batch: a
    braid of 256 walks (8 layers, owned by its entrance so that the walk finds all of them) and one of 512 (9 layers) come out as
    the oracle writes them.  Walked from the other end the reference's two-stack walk loses walks in a braid (its backtracking rule:
    "is the next vertex a successor of the top of the path stack?") -- and so does K-PATHS, to the same rows."""
    pyoracle.build()
    seen_rows = []
    for layers, seed, n_rows in ((7, 11, 128), (8, 11, 32), (8, 12, 256), (9, 12, 256), (9, 16, 512)):
        sub = tmp_path / ("braid_%d_%d" % (layers, seed))
        sub.mkdir()
        gfa, db, n = _braid(sub, 25, layers, seed=seed)
        r = subprocess.run([CLI, "-g", gfa, "-d", db, "-o", "g", "-z", "40", "-l", "1", "-u", "1000", "-t", "2"], cwd=sub, capture_output=True, text=True)
        assert r.returncode == 0, (r.stdout[-400:], r.stderr[-400:])
        rows = open(os.path.join(str(sub), "PloidyFrost_output", "g_alignseq.txt")).read()
        assert n_rows is None or rows.count("\n") == n_rows, rows.count("\n")
        seen_rows.append(rows.count("\n"))
        want = sub / "oracle"
        want.mkdir()
        ro = subprocess.run([pyoracle.CLI, "-g", gfa, "-d", db, "-o", "g", "-z", "40", "-l", "1", "-u", "1000", "-O", str(want / "PloidyFrost_output")], cwd=want,
                            capture_output=True, text=True)
        assert ro.returncode == 0, ro.stderr[-300:]
        assert not compare_outputs(str(want / "PloidyFrost_output"), os.path.join(str(sub), "PloidyFrost_output")), (layers, seed)
    assert max(seen_rows) >= 512, seen_rows   # (one of the 9-layer braids is walked from its entrance)


@pytest.mark.parametrize("case", ["weird12k", "giant7k", "hex30k", "tet_frac"])
@pytest.mark.parametrize("mode", ["limit0", "limit6", "host", "seq"])
def test_commits_on_the_device_the_host_and_both(case, mode, tmp_path):
    """findSuperBubble's commits: on the device one thread per component (default, the other tests), with the components above a
    tiny limit -- or all of them -- committed on the host and patched into the device state, on host threads, and in the
    sequential loop: the same twelve files."""
    meta = load_case(case)
    env = dict(os.environ)
    env.pop("PF_REPLAY", None)
    if mode.startswith("limit"):
        env["PF_REPLAY_SMALL_LIMIT"] = mode[5:]
    else:
        env["PF_REPLAY"] = mode
    r = subprocess.run([CLI, "-g", meta["gfa"], "-d", meta["db"], "-o", "g", "-t", "4"] + meta["args"], cwd=tmp_path, capture_output=True, text=True, env=env)
    assert r.returncode == 0, r.stderr[-400:] + r.stdout[-400:]
    assert not compare_outputs(os.path.join(meta["dir"], "expected"), os.path.join(str(tmp_path), "PloidyFrost_output")), (case, mode)


@pytest.mark.parametrize("case", ["weird12k", "giant7k", "hex30k"])
@pytest.mark.parametrize("knob", ["PF_BFS_HINT_AT=9", "PF_BFS_WAVE_CAP=16", "PF_BFS_HINT_AT=9 PF_BFS_WAVE_CAP=16"])
def test_live_notices_and_early_give_ups_do_not_change_a_byte(case, knob, tmp_path):
    """K-BFS's wave tier enters a traversal into the live list the host walkers poll when it reaches 48 vertices -- a notice: the
    traversal goes on on the device, and a walk whose traversal ends there is dropped.  With the notice at 9 vertices most notices are
    of that kind; with the tier giving up at 16 vertices instead of 128 many more traversals are walked on the host and their
    components committed there.  The same twelve files either way."""
    meta = load_case(case)
    env = dict(os.environ)
    env.update(dict(kv.split("=") for kv in knob.split()))
    env["PF_TRACE_FIND"] = "1"
    r = subprocess.run([CLI, "-g", meta["gfa"], "-d", meta["db"], "-o", "g", "-t", "4"] + meta["args"], cwd=tmp_path, capture_output=True, text=True, env=env)
    assert r.returncode == 0, r.stderr[-400:] + r.stdout[-400:]
    assert not compare_outputs(os.path.join(meta["dir"], "expected"), os.path.join(str(tmp_path), "PloidyFrost_output")), (case, knob)
    import re
    m = re.search(r"(\d+) notices in the live list, (\d+) traversals given up", r.stderr)
    assert m, r.stderr[-400:]
    notices, given_up = int(m.group(1)), int(m.group(2))
    assert notices >= given_up, (notices, given_up)
    if case == "giant7k":
        assert given_up > 0
    if case == "weird12k" and "WAVE_CAP" not in knob:
        assert notices > given_up, (notices, given_up)   # noticed traversals that ended on the device: their walks were dropped


def test_graph_without_a_candidate_entrance(tmp_path):
    """three unitigs that share no k-mer overlap: no vertex has two successors, findSuperBubble has nothing to traverse or commit"""
    from ploidyfrost_amd import synth
    rng = np.random.default_rng(3)
    segs = [bytes(rng.choice(list(b"ACGT"), size=n).tolist()) for n in (80, 25, 140)]
    gfa = str(tmp_path / "lone.gfa")
    with open(gfa, "wb") as f:
        f.write(b"H\tVN:Z:1.0\tKL:Z:25\tML:Z:17\n")
        for i, sq in enumerate(segs):
            f.write(b"S\t%d\t%s\n" % (i + 1, sq))
    code = {ord("A"): 0, ord("C"): 1, ord("G"): 2, ord("T"): 3}
    km, mult = synth.canonical_counts([np.array([code[c] for c in sq], dtype=np.uint8) for sq in segs], 25)
    db = str(tmp_path / "lone_db")
    synth.write_kmc1(db, km, np.full(len(km), 20, dtype=np.uint32), 25)
    for mode in ("device", "host"):
        d = tmp_path / mode
        d.mkdir()
        r = subprocess.run([CLI, "-g", gfa, "-d", db, "-o", "g", "-l", "1", "-u", "1000", "-t", "2"], cwd=d, capture_output=True, text=True,
                           env=dict(os.environ, PF_REPLAY=mode))
        # (no site at all: the reference divides by zero at the very end, after its files are written; this CLI skips that line)
        assert r.returncode == 0, (mode, r.stdout[-300:], r.stderr[-300:])
        sb = open(os.path.join(str(d), "PloidyFrost_output", "g_super_bubble.txt")).read()
        assert sb == "BubbleId\tEntrance\tStrand\tExit\tisSimple\tisComplex\n"
        assert open(os.path.join(str(d), "PloidyFrost_output", "g_Unitig_Id.txt")).read().count("\n") == 3
