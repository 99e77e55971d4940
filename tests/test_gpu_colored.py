"""Colored (multi-sample) path, reference src/CCDBG.cpp: the HIP kernels and the CCDBG mirror against the oracle and
the committed outputs of the real reference binary on the colored fixtures."""
import os
import subprocess
import sys

import numpy as np
import pytest

from conftest import ROOT, colored_cases, compare_outputs, load_case

sys.path.insert(0, os.path.join(ROOT, "oracle"))
import pyoracle  # noqa: E402

from ploidyfrost_amd import hipapi, hostapi, synth  # noqa: E402

pytestmark = pytest.mark.gpu
CLI = os.path.join(ROOT, "ploidyfrost_amd", "csrc", "ploidyfrost")


def _device_with_colours(meta):
    o = pyoracle.ColoredOracle(meta["gfa"], meta["colors_dump"], meta["dbs"], os.environ.get("TMPDIR", "/tmp"))
    seqs = o.sequences()
    words, off, lens = hipapi.pack_unitigs(seqs)
    dev = hipapi.Device(0)
    dev.upload_graph(words, off, lens, o.k)
    dbs = []
    for p in meta["dbs"]:
        km, cnt, hdr = synth.read_kmc(p)
        dbs.append((km, cnt, hdr["both_strands"]))
    dev.upload_counts_colored(dbs)
    return o, dev, seqs, dbs


@pytest.mark.parametrize("case", colored_cases())
def test_unitig_coverage_per_colour_matches_oracle(case):
    """K-COV-C == readCovUni (src/CCDBG.cpp:123-156) for every unitig and colour, under two cutoff pairs."""
    meta = load_case(case)
    o, dev, seqs, _ = _device_with_colours(meta)
    s, lo, hi, miss = dev.unitig_cov_colored()
    assert s.shape == (o.n_colors, o.n)
    for cut in ([(5, 1000)] * o.n_colors, meta["cutoffs"], [(25, 45)] * o.n_colors):
        for c in range(o.n_colors):
            low, up = cut[c]
            for u in range(o.n):
                mean, ok = o.unitig_cov_color(c, u, low, up)
                got_ok = (not miss[c, u]) and lo[c, u] > low and hi[c, u] < up
                assert bool(ok) == bool(got_ok), (case, c, u)
                if ok:
                    assert mean == float(s[c, u]) / (len(seqs[u]) - o.k + 1)


@pytest.mark.parametrize("case", colored_cases())
def test_unitig_coverage_streamed_equals_probed(case):
    """K-COV-C streams the colour-major coverage SoA joined at load; pf_unitig_cov_colored_probe looks every k-mer up at call
    time.  Same function (src/CCDBG.cpp:123-156), two routes: whole range, ragged sub-ranges, single unitigs."""
    meta = load_case(case)
    o, dev, seqs, _ = _device_with_colours(meta)
    full = dev.unitig_cov_colored()
    probed = dev.unitig_cov_colored(probe=True)
    for a, b in zip(full, probed):
        assert np.array_equal(a, b)
    rng = np.random.default_rng(3)
    spans = [(int(a), int(b)) for a, b in ((rng.integers(0, dev.n), 0) for _ in range(5))]
    spans = [(a, int(rng.integers(a + 1, dev.n + 1))) for a, _ in spans] + [(0, 1), (dev.n - 1, dev.n)]
    for u0, u1 in spans:
        part = dev.unitig_cov_colored(u0, u1)
        for a, b in zip(part, full):
            assert np.array_equal(a, b[:, u0:u1]), (case, u0, u1)
    # a new graph under the same table is joined again on the next call
    words, off, lens = hipapi.pack_unitigs(seqs)
    dev.upload_graph(words, off, lens, o.k)
    again = dev.unitig_cov_colored()
    for a, b in zip(again, full):
        assert np.array_equal(a, b)
    dev.close()


@pytest.mark.parametrize("case", colored_cases())
def test_string_coverage_per_colour_matches_oracle(case):
    """K-STRCOV-C == readCov(string, low, up, colour) (src/CCDBG.cpp:89-122)."""
    meta = load_case(case)
    o, dev, seqs, _ = _device_with_colours(meta)
    rng = np.random.default_rng(5)
    k = o.k
    strings = []
    for _ in range(400):
        u = int(rng.integers(0, o.n))
        sq = seqs[u]
        L = int(rng.integers(k, k + 9))
        if len(sq) < L:
            L = len(sq)
        a = int(rng.integers(0, len(sq) - L + 1))
        sub = sq[a: a + L]
        if rng.random() < 0.3:   # reverse strand
            sub = sub[::-1].translate(bytes.maketrans(b"ACGT", b"TGCA"))
        if rng.random() < 0.1:   # a k-mer no database holds
            sub = sub[:-1] + (b"A" if sub[-1:] != b"A" else b"C")
        strings.append(sub)
    cut = meta["cutoffs"]
    s, ok = dev.string_cov_colored(strings, [c[0] for c in cut], [c[1] for c in cut])
    for i, sx in enumerate(strings):
        for c in range(o.n_colors):
            mean, eok = o.string_cov_color(c, sx, cut[c][0], cut[c][1])
            assert bool(eok) == bool(ok[i, c]), (i, c, sx)
            if eok:
                assert mean == float(s[i, c]) / (len(sx) - k + 1)
            else:
                assert s[i, c] == 0


def test_joined_table_keeps_colours_apart():
    """One slot per k-mer, one count per colour: a k-mer two samples share keeps both counts, and absence in a colour
    is reported for that colour only."""
    k = 25
    rng = np.random.default_rng(11)
    seq = rng.integers(0, 4, size=600, dtype=np.uint8)
    fw, rc = synth.kmers_u64(seq, k)
    can = np.unique(np.minimum(fw, rc))
    a, b = can[: len(can) * 2 // 3], can[len(can) // 3:]
    dev = hipapi.Device(0)
    words, off, lens = hipapi.pack_unitigs([synth.BASES[seq].tobytes()])
    dev.upload_graph(words, off, lens, k)
    dev.upload_counts_colored([(a, np.arange(len(a), dtype=np.uint32) + 7), (b, np.arange(len(b), dtype=np.uint32) + 1000),
                               (can[:0], np.zeros(0, dtype=np.uint32))])
    strings = [synth.BASES[seq[i: i + k]].tobytes() for i in range(len(seq) - k + 1)]
    s, ok = dev.string_cov_colored(strings, [0, 0, 0], [1 << 30] * 3)
    ia = {int(x): j for j, x in enumerate(a)}
    ib = {int(x): j for j, x in enumerate(b)}
    for i in range(len(strings)):
        key = int(min(fw[i], rc[i]))
        assert (ok[i, 0], s[i, 0]) == ((1, ia[key] + 7) if key in ia else (0, 0))
        assert (ok[i, 1], s[i, 1]) == ((1, ib[key] + 1000) if key in ib else (0, 0))
        assert (ok[i, 2], s[i, 2]) == (0, 0)


@pytest.mark.parametrize("case", colored_cases())
def test_cli_outputs_match_reference(case, tmp_path):
    meta = load_case(case)
    (tmp_path / "dbs.txt").write_text("".join(p + "\n" for p in meta["dbs"]))
    (tmp_path / "cutoffs.txt").write_text("".join("%d\t%d\n" % tuple(c) for c in meta["cutoffs"]))
    r = subprocess.run([CLI, "-g", meta["gfa"], "-f", meta["colors"], "-d", str(tmp_path / "dbs.txt"), "-C", str(tmp_path / "cutoffs.txt"),
                        "-o", "g", "-t", "1"] + meta["args"], cwd=tmp_path, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
    assert r.returncode == 0, r.stdout
    bad = compare_outputs(os.path.join(meta["dir"], "expected"), os.path.join(tmp_path, "PloidyFrost_output"))
    assert not bad, "files differ from the reference: %s\n%s" % (bad, r.stdout)
    for line in meta["reference_log"]:
        assert line.strip() in r.stdout.replace("\r", ""), line


@pytest.mark.parametrize("case", colored_cases())
def test_state_after_find_superbubbles_matches_oracle(case, tmp_path):
    meta = load_case(case)
    op = meta["opts"]
    run = hostapi.ColoredRun(meta["gfa"], meta["colors"], meta["dbs"], str(tmp_path), z=int(op["-z"]))
    run.set_output_dir(str(tmp_path / "out"))
    run.set_unitig_id("g")
    run.find_superbubbles("g")
    f, p, m = run.state()
    o = pyoracle.ColoredOracle(meta["gfa"], meta["colors_dump"], meta["dbs"], str(tmp_path))
    nb = o.find_superbubbles(z=int(op["-z"]))
    ef, ep, em = o.state()
    assert np.array_equal(f, ef) and np.array_equal(p, ep) and np.array_equal(m, em)
    assert run.times()["superbubbles"] == nb


@pytest.mark.parametrize("threads", [1, 6])
def test_facade_run_and_host_threads(threads, tmp_path):
    meta = load_case("col4_mix")
    op = meta["opts"]
    run = hostapi.ColoredRun(meta["gfa"], meta["colors"], meta["dbs"], str(tmp_path), z=int(op["-z"]), threads=threads)
    run.set_threads(threads)
    run.set_output_dir(str(tmp_path / "out"))
    run.set_unitig_id("g")
    run.find_superbubbles("g")
    run.ploidy_estimation("g", meta["cutoffs"])
    assert not compare_outputs(os.path.join(meta["dir"], "expected"), str(tmp_path / "out"))
    # a second pass over the resident graph gives the same bytes (bench.py repeats passes)
    run.find_superbubbles("g")
    run.ploidy_estimation("g", meta["cutoffs"])
    assert not compare_outputs(os.path.join(meta["dir"], "expected"), str(tmp_path / "out"))


def test_wrong_number_of_cutoffs_is_an_error(tmp_path):
    meta = load_case("col3_dip")
    run = hostapi.ColoredRun(meta["gfa"], meta["colors"], meta["dbs"], str(tmp_path))
    run.set_output_dir(str(tmp_path / "out"))
    run.set_unitig_id("g")
    run.find_superbubbles("g")
    with pytest.raises(hipapi.DeviceError, match="cutoff"):
        run.ploidy_estimation("g", meta["cutoffs"][:-1])


@pytest.mark.parametrize("aligners", ["1", "3"])
def test_pipeline_batches_colored(aligners, tmp_path, monkeypatch):
    monkeypatch.setenv("PF_ALIGN_RANGES", "7")
    monkeypatch.setenv("PF_ALIGN_THREADS", aligners)   # (aligners side by side, a lane each: walk pools and colour tables per lane)
    meta = load_case("col2_weird")
    op = meta["opts"]
    run = hostapi.ColoredRun(meta["gfa"], meta["colors"], meta["dbs"], str(tmp_path), z=int(op["-z"]), M=float(op["-M"]),
                             D=float(op["-D"]), G=float(op["-G"]))
    run.set_batch_bubbles(16)
    run.set_threads(4)
    run.set_output_dir(str(tmp_path / "out"))
    run.set_unitig_id("g")
    for _ in range(3):
        run.find_superbubbles("g")
        run.ploidy_estimation("g", meta["cutoffs"])
        assert not compare_outputs(os.path.join(meta["dir"], "expected"), str(tmp_path / "out"))


@pytest.mark.parametrize("case", colored_cases())
def test_colored_calling_on_the_resident_pipeline_and_on_the_host(case, tmp_path, monkeypatch):
    """CCDBG::ploidyEstimation_ptr runs on the resident pipeline (pf_call_set_colours: colored K-SCAN, K-SITES, K-TEXT around the
    shared alignment tiers); PF_CALL=host keeps round 1's host-threaded pipeline.  Both give the reference's files."""
    meta = load_case(case)
    op = meta["opts"]
    kw = dict(z=int(op["-z"]), M=float(op["-M"]), D=float(op["-D"]), G=float(op["-G"]))
    jobs = {}
    for mode in ("resident", "host"):
        if mode == "host":
            monkeypatch.setenv("PF_CALL", "host")
        else:
            monkeypatch.delenv("PF_CALL", raising=False)
        out = tmp_path / mode
        run = hostapi.ColoredRun(meta["gfa"], meta["colors"], meta["dbs"], str(tmp_path), **kw)
        run.set_output_dir(str(out))
        run.set_unitig_id("g")
        run.find_superbubbles("g")
        run.ploidy_estimation("g", meta["cutoffs"])
        assert not compare_outputs(os.path.join(meta["dir"], "expected"), str(out)), mode
        t = run.times()
        jobs[mode] = (t["align_jobs"], t["snp_jobs"] + t["pair_jobs"] + t["stack_jobs"] + t["wave_jobs"])
    # (col100: more colours than one 64-bit word holds -- the reference has no limit, src/CCDBG.cpp:2759-2853; colour sets are
    # (C + 63) / 64 words on the device, and the graph runs on the resident pipeline like any other)
    assert jobs["resident"][0] > 0 and jobs["resident"][1] == jobs["resident"][0], jobs   # every job accounted to a device tier
    assert jobs["host"][1] == 0, jobs


@pytest.mark.parametrize("case,world", [("col4_mix", 3), ("col2_weird", 2)])
def test_colored_graph_cut_over_ranks_in_process(case, world, tmp_path):
    """ploidyfrost_amd/dist.py with a CCDBG: `world` ranks (here: runs in one process) hold the same colored graph, find their shards
    of findSuperBubble, replay all of them, select the same bubble list (one cutoff pair per colour), align and format their slices
    and write their slabs into shared files -- which must be the reference's."""
    from ploidyfrost_amd import dist as pfdist
    meta = load_case(case)
    op = meta["opts"]
    kw = dict(z=int(op["-z"]), M=float(op["-M"]), D=float(op["-D"]), G=float(op["-G"]))
    out = str(tmp_path / "shared")
    runs = []
    for rank in range(world):
        run = hostapi.ColoredRun(meta["gfa"], meta["colors"], meta["dbs"], str(tmp_path), **kw)
        run.set_threads(3)
        run.set_output_dir(out)
        runs.append(run)
    runs[0].set_unitig_id("g")
    n = runs[0].times()["unitigs"]
    shards = [runs[r].find_shard(*pfdist.shard_range(n, r, world)) for r in range(world)]
    for r in range(world):
        runs[r].find_replay("g", [s[0] for s in shards], [s[1] for s in shards], write_file=r == 0)
    nb = [run.ploidy_select(meta["cutoffs"]) for run in runs]
    assert len(set(nb)) == 1 and nb[0] > 0
    slices = [pfdist.shard_range(nb[0], r, world) for r in range(world)]
    called = [runs[r].ploidy_align(*slices[r]) for r in range(world)]
    texts = [runs[r].ploidy_text(int(sum(called[:r]))) for r in range(world)]
    sizes = np.array([t[0] for t in texts], dtype=np.uint64)
    totals = sizes.sum(axis=0)
    for r in reversed(range(world)):
        runs[r].ploidy_write("g", sizes[:r].sum(axis=0), totals, truncate=True)
    assert not compare_outputs(os.path.join(meta["dir"], "expected"), out)
    for run in runs:
        run.close()


def test_colored_calling_refusals_are_by_name(tmp_path):
    """what the colored resident pipeline does not do says so: the -t > 1 text format and the kernel-level view are the single-sample
    path's, the colour sets must match the uploaded databases, cutoffs come after the colour sets -- an error code and a message,
    never a crash or a silent fallback"""
    import ctypes as C
    meta = load_case("col3_dip")
    dead = hostapi.ColoredRun(meta["gfa"], meta["colors"], meta["dbs"], str(tmp_path))
    with pytest.raises(hipapi.DeviceError, match="single-sample"):
        dead.set_reference_threads(4)   # (an error is sticky, as the reference's exit() is final: this run is over)
    dead.close()
    run = hostapi.ColoredRun(meta["gfa"], meta["colors"], meta["dbs"], str(tmp_path))
    run.set_output_dir(str(tmp_path / "out"))
    run.set_unitig_id("g")
    run.find_superbubbles("g")
    nb = run.ploidy_select(meta["cutoffs"])
    assert nb > 0 and run.ploidy_align(0, nb) > 0
    L = hipapi.load_library()
    ctx = C.c_void_p(run.device_ctx())
    used = (C.c_uint64 * 6)()
    dummy = (C.c_uint64 * 5)(1 << 40, 1 << 40, 1 << 40, 1 << 40, 1 << 40)
    buf = (C.c_char * 64)()
    rc = L.pf_call_peek(ctx, 0, C.cast(buf, C.c_void_p), None, None, 1 << 30, None, None, None, None, None, C.cast(dummy, C.c_void_p), C.cast(used, C.c_void_p))
    assert rc != 0 and b"single-sample" in L.pf_last_error(ctx)
    lo = (C.c_uint32 * 8)(*([5] * 8))
    hi = (C.c_uint32 * 8)(*([1000] * 8))
    assert L.pf_call_set_cutoffs(ctx, 2, C.cast(lo, C.c_void_p), C.cast(hi, C.c_void_p)) != 0   # (three colours were uploaded)
    assert b"per colour" in L.pf_last_error(ctx)
    n = run.times()["unitigs"]
    full = (C.c_uint64 * n)()
    first = (C.c_uint32 * (n + 1))()
    assert L.pf_call_set_colours(ctx, 5, C.cast(full, C.c_void_p), C.cast(full, C.c_void_p), C.cast(first, C.c_void_p), None, None, None, 0, 0) != 0
    assert b"colour sets" in L.pf_last_error(ctx)
    run.close()
