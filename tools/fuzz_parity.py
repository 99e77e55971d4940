#!/usr/bin/env python3
"""Differential fuzzing on a GPU box: random small inputs (ploidy, k, -z, variant density, insertion lengths, repeat-rich
genomes, integral and fractional scores, tight and loose cutoffs; single-sample and colored) through this repository's CLI
and through the CPU oracle (oracle/_build/pf_oracle_cli, itself pinned to the reference on the golden fixtures); all twelve
output files must be byte-identical.   usage: tools/fuzz_parity.py [n_cases] [first_seed]
Graphs come from ploidyfrost_amd.cdbg_build (seeds whose repeat structure it cannot compact are skipped); colour sets are
written by ploidyfrost_amd.bfg_colors and read back for the oracle by oracle/_ref/colors_dump (the real Bifrost).
With PF_FUZZ_BIFROST=1 the graphs (and colour files) are built by the reference's own `Bifrost build` (oracle/_ref/Bifrost)
instead: any repeat structure, samples that start and stop at different places (colours on part of a unitig).
With PF_FUZZ_REFERENCE=1 the comparator is the reference binary itself (oracle/_ref/PloidyFrost -t 1) instead of the oracle.
With PF_FUZZ_CROWD=1 (implies both of the above) a g-mer with a very low minimizer hash is planted at 20-120 places of the
genome, so that Bifrost files some k-length unitigs as "abundant" k-mers and numbers them last, in hash-table order -- the
part of the unitig numbering the oracle's loader does not restate (pf_host_minz.hpp).
With PF_FUZZ_GIANT=1 the genomes are 250-400 kb long and carry one to three copied segments (k .. 2k bp) near their start: the
shared unitig's traversal cannot close before it has walked a whole locus to the end of the chromosome -- traversals of more
than 4096 vertices, the third K-BFS tier (host walkers by default; PF_FUZZ_GIANT=device switches every other case to k_bfs_huge).
With PF_FUZZ_COLORED=1 every case is a colored one (two to three samples; CCDBG's calling phase on the resident pipeline);
PF_FUZZ_SAMPLES=lo,hi with it: lo .. hi diploid samples over the case's four or six haplotypes, graph and colour file by the
reference's Bifrost (colour sets of more than one 64-bit word).
With PF_FUZZ_SCORES=wide the scores are drawn from the whole region the reference accepts (draw_scores: D <= M, G <= M -- zero and
positive gap scores, D = M, G = M, fractions, a negative match, magnitudes to 1e5); both sides then often end the run themselves
(a site string's k-mer that is in no database): "oracle rc 1, product rc 1" is agreement."""
import os
import subprocess
import sys
import tempfile

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "oracle"))
import bench  # noqa: E402
import pyoracle  # noqa: E402
from ploidyfrost_amd import cdbg_build, synth  # noqa: E402

CLI = os.path.join(ROOT, "ploidyfrost_amd", "csrc", "ploidyfrost")
FILES = ["Unitig_Id", "super_bubble", "alignseq", "allele_frequency", "bicov", "bifre", "tricov", "trifre", "tetracov", "tetrafre",
         "pentacov", "pentafre"]


def repeat_rich(rng, n):
    base = rng.integers(0, 4, size=n, dtype=np.uint8)
    parts, pos = [], 0
    while pos < n:
        step = int(rng.integers(150, 500))
        seg = base[pos: pos + step]
        parts.append(seg)
        if len(seg) > 120 and rng.random() < 0.6:
            a = int(rng.integers(0, len(seg) - 100))
            ln = int(rng.integers(30, 100))
            piece = seg[a: a + ln]
            kind = int(rng.integers(0, 3))
            parts += [piece, piece] if kind == 0 else [(3 - piece)[::-1]] if kind == 1 else [base[a: a + ln]]
        pos += step
    return np.concatenate(parts)


def draw_scores(rng):
    """(M, D, G) from the whole region the reference accepts (src/Main.cpp:470-479: D <= M and G <= M, nothing else): zero and
    positive gap scores, D = M, G = M, fractions, negative match scores, magnitudes to 1e5."""
    M = float(rng.choice([0, 0.5, 1, 1, 1.5, 2, 2, 2, 3, 5, 10, 100, 1e5, -1, -0.5]))
    d_pool = [M, M - 0.5, M - 1, M - 3, 0.0, -1.0, -M, -2.5, -7.0, -1e5, M - 0.25]
    g_pool = [M, M - 1, M - 5, 1.0, 0.5, 0.0, -0.5, -1.0, -2.0, -3.0, -2.25, -10.0, -1e5, M - 0.75]
    D = float(rng.choice([x for x in d_pool if x <= M]))
    G = float(rng.choice([x for x in g_pool if x <= M]))
    f = lambda x: int(x) if x == int(x) else x  # noqa: E731
    return (f(M), f(D), f(G))


def build_case(seed, tmp, dev, force_colored=None, force_giant=None, wide_scores=None):
    """inputs of one random case: a dict, or a string when the seed is skipped.  force_colored / force_giant (tests/test_gpu_fuzz.py):
    the kind of case is chosen by the caller instead of by the seed / the environment."""
    rng = np.random.default_rng(seed)
    k = int(rng.choice([21, 25, 25, 31]))
    ploidy = int(rng.integers(2, 7))
    if force_colored:
        ploidy = 4 if ploidy < 5 else 6
    z = int(rng.integers(4, 13))
    L = int(rng.integers(8000, 40000))
    if os.environ.get("PF_FUZZ_SAMPLES"):
        L = 4000 + L % 8000   # (a hundred databases: a short genome)
    spec = synth.HapSpec(L, ploidy, seed=seed, gap_lo=int(rng.integers(5, 30)), gap_hi=int(rng.integers(40, 500)),
                         p_multi=float(rng.choice([0.0, 0.05, 0.15])), max_ins=int(rng.choice([3, 6, 12, 30])),
                         p_snp=float(rng.choice([0.5, 0.75, 0.9])), p_del=0.1)
    crowd = os.environ.get("PF_FUZZ_CROWD") == "1"
    giant = os.environ.get("PF_FUZZ_GIANT") if force_giant is None else ("1" if force_giant else None)
    edit = None
    if giant:
        L = int(rng.integers(250000, 400000))
        spec.genome_len = L

        def edit(base, rng2=np.random.default_rng(seed + 11), k=k):
            out = base.copy()
            for _ in range(int(rng2.integers(1, 4))):
                ln = int(rng2.integers(k, 2 * k))
                a = int(rng2.integers(2000, 12000))
                b = int(rng2.integers(14000, 30000))
                seg = out[a: a + ln] if rng2.random() < 0.8 else (3 - out[a: a + ln])[::-1]   # sometimes inverted
                out[b: b + ln] = seg
            return out
    if crowd:
        g_len = k - 8 if k >= 15 else k - 4   # Bifrost's default minimizer length (CompactedDBG.hpp DEFAULT_G_DEC1/2)
        copies = int(rng.integers(20, 120))
        edit = lambda base: synth.plant_crowded_minimizer(np.random.default_rng(seed + 7), base, g_len, copies, tries=1500)  # noqa: E731
    haps = synth.make_haplotypes(spec, edit)
    if rng.random() < 0.3 and not giant:  # splice the variants onto a repeat-rich genome: cycles, hairpins, tips
        rep = repeat_rich(rng, L)
        haps = [np.concatenate([rep[: len(rep) // 2], h[200:-200], rep[len(rep) // 2:]]) for h in haps]
    scores = [(2, -1, -3), (2, -1, -3), (1, -1, -1), (3, -2, -4), (1.5, -0.5, -2.25), (2, -1, -2)][int(rng.integers(0, 6))]
    lower, upper = [(5, 1000), (5, 1000), (15, 70), (1, 100000)][int(rng.integers(0, 4))]
    if wide_scores if wide_scores is not None else os.environ.get("PF_FUZZ_SCORES") == "wide":
        scores = draw_scores(np.random.default_rng(seed + 77))   # its own stream: the rest of the case stays what the seed made it
    colored = rng.random() < 0.35 and ploidy % 2 == 0 and ploidy >= 4 and not giant
    if force_colored is not None:
        colored = bool(force_colored) and not giant
    many = tuple(int(x) for x in os.environ["PF_FUZZ_SAMPLES"].split(",")) if os.environ.get("PF_FUZZ_SAMPLES") and colored else None
    use_bifrost = os.environ.get("PF_FUZZ_BIFROST") == "1" or crowd or bool(giant) or bool(many)
    use_reference = os.environ.get("PF_FUZZ_REFERENCE") == "1" or crowd
    try:
        if use_bifrost:
            groups = [haps[2 * i: 2 * i + 2] for i in range(ploidy // 2)] if colored else [haps]
            if colored and many:
                # lo .. hi diploid samples over the haplotypes drawn above, each short of a few bases at its ends; the last ones with a
                # private tail (unitigs that lack most colours, their own colours in the last word of a colour set)
                n_samples = int(np.random.default_rng(seed + 5).integers(many[0], many[1] + 1))
                tail_rng = np.random.default_rng(seed + 6)
                groups = []
                for i in range(n_samples):
                    hs = [haps[i % ploidy], haps[(i + 1 + i // ploidy) % ploidy]]
                    hs = [h[(5 * i + 3 * j) % 31: len(h) - ((3 * i + 7 * j) % 43)] for j, h in enumerate(hs)]
                    if i >= n_samples - 5:
                        hs = [np.concatenate([h, tail_rng.integers(0, 4, size=80, dtype=np.uint8)]) for h in hs]
                    groups.append(hs)
            elif colored:  # samples that do not span the whole genome
                groups = [[h[int(rng.integers(0, 1500)): len(h) - int(rng.integers(0, 1500))] for h in hs] if i else hs for i, hs in enumerate(groups)]
            fas = []
            for i, hs in enumerate(groups):
                fa = os.path.join(tmp, "s%d.fa" % i)
                synth.write_fasta(fa, hs)
                fas.append(fa)
            refs = os.path.join(tmp, "refs.txt")
            open(refs, "w").write("".join(f + "\n" for f in fas))
            r = subprocess.run([pyoracle.REF_BIFROST, "build", "-r", refs, "-k", str(k), "-o", os.path.join(tmp, "g"), "-t", "1"] + (["-c"] if colored else []),
                               capture_output=True, text=True)
            if r.returncode != 0:
                return "skipped (Bifrost build failed)"
            gfa = os.path.join(tmp, "g.gfa")
            n_unitigs = sum(1 for line in open(gfa) if line.startswith("S\t"))
            if colored:
                colors = os.path.join(tmp, "g.bfg_colors")
                dbs = []
                for i, hs in enumerate(groups):
                    km, mult = synth.canonical_counts(hs, k)
                    dbs.append(os.path.join(tmp, "db%d" % i))
                    synth.write_kmc1(dbs[-1], km, synth.synth_counts(km, mult), k)
            else:
                km, mult = synth.canonical_counts(haps, k)
                db = os.path.join(tmp, "g_kmc")
                synth.write_kmc1(db, km, synth.synth_counts(km, mult), k)
        elif colored:
            # make_colored_inputs draws its own haplotypes: samples x 2 on one genome
            gfa, colors, dbs, n_unitigs, _ = bench.make_colored_inputs(tmp, "g", L, seed, dev, k=k, samples=ploidy // 2, ploidy=2,
                                                                      max_ins=spec.max_ins)
        else:
            g = cdbg_build.build_cdbg(haps, k, dev)
            gfa = os.path.join(tmp, "g.gfa")
            n_unitigs = cdbg_build.write_gfa(gfa, g)
            db = os.path.join(tmp, "g_kmc")
            synth.write_kmc1(db, g["kmers"], synth.synth_counts(g["kmers"], g["mult"]), k)
    except RuntimeError as e:
        return "skipped (%s)" % str(e)[:60]
    return dict(k=k, ploidy=ploidy, z=z, L=L, scores=scores, lower=lower, upper=upper, colored=colored, gfa=gfa, n_unitigs=n_unitigs,
                db=None if colored else db, dbs=dbs if colored else None, colors=colors if colored else None, use_reference=use_reference,
                giant=giant, crowd=crowd)


def one_case(seed, tmp, dev, force_colored=None, force_giant=None, wide_scores=None):
    c = build_case(seed, tmp, dev, force_colored, force_giant, wide_scores)
    if isinstance(c, str):
        return c
    k, ploidy, z, L, scores, lower, upper, colored, gfa, n_unitigs = (c[x] for x in ("k", "ploidy", "z", "L", "scores", "lower", "upper", "colored",
                                                                                      "gfa", "n_unitigs"))
    db, dbs, colors, use_reference, giant, crowd = c["db"], c["dbs"], c["colors"], c["use_reference"], c["giant"], c["crowd"]
    common = ["-o", "x", "-z", str(z), "-M", repr(scores[0]), "-D", repr(scores[1]), "-G", repr(scores[2])]
    ub_log = os.path.join(tmp, "ub_cells.txt")   # the cells the reference leaves undefined (pfo::indel_len_at), listed by the oracle
    oenv = dict(os.environ, PFO_UB_LOG=ub_log)
    og, gg = os.path.join(tmp, "oracle"), os.path.join(tmp, "gpu")
    os.makedirs(og), os.makedirs(gg)
    if colored:
        dump = os.path.join(tmp, "colors.txt")
        with open(dump, "w") as f:
            subprocess.run([pyoracle.REF_COLORS_DUMP, gfa, colors], check=True, stdout=f)
        lst, cut = os.path.join(tmp, "dbs.txt"), os.path.join(tmp, "cut.txt")
        open(lst, "w").write("".join(d + "\n" for d in dbs))
        open(cut, "w").write(("%d\t%d\n" % (lower, upper)) * len(dbs))
        if use_reference:
            ro = subprocess.run([pyoracle.REF_BIN, "-g", gfa, "-f", colors, "-d", lst, "-C", cut, "-t", "1"] + common, cwd=og, capture_output=True, text=True)
        else:
            ro = subprocess.run([pyoracle.CLI, "-g", gfa, "-f", dump, "-d", lst, "-C", cut, "-O", os.path.join(og, "PloidyFrost_output")] + common,
                                cwd=og, capture_output=True, text=True, env=oenv)
        rg = subprocess.run([CLI, "-g", gfa, "-f", colors, "-d", lst, "-C", cut, "-t", "8"] + common, cwd=gg, capture_output=True, text=True)
    else:
        if use_reference:
            ro = subprocess.run([pyoracle.REF_BIN, "-g", gfa, "-d", db, "-l", str(lower), "-u", str(upper), "-t", "1"] + common, cwd=og,
                                capture_output=True, text=True)
        else:
            ro = subprocess.run([pyoracle.CLI, "-g", gfa, "-d", db, "-l", str(lower), "-u", str(upper), "-O", os.path.join(og, "PloidyFrost_output")]
                                + common, cwd=og, capture_output=True, text=True, env=oenv)
        env = dict(os.environ)
        if giant == "device" and seed % 2:
            env["PF_BFS_HUGE_ON_DEVICE"] = "1"
        rg = subprocess.run([CLI, "-g", gfa, "-d", db, "-l", str(lower), "-u", str(upper), "-t", "8", "-v"] + common, cwd=gg, capture_output=True,
                            text=True, env=env)
    cells = {}
    if use_reference:
        # the reference's files hold heap garbage in the cells it leaves undefined: the oracle lists them (and is itself held to the
        # reference on every other byte by tools/fuzz_scores_cpu.py)
        side = os.path.join(tmp, "oracle_side")
        os.makedirs(side)
        if colored:
            subprocess.run([pyoracle.CLI, "-g", gfa, "-f", dump, "-d", lst, "-C", cut, "-O", os.path.join(side, "out")] + common, cwd=side,
                           capture_output=True, text=True, env=oenv)
        else:
            subprocess.run([pyoracle.CLI, "-g", gfa, "-d", db, "-l", str(lower), "-u", str(upper), "-O", os.path.join(side, "out")] + common,
                           cwd=side, capture_output=True, text=True, env=oenv)
        cells = pyoracle.read_ub_log(ub_log)
    desc = "k=%d ploidy=%d z=%d L=%d scores=%s cut=%d/%d %s unitigs=%d" % (k, ploidy, z, L, scores, lower, upper,
                                                                          "colored" if colored else "single", n_unitigs)
    if giant and not colored:
        import re
        mm = re.search(r"traversals > 4096 unitigs: (\d+) \(sum of seen \d+, largest (\d+)\)", rg.stdout)
        desc += " giants=%s largest=%s" % (mm.group(1), mm.group(2)) if mm else " giants=?"
    if crowd:
        from ploidyfrost_amd import hostapi
        desc += " abundant=%d" % hostapi.load_library().pfh_gfa_abundant_kmers(gfa.encode())
    if use_reference and ro.returncode == -8 and rg.returncode == 0:
        # the reference divides by the number of sites it found (src/CDBG.cpp:1703): no site -> SIGFPE before its files are flushed
        return "%s: skipped (the reference died of its own division by zero: no site passed the cutoffs)" % desc
    if use_reference and ro.returncode in (-11, -6) and rg.returncode == 0 and cells:
        return "%s: skipped (the reference died of its own out-of-bounds read: %d undefined cells)" % (desc, sum(len(v) for v in cells.values()))
    if ro.returncode != 0 or rg.returncode != 0:
        # both must fail alike (e.g. a k-mer missing from the database)
        return "%s: oracle rc %d, product rc %d%s" % (desc, ro.returncode, rg.returncode, "" if (ro.returncode != 0) == (rg.returncode != 0)
                                                      else "  MISMATCH\n" + ro.stderr[-300:] + rg.stderr[-300:] + rg.stdout[-300:])
    bad = pyoracle.compare_outputs(os.path.join(og, "PloidyFrost_output"), os.path.join(gg, "PloidyFrost_output"), "x", cells, colored)
    return "%s: %s" % (desc, "identical" if not bad else "DIFFERENT " + ",".join(bad))


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 30
    first = int(sys.argv[2]) if len(sys.argv) > 2 else 1
    import torch
    dev = "cuda" if torch.cuda.is_available() else "cpu"
    pyoracle.build()
    failures = 0
    for seed in range(first, first + n):
        with tempfile.TemporaryDirectory() as tmp:
            msg = one_case(seed, tmp, dev, force_colored=True if os.environ.get("PF_FUZZ_COLORED") == "1" else None)   # (PF_FUZZ_SCORES=wide: draw_scores)
            if ("DIFFERENT" in msg or "MISMATCH" in msg) and os.environ.get("PF_FUZZ_KEEP"):   # the inputs of a failing case, for a closer look
                import shutil
                shutil.copytree(tmp, os.path.join(os.environ["PF_FUZZ_KEEP"], "seed%d" % seed), dirs_exist_ok=True)
        print("seed %d: %s" % (seed, msg), flush=True)
        failures += "DIFFERENT" in msg or "MISMATCH" in msg
    print("%d cases, %d failures" % (n, failures))
    return 1 if failures else 0


if __name__ == "__main__":
    sys.exit(main())
