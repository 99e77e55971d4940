"""`ploidyfrost filter` / `filter-multi`: the row predicates of the reference's R scripts (script/Filter.R:1-159,
script/Filter-multi.R:1-186) over the *cov.txt files of the path.

PARITY UNPINNED: R is not installed in the build image, so nothing the reference's scripts wrote exists to hold this to.  Every
expected row below is worked out BY HAND from the scripts' text and from R's documented rules for read.table / write.table
(column types by type.convert, every number on its own with 15 significant digits, fixed notation unless scientific is narrower)."""
import ctypes as C
import os
import subprocess

import pytest

from conftest import ROOT, load_case

CLI = os.path.join(ROOT, "ploidyfrost_amd", "csrc", "ploidyfrost")

BICOV = (  # CovA CovB isStrict VarType VarId VarNum VarDis   (rows end in a tab, as the path writes them)
    "60.2174\t59.6429\t1\t0\t1\t2\t40\t\n"      # r1 SNP, strict
    "100.36\t19.64\t1\t3\t2\t1\t25\t\n"         # r2 indel of length 3, strict
    "30\t30\t0\t0\t3\t4\t7\t\n"                  # r3 SNP of a branching bubble, 4 sites, distance 7
    "4.5\t70\t1\t0\t4\t1\t25\t\n"               # r4 CovA below the default -l of the tests (5)
    "61\t1200.5\t0\t12\t5\t1\t3\t\n"            # r5 CovB above -u 1000; indel of 12
    "100000\t250000\t1\t0\t6\t1\t9\t\n"         # r6 large coverages: R writes 1e+05 and 250000 (doubles: narrower notation wins)
)
TRICOV = "20\t20\t20\t1\t0\t7\t1\t30\t\n40.5\t20.25\t20.25\t0\t2\t8\t3\t11\t\n"
TETRACOV = "20\t20\t20\t20\t1\t0\t9\t1\t30\t\n300\t300\t300\t300\t1\t0\t10\t1\t30\t\n"   # second row: A+B+C+D = 1200 is not < -u 1000
PENTACOV = ""   # (empty, as for most inputs: read.table with col.names gives a table of no rows)


def run(args, cwd):
    return subprocess.run([CLI] + args, cwd=cwd, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True)


@pytest.fixture
def tables(tmp_path):
    for name, text in (("bicov", BICOV), ("tricov", TRICOV), ("tetracov", TETRACOV), ("pentacov", PENTACOV)):
        (tmp_path / ("in_%s.txt" % name)).write_text(text)
    return tmp_path


def rows(p):
    return open(p).read().splitlines()


def test_default_predicates_and_the_numbers_as_r_writes_them(tables):
    r = run(["filter", "-i", "in", "-o", "out", "-l", "5", "-u", "1000"], tables)
    assert r.returncode == 0, r.stderr
    # r4 (CovA 4.5 <= 5), r5 (CovB >= 1000) and r6 (both >= 1000) go; write.table leaves no trailing tab; CovA/CovB are doubles
    # (colClasses), the other columns integers
    assert rows(tables / "out_bicov.txt") == ["60.2174\t59.6429\t1\t0\t1\t2\t40", "100.36\t19.64\t1\t3\t2\t1\t25", "30\t30\t0\t0\t3\t4\t7"]
    assert rows(tables / "out_tricov.txt") == ["20\t20\t20\t1\t0\t7\t1\t30", "40.5\t20.25\t20.25\t0\t2\t8\t3\t11"]
    assert rows(tables / "out_tetracov.txt") == ["20\t20\t20\t20\t1\t0\t9\t1\t30"]     # Filter.R:106-108: the sum of the four < up
    assert rows(tables / "out_pentacov.txt") == []
    # frequencies: c(bifre[1,], bifre[2,]) -- first alleles of all bi rows, then second alleles -- then the tri table allele by
    # allele, then tetra; kept in (0.05, 0.95), rounded to 7 places
    a = [60.2174 / (60.2174 + 59.6429), 100.36 / (100.36 + 19.64), 0.5]
    b = [59.6429 / (60.2174 + 59.6429), 19.64 / (100.36 + 19.64), 0.5]
    tri = [20 / 60, 40.5 / 81, 20 / 60, 20.25 / 81, 20 / 60, 20.25 / 81]
    want = ["%.7f" % x for x in a + b + tri + [0.25] * 4]
    want = [w.rstrip("0").rstrip(".") for w in want]
    assert rows(tables / "out_allele_frequency.txt") == want
    assert want[0] == "0.5023965" and want[1] == "0.8363333" and want[2] == "0.5" and want[6] == "0.3333333"


def test_flags(tables):
    assert run(["filter", "-i", "in", "-o", "s", "-l", "5", "-u", "1000", "-S"], tables).returncode == 0
    assert [x.split("\t")[4] for x in rows(tables / "s_bicov.txt")] == ["1", "2"]            # isStrict == 1
    assert run(["filter", "-i", "in", "-o", "i", "-l", "5", "-u", "1000", "-I"], tables).returncode == 0
    assert [x.split("\t")[4] for x in rows(tables / "i_bicov.txt")] == ["1", "3"]            # "filter indel": VarType == 0 stays
    assert run(["filter", "-i", "in", "-o", "p", "-l", "5", "-u", "1000", "--snp"], tables).returncode == 0
    assert [x.split("\t")[4] for x in rows(tables / "p_bicov.txt")] == ["2"]                 # "filter snp": VarType > 0 stays
    assert rows(tables / "p_tricov.txt") == ["40.5\t20.25\t20.25\t0\t2\t8\t3\t11"]
    assert run(["filter", "-i", "in", "-o", "n", "-l", "5", "-u", "1000", "-n", "3", "-d", "10", "--size=3"], tables).returncode == 0
    assert [x.split("\t")[4] for x in rows(tables / "n_bicov.txt")] == ["1"]   # VarNum < 3, VarDis > 10, VarType < 3: r2 (type 3), r3 (4 sites) go
    # defaults: -l 0 -u 10000: r4 and r5 stay, r6 still goes
    assert run(["filter", "-i", "in", "-o", "d"], tables).returncode == 0
    assert [x.split("\t")[4] for x in rows(tables / "d_bicov.txt")] == ["1", "2", "3", "4", "5"]
    assert rows(tables / "d_bicov.txt")[4] == "61\t1200.5\t0\t12\t5\t1\t3"
    # everything: R's doubles in the narrower notation
    assert run(["filter", "-i", "in", "-o", "e", "-u", "1000000"], tables).returncode == 0
    assert rows(tables / "e_bicov.txt")[5] == "1e+05\t250000\t1\t0\t6\t1\t9"
    # -q: a wider band drops more
    assert run(["filter", "-i", "in", "-o", "q", "-l", "5", "-u", "1000", "-q", "0.3"], tables).returncode == 0
    got = [float(x) for x in rows(tables / "q_allele_frequency.txt")]
    assert got and all(0.3 < x < 0.7 for x in got) and 0.8363333 not in got


def test_the_scripts_own_exits(tables, tmp_path):
    r = run(["filter", "-i", "in", "-o", "x", "-q", "0.6"], tables)
    assert r.returncode == 0 and "frequency should < 0.5" in r.stderr and not (tables / "x_bicov.txt").exists()
    r = run(["filter", "-i", "nowhere", "-o", "x"], tables)
    assert r.returncode == 0 and "does not exists" in r.stderr
    # nothing kept anywhere: the four (empty) tables are written, then round(NULL, 7) is an R error and no frequency file is made
    r = run(["filter", "-i", "in", "-o", "z", "-l", "5000"], tables)
    assert r.returncode != 0 and "non-numeric argument" in r.stderr
    assert (tables / "z_bicov.txt").exists() and not (tables / "z_allele_frequency.txt").exists()


def test_filter_multi(tmp_path):
    # CovA CovB color isStrict VarType VarId VarNum Cramer VarDis
    (tmp_path / "m_bicov.txt").write_text("60\t20\t0\t1\t0\t1\t1\t0.5\t25\t\n30\t30\t1\t1\t0\t1\t1\t0.5\t25\t\n50\t25\t1\t0\t4\t2\t2\t0.0816497\t9\t\n")
    (tmp_path / "m_tricov.txt").write_text("")
    (tmp_path / "m_tetracov.txt").write_text("300\t300\t300\t300\t2\t1\t0\t3\t1\t0.9\t30\t\n")   # no sum clause in Filter-multi.R
    (tmp_path / "m_pentacov.txt").write_text("")
    r = run(["filter-multi", "-i", "m", "-o", "o", "-l", "5", "-u", "1000"], tmp_path)
    assert r.returncode == 0, r.stderr
    assert rows(tmp_path / "o_bicov.txt") == ["60\t20\t0\t1\t0\t1\t1\t0.5\t25", "30\t30\t1\t1\t0\t1\t1\t0.5\t25", "50\t25\t1\t0\t4\t2\t2\t0.0816497\t9"]
    assert rows(tmp_path / "o_tetracov.txt") == ["300\t300\t300\t300\t2\t1\t0\t3\t1\t0.9\t30"]
    assert run(["filter-multi", "-i", "m", "-o", "c", "-l", "5", "-u", "1000", "-c", "1", "-v", "0.1"], tmp_path).returncode == 0
    assert rows(tmp_path / "c_bicov.txt") == ["30\t30\t1\t1\t0\t1\t1\t0.5\t25"]      # colour 1 and Cramer's V > 0.1
    assert rows(tmp_path / "c_tetracov.txt") == []
    assert rows(tmp_path / "c_allele_frequency.txt") == ["0.5", "0.5"]
    assert run(["filter", "-i", "m", "-o", "w", "-c", "1"], tmp_path).returncode != 0   # -c belongs to filter-multi


def test_on_the_paths_own_files(tmp_path):
    """a fixture's files as the reference wrote them: with the defaults every row whose coverages lie below 10000 comes back, number
    for number (six significant digits in, the same text out)"""
    meta = load_case("hex30k")
    exp = os.path.join(meta["dir"], "expected", "g")
    r = run(["filter", "-i", exp, "-o", "f"], tmp_path)
    assert r.returncode == 0, r.stderr
    for name in ("bicov", "tricov", "tetracov", "pentacov"):
        want = [ln.rstrip("\t") for ln in rows(exp + "_%s.txt" % name)]
        assert rows(tmp_path / ("f_%s.txt" % name)) == want, name
    n = sum(len(rows(exp + "_%s.txt" % name)) * a for name, a in (("bicov", 2), ("tricov", 3), ("tetracov", 4), ("pentacov", 5)))
    assert 0 < len(rows(tmp_path / "f_allele_frequency.txt")) <= n


def test_numbers_as_write_table_renders_them():
    from ploidyfrost_amd import hostapi
    L = hostapi.load_library()

    def f(x):
        buf = C.create_string_buffer(64)
        L.pfh_r_format_double(x, buf, 64)
        return buf.value.decode()
    # fixed unless scientific is narrower (ties: fixed); the fewest digits that give the value at 15 significant digits
    assert f(60.2174) == "60.2174" and f(0.5) == "0.5" and f(30.0) == "30" and f(0.0) == "0"
    assert f(100000.0) == "1e+05" and f(250000.0) == "250000" and f(1500000.0) == "1500000" and f(1e6) == "1e+06"
    assert f(123456.0) == "123456" and f(0.0001) == "1e-04" and f(0.00012) == "0.00012" and f(1e-5) == "1e-05"
    assert f(1 / 3) == "0.333333333333333" and f(2 / 3) == "0.666666666666667" and f(-2.5) == "-2.5"
    assert f(0.3333333) == "0.3333333" and f(1e15) == "1e+15" and f(123456789012.0) == "123456789012"


@pytest.mark.parametrize("cell", ["nan", "-nan", "inf", "NA", "0x1p3"])
def test_a_cell_r_would_read_as_na_is_refused_by_name(tables, cell):
    """R's scan() turns such a cell into NA (or stops), and `x[cond, ]` with an NA condition writes a row of NAs; with no R here to
    pin that, the cell is an error -- never a silently different filter (pf_filter.hpp, PARITY UNPINNED)"""
    lines = BICOV.splitlines()
    f = lines[1].split("\t")
    f[0] = cell
    lines[1] = "\t".join(f)
    (tables / "in_bicov.txt").write_text("\n".join(lines) + "\n")
    r = run(["filter", "-i", "in", "-o", "out", "-l", "5", "-u", "1000"], tables)
    assert r.returncode != 0 and cell in r.stderr and "line 2" in r.stderr, r.stderr
