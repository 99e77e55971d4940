"""CPU-only checks of the boundary: both shared libraries load and export exactly the entry
points their headers declare; no compute call is made (there is no GPU here) and asking for a
device fails loudly instead of falling back."""
import ctypes as C
import os
import re
import subprocess

import numpy as np
import pytest

from conftest import ROOT

from ploidyfrost_amd import build, hipapi, hostapi


@pytest.fixture(scope="module", autouse=True)
def built():
    build.build_device()


def declared(header):
    text = open(os.path.join(ROOT, "include", header)).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(pfh?_[a-z_0-9]+)\s*\(", text)))


def exported(lib):
    out = subprocess.run(["nm", "-D", "--defined-only", lib], stdout=subprocess.PIPE, text=True, check=True).stdout
    return sorted(set(re.findall(r" T (pfh?_[a-z_0-9]+)$", out, flags=re.M)))


def test_device_library_exports_every_declared_symbol():
    want = declared("ploidyfrost_hip.h")
    have = exported(hipapi.LIB_PATH)
    assert want and set(want) <= set(have), set(want) - set(have)
    assert set(hipapi.DECLARED_SYMBOLS) == set(want)
    hipapi.load_library()


def test_host_library_exports_every_declared_symbol():
    want = declared("ploidyfrost_host.h")
    have = exported(hostapi.LIB_PATH)
    assert want and set(want) <= set(have), set(want) - set(have)
    assert set(hostapi.DECLARED_SYMBOLS) == set(want)
    hostapi.load_library()


def test_device_library_is_gfx950_code_object():
    out = subprocess.run(["strings", "-n", "6", hipapi.LIB_PATH], stdout=subprocess.PIPE, text=True).stdout
    targets = set(re.findall(r"amdgcn-amd-amdhsa--(gfx[0-9a-f]+)", out))
    assert targets == {"gfx950"}, targets  # one code object, CDNA4 only


def test_record_layouts_match_the_header():
    assert hipapi.BFS_RECORD.itemsize == 32 and hipapi.ALIGN_JOB.itemsize == 24 and hipapi.ALIGN_HIT.itemsize == 40
    assert hipapi.BFS_RECORD.fields["list_off"][1] == 16 and hipapi.BFS_RECORD.fields["outcome"][1] == 24
    assert hipapi.ALIGN_HIT.fields["score"][1] == 24


def test_no_gpu_means_loud_failure_not_fallback():
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    with pytest.raises(hipapi.DeviceError) as e:
        hipapi.Device(0)
    assert e.value.status == hipapi.PF_ERR_NO_DEVICE
    with pytest.raises(RuntimeError):
        hostapi.Run(os.path.join(ROOT, "tests/golden/dip20k/graph.gfa"), os.path.join(ROOT, "tests/golden/dip20k/db"))
    cli = os.path.join(ROOT, "ploidyfrost_amd", "csrc", "ploidyfrost")
    r = subprocess.run([cli, "-g", os.path.join(ROOT, "tests/golden/dip20k/graph.gfa"), "-d",
                        os.path.join(ROOT, "tests/golden/dip20k/db"), "-o", "x"], stdout=subprocess.PIPE, stderr=subprocess.STDOUT,
                       text=True, cwd="/tmp")
    assert r.returncode != 0 and "no CPU fallback" in r.stdout


def test_product_never_touches_the_oracle():
    """ploidyfrost_amd/ must not import, link or execute anything under oracle/."""
    pkg = os.path.join(ROOT, "ploidyfrost_amd")
    for dp, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".cpp", ".hpp", ".hip", ".h", "Makefile")):
                text = open(os.path.join(dp, f), errors="replace").read()
                assert "pyoracle" not in text and "pf_oracle" not in text and "oracle/_" not in text, os.path.join(dp, f)
    for lib in (hipapi.LIB_PATH, hostapi.LIB_PATH):
        out = subprocess.run(["ldd", lib], stdout=subprocess.PIPE, text=True).stdout
        assert "oracle" not in out


def test_cli_option_errors_mirror_the_reference():
    cli = os.path.join(ROOT, "ploidyfrost_amd", "csrc", "ploidyfrost")
    r = subprocess.run([cli, "-g", "/nonexistent.gfa", "-d", "/nonexistent", "-o", "x", "-z", "3", "-M", "1", "-D", "2"],
                       stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
    assert r.returncode == 0  # the reference prints the usage and returns 0 (src/Main.cpp:765-769)
    for msg in ("Could not read the input kmc database", "Maximum number of unitigs in superbubble is at least 4",
                "Mismatch penalty should be smaller than match score", "The graph file does not exist", "Usage: PloidyFrost"):
        assert msg in r.stdout, msg


def test_cli_cutoff_subcommands(tmp_path):
    """cutoffL / cutoffU (reference src/Main.cpp:200-277, 721-762): thresholds from a k-mer histogram.  Expected values
    are worked out by hand from the definitions; where the reference binary exists it must print the same bytes."""
    cli = os.path.join(ROOT, "ploidyfrost_amd", "csrc", "ploidyfrost")
    hist = tmp_path / "hist.txt"
    counts = [900, 500, 200, 80, 60, 90, 150, 300, 420, 380, 260, 140, 70, 30, 12, 5, 2, 1]
    hist.write_text("".join("%d\t%d\n" % (i + 1, c) for i, c in enumerate(counts)))
    out = lambda *a: subprocess.run([cli] + list(a), stdout=subprocess.PIPE, text=True).stdout  # noqa: E731
    # first rise is at index 5 (60 -> 90): round(1.25 * 4) = 5 -> max(10, 5)
    assert out("cutoffL", str(hist)) == "10\n"
    # k-mers beyond the first bin: 2700; 0.998 of them + 900 = 3594 -> the first prefix sum above it is that of bin 16 (3597)
    assert out("cutoffU", str(hist)) == "16\n"
    assert out("cutoffU", str(hist), "0.5") == "8"      # 900 + 1350 = 2250 < prefix(8) = 2280; no newline, as the reference
    assert "Usage:PloidyFrost cutoffU" in out("cutoffU", str(hist), "1.5")
    assert "Usage:PloidyFrost cutoffL" in out("cutoffL")
    ref = os.path.join(ROOT, "oracle", "_ref", "PloidyFrost")
    if os.path.exists(ref):
        for args in (["cutoffL", str(hist)], ["cutoffU", str(hist)], ["cutoffU", str(hist), "0.5"], ["cutoffU", str(hist), "0.9"]):
            assert subprocess.run([ref] + args, stdout=subprocess.PIPE, text=True).stdout == out(*args)
