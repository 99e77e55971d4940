"""Build helper: compile the gfx950 device library + C++ host layer in-tree.
hipcc cross-compiles without a GPU.  (The CPU checker is test infrastructure; it has its own
build script outside this package and is never built or loaded from here.)"""
from __future__ import annotations

import os
import subprocess

PKG = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(PKG)
CSRC = os.path.join(PKG, "csrc")
DEVICE_LIB = os.path.join(CSRC, "libploidyfrost_hip.so")


def _newer(target: str, sources: list[str]) -> bool:
    if not os.path.exists(target):
        return False
    t = os.path.getmtime(target)
    return all(os.path.getmtime(s) <= t for s in sources if os.path.exists(s))


def _run(cmd: list[str], cwd: str | None = None) -> None:
    r = subprocess.run(cmd, cwd=cwd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
    if r.returncode != 0:
        raise RuntimeError("build failed: %s\n%s" % (" ".join(cmd), r.stdout))


def build_device(force: bool = False) -> str:
    """hipcc --offload-arch=gfx950 -> ploidyfrost_amd/csrc/libploidyfrost_hip.so (+ host layer)."""
    r = subprocess.run(["make", "-C", CSRC] + (["-B"] if force else []), stdout=subprocess.PIPE,
                       stderr=subprocess.STDOUT, text=True)
    if r.returncode != 0:
        raise RuntimeError("device build failed:\n" + r.stdout)
    return DEVICE_LIB
