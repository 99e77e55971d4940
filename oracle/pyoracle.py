"""ctypes binding of the CPU ORACLE (oracle/pf_oracle.h) -- TEST INFRASTRUCTURE ONLY.

May be imported by tests/, by __graft_entry__.smoke() and by the cpu_baseline leg of
bench.py; never by anything under ploidyfrost_amd/.
"""
from __future__ import annotations

import ctypes as C
import os
import shutil
import subprocess

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
LIB = os.path.join(HERE, "_build", "libpf_oracle.so")
CLI = os.path.join(HERE, "_build", "pf_oracle_cli")
REF_BIN = os.path.join(HERE, "_ref", "PloidyFrost")
REF_BIFROST = os.path.join(HERE, "_ref", "Bifrost")
REF_COLORS_DUMP = os.path.join(HERE, "_ref", "colors_dump")

NONE = 0xFFFFFFFF
OUTCOMES = {0: "none", 1: "cycle_exit", 2: "reject", 3: "accept"}


def build() -> str:
    """g++ the restatement into oracle/_build (no-op when up to date / no compiler)."""
    if shutil.which("make") and shutil.which("g++"):
        r = subprocess.run(["make", "-C", HERE], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
        if r.returncode != 0:
            raise RuntimeError("oracle build failed:\n" + r.stdout)
    if not os.path.exists(LIB):
        raise RuntimeError("oracle library missing and cannot be built")
    return LIB


def build_reference() -> str | None:
    """oracle/_ref from /root/reference -- only where that tree exists (the build container)."""
    ref = os.environ.get("PF_REFERENCE", "/root/reference")
    if not os.path.isdir(os.path.join(ref, "src")):
        return REF_BIN if os.path.exists(REF_BIN) else None
    r = subprocess.run(["make", "-f", os.path.join("oracle", "Makefile.ref"), "-j8", "REF=" + ref], cwd=ROOT,
                       stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
    if r.returncode != 0:
        raise RuntimeError("reference build failed:\n" + r.stdout[-4000:])
    return REF_BIN


_lib = None


def lib() -> C.CDLL:
    global _lib
    if _lib is not None:
        return _lib
    build()
    L = C.CDLL(LIB)
    vp, u32, u64 = C.c_void_p, C.c_uint32, C.c_uint64
    L.pfo_open.restype = vp
    L.pfo_open.argtypes = [C.c_char_p, C.c_char_p]
    L.pfo_close.argtypes = [vp]
    L.pfo_last_error.restype = C.c_char_p
    L.pfo_k.argtypes = [vp]
    L.pfo_num_unitigs.restype = u32
    L.pfo_num_unitigs.argtypes = [vp]
    L.pfo_num_kmers.restype = u64
    L.pfo_num_kmers.argtypes = [vp]
    L.pfo_unitig_seq.restype = u32
    L.pfo_unitig_seq.argtypes = [vp, u32, C.c_char_p, u32]
    L.pfo_adjacency.argtypes = [vp, vp, vp]
    L.pfo_unitig_cov.argtypes = [vp, u32, C.POINTER(u64), C.POINTER(u32)]
    L.pfo_string_cov.argtypes = [vp, C.c_char_p, u32, u32, u32, C.POINTER(u64), C.POINTER(C.c_int)]
    L.pfo_kmer_count.argtypes = [vp, C.c_char_p, C.POINTER(u32)]
    L.pfo_extract.argtypes = [vp, u32, C.POINTER(u32), C.POINTER(u32), vp, C.POINTER(u32), vp, u32,
                              C.POINTER(C.c_int), C.POINTER(C.c_int)]
    L.pfo_seq_align.argtypes = [C.c_double, C.c_double, C.c_double, C.POINTER(C.c_char_p), C.c_int, C.c_char_p, u32,
                                C.POINTER(u32), vp, C.POINTER(u32), vp, C.POINTER(u32), vp, C.POINTER(u32), vp, u32, u32]
    L.pfo_pairwise.argtypes = [C.c_double, C.c_double, C.c_double, C.c_char_p, C.c_char_p, C.c_char_p, u32]
    L.pfo_set_unitig_id.argtypes = [vp, C.c_char_p, C.c_char_p]
    L.pfo_find_superbubbles.argtypes = [vp, C.c_char_p, C.c_char_p, u32, C.POINTER(u64)]
    L.pfo_ploidy_estimation.argtypes = [vp, C.c_char_p, C.c_char_p, C.c_int, C.c_int, C.c_double, C.c_double, C.c_double,
                                        C.POINTER(u64), C.POINTER(u64), C.POINTER(u64)]
    L.pfo_state.argtypes = [vp, vp, vp, vp]
    L.pfo_open_colored.restype = vp
    L.pfo_open_colored.argtypes = [C.c_char_p, C.c_char_p, C.c_char_p]
    L.pfo_num_colors.restype = u32
    L.pfo_num_colors.argtypes = [vp]
    L.pfo_unitig_colors.restype = u64
    L.pfo_unitig_colors.argtypes = [vp, u32, vp, C.POINTER(u32)]
    L.pfo_unitig_cov_color.argtypes = [vp, u32, u32, u32, u32, C.POINTER(C.c_double), C.POINTER(C.c_int)]
    L.pfo_string_cov_color.argtypes = [vp, u32, C.c_char_p, u32, u32, u32, C.POINTER(C.c_double), C.POINTER(C.c_int)]
    L.pfo_find_unitig.argtypes = [vp, C.c_char_p, u32, C.POINTER(u32), C.POINTER(u32), C.POINTER(u32)]
    L.pfo_ploidy_estimation_colored.argtypes = [vp, C.c_char_p, C.c_char_p, C.POINTER(C.c_int), C.POINTER(C.c_int), C.c_double,
                                                C.c_double, C.c_double, C.POINTER(u64), C.POINTER(u64), C.POINTER(u64)]
    _lib = L
    return L


class Oracle:
    """One loaded graph + k-mer database."""

    def __init__(self, gfa: str, kmc_prefix: str | None):
        self.L = lib()
        self.h = self.L.pfo_open(gfa.encode(), (kmc_prefix or "").encode())
        if not self.h:
            raise RuntimeError("oracle: " + self.L.pfo_last_error().decode())
        self.k = self.L.pfo_k(self.h)
        self.n = self.L.pfo_num_unitigs(self.h)
        self.n_kmers = self.L.pfo_num_kmers(self.h)

    def close(self):
        if self.h:
            self.L.pfo_close(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def sequences(self) -> list[bytes]:
        out = []
        buf = C.create_string_buffer(1 << 16)
        for u in range(self.n):
            n = self.L.pfo_unitig_seq(self.h, u, buf, len(buf))
            if n > len(buf):
                buf = C.create_string_buffer(n + 16)
                self.L.pfo_unitig_seq(self.h, u, buf, len(buf))
            out.append(buf.raw[:n])
        return out

    def adjacency(self):
        succ = np.empty(self.n * 8, dtype=np.uint32)
        pred = np.empty(self.n * 8, dtype=np.uint32)
        self.L.pfo_adjacency(self.h, succ.ctypes.data, pred.ctypes.data)
        return succ.reshape(-1, 4), pred.reshape(-1, 4)

    def unitig_cov(self):
        s = np.zeros(self.n, dtype=np.uint64)
        m = np.zeros(self.n, dtype=np.uint32)
        miss = np.zeros(self.n, dtype=np.uint8)
        a, b = C.c_uint64(), C.c_uint32()
        for u in range(self.n):
            miss[u] = self.L.pfo_unitig_cov(self.h, u, C.byref(a), C.byref(b))
            s[u], m[u] = a.value, b.value
        return s, m, miss

    def string_cov(self, text: bytes, low: int, up: int):
        a, ok = C.c_uint64(), C.c_int()
        miss = self.L.pfo_string_cov(self.h, text, len(text), low, up, C.byref(a), C.byref(ok))
        return a.value, ok.value, miss

    def extract(self, s_ov: int, cap: int = 1 << 16):
        ex, ns, nc = C.c_uint32(), C.c_uint32(), C.c_uint32()
        fc, ft = C.c_int(), C.c_int()
        seen = np.empty(cap, dtype=np.uint32)
        cyc = np.empty(cap, dtype=np.uint32)
        oc = self.L.pfo_extract(self.h, s_ov, C.byref(ex), C.byref(ns), seen.ctypes.data, C.byref(nc), cyc.ctypes.data,
                                cap, C.byref(fc), C.byref(ft))
        return dict(outcome=oc, exit=ex.value, seen=seen[: ns.value].copy(), cyc=cyc[: nc.value].copy(),
                    flag_cycle=fc.value, flag_tip=ft.value)

    def run(self, outdir: str, prefix: str, z=8, lower=10, upper=1000, M=2.0, D=-1.0, G=-3.0):
        L = self.L
        if L.pfo_set_unitig_id(self.h, outdir.encode(), prefix.encode()):
            raise RuntimeError(L.pfo_last_error().decode())
        nb = C.c_uint64()
        if L.pfo_find_superbubbles(self.h, outdir.encode(), prefix.encode(), z, C.byref(nb)):
            raise RuntimeError(L.pfo_last_error().decode())
        allele = (C.c_uint64 * 4)()
        cc, cn = C.c_uint64(), C.c_uint64()
        rc = L.pfo_ploidy_estimation(self.h, outdir.encode(), prefix.encode(), lower, upper, M, D, G, allele, C.byref(cc),
                                     C.byref(cn))
        if rc:
            raise RuntimeError(L.pfo_last_error().decode())
        return dict(bubbles=nb.value, allele=list(allele), core_cov=cc.value, core_num=cn.value)

    def find_superbubbles(self, z=8):
        nb = C.c_uint64()
        if self.L.pfo_find_superbubbles(self.h, None, None, z, C.byref(nb)):
            raise RuntimeError(self.L.pfo_last_error().decode())
        return nb.value

    def state(self):
        f = np.empty(self.n, dtype=np.uint8)
        p = np.empty(self.n, dtype=np.uint32)
        m = np.empty(self.n, dtype=np.uint32)
        self.L.pfo_state(self.h, f.ctypes.data, p.ctypes.data, m.ctypes.data)
        return f, p, m


class ColoredOracle(Oracle):
    """Colored twin (reference src/CCDBG.cpp): graph + the real Bifrost's colour dump + one database per colour."""

    def __init__(self, gfa: str, colors_dump: str, db_prefixes: list[str], workdir: str):
        self.L = lib()
        lst = os.path.join(workdir, "oracle_dbs.txt")
        with open(lst, "w") as f:
            f.write("".join(p + "\n" for p in db_prefixes))
        self.h = self.L.pfo_open_colored(gfa.encode(), colors_dump.encode(), lst.encode() if db_prefixes else b"")
        if not self.h:
            raise RuntimeError("oracle: " + self.L.pfo_last_error().decode())
        self.k = self.L.pfo_k(self.h)
        self.n = self.L.pfo_num_unitigs(self.h)
        self.n_kmers = self.L.pfo_num_kmers(self.h)
        self.n_colors = self.L.pfo_num_colors(self.h)

    def unitig_colors(self, u: int, n_kmers: int):
        """(presence[colour, kmer] uint8, UnitigColors::size, n_full_enc)"""
        out = np.zeros((self.n_colors, n_kmers), dtype=np.uint8)
        nf = C.c_uint32()
        sz = self.L.pfo_unitig_colors(self.h, u, out.ctypes.data, C.byref(nf))
        return out, sz, nf.value

    def unitig_cov_color(self, colour: int, u: int, low: int, up: int):
        m, ok = C.c_double(), C.c_int()
        self.L.pfo_unitig_cov_color(self.h, colour, u, low, up, C.byref(m), C.byref(ok))
        return m.value, ok.value

    def string_cov_color(self, colour: int, text: bytes, low: int, up: int):
        m, ok = C.c_double(), C.c_int()
        self.L.pfo_string_cov_color(self.h, colour, text, len(text), low, up, C.byref(m), C.byref(ok))
        return m.value, ok.value

    def find_unitig(self, text: bytes):
        u, d, n = C.c_uint32(), C.c_uint32(), C.c_uint32()
        if not self.L.pfo_find_unitig(self.h, text, len(text), C.byref(u), C.byref(d), C.byref(n)):
            return None
        return u.value, d.value, n.value

    def run(self, outdir: str, prefix: str, cutoffs, z=8, M=2.0, D=-1.0, G=-3.0):
        L = self.L
        if L.pfo_set_unitig_id(self.h, outdir.encode(), prefix.encode()):
            raise RuntimeError(L.pfo_last_error().decode())
        nb = C.c_uint64()
        if L.pfo_find_superbubbles(self.h, outdir.encode(), prefix.encode(), z, C.byref(nb)):
            raise RuntimeError(L.pfo_last_error().decode())
        lo = (C.c_int * self.n_colors)(*[c[0] for c in cutoffs])
        up = (C.c_int * self.n_colors)(*[c[1] for c in cutoffs])
        allele = (C.c_uint64 * 4)()
        cc, cn = C.c_uint64(), C.c_uint64()
        rc = L.pfo_ploidy_estimation_colored(self.h, outdir.encode(), prefix.encode(), lo, up, M, D, G, allele, C.byref(cc),
                                             C.byref(cn))
        if rc:
            raise RuntimeError(L.pfo_last_error().decode())
        return dict(bubbles=nb.value, allele=list(allele), core_cov=cc.value, core_num=cn.value)


def seq_align(strs: list[bytes], M=2.0, D=-1.0, G=-3.0):
    L = lib()
    arr = (C.c_char_p * len(strs))(*strs)
    cap = 1 << 20
    out = C.create_string_buffer(cap)
    pos_cap, part_cap = 1 << 14, 1 << 18
    n_snp, n_indel, n_len, n_cols = C.c_uint32(), C.c_uint32(), C.c_uint32(), C.c_uint32()
    snp = np.zeros(pos_cap, dtype=np.uint32)
    indel = np.zeros(pos_cap, dtype=np.uint32)
    ilen = np.zeros(pos_cap, dtype=np.uint32)
    part = np.zeros(part_cap, dtype=np.uint16)
    rows = L.pfo_seq_align(M, D, G, arr, len(strs), out, cap, C.byref(n_snp), snp.ctypes.data, C.byref(n_indel),
                           indel.ctypes.data, C.byref(n_len), ilen.ctypes.data, C.byref(n_cols), part.ctypes.data,
                           pos_cap, part_cap)
    text = out.value.split(b"\n")[:rows]
    return dict(rows=text, snp_pos=snp[: n_snp.value].copy(), indel_pos=indel[: n_indel.value].copy(),
                indel_len=ilen[: n_len.value].copy(),
                partition=part[: n_cols.value * max(rows, 1)].reshape(n_cols.value, max(rows, 1)).copy() if rows else None)


def pairwise(a: bytes, b: bytes, M=2.0, D=-1.0, G=-3.0):
    """[(a_row, b_row, gap_pos list, score, n_pos, indel)] in traceback order."""
    L = lib()
    cap = 1 << 16
    while True:
        out = C.create_string_buffer(cap)
        n = L.pfo_pairwise(M, D, G, a, b, out, cap)
        if n >= 0:
            break
        cap *= 8
    res = []
    for line in out.value.split(b"\n")[:n]:
        f = line.split(b"\t")
        gp = [int(x) for x in f[5].split(b",")] if f[5] else []
        res.append((f[0], f[1], gp, int(f[2]), int(f[3]), int(f[4])))
    return res


class GmmOracle:
    """`PloidyFrost model` restated on the CPU (oracle/pf_oracle_gmm.cpp; reference src/GmmModel.cpp)."""

    def __init__(self):
        self.L = lib()
        L, vp, d = self.L, C.c_void_p, C.c_double
        L.pfo_gmm_open.restype = vp
        L.pfo_gmm_close.argtypes = [vp]
        L.pfo_gmm_error.restype = C.c_char_p
        L.pfo_gmm_error.argtypes = [vp]
        L.pfo_gmm_read_fre.argtypes = [vp, C.c_char_p, d]
        L.pfo_gmm_read_cov.argtypes = [vp, C.c_char_p, d]
        L.pfo_gmm_set_values.argtypes = [vp, vp, C.c_uint64]
        L.pfo_gmm_size.restype = C.c_uint64
        L.pfo_gmm_size.argtypes = [vp]
        L.pfo_gmm_values.argtypes = [vp, vp]
        L.pfo_gmm_fit.argtypes = [vp, C.c_uint32, d, d, C.c_int, d, vp, vp, vp, C.POINTER(d), C.POINTER(d), C.POINTER(C.c_uint32)]
        L.pfo_gmm_run.argtypes = [vp, C.c_int, C.c_int, d, d, C.c_int, d, C.c_char_p]
        self.h = L.pfo_gmm_open()

    def __del__(self):
        if getattr(self, "h", None):
            self.L.pfo_gmm_close(self.h)
            self.h = None

    def _check(self, rc):
        if rc != 0:
            raise RuntimeError(self.L.pfo_gmm_error(self.h).decode())

    def read_fre(self, path, min_frequency=0.0):
        self._check(self.L.pfo_gmm_read_fre(self.h, path.encode(), min_frequency))

    def read_cov(self, prefix, min_frequency=0.0):
        self._check(self.L.pfo_gmm_read_cov(self.h, prefix.encode(), min_frequency))

    def set_values(self, v):
        v = np.ascontiguousarray(v, dtype=np.float64)
        self.L.pfo_gmm_set_values(self.h, v.ctypes.data, len(v))

    def values(self):
        out = np.zeros(self.L.pfo_gmm_size(self.h), dtype=np.float64)
        self.L.pfo_gmm_values(self.h, out.ctypes.data)
        return out

    def fit(self, gauss, m_thre=5.0, n_thre=2.0, max_iter=1000, max_delta=0.01):
        w, mean, var = (np.zeros(gauss) for _ in range(3))
        ll, aic, it = C.c_double(), C.c_double(), C.c_uint32()
        self.L.pfo_gmm_fit(self.h, gauss, m_thre, n_thre, max_iter, max_delta, w.ctypes.data, mean.ctypes.data, var.ctypes.data,
                           C.byref(ll), C.byref(aic), C.byref(it))
        return {"weights": w, "means": mean, "vars": var, "loglik": ll.value, "aic": aic.value, "iterations": it.value}

    def run(self, outprefix, lo=1, hi=9, m_thre=5.0, n_thre=2.0, max_iter=1000, max_delta=0.01):
        self._check(self.L.pfo_gmm_run(self.h, lo, hi, m_thre, n_thre, max_iter, max_delta, outprefix.encode()))


# ---- cells the reference leaves undefined (pfo::indel_len_at, pf_oracle_align.hpp) ---------------------------------------

OUTPUT_SUFFIXES = ["Unitig_Id", "super_bubble", "alignseq", "allele_frequency", "bicov", "bifre", "tricov", "trifre", "tetracov",
                   "tetrafre", "pentacov", "pentafre"]


def read_ub_log(path: str) -> dict:
    """{'bicov': {line, ...}, ...} from a PFO_UB_LOG file (1-based lines)."""
    cells: dict = {}
    if path and os.path.exists(path):
        for ln in open(path):
            name, line = ln.split("\t")
            cells.setdefault(name, set()).add(int(line))
    return cells


def mask_ub(data: bytes, lines: set, colored: bool) -> bytes:
    """A *cov.txt file with the indel-length field of the given lines replaced by '?'.  Field position counted from the
    row's end (rows end in a tab): single-sample ... strict, LEN, var_count, sites, dist, '' ; colored ... strict, LEN,
    var_count, sites, Cramer V, dist, ''."""
    if not lines:
        return data
    rows = data.split(b"\n")
    at = -6 if colored else -5
    for ln in lines:
        if ln - 1 < len(rows):
            f = rows[ln - 1].split(b"\t")
            if len(f) >= -at:
                f[at] = b"?"
                rows[ln - 1] = b"\t".join(f)
    return b"\n".join(rows)


def compare_outputs(dir_a: str, dir_b: str, prefix: str, ub_cells: dict | None = None, colored: bool = False) -> list:
    """Suffixes of the twelve files that differ between two output directories, the undefined cells left out."""
    bad = []
    for suf in OUTPUT_SUFFIXES:
        a = open(os.path.join(dir_a, "%s_%s.txt" % (prefix, suf)), "rb").read()
        b = open(os.path.join(dir_b, "%s_%s.txt" % (prefix, suf)), "rb").read()
        if ub_cells and suf in ub_cells:
            a, b = mask_ub(a, ub_cells[suf], colored), mask_ub(b, ub_cells[suf], colored)
        if a != b:
            bad.append(suf)
    return bad
