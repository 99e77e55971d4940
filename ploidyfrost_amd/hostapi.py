"""ctypes binding of the C facade over the C++ host layer (include/ploidyfrost_host.h)."""
from __future__ import annotations

import ctypes as C
import os

import numpy as np

from . import hipapi

PKG = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(PKG, "csrc", "libploidyfrost_host.so")


class Times(C.Structure):
    _fields_ = [(n, C.c_double) for n in ("load_s", "upload_s", "bfs_device_s", "replay_s", "bubble_write_s", "find_total_s",
                                          "cov_device_s", "tasks_s", "align_s", "sites_s", "format_s", "write_s",
                                          "ploidy_total_s")] + \
               [(n, C.c_uint64) for n in ("unitigs", "kmers", "candidates", "superbubbles", "tasks", "align_jobs", "site_strings",
                                          "output_bytes")] + [("allele", C.c_uint64 * 4), ("core_cov", C.c_uint64),
                                                              ("core_num", C.c_uint64), ("scan_s", C.c_double), ("scan_serial_s", C.c_double),
                                                              ("bfs_large", C.c_uint64), ("bfs_max_seen", C.c_uint64), ("bfs_deferred", C.c_uint64)]


_lib = None
DECLARED_SYMBOLS = ["pfh_open", "pfh_close", "pfh_last_error", "pfh_set_output_dir", "pfh_set_write_files", "pfh_set_threads", "pfh_set_batch_bubbles", "pfh_set_overlap_output", "pfh_set_third_tier_on_host", "pfh_set_partition", "pfh_set_unitig_id",
                    "pfh_find_superbubbles", "pfh_ploidy_estimation", "pfh_get_times", "pfh_device_ctx", "pfh_state", "pfh_last_allele_frequency",
                    "pfh_open_colored", "pfh_num_colors", "pfh_ploidy_estimation_colored",
                    "pfh_colors_open", "pfh_colors_close", "pfh_colors_count", "pfh_colors_unitigs", "pfh_colors_name",
                    "pfh_colors_unitig", "pfh_bifrost_kmer_hash", "pfh_gfa_abundant_kmers", "pfh_gfa_write_unitig_ids", "pfh_gfa_numbering_replays", "pfh_gfa_minimizer_counts", "pfh_host_walk",
                    "pfh_gmm_open", "pfh_gmm_close", "pfh_gmm_last_error", "pfh_gmm_read_fre", "pfh_gmm_read_cov", "pfh_gmm_set_values",
                    "pfh_gmm_size", "pfh_gmm_values", "pfh_gmm_fit", "pfh_gmm_run", "pfh_gmm_kernel_time"]


def load_library() -> C.CDLL:
    global _lib
    if _lib is not None:
        return _lib
    hipapi.load_library()  # device layer first (and torch's HIP runtime before it)
    if not os.path.exists(LIB_PATH):
        raise RuntimeError("%s is missing: build with `make -C ploidyfrost_amd/csrc`" % LIB_PATH)
    L = C.CDLL(LIB_PATH)
    vp = C.c_void_p
    L.pfh_open.restype = vp
    L.pfh_open.argtypes = [C.c_char_p, C.c_char_p, C.c_uint32, C.c_double, C.c_double, C.c_double, C.c_int]
    L.pfh_close.argtypes = [vp]
    L.pfh_last_error.restype = C.c_char_p
    L.pfh_last_error.argtypes = [vp]
    L.pfh_set_output_dir.argtypes = [vp, C.c_char_p]
    L.pfh_set_write_files.argtypes = [vp, C.c_int]
    L.pfh_set_threads.argtypes = [vp, C.c_uint32]
    L.pfh_set_batch_bubbles.argtypes = [vp, C.c_uint64]
    L.pfh_set_overlap_output.argtypes = [vp, C.c_int]
    L.pfh_set_third_tier_on_host.argtypes = [vp, C.c_int]
    L.pfh_set_partition.argtypes = [vp, C.c_uint32, C.c_uint32]
    L.pfh_set_unitig_id.argtypes = [vp, C.c_char_p]
    L.pfh_find_superbubbles.argtypes = [vp, C.c_char_p]
    L.pfh_ploidy_estimation.argtypes = [vp, C.c_char_p, C.c_int, C.c_int]
    L.pfh_get_times.argtypes = [vp, C.POINTER(Times)]
    L.pfh_device_ctx.restype = vp
    L.pfh_device_ctx.argtypes = [vp]
    L.pfh_state.argtypes = [vp, vp, vp, vp]
    L.pfh_last_allele_frequency.restype = C.c_void_p
    L.pfh_last_allele_frequency.argtypes = [vp, C.POINTER(C.c_uint64)]
    L.pfh_open_colored.restype = vp
    L.pfh_open_colored.argtypes = [C.c_char_p, C.c_char_p, C.c_char_p, C.c_uint32, C.c_double, C.c_double, C.c_double, C.c_uint32, C.c_int]
    L.pfh_num_colors.restype = C.c_uint32
    L.pfh_num_colors.argtypes = [vp]
    L.pfh_ploidy_estimation_colored.argtypes = [vp, C.c_char_p, C.POINTER(C.c_int), C.POINTER(C.c_int), C.c_uint32]
    L.pfh_colors_open.restype = vp
    L.pfh_colors_open.argtypes = [C.c_char_p, C.c_char_p, C.c_uint32]
    L.pfh_colors_close.argtypes = [vp]
    L.pfh_colors_count.restype = C.c_uint32
    L.pfh_colors_count.argtypes = [vp]
    L.pfh_colors_unitigs.restype = C.c_uint32
    L.pfh_colors_unitigs.argtypes = [vp]
    L.pfh_colors_name.restype = C.c_char_p
    L.pfh_colors_name.argtypes = [vp, C.c_uint32]
    L.pfh_colors_unitig.restype = C.c_uint64
    L.pfh_colors_unitig.argtypes = [vp, C.c_uint32, vp, C.POINTER(C.c_uint32), C.POINTER(C.c_uint32)]
    L.pfh_gfa_abundant_kmers.restype = C.c_uint64
    L.pfh_gfa_abundant_kmers.argtypes = [C.c_char_p]
    L.pfh_gfa_numbering_replays.restype = C.c_uint32
    L.pfh_gfa_numbering_replays.argtypes = [C.c_char_p]
    L.pfh_gfa_minimizer_counts.restype = C.c_uint64
    L.pfh_gfa_minimizer_counts.argtypes = [C.c_char_p, C.c_void_p, C.c_uint64]
    L.pfh_gfa_write_unitig_ids.restype = C.c_int
    L.pfh_gfa_write_unitig_ids.argtypes = [C.c_char_p, C.c_char_p]
    L.pfh_bifrost_kmer_hash.restype = C.c_uint64
    L.pfh_bifrost_kmer_hash.argtypes = [C.c_uint64, C.c_uint64]
    _lib = L
    return L


class Colors:
    """The product's reading of a colored graph's .bfg_colors (host only, no GPU)."""

    def __init__(self, gfa: str, colors: str, threads: int = 1):
        self.L = load_library()
        self.h = self.L.pfh_colors_open(gfa.encode(), colors.encode(), threads)
        if not self.h:
            raise RuntimeError("ploidyfrost host layer: " + self.L.pfh_last_error(None).decode())
        self.n_colors = self.L.pfh_colors_count(self.h)
        self.n = self.L.pfh_colors_unitigs(self.h)
        self.names = [self.L.pfh_colors_name(self.h, c).decode() for c in range(self.n_colors)]

    def close(self):
        if getattr(self, "h", None):
            self.L.pfh_colors_close(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def unitig(self, u: int):
        """(presence[colour, kmer] uint8, UnitigColors::size(um), n_full_enc)"""
        km, nf = C.c_uint32(), C.c_uint32()
        self.L.pfh_colors_unitig(self.h, u, None, C.byref(km), C.byref(nf))
        out = np.zeros((self.n_colors, km.value), dtype=np.uint8)
        sz = self.L.pfh_colors_unitig(self.h, u, out.ctypes.data, C.byref(km), C.byref(nf))
        return out, sz, nf.value


class Run:
    """CompactedDBG::read + CDBG::CDBG of the reference's main(): graph and count table in HBM."""

    def __init__(self, gfa: str, kmc_prefix: str, z: int = 8, M: float = 2.0, D: float = -1.0, G: float = -3.0,
                 device: int = 0):
        self.L = load_library()
        self.h = self.L.pfh_open(gfa.encode(), kmc_prefix.encode(), z, M, D, G, device)
        if not self.h:
            raise RuntimeError("ploidyfrost host layer: " + self.L.pfh_last_error(None).decode())

    def close(self):
        if getattr(self, "h", None):
            self.L.pfh_close(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _check(self, st):
        if st != 0:
            raise hipapi.DeviceError(st, self.L.pfh_last_error(self.h).decode())

    def set_output_dir(self, d: str):
        self.L.pfh_set_output_dir(self.h, d.encode())

    def set_write_files(self, on: bool):
        self.L.pfh_set_write_files(self.h, int(on))

    def set_threads(self, threads: int):
        self.L.pfh_set_threads(self.h, threads)

    def set_overlap_output(self, on: bool):
        self.L.pfh_set_overlap_output(self.h, int(on))

    def set_third_tier_on_host(self, on: bool):
        self.L.pfh_set_third_tier_on_host(self.h, int(on))

    def set_partition(self, rank: int, world: int):
        """one graph on `world` GPUs: this run handles slice `rank` of the bubble list in ploidy_estimation"""
        self.L.pfh_set_partition(self.h, rank, world)

    def set_batch_bubbles(self, n: int):
        self.L.pfh_set_batch_bubbles(self.h, n)

    def set_unitig_id(self, outpre: str):
        self._check(self.L.pfh_set_unitig_id(self.h, outpre.encode()))

    def find_superbubbles(self, outpre: str):
        self._check(self.L.pfh_find_superbubbles(self.h, outpre.encode()))

    def ploidy_estimation(self, outpre: str, lower: int = 10, upper: int = 1000):
        self._check(self.L.pfh_ploidy_estimation(self.h, outpre.encode(), lower, upper))

    def times(self) -> dict:
        t = Times()
        self.L.pfh_get_times(self.h, C.byref(t))
        d = {n: getattr(t, n) for n, _ in Times._fields_ if n != "allele"}
        d["allele"] = list(t.allele)
        return d

    def last_allele_frequency(self) -> np.ndarray:
        """bytes of <outpre>_allele_frequency.txt of the last run (copy)"""
        n = C.c_uint64()
        p = self.L.pfh_last_allele_frequency(self.h, C.byref(n))
        if not n.value:
            return np.zeros(0, dtype=np.uint8)
        return np.ctypeslib.as_array(C.cast(p, C.POINTER(C.c_uint8)), shape=(n.value,)).copy()

    def device_ctx(self) -> int:
        return self.L.pfh_device_ctx(self.h)

    def state(self):
        n = self.times()["unitigs"]
        f = np.empty(n, dtype=np.uint8)
        p = np.empty(n, dtype=np.uint32)
        m = np.empty(n, dtype=np.uint32)
        self.L.pfh_state(self.h, f.ctypes.data, p.ctypes.data, m.ctypes.data)
        return f, p, m


class ColoredRun(Run):
    """ColoredCDBG::read + CCDBG::CCDBG (reference src/Main.cpp:775-795): graph, colour sets, and the count
    databases of all colours joined in one HBM table."""

    def __init__(self, gfa: str, colors: str, db_prefixes: list[str], workdir: str, z: int = 8, M: float = 2.0, D: float = -1.0,
                 G: float = -3.0, threads: int = 1, device: int = 0):
        self.L = load_library()
        lst = os.path.join(workdir, "kmc_databases.txt")
        with open(lst, "w") as f:
            f.write("".join(p + "\n" for p in db_prefixes))
        self.h = self.L.pfh_open_colored(gfa.encode(), colors.encode(), lst.encode(), z, M, D, G, threads, device)
        if not self.h:
            raise RuntimeError("ploidyfrost host layer: " + self.L.pfh_last_error(None).decode())
        self.n_colors = self.L.pfh_num_colors(self.h)

    def ploidy_estimation(self, outpre: str, cutoffs):
        """cutoffs: one (lower, upper) per colour, the reference's -C file"""
        lo = (C.c_int * self.n_colors)(*[int(c[0]) for c in cutoffs])
        up = (C.c_int * self.n_colors)(*[int(c[1]) for c in cutoffs])
        self._check(self.L.pfh_ploidy_estimation_colored(self.h, outpre.encode(), lo, up, len(cutoffs)))


class Gmm:
    """`PloidyFrost model` (reference class GmmModel, src/GmmModel.hpp): readers on the host, the EM fit on the GPU."""

    def __init__(self, device: int = 0):
        self.L = load_library()
        L, vp, d = self.L, C.c_void_p, C.c_double
        L.pfh_gmm_open.restype = vp
        L.pfh_gmm_open.argtypes = [C.c_int]
        L.pfh_gmm_close.argtypes = [vp]
        L.pfh_gmm_last_error.restype = C.c_char_p
        L.pfh_gmm_last_error.argtypes = [vp]
        L.pfh_gmm_read_fre.argtypes = [vp, C.c_char_p, d]
        L.pfh_gmm_read_cov.argtypes = [vp, C.c_char_p, d]
        L.pfh_gmm_set_values.argtypes = [vp, vp, C.c_uint64]
        L.pfh_gmm_size.restype = C.c_uint64
        L.pfh_gmm_size.argtypes = [vp]
        L.pfh_gmm_values.argtypes = [vp, vp]
        L.pfh_gmm_fit.argtypes = [vp, C.c_uint32, d, d, C.c_int32, d, vp, vp, vp, C.POINTER(d), C.POINTER(d), C.POINTER(C.c_uint32)]
        L.pfh_gmm_run.argtypes = [vp, C.c_int, C.c_int, d, d, C.c_int32, d, C.c_char_p]
        L.pfh_gmm_kernel_time.argtypes = [vp, C.c_int, C.POINTER(d), C.POINTER(C.c_uint64)]
        self.h = L.pfh_gmm_open(device)
        if not self.h:
            raise RuntimeError(L.pfh_gmm_last_error(None).decode())

    def close(self):
        if getattr(self, "h", None):
            self.L.pfh_gmm_close(self.h)
            self.h = None

    __del__ = close

    def _check(self, rc):
        if rc != 0:
            raise RuntimeError(self.L.pfh_gmm_last_error(self.h).decode())

    def read_fre(self, path, min_frequency=0.0):
        self._check(self.L.pfh_gmm_read_fre(self.h, path.encode(), min_frequency))

    def read_cov(self, prefix, min_frequency=0.0):
        self._check(self.L.pfh_gmm_read_cov(self.h, prefix.encode(), min_frequency))

    def set_values(self, v):
        import numpy as np
        v = np.ascontiguousarray(v, dtype=np.float64)
        self._check(self.L.pfh_gmm_set_values(self.h, v.ctypes.data, len(v)))

    def values(self):
        import numpy as np
        out = np.zeros(self.L.pfh_gmm_size(self.h), dtype=np.float64)
        self._check(self.L.pfh_gmm_values(self.h, out.ctypes.data if len(out) else None) if len(out) else 0)
        return out

    def fit(self, gauss, m_thre=5.0, n_thre=2.0, max_iter=1000, max_delta=0.01):
        import numpy as np
        w, mean, var = (np.zeros(gauss) for _ in range(3))
        ll, aic, it = C.c_double(), C.c_double(), C.c_uint32()
        self._check(self.L.pfh_gmm_fit(self.h, gauss, m_thre, n_thre, max_iter, max_delta, w.ctypes.data, mean.ctypes.data,
                                       var.ctypes.data, C.byref(ll), C.byref(aic), C.byref(it)))
        return {"weights": w, "means": mean, "vars": var, "loglik": ll.value, "aic": aic.value, "iterations": it.value}

    def run(self, outprefix, lo=1, hi=9, m_thre=5.0, n_thre=2.0, max_iter=1000, max_delta=0.01):
        self._check(self.L.pfh_gmm_run(self.h, lo, hi, m_thre, n_thre, max_iter, max_delta, outprefix.encode()))

    def enable_timing(self, on=True):
        self._check(self.L.pfh_gmm_kernel_time(self.h, 1 if on else 0, None, None))

    def kernel_time(self):
        ms, n = C.c_double(), C.c_uint64()
        self._check(self.L.pfh_gmm_kernel_time(self.h, -1, C.byref(ms), C.byref(n)))
        return ms.value, n.value
