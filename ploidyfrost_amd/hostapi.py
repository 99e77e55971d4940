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
                                                              ("bfs_large", C.c_uint64), ("bfs_max_seen", C.c_uint64), ("bfs_deferred", C.c_uint64),
                                                              ("snp_jobs", C.c_uint64), ("pair_jobs", C.c_uint64), ("wave_jobs", C.c_uint64), ("stack_jobs", C.c_uint64),
                                                              ("host_commit_records", C.c_uint64), ("host_walk_vertices", C.c_uint64)]


_lib = None
DECLARED_SYMBOLS = ["pfh_open", "pfh_close", "pfh_last_error", "pfh_set_output_dir", "pfh_set_write_files", "pfh_set_threads", "pfh_set_batch_bubbles", "pfh_set_align_pieces", "pfh_set_reference_threads", "pfh_set_overlap_output", "pfh_set_third_tier_on_host", "pfh_set_unitig_id",
                    "pfh_find_superbubbles", "pfh_ploidy_estimation", "pfh_get_times", "pfh_load_trace", "pfh_filter", "pfh_r_format_double", "pfh_device_ctx", "pfh_state", "pfh_last_allele_frequency",
                    "pfh_open_colored", "pfh_num_colors", "pfh_ploidy_estimation_colored",
                    "pfh_colors_open", "pfh_colors_close", "pfh_colors_count", "pfh_colors_unitigs", "pfh_colors_name",
                    "pfh_colors_unitig", "pfh_bifrost_kmer_hash", "pfh_gfa_abundant_kmers", "pfh_gfa_write_unitig_ids", "pfh_gfa_write_unitig_ids_given_inputs", "pfh_gfa_numbering_replays", "pfh_gfa_minimizer_counts", "pfh_host_walk", "pfh_host_walk_range", "pfh_replay_open", "pfh_replay_close", "pfh_replay_apply", "pfh_replay_state", "pfh_replay_apply_parallel", "pfh_side_components", "pfh_replay_check_footprints", "pfh_colors_check_footprints",
                    "pfh_find_shard", "pfh_shard_records", "pfh_shard_pool", "pfh_find_replay", "pfh_set_replay_threads", "pfh_set_write_super_bubble", "pfh_ploidy_select", "pfh_ploidy_select_colored", "pfh_ploidy_align", "pfh_ploidy_text", "pfh_ploidy_write",
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
    L.pfh_set_align_pieces.argtypes = [vp, C.c_uint64]
    L.pfh_set_reference_threads.argtypes = [vp, C.c_uint32]
    L.pfh_set_overlap_output.argtypes = [vp, C.c_int]
    L.pfh_set_third_tier_on_host.argtypes = [vp, C.c_int]
    L.pfh_set_unitig_id.argtypes = [vp, C.c_char_p]
    L.pfh_find_superbubbles.argtypes = [vp, C.c_char_p]
    L.pfh_ploidy_estimation.argtypes = [vp, C.c_char_p, C.c_int, C.c_int]
    L.pfh_get_times.argtypes = [vp, C.POINTER(Times)]
    L.pfh_r_format_double.restype = C.c_uint64
    L.pfh_r_format_double.argtypes = [C.c_double, C.c_char_p, C.c_uint64]
    L.pfh_load_trace.restype = C.c_uint64
    L.pfh_load_trace.argtypes = [C.c_char_p, C.c_uint64, C.c_int]
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
    L.pfh_gfa_write_unitig_ids_given_inputs.restype = C.c_int
    L.pfh_gfa_write_unitig_ids_given_inputs.argtypes = [C.c_char_p, C.c_char_p, C.c_int]
    u32, u64 = C.c_uint32, C.c_uint64
    L.pfh_find_shard.argtypes = [vp, u32, u32]
    L.pfh_shard_records.restype = vp
    L.pfh_shard_records.argtypes = [vp, C.POINTER(u64)]
    L.pfh_shard_pool.restype = vp
    L.pfh_shard_pool.argtypes = [vp, C.POINTER(u64)]
    L.pfh_find_replay.argtypes = [vp, C.c_char_p, u32, vp, vp, vp, C.c_int, vp, vp, vp]
    L.pfh_set_replay_threads.argtypes = [vp, C.c_int]
    L.pfh_set_replay_threads.restype = None
    L.pfh_set_write_super_bubble.argtypes = [vp, C.c_int]
    L.pfh_set_write_super_bubble.restype = None
    L.pfh_ploidy_select.argtypes = [vp, C.c_int, C.c_int, C.POINTER(u64)]
    L.pfh_ploidy_select_colored.argtypes = [vp, vp, vp, C.c_int, C.POINTER(u64)]
    L.pfh_ploidy_align.argtypes = [vp, u64, u64, C.POINTER(u64)]
    L.pfh_ploidy_text.argtypes = [vp, u64, vp, vp]
    L.pfh_ploidy_write.argtypes = [vp, C.c_char_p, vp, vp, C.c_int]
    L.pfh_host_walk_range.restype = u64
    L.pfh_host_walk_range.argtypes = [vp, vp, u32, u32, u32, vp, u64, vp, u64, C.POINTER(u64)]
    L.pfh_replay_open.restype = vp
    L.pfh_replay_open.argtypes = [u32, u32]
    L.pfh_replay_close.argtypes = [vp]
    L.pfh_replay_apply.argtypes = [vp, vp, u64, vp]
    L.pfh_replay_apply_parallel.argtypes = [vp, vp, u64, vp, C.c_uint32]
    L.pfh_replay_apply_parallel.restype = C.c_int
    L.pfh_side_components.argtypes = [vp, u64, vp, C.c_uint32, vp]
    L.pfh_side_components.restype = None
    L.pfh_replay_check_footprints.argtypes = [vp, u64, vp, C.c_uint32, C.c_uint32, u64, C.POINTER(u64)]
    L.pfh_replay_check_footprints.restype = u64
    L.pfh_colors_check_footprints.argtypes = [vp, vp, vp, u64, vp, C.c_uint32, u64, C.POINTER(u64)]
    L.pfh_colors_check_footprints.restype = u64
    L.pfh_replay_state.argtypes = [vp, vp, vp, vp]
    L.pfh_bifrost_kmer_hash.restype = C.c_uint64
    L.pfh_bifrost_kmer_hash.argtypes = [C.c_uint64, C.c_uint64]
    _lib = L
    return L


def host_walk_range(succ: np.ndarray, pred: np.ndarray, u0: int, u1: int):
    """Records + vertex pool of every candidate entrance on unitigs [u0, u1) from the host walker alone (no device)."""
    L = load_library()
    succ = np.ascontiguousarray(succ, dtype=np.uint32).reshape(-1, 4)
    pred = np.ascontiguousarray(pred, dtype=np.uint32).reshape(-1, 4)
    n = succ.shape[0] // 2
    used = C.c_uint64()
    cnt = int(((succ[2 * u0: 2 * u1] != hipapi.NONE).sum(axis=1) > 1).sum())
    rec = np.zeros(max(cnt, 1), dtype=hipapi.BFS_RECORD)
    cap = 64 * max(cnt, 1) + 1024
    while True:
        pool = np.zeros(cap, dtype=np.uint32)
        got = L.pfh_host_walk_range(succ.ctypes.data, pred.ctypes.data, n, u0, u1, rec.ctypes.data, len(rec), pool.ctypes.data, cap, C.byref(used))
        if got != 0xFFFFFFFFFFFFFFFF:
            return rec[:got].copy(), pool[: used.value].copy()
        if used.value <= cap:
            raise RuntimeError("pfh_host_walk_range failed")
        cap = used.value + 1024


def side_components(records: np.ndarray, pool: np.ndarray, n_unitigs: int) -> np.ndarray:
    """component label of every record's entrance side (host union-find over the records' footprints)"""
    L = load_library()
    records = np.ascontiguousarray(records)
    pool = np.ascontiguousarray(pool, dtype=np.uint32) if len(pool) else np.zeros(1, dtype=np.uint32)
    labels = np.zeros(len(records), dtype=np.uint32)
    L.pfh_side_components(records.ctypes.data if len(records) else None, len(records), pool.ctypes.data, n_unitigs, labels.ctypes.data)
    return labels


def check_footprints(records: np.ndarray, pool: np.ndarray, n_unitigs: int, complex_size: int = 8, slice_len: int = 0):
    """(accesses outside the running record's component, index of the first such record or None)"""
    L = load_library()
    records = np.ascontiguousarray(records)
    pool = np.ascontiguousarray(pool, dtype=np.uint32) if len(pool) else np.zeros(1, dtype=np.uint32)
    first = C.c_uint64()
    bad = L.pfh_replay_check_footprints(records.ctypes.data if len(records) else None, len(records), pool.ctypes.data, n_unitigs,
                                        complex_size, slice_len, C.byref(first))
    return int(bad), (None if first.value == 0xFFFFFFFFFFFFFFFF else int(first.value))


class Replay:
    """The commit replay on a bare MyUnitig state (no device): shards of records in, state out."""

    def __init__(self, n_unitigs: int, complex_size: int = 8):
        self.L = load_library()
        self.n = n_unitigs
        self.h = self.L.pfh_replay_open(n_unitigs, complex_size)

    def apply(self, records: np.ndarray, pool: np.ndarray, threads: int = 0):
        """threads > 0: the parallel replay (components of record footprints side by side); 0: the sequential one"""
        records = np.ascontiguousarray(records)
        pool = np.ascontiguousarray(pool, dtype=np.uint32) if len(pool) else np.zeros(1, dtype=np.uint32)
        if threads:
            rc = self.L.pfh_replay_apply_parallel(self.h, records.ctypes.data if len(records) else None, len(records), pool.ctypes.data, threads)
        else:
            rc = self.L.pfh_replay_apply(self.h, records.ctypes.data if len(records) else None, len(records), pool.ctypes.data)
        if rc:
            raise RuntimeError("pfh_replay_apply: %d (shards must arrive in entrance order)" % rc)

    def state(self):
        f = np.empty(self.n, dtype=np.uint8)
        p = np.empty(self.n, dtype=np.uint32)
        m = np.empty(self.n, dtype=np.uint32)
        self.L.pfh_replay_state(self.h, f.ctypes.data, p.ctypes.data, m.ctypes.data)
        return f, p, m

    def close(self):
        if getattr(self, "h", None):
            self.L.pfh_replay_close(self.h)
            self.h = None


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

    def check_footprints(self, succ: np.ndarray, records: np.ndarray, pool: np.ndarray, complex_size: int = 8, slice_len: int = 0):
        """the footprint check of the parallel commits with the colored gate: (accesses outside the component, first such record)"""
        succ = np.ascontiguousarray(succ, dtype=np.uint32)
        records = np.ascontiguousarray(records)
        pool = np.ascontiguousarray(pool, dtype=np.uint32) if len(pool) else np.zeros(1, dtype=np.uint32)
        first = C.c_uint64()
        bad = self.L.pfh_colors_check_footprints(self.h, succ.ctypes.data, records.ctypes.data if len(records) else None, len(records), pool.ctypes.data,
                                                 complex_size, slice_len, C.byref(first))
        return int(bad), (None if first.value == 0xFFFFFFFFFFFFFFFF else int(first.value))

    def unitig(self, u: int):
        """(presence[colour, kmer] uint8, UnitigColors::size(um), n_full_enc)"""
        km, nf = C.c_uint32(), C.c_uint32()
        self.L.pfh_colors_unitig(self.h, u, None, C.byref(km), C.byref(nf))
        out = np.zeros((self.n_colors, km.value), dtype=np.uint8)
        sz = self.L.pfh_colors_unitig(self.h, u, out.ctypes.data, C.byref(km), C.byref(nf))
        return out, sz, nf.value


def load_trace(reset: bool = True) -> list:
    """[(step, seconds)] of the loads of this process since the last reset (pfh_load_trace)"""
    L = load_library()
    n = L.pfh_load_trace(None, 0, 0)
    buf = C.create_string_buffer(int(n) + 1)
    L.pfh_load_trace(buf, n + 1, int(reset))
    out = []
    for ln in buf.value.decode().splitlines():
        a, b = ln.rsplit("\t", 1)
        out.append((a, float(b)))
    return out


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

    def set_align_pieces(self, n: int):
        self.L.pfh_set_align_pieces(self.h, n)

    def set_reference_threads(self, n: int):
        """n > 1: the text format of the reference's `-t n` run (deterministic row order)"""
        self._check(self.L.pfh_set_reference_threads(self.h, n))

    def set_overlap_output(self, on: bool):
        self.L.pfh_set_overlap_output(self.h, int(on))

    def set_third_tier_on_host(self, on: bool):
        self.L.pfh_set_third_tier_on_host(self.h, int(on))

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

    # ---- one graph over several GPUs (include/ploidyfrost_host.h; the exchange itself lives in ploidyfrost_amd/dist.py) ----
    def find_shard(self, u0: int, u1: int):
        """K-BFS + host walkers of the entrances on unitigs [u0, u1): (records as BFS_RECORD array, pool u32) copies"""
        self._check(self.L.pfh_find_shard(self.h, u0, u1))
        n = C.c_uint64()
        p = self.L.pfh_shard_records(self.h, C.byref(n))
        rec = np.ctypeslib.as_array(C.cast(p, C.POINTER(C.c_uint8)), shape=(n.value * hipapi.BFS_RECORD.itemsize,)).copy().view(hipapi.BFS_RECORD) \
            if n.value else np.zeros(0, dtype=hipapi.BFS_RECORD)
        p = self.L.pfh_shard_pool(self.h, C.byref(n))
        pool = np.ctypeslib.as_array(C.cast(p, C.POINTER(C.c_uint32)), shape=(n.value,)).copy() if n.value else np.zeros(0, dtype=np.uint32)
        return rec, pool

    def find_replay(self, outpre: str, records: list, pools: list, write_file: bool = True, dev_records: list | None = None,
                    dev_pools: list | None = None):
        """dev_records / dev_pools: device addresses (ints) of the same shards where they already lie in device memory"""
        if records is None:   # device arrays only: (n_records, pool_len) per shard in `pools`
            n = len(pools)
            rp = pp = None
            nr = (C.c_uint64 * n)(*[int(a) for a, _ in pools])
            pl = (C.c_uint64 * n)(*[int(b) for _, b in pools])
        else:
            recs = [np.ascontiguousarray(r) for r in records]
            pls = [np.ascontiguousarray(p, dtype=np.uint32) if len(p) else np.zeros(1, dtype=np.uint32) for p in pools]
            n = len(recs)
            rp = (C.c_void_p * n)(*[r.ctypes.data if len(r) else None for r in recs])
            pp = (C.c_void_p * n)(*[p.ctypes.data for p in pls])
            nr = (C.c_uint64 * n)(*[len(r) for r in recs])
            pl = (C.c_uint64 * n)(*[len(p) for p in pools])
        dr = (C.c_void_p * n)(*[int(a) if a else None for a in dev_records]) if dev_records else None
        dp = (C.c_void_p * n)(*[int(a) if a else None for a in dev_pools]) if dev_pools else None
        self._check(self.L.pfh_find_replay(self.h, outpre.encode(), n, rp, nr, pp, int(write_file), pl, dr, dp))

    def set_write_super_bubble(self, on: bool):
        """several ranks running findSuperBubble on one graph into one directory: off on all but the rank that writes the file"""
        self.L.pfh_set_write_super_bubble(self.h, int(on))

    def set_replay_threads(self, threads: int):
        """host threads of the commit replay: 0 = sequential, -1 = default"""
        self.L.pfh_set_replay_threads(self.h, threads)

    def ploidy_select(self, lower: int, upper: int) -> int:
        n = C.c_uint64()
        self._check(self.L.pfh_ploidy_select(self.h, lower, upper, C.byref(n)))
        return n.value

    def ploidy_align(self, t0: int, t1: int) -> int:
        n = C.c_uint64()
        self._check(self.L.pfh_ploidy_align(self.h, t0, t1, C.byref(n)))
        return n.value

    def ploidy_text(self, var_count_base: int):
        """-> (sizes of the ten slabs, [sites with 2..5 alleles, coreCov, coreNum, bubbles called, bubbles])"""
        sizes = np.zeros(10, dtype=np.uint64)
        counters = np.zeros(8, dtype=np.uint64)
        self._check(self.L.pfh_ploidy_text(self.h, var_count_base, sizes.ctypes.data, counters.ctypes.data))
        return sizes, counters

    def ploidy_write(self, outpre: str, offsets, totals, truncate: bool = True):
        o = np.ascontiguousarray(offsets, dtype=np.uint64)
        t = np.ascontiguousarray(totals, dtype=np.uint64)
        self._check(self.L.pfh_ploidy_write(self.h, outpre.encode(), o.ctypes.data, t.ctypes.data, int(truncate)))

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

    def ploidy_select(self, cutoffs) -> int:
        """scan + sequential pass with one (lower, upper) per colour; then ploidy_align / ploidy_text / ploidy_write as for Run"""
        lo = (C.c_int * len(cutoffs))(*[int(c[0]) for c in cutoffs])
        up = (C.c_int * len(cutoffs))(*[int(c[1]) for c in cutoffs])
        n = C.c_uint64()
        self._check(self.L.pfh_ploidy_select_colored(self.h, C.cast(lo, C.c_void_p), C.cast(up, C.c_void_p), len(cutoffs), C.byref(n)))
        return n.value


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
