"""ctypes binding of the gfx950 device layer (include/ploidyfrost_hip.h).

The library is the product; this module is plumbing so that Python callers (bench.py, the
tests, torch.distributed launchers) can drive it.  torch is imported first on purpose: PyTorch
ships its own libamdhip64.so.7, and loading it first makes this library bind to the same HIP
runtime, so torch tensors and torch streams can be handed straight to the C ABI.

There is no CPU fallback: a missing library or a missing GPU raises.
"""
from __future__ import annotations

import ctypes as C
import os

import numpy as np

PKG = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(PKG, "csrc", "libploidyfrost_hip.so")

NONE = 0xFFFFFFFF
PF_OK, PF_ERR_ARG, PF_ERR_HIP, PF_ERR_NO_DEVICE, PF_ERR_OVERFLOW, PF_ERR_MISSING_KMER = range(6)
KERNELS = ["k_table_build", "k_adj_insert", "k_adj_probe", "k_cov", "k_bfs", "k_bfs_big", "k_align", "k_align_big",
           "k_strcov", "k_bubble", "k_bubble_big", "k_cov_colored", "k_strcov_colored", "k_gmm", "k_kmc_decode", "k_minz_count", "k_cov_join",
           "k_call_sides", "k_call_prep", "k_call_paths", "k_call_sites", "k_call_format", "k_call_snp", "k_bfs_thread", "k_call_pair", "k_call_stack", "k_cov_join_rest", "copy_text_to_host"]

BFS_RECORD = np.dtype([("entrance", "<u4"), ("exit", "<u4"), ("n_seen", "<u4"), ("n_list", "<u4"), ("list_off", "<u8"),
                       ("outcome", "u1"), ("flag_cycle", "u1"), ("flag_tip", "u1"), ("strict", "u1"), ("pad", "<u4")])
ALIGN_JOB = np.dtype([("a_off", "<u8"), ("b_off", "<u8"), ("a_len", "<u4"), ("b_len", "<u4")])
ALIGN_HIT = np.dtype([("text_off", "<u8"), ("gap_off", "<u8"), ("len", "<u4"), ("n_gaps", "<u4"), ("score", "<i8"),
                      ("n_pos", "<u4"), ("n_indel", "<u4")])
CALL_SIDE = np.dtype([("u", "<u4"), ("exit_ov", "<u4"), ("err_unitig", "<u4"), ("plus_side", "u1"), ("kind", "u1"), ("aligned", "u1"),
                      ("err", "u1")])
CALL_RESULT = np.dtype([("text_len", "<u8", (10,)), ("allele", "<u8", (4,)), ("core_cov", "<u8"), ("core_num", "<u8"), ("n_called", "<u8"),
                        ("align_jobs", "<u8"), ("site_strings", "<u8"), ("n_branching", "<u8"),
                        ("snp_jobs", "<u8"), ("pair_jobs", "<u8"), ("wave_jobs", "<u8"), ("stack_jobs", "<u8"),
                        ("alignseq_packed_len", "<u8"), ("numeric_packed", "<u8")])
CALL_STREAMS = ["allele_frequency", "alignseq", "bifre", "trifre", "tetrafre", "pentafre", "bicov", "tricov", "tetracov", "pentacov"]
BUBBLE_PATH = np.dtype([("text_off", "<u8"), ("len", "<u4"), ("ov", "<u4")])
BUBBLE_TASK = np.dtype([("path_first", "<u8"), ("n_paths", "<u4"), ("pad", "<u4")])
BUBBLE_SITE = np.dtype([("col", "<u4"), ("is_indel", "u1"), ("maxnum", "u1"), ("pad", "<u2")])
BUBBLE_RESULT = np.dtype([("rows_off", "<u8"), ("site_off", "<u8"), ("group_off", "<u8"), ("ilen_off", "<u8"), ("n_rows", "<u4"),
                          ("n_cols", "<u4"), ("n_sites", "<u4"), ("n_indel_len", "<u4")])
CALL_BUBBLE = np.dtype([("entrance_ov", "<u4"), ("exit_ov", "<u4"), ("strict", "<u4"), ("n_inner", "<u4"), ("inner", "<u4", (4,)),
                        ("cov", "<f8", (4,)), ("core_mean", "<f8"), ("cov_sum", "<f8")])
assert CALL_BUBBLE.itemsize == 80
assert BFS_RECORD.itemsize == 32 and ALIGN_JOB.itemsize == 24 and ALIGN_HIT.itemsize == 40
assert BUBBLE_PATH.itemsize == 16 and BUBBLE_TASK.itemsize == 16 and BUBBLE_SITE.itemsize == 8 and BUBBLE_RESULT.itemsize == 48

_lib = None


class DeviceError(RuntimeError):
    def __init__(self, status: int, msg: str):
        super().__init__("ploidyfrost_hip status %d: %s" % (status, msg))
        self.status = status


def load_library() -> C.CDLL:
    """Load libploidyfrost_hip.so (fails loudly when it has not been built)."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise RuntimeError("%s is missing: run `python -c 'import __graft_entry__ as g; g.build()'` "
                           "(there is no CPU fallback for the device layer)" % LIB_PATH)
    try:
        import torch  # noqa: F401  (binds libamdhip64.so.7 first, see module docstring)
    except Exception:  # pragma: no cover - torch is optional for pure ctypes use
        pass
    L = C.CDLL(LIB_PATH)
    vp, u32, u64, i = C.c_void_p, C.c_uint32, C.c_uint64, C.c_int
    sig = {
        "pf_create": (i, [i, C.POINTER(vp)]),
        "pf_warmup": (i, [i]),
        "pf_destroy": (None, [vp]),
        "pf_last_error": (C.c_char_p, [vp]),
        "pf_set_stream": (i, [vp, vp]),
        "pf_synchronize": (i, [vp]),
        "pf_enable_timing": (i, [vp, i]),
        "pf_kernel_time": (i, [vp, i, C.POINTER(C.c_double), C.POINTER(u64)]),
        "pf_reset_timing": (i, [vp]),
        "pf_device_busy": (i, [vp, vp, vp]),
        "pf_kernel_units": (i, [vp, i, C.POINTER(u64)]),
        "pf_side_components": (i, [vp, i, vp, u64, vp, u64, vp, u64, vp, u64]),
        "pf_bfs_candidates_begin": (i, [vp, C.c_uint32, C.c_uint32, vp, u64, vp, u64, C.POINTER(u64), C.POINTER(u64), vp, vp, u64, C.POINTER(u64)]),
        "pf_bfs_candidates_end": (i, [vp]),
        "pf_fetch": (i, [vp, vp, vp, u64]),
        "pf_bfs_candidates_resident": (i, [vp, C.c_uint32, C.c_uint32, C.POINTER(u64), C.POINTER(u64), vp, vp, u64, C.POINTER(u64)]),
        "pf_replay_device": (i, [vp, C.c_uint32, C.c_uint32, C.POINTER(u64), C.POINTER(u64)]),
        "pf_replay_big_fetch": (i, [vp, vp, vp, vp]),
        "pf_replay_set_colours": (i, [vp, C.c_uint32, vp, vp, vp]),
        "pf_replay_finish": (i, [vp, vp, vp, vp, u64]),
        "pf_call_get_state": (i, [vp, vp, vp, vp]),
        "pf_gfa_ingest": (i, [vp, vp, u64, i, i, C.POINTER(C.c_uint32), C.POINTER(C.c_uint32)]),
        "pf_gfa_parse": (i, [vp, vp, u64, i, i, C.POINTER(C.c_uint32), C.POINTER(C.c_uint32)]),
        "pf_gfa_upload": (i, [vp]),
        "pf_gfa_error": (C.c_char_p, [vp]),
        "pf_gfa_segments": (i, [vp, vp, vp, vp, vp, vp, C.POINTER(C.c_int)]),
        "pf_replay_order": (i, [vp, C.c_uint32, vp, vp, vp]),
        "pf_kernel_name": (C.c_char_p, [i]),
        "pf_upload_graph": (i, [vp, vp, vp, vp, u32, i]),
        "pf_build_adjacency": (i, [vp, vp, vp]),
        "pf_upload_counts": (i, [vp, vp, vp, u64, u32, u64, u64, i]),
        "pf_join_counts": (i, [vp]),
        "pf_join_counts_begin": (i, [vp]),
        "pf_join_counts_end": (i, [vp]),
        "pf_lookup_kmers": (i, [vp, vp, u64, vp, vp]),
        "pf_unitig_cov": (i, [vp, u32, u32, vp, vp, vp]),
        "pf_unitig_cov_probe": (i, [vp, u32, u32, vp, vp, vp]),
        "pf_count_candidates": (i, [vp, u32, u32, C.POINTER(u64)]),
        "pf_bfs_candidates": (i, [vp, u32, u32, vp, u64, vp, u64, C.POINTER(u64), C.POINTER(u64)]),
        "pf_align_batch": (i, [vp, vp, u64, vp, u32, C.c_double, C.c_double, C.c_double, vp, vp, vp, u64, vp, u64, vp, u64,
                               C.POINTER(u64)]),
        "pf_align_bubbles": (i, [vp, vp, u64, vp, u64, vp, u32, C.c_double, C.c_double, C.c_double, vp, vp, u64, vp, u64, vp, u64,
                                 vp, u64, C.POINTER(u64)]),
        "pf_string_cov": (i, [vp, vp, vp, u32, u32, u32, vp, vp, vp]),
        "pf_host_alloc": (i, [vp, C.c_size_t, C.POINTER(vp)]),
        "pf_selftest_scan": (i, [vp, u64, C.c_uint32]),
        "pf_call_set_numeric_packed": (i, [vp, i]),
        "pf_call_fetch_text": (i, [vp, i, i, vp, u64]),
        "pf_host_free": (None, [vp, vp]),
        "pf_device_name": (i, [vp, C.c_char_p, C.c_size_t]),
        "pf_device_pci_bus_id": (i, [vp, C.c_char_p, C.c_size_t]),
        "pf_table_capacity": (u64, [vp]),
        "pf_num_kmers": (u64, [vp]),
        "pf_upload_counts_colored": (i, [vp, u32, vp, vp, vp, vp, vp, vp]),
        "pf_num_colors": (u32, [vp]),
        "pf_unitig_cov_colored": (i, [vp, u32, u32, vp, vp, vp, vp]),
        "pf_unitig_cov_colored_probe": (i, [vp, u32, u32, vp, vp, vp, vp]),
        "pf_string_cov_colored": (i, [vp, vp, vp, u32, vp, vp, vp, vp]),
        "pf_unitig_cov_exact": (i, [vp, u32, u32, i, vp, vp, vp]),
        "pf_minimizer_table_slots": (u64, [u64]),
        "pf_minimizer_crowding": (i, [vp, i, u32, vp, vp, vp]),
        "pf_minimizer_replay_inputs": (i, [vp, i, C.c_uint32, vp, vp]),
        "pf_kmc_decode": (i, [vp, vp, u64, u32, u32, vp, u64, u32, u32, vp, vp]),
        "pf_device_free": (None, [vp, vp]),
        "pf_copy_to_host": (i, [vp, vp, vp, C.c_size_t]),
        "pf_gmm_upload": (i, [vp, vp, u64]),
        "pf_gmm_count": (u64, [vp]),
        "pf_gmm_fit": (i, [vp, u32, C.c_double, C.c_double, C.c_int32, C.c_double, vp, vp, vp, vp, vp]),
        "pf_call_set_state": (i, [vp, vp, vp, vp]),
        "pf_call_coverage": (i, [vp]),
        "pf_call_set_format": (i, [vp, i]),
        "pf_superbubble_rows": (i, [vp, i, C.POINTER(u64), C.POINTER(u64)]),
        "pf_superbubble_fetch": (i, [vp, vp, u64]),
        "pf_call_scan": (i, [vp, u32, u32, C.POINTER(u64)]),
        "pf_call_sides": (i, [vp, vp, u64]),
        "pf_call_resolve": (i, [vp, C.POINTER(u64), C.POINTER(u32), C.POINTER(u32)]),
        "pf_call_select": (i, [vp, vp, u64]),
        "pf_call_run": (i, [vp, i, u64, u64, u64, u32, C.c_double, C.c_double, C.c_double, vp]),
        "pf_call_align": (i, [vp, u64, u64, u32, C.c_double, C.c_double, C.c_double, vp]),
        "pf_call_text": (i, [vp, i, u64, vp]),
        "pf_call_set_alignseq_packed": (i, [vp, i]),
        "pf_comm_unique_id": (i, [vp]),
        "pf_comm_init": (i, [vp, vp, i, i]),
        "pf_gather": (i, [vp, vp, C.c_uint32, vp]),
        "pf_comm_destroy": (None, [vp]),
        "pf_call_peek": (i, [vp, i, vp, vp, vp, u64, vp, vp, vp, vp, vp, vp, vp]),
        "pf_call_reserve": (i, [vp, u64, C.c_uint32]),
        "pf_call_reserve_lanes": (i, [vp, u64, C.c_uint32, C.c_int]),
        "pf_timing_select": (i, [vp, u64]),
        "pf_kernel_busy": (i, [vp, u64, C.POINTER(C.c_double)]),
        "pf_call_reserve_text": (i, [vp, u64]),
        "pf_find_reserve": (i, [vp, u64]),
        "pf_call_set_colours": (i, [vp, u32, vp, vp, vp, vp, vp, vp, u64, u64]),
        "pf_call_set_cutoffs": (i, [vp, u32, vp, vp]),
        "pf_call_text_range": (i, [vp, i, u64, u64, u64, vp]),
        "pf_call_align_lane": (i, [vp, i, u64, u64, u32, C.c_double, C.c_double, C.c_double, vp]),
        "pf_call_text_range_lane": (i, [vp, i, i, u64, u64, u64, vp]),
        "pf_call_text_sizes": (i, [vp, i, u64, u64, u64, vp]),
        "pf_bfs_live_deferred": (i, [vp, u64, vp]),
        "pf_bfs_live_count": (i, [vp, vp]),
        "pf_call_fetch": (i, [vp, i, i, vp, u64]),
        "pf_call_fetch_slab": (i, [vp, i, vp, vp]),
        "pf_call_fetch_range": (i, [vp, i, u64, vp, u64, i]),
        "pf_call_fetch_wait": (i, [vp, i]),
        "pf_format_doubles": (i, [vp, vp, u64, vp, vp]),
    }
    for name, (res, args) in sig.items():
        fn = getattr(L, name)  # AttributeError = header / library mismatch
        fn.restype = res
        fn.argtypes = args
    _lib = L
    return L


DECLARED_SYMBOLS = ["pf_create", "pf_warmup", "pf_destroy", "pf_last_error", "pf_set_stream", "pf_synchronize", "pf_enable_timing",
                    "pf_kernel_time", "pf_reset_timing", "pf_device_busy", "pf_kernel_units", "pf_side_components", "pf_replay_order", "pf_bfs_candidates_begin", "pf_bfs_candidates_end", "pf_fetch", "pf_bfs_candidates_resident", "pf_bfs_live_deferred", "pf_bfs_live_count", "pf_replay_device", "pf_replay_big_fetch", "pf_replay_set_colours", "pf_replay_finish", "pf_call_get_state", "pf_gfa_ingest", "pf_gfa_segments", "pf_gfa_parse", "pf_gfa_upload", "pf_gfa_error", "pf_kernel_name", "pf_upload_graph", "pf_build_adjacency",
                    "pf_upload_counts", "pf_join_counts", "pf_join_counts_begin", "pf_join_counts_end", "pf_lookup_kmers", "pf_unitig_cov", "pf_count_candidates", "pf_bfs_candidates",
                    "pf_align_batch", "pf_align_bubbles", "pf_string_cov", "pf_host_alloc", "pf_host_free", "pf_device_name", "pf_device_pci_bus_id", "pf_table_capacity", "pf_num_kmers",
                    "pf_upload_counts_colored", "pf_num_colors", "pf_unitig_cov_colored", "pf_string_cov_colored",
                    "pf_gmm_upload", "pf_gmm_count", "pf_gmm_fit", "pf_kmc_decode", "pf_device_free", "pf_copy_to_host",
                    "pf_minimizer_table_slots", "pf_minimizer_crowding", "pf_minimizer_replay_inputs", "pf_bfs_candidates_split", "pf_unitig_cov_exact", "pf_unitig_cov_probe", "pf_unitig_cov_colored_probe",
                    "pf_call_set_state", "pf_call_set_format", "pf_superbubble_rows", "pf_superbubble_fetch", "pf_call_coverage", "pf_call_scan", "pf_call_sides", "pf_call_resolve", "pf_call_select", "pf_call_run", "pf_call_align",
                    "pf_call_text", "pf_call_set_alignseq_packed", "pf_comm_unique_id", "pf_comm_init", "pf_gather", "pf_comm_destroy", "pf_call_reserve", "pf_call_reserve_lanes", "pf_timing_select", "pf_kernel_busy", "pf_call_reserve_text", "pf_selftest_scan", "pf_call_set_numeric_packed", "pf_call_fetch_text", "pf_find_reserve", "pf_call_set_colours", "pf_call_set_cutoffs", "pf_call_peek", "pf_call_text_range", "pf_call_align_lane", "pf_call_text_range_lane", "pf_call_text_sizes", "pf_call_fetch", "pf_call_fetch_slab", "pf_call_fetch_range", "pf_call_fetch_wait", "pf_format_doubles"]


def call_peek(ctx_handle, lane: int = 0):
    """pf_call_peek on a raw pf_ctx (e.g. hostapi.Run.device_ctx()): what pf_call_align left resident, per bubble a dict with
    entrance_ov / exit_ov / strict / inner / cov, and -- when an alignment survived -- rows, sites [(col, is_indel, maxnum, groups,
    ok)], indel_len and, for a branching bubble, site_cov = per site (group coverages, their sum)."""
    L = load_library()
    h = C.c_void_p(ctx_handle)
    used = (C.c_uint64 * 6)()
    st = L.pf_call_peek(h, lane, None, None, None, 0, None, None, None, None, None, None, used)
    if st != PF_OK:
        raise DeviceError(st, L.pf_last_error(h).decode())
    nb = int(used[0])
    if nb == 0:
        return []
    cap = (C.c_uint64 * 5)(*[int(used[x + 1]) for x in range(5)])
    bub = np.zeros(nb, dtype=CALL_BUBBLE)
    res = np.zeros(nb, dtype=BUBBLE_RESULT)
    sv_off = np.zeros(nb, dtype=np.uint64)
    text = np.zeros(max(1, cap[0]), dtype=np.uint8)
    sites = np.zeros(max(1, cap[1]), dtype=BUBBLE_SITE)
    groups = np.zeros(max(1, cap[2]), dtype=np.uint8)
    ilen = np.zeros(max(1, cap[3]), dtype=np.uint32)
    sv = np.zeros(max(1, cap[4]), dtype=np.float64)
    st = L.pf_call_peek(h, lane, bub.ctypes.data, res.ctypes.data, sv_off.ctypes.data, nb, text.ctypes.data, sites.ctypes.data,
                        groups.ctypes.data, ilen.ctypes.data, sv.ctypes.data, cap, used)
    if st != PF_OK:
        raise DeviceError(st, L.pf_last_error(h).decode())
    tb = text.tobytes()
    out = []
    for j in range(nb):
        b, r = bub[j], res[j]
        d = dict(entrance_ov=int(b["entrance_ov"]), exit_ov=int(b["exit_ov"]), strict=bool(b["strict"]),
                 inner=[int(x) for x in b["inner"][: int(b["n_inner"])]], cov=[float(x) for x in b["cov"][: int(b["n_inner"])]],
                 core_mean=float(b["core_mean"]), cov_sum=float(b["cov_sum"]), rows=None)
        R, Lc = int(r["n_rows"]), int(r["n_cols"])
        if R:
            o = int(r["rows_off"])
            d["rows"] = [tb[o + i * Lc: o + (i + 1) * Lc] for i in range(R)]
            d["sites"] = []
            d["site_cov"] = []
            v = int(sv_off[j])
            for i in range(int(r["n_sites"])):
                srec = sites[int(r["site_off"]) + i]
                g0 = int(r["group_off"]) + i * R
                mx = int(srec["maxnum"])
                d["sites"].append((int(srec["col"]), int(srec["is_indel"]), mx, groups[g0: g0 + R].tolist(), int(srec["pad"])))
                if not d["strict"]:
                    d["site_cov"].append((sv[v: v + mx].tolist(), float(sv[v + mx])))
                    v += mx + 1
            d["indel_len"] = ilen[int(r["ilen_off"]): int(r["ilen_off"]) + int(r["n_indel_len"])].tolist()
        out.append(d)
    return out


def pack_unitigs(seqs: list[bytes]):
    """2-bit pack (layout of pf_upload_graph): returns (words u64, off u64[N+1], len u32[N])."""
    n = len(seqs)
    lens = np.fromiter((len(s) for s in seqs), dtype=np.uint32, count=n)
    nwords = (lens.astype(np.uint64) + 31) // 32
    off = np.zeros(n + 1, dtype=np.uint64)
    np.cumsum(nwords, out=off[1:])
    total = int(off[-1])
    codes = np.zeros(total * 32, dtype=np.uint8)
    lut = np.zeros(256, dtype=np.uint8)
    for ch, v in zip(b"ACGTacgt", [0, 1, 2, 3, 0, 1, 2, 3]):
        lut[ch] = v
    for u, s in enumerate(seqs):
        b = int(off[u]) * 32
        codes[b : b + len(s)] = lut[np.frombuffer(s, dtype=np.uint8)]
    c = codes.reshape(total, 32).astype(np.uint64)
    shifts = (np.uint64(62) - np.arange(32, dtype=np.uint64) * np.uint64(2))
    words = (c << shifts).sum(axis=1, dtype=np.uint64) if total else np.zeros(0, dtype=np.uint64)
    return np.ascontiguousarray(words), off, lens


def _ptr(a):
    """Address of a numpy array or a torch tensor (host or device)."""
    if a is None:
        return None
    if isinstance(a, np.ndarray):
        assert a.flags["C_CONTIGUOUS"]
        return a.ctypes.data
    return a.data_ptr()  # torch.Tensor


class Device:
    """One pf_ctx: graph + count table resident in the HBM of one MI355X."""

    def __init__(self, device: int = 0):
        self.L = load_library()
        h = C.c_void_p()
        st = self.L.pf_create(device, C.byref(h))
        if st != PF_OK:
            raise DeviceError(st, self.L.pf_last_error(None).decode())
        self.h = h
        self.n = 0
        self.k = 0

    def close(self):
        if getattr(self, "h", None):
            self.L.pf_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _check(self, st: int, allow=()):
        if st != PF_OK and st not in allow:
            raise DeviceError(st, self.L.pf_last_error(self.h).decode())
        return st

    @property
    def name(self) -> str:
        buf = C.create_string_buffer(256)
        self._check(self.L.pf_device_name(self.h, buf, 256))
        return buf.value.decode()

    def set_stream(self, stream_ptr: int | None):
        self._check(self.L.pf_set_stream(self.h, stream_ptr))

    def synchronize(self):
        self._check(self.L.pf_synchronize(self.h))

    def enable_timing(self, on=True):
        self._check(self.L.pf_enable_timing(self.h, int(on)))

    def reset_timing(self):
        self._check(self.L.pf_reset_timing(self.h))

    def timing_select(self, kernels=None):
        """pf_timing_select: time the launches of these kernels only (names of KERNELS; None = all)"""
        mask = (1 << 64) - 1 if kernels is None else sum(1 << KERNELS.index(k) for k in kernels)
        self._check(self.L.pf_timing_select(self.h, mask))

    def kernel_times(self) -> dict:
        out = {}
        for i, name in enumerate(KERNELS):
            ms, n = C.c_double(), C.c_uint64()
            self._check(self.L.pf_kernel_time(self.h, i, C.byref(ms), C.byref(n)))
            if n.value:
                out[name] = (ms.value, n.value)
        return out

    def format_doubles(self, values: np.ndarray) -> list[bytes]:
        """printf("%g") of every value, formatted on the device (the result rows' number formatting, pf_format_doubles)"""
        v = np.ascontiguousarray(values, dtype=np.float64)
        text = np.zeros((len(v), 32), dtype=np.uint8)
        ln = np.zeros(len(v), dtype=np.uint8)
        self._check(self.L.pf_format_doubles(self.h, _ptr(v), len(v), _ptr(text), _ptr(ln)))
        return [text[i, :ln[i]].tobytes() for i in range(len(v))]

    # ---- uploads
    def upload_graph(self, words, off, lens, k: int):
        self.n = int(len(lens))
        self.k = k
        self._keep = (words, off, lens)
        self._check(self.L.pf_upload_graph(self.h, _ptr(words), _ptr(off), _ptr(lens), self.n, k))

    def build_adjacency(self, want_host=True):
        if want_host:
            succ = np.empty((self.n * 2, 4), dtype=np.uint32)
            pred = np.empty((self.n * 2, 4), dtype=np.uint32)
            self._check(self.L.pf_build_adjacency(self.h, succ.ctypes.data, pred.ctypes.data))
            return succ, pred
        self._check(self.L.pf_build_adjacency(self.h, None, None))
        return None

    def upload_counts(self, kmers, counts, min_count=1, max_count=0xFFFFFFFF, both_strands=True, k=None):
        """k: the database's k-mer length (default: the uploaded graph's) -- the table is addressed by minimizers of its keys"""
        self.both_strands = bool(both_strands)
        n = int(kmers.shape[0])
        k = int(k if k is not None else getattr(self, "k", 0))
        if not k:
            raise ValueError("upload_counts before upload_graph needs k")
        self._check(self.L.pf_upload_counts(self.h, _ptr(kmers), _ptr(counts), n, k, min_count, max_count, int(both_strands)))

    def join_counts(self):
        """pf_join_counts: K-COV-JOIN once more (every graph k-mer looked up in the count table)"""
        self._check(self.L.pf_join_counts(self.h))

    def minimizer_crowding(self, g: int, limit: int = 15, want_table: bool = False):
        """K-MINZ: (max occurrences of a minimizer slot, slots that reached `limit`[, the u32 counter table])."""
        mx, crowded = C.c_uint32(), C.c_uint64()
        table = np.zeros(self.L.pf_minimizer_table_slots(self.L.pf_num_kmers(self.h)), dtype=np.uint32) if want_table else None
        self._check(self.L.pf_minimizer_crowding(self.h, g, limit, C.byref(mx), C.byref(crowded), _ptr(table) if want_table else None))
        return (mx.value, crowded.value, table) if want_table else (mx.value, crowded.value)

    def kmc_decode(self, records: np.ndarray, n: int, suffix_bytes: int, counter_bytes: int, lut: np.ndarray, p: int, k: int):
        """K-KMC on raw .kmc_suf records (pf_kmc_decode): returns (kmers u64, counts u32) copied back to the host."""
        records = np.ascontiguousarray(records, dtype=np.uint8)
        lut = np.ascontiguousarray(lut, dtype=np.uint64)
        dk, dc = C.c_void_p(), C.c_void_p()
        self._check(self.L.pf_kmc_decode(self.h, _ptr(records), n, suffix_bytes, counter_bytes, _ptr(lut), len(lut) - 1, p, k,
                                         C.byref(dk), C.byref(dc)))
        km, ct = np.zeros(n, dtype=np.uint64), np.zeros(n, dtype=np.uint32)
        try:
            if n:
                self._check(self.L.pf_copy_to_host(self.h, _ptr(km), dk, n * 8))
                self._check(self.L.pf_copy_to_host(self.h, _ptr(ct), dc, n * 4))
        finally:
            self.L.pf_device_free(self.h, dk)
            self.L.pf_device_free(self.h, dc)
        return km, ct

    def lookup(self, kmers: np.ndarray):
        n = len(kmers)
        c = np.zeros(n, dtype=np.uint32)
        f = np.zeros(n, dtype=np.uint8)
        self._check(self.L.pf_lookup_kmers(self.h, _ptr(np.ascontiguousarray(kmers, dtype=np.uint64)), n, _ptr(c), _ptr(f)))
        return c, f

    def upload_counts_colored(self, dbs, min_count=1, max_count=0xFFFFFFFF):
        """dbs: list of (kmers u64, counts u32[, both_strands]) per colour -> one joined table (pf_upload_counts_colored)."""
        nc = len(dbs)
        self.n_colors = nc
        km = [np.ascontiguousarray(d[0], dtype=np.uint64) for d in dbs]
        ct = [np.ascontiguousarray(d[1], dtype=np.uint32) for d in dbs]
        pk = (C.c_void_p * nc)(*[a.ctypes.data for a in km])
        pc = (C.c_void_p * nc)(*[a.ctypes.data for a in ct])
        n = np.array([len(a) for a in km], dtype=np.uint64)
        mn = np.full(nc, min_count, dtype=np.uint64)
        mx = np.full(nc, max_count, dtype=np.uint64)
        both = np.array([int(d[2]) if len(d) > 2 else 1 for d in dbs], dtype=np.int32)
        self._check(self.L.pf_upload_counts_colored(self.h, nc, pk, pc, n.ctypes.data, mn.ctypes.data, mx.ctypes.data, both.ctypes.data))

    def unitig_cov_colored(self, u0=0, u1=None, probe=False):
        """(sum, min, max, miss), each [n_colors, u1 - u0]; probe: pf_unitig_cov_colored_probe (table look-ups at call time)"""
        u1 = self.n if u1 is None else u1
        shape = (self.n_colors, u1 - u0)
        s = np.zeros(shape, dtype=np.uint64)
        lo = np.zeros(shape, dtype=np.uint32)
        hi = np.zeros(shape, dtype=np.uint32)
        x = np.zeros(shape, dtype=np.uint8)
        fn = self.L.pf_unitig_cov_colored_probe if probe else self.L.pf_unitig_cov_colored
        self._check(fn(self.h, u0, u1, _ptr(s), _ptr(lo), _ptr(hi), _ptr(x)))
        return s, lo, hi, x

    def string_cov_colored(self, strings: list[bytes], low, up):
        """(sum, ok), each [n_strings, n_colors]; low / up: one cutoff per colour"""
        n = len(strings)
        off = np.zeros(n + 1, dtype=np.uint64)
        np.cumsum([len(s) for s in strings], out=off[1:])
        text = np.frombuffer(b"".join(strings) + b"\0", dtype=np.uint8).copy()
        lo = np.ascontiguousarray(low, dtype=np.uint32)
        hi = np.ascontiguousarray(up, dtype=np.uint32)
        s = np.zeros((n, self.n_colors), dtype=np.uint64)
        ok = np.zeros((n, self.n_colors), dtype=np.uint8)
        self._check(self.L.pf_string_cov_colored(self.h, text.ctypes.data, off.ctypes.data, n, lo.ctypes.data, hi.ctypes.data,
                                                 s.ctypes.data, ok.ctypes.data))
        return s, ok

    # ---- kernels
    def unitig_cov(self, u0=0, u1=None, out=None):
        u1 = self.n if u1 is None else u1
        if out is None:
            s = np.zeros(u1 - u0, dtype=np.uint64)
            m = np.zeros(u1 - u0, dtype=np.uint32)
            x = np.zeros(u1 - u0, dtype=np.uint8)
        else:
            s, m, x = out
        if getattr(self, "both_strands", True):
            st = self._check(self.L.pf_unitig_cov(self.h, u0, u1, _ptr(s), _ptr(m), _ptr(x)), allow=(PF_ERR_MISSING_KMER,))
        else:  # database without canonical counting: the unitigs as stored
            st = self._check(self.L.pf_unitig_cov_exact(self.h, u0, u1, 0, _ptr(s), _ptr(m), _ptr(x)), allow=(PF_ERR_MISSING_KMER,))
        return s, m, x, st

    def unitig_cov_probe(self, u0=0, u1=None):
        """pf_unitig_cov_probe: C1 with every k-mer looked up in the hash table at call time (never the joined SoA)"""
        u1 = self.n if u1 is None else u1
        s = np.zeros(u1 - u0, dtype=np.uint64)
        m = np.zeros(u1 - u0, dtype=np.uint32)
        x = np.zeros(u1 - u0, dtype=np.uint8)
        st = self._check(self.L.pf_unitig_cov_probe(self.h, u0, u1, _ptr(s), _ptr(m), _ptr(x)), allow=(PF_ERR_MISSING_KMER,))
        return s, m, x, st

    def unitig_cov_exact(self, reverse: bool, u0=0, u1=None):
        """pf_unitig_cov_exact: every k-mer looked up as it reads in the given orientation"""
        u1 = self.n if u1 is None else u1
        s = np.zeros(u1 - u0, dtype=np.uint64)
        m = np.zeros(u1 - u0, dtype=np.uint32)
        x = np.zeros(u1 - u0, dtype=np.uint8)
        st = self._check(self.L.pf_unitig_cov_exact(self.h, u0, u1, int(reverse), _ptr(s), _ptr(m), _ptr(x)), allow=(PF_ERR_MISSING_KMER,))
        return s, m, x, st

    def count_candidates(self, u0=0, u1=None) -> int:
        u1 = self.n if u1 is None else u1
        n = C.c_uint64()
        self._check(self.L.pf_count_candidates(self.h, u0, u1, C.byref(n)))
        return n.value

    def bfs(self, u0=0, u1=None, pool_cap=None):
        u1 = self.n if u1 is None else u1
        n = self.count_candidates(u0, u1)
        rec = np.zeros(max(n, 1), dtype=BFS_RECORD)
        pool_cap = pool_cap or max(1024, n * 8)
        while True:
            pool = np.zeros(pool_cap, dtype=np.uint32)
            nr, used = C.c_uint64(), C.c_uint64()
            st = self.L.pf_bfs_candidates(self.h, u0, u1, rec.ctypes.data, len(rec), pool.ctypes.data, pool_cap, C.byref(nr),
                                          C.byref(used))
            if st == PF_ERR_OVERFLOW and used.value > pool_cap:
                pool_cap = int(used.value)
                continue
            self._check(st)
            return rec[: nr.value], pool[: used.value]

    def side_components(self, records=None, pool=None, n_records=None, reset=True, extra=None, extra_pool=None):
        """K-CC: the records (default: the ones the last bfs() call left on the device) join the union-find over unitig sides"""
        if records is None:
            rp, pp, n, pl = None, None, int(n_records), 0
        else:
            records = np.ascontiguousarray(records)
            pool = np.ascontiguousarray(pool, dtype=np.uint32) if len(pool) else np.zeros(1, dtype=np.uint32)
            rp, pp, n, pl = records.ctypes.data, pool.ctypes.data, len(records), len(pool)
        xr = xp = None
        nx = nxp = 0
        if extra is not None and len(extra):
            extra = np.ascontiguousarray(extra)
            extra_pool = np.ascontiguousarray(extra_pool, dtype=np.uint32) if len(extra_pool) else np.zeros(1, dtype=np.uint32)
            xr, xp, nx, nxp = extra.ctypes.data, extra_pool.ctypes.data, len(extra), len(extra_pool)
        self._check(self.L.pf_side_components(self.h, int(reset), rp, n, pp, pl, xr, nx, xp, nxp))
        self._cc_n = n

    def replay_order(self, n_classes: int = 64):
        """(order, class_off, labels) of the records of the last side_components call"""
        n = self._cc_n
        order = np.zeros(max(n, 1), dtype=np.uint32)
        labels = np.zeros(max(n, 1), dtype=np.uint32)
        off = np.zeros(n_classes + 1, dtype=np.uint32)
        self._check(self.L.pf_replay_order(self.h, n_classes, order.ctypes.data, off.ctypes.data, labels.ctypes.data))
        return order[:n], off, labels[:n]

    def string_cov(self, strings: list[bytes], low: int, up: int):
        n = len(strings)
        off = np.zeros(n + 1, dtype=np.uint64)
        np.cumsum([len(s) for s in strings], out=off[1:])
        text = np.frombuffer(b"".join(strings) + b"\0", dtype=np.uint8).copy()
        s = np.zeros(n, dtype=np.uint64)
        ok = np.zeros(n, dtype=np.uint8)
        miss = np.zeros(n, dtype=np.uint8)
        self._check(self.L.pf_string_cov(self.h, text.ctypes.data, off.ctypes.data, n, low, up, s.ctypes.data, ok.ctypes.data,
                                         miss.ctypes.data))
        return s, ok, miss

    def align_batch(self, pairs: list[tuple[bytes, bytes]], M=2.0, D=-1.0, G=-3.0):
        """Returns, per job, the list of (a_row, b_row, gap_pos, score, n_pos, n_indel)."""
        n = len(pairs)
        jobs = np.zeros(n, dtype=ALIGN_JOB)
        chunks = []
        pos = 0
        for j, (a, b) in enumerate(pairs):
            jobs[j] = (pos, pos + len(a), len(a), len(b))
            chunks += [a, b]
            pos += len(a) + len(b)
        text = np.frombuffer(b"".join(chunks) + b"\0", dtype=np.uint8).copy()
        hit_cap, text_cap, gap_cap = max(64, 4 * n), max(4096, 8 * pos), max(1024, 4 * n * 8)
        while True:
            first = np.zeros(n, dtype=np.uint64)
            count = np.zeros(n, dtype=np.uint32)
            hits = np.zeros(hit_cap, dtype=ALIGN_HIT)
            otext = np.zeros(text_cap, dtype=np.uint8)
            ogap = np.zeros(gap_cap, dtype=np.uint32)
            used = (C.c_uint64 * 3)()
            st = self.L.pf_align_batch(self.h, text.ctypes.data, pos, jobs.ctypes.data, n, M, D, G, first.ctypes.data,
                                       count.ctypes.data, hits.ctypes.data, hit_cap, otext.ctypes.data, text_cap, ogap.ctypes.data, gap_cap, used)
            if st == PF_ERR_OVERFLOW and (used[0] > hit_cap or used[1] > text_cap or used[2] > gap_cap):
                hit_cap, text_cap, gap_cap = max(hit_cap, used[0]), max(text_cap, used[1]), max(gap_cap, used[2])
                continue
            self._check(st)
            break
        out = []
        tb = otext.tobytes()
        for j in range(n):
            lst = []
            for h in hits[int(first[j]) : int(first[j]) + int(count[j])]:
                o, ln = int(h["text_off"]), int(h["len"])
                gp = ogap[int(h["gap_off"]) : int(h["gap_off"]) + int(h["n_gaps"])].copy()
                lst.append((tb[o : o + ln], tb[o + ln : o + 2 * ln], gp, int(h["score"]), int(h["n_pos"]), int(h["n_indel"])))
            out.append(lst)
        return out

    def align_bubbles(self, bubbles: list[list], M=2.0, D=-1.0, G=-3.0):
        """bubbles[t] = list of paths, each either bytes (ASCII) or an int oriented-unitig id.
        Returns per bubble None (no alignment) or dict(rows, sites=[(col, is_indel, groups)], indel_len)."""
        nt = len(bubbles)
        tasks = np.zeros(nt, dtype=BUBBLE_TASK)
        paths = []
        chunks = []
        pos = 0
        lens = self._keep[2]
        for t, b in enumerate(bubbles):
            tasks[t] = (len(paths), len(b), 0)
            for x in b:
                if isinstance(x, (bytes, bytearray)):
                    paths.append((pos, len(x), NONE))
                    chunks.append(bytes(x))
                    pos += len(x)
                else:
                    paths.append((0, int(lens[int(x) >> 1]), int(x)))
        parr = np.array(paths, dtype=BUBBLE_PATH)
        text = np.frombuffer(b"".join(chunks) + b"\0", dtype=np.uint8).copy()
        caps = [max(4096, 4 * int(parr["len"].sum())), max(256, 8 * nt), max(1024, 32 * nt), max(256, 4 * nt)]
        while True:
            res = np.zeros(nt, dtype=BUBBLE_RESULT)
            otext = np.zeros(caps[0], dtype=np.uint8)
            osites = np.zeros(caps[1], dtype=BUBBLE_SITE)
            ogroups = np.zeros(caps[2], dtype=np.uint8)
            oilen = np.zeros(caps[3], dtype=np.uint32)
            used = (C.c_uint64 * 4)()
            st = self.L.pf_align_bubbles(self.h, text.ctypes.data, pos, parr.ctypes.data, len(parr), tasks.ctypes.data, nt, M, D, G,
                                         res.ctypes.data, otext.ctypes.data, caps[0], osites.ctypes.data, caps[1],
                                         ogroups.ctypes.data, caps[2], oilen.ctypes.data, caps[3], used)
            if st == PF_ERR_OVERFLOW and any(used[i] > caps[i] for i in range(4)):
                caps = [max(caps[i], int(used[i])) for i in range(4)]
                continue
            self._check(st)
            break
        out = []
        tb = otext.tobytes()
        for r in res:
            R, L = int(r["n_rows"]), int(r["n_cols"])
            if R == 0:
                out.append(None)
                continue
            o = int(r["rows_off"])
            rows = [tb[o + i * L : o + (i + 1) * L] for i in range(R)]
            sites = []
            for i in range(int(r["n_sites"])):
                srec = osites[int(r["site_off"]) + i]
                g0 = int(r["group_off"]) + i * R
                sites.append((int(srec["col"]), int(srec["is_indel"]), int(srec["maxnum"]), ogroups[g0 : g0 + R].tolist()))
            il = oilen[int(r["ilen_off"]) : int(r["ilen_off"]) + int(r["n_indel_len"])].tolist()
            out.append(dict(rows=rows, sites=sites, indel_len=il))
        return out
