"""Writer of Bifrost ``.bfg_colors`` files for synthetic colored graphs (inputs for tests and bench.py).

Like ``synth.py`` this is an input generator, not product code: it lays colour sets out the way
``DataStorage<U>::write`` does (bifrost/src/DataStorage.tcc:545-787, format version 2) so that both the
reference's reader (``ColoredCDBG::read``) and the product's own reader
(``csrc/host/pf_host_colors.cpp``) can be fed graphs of any size without running ``Bifrost build -c``,
and it can emit *every* colour-set encoding ``UnitigColors::write`` has (bifrost/src/ColorSet.cpp:1174-1226)
-- including the ones a small ``Bifrost build`` never produces (Roaring bitmaps, the {full colours, rest}
pair that ``optimizeFullColors`` creates on > 1 GiB colourings).  ``tests/test_colors_cpu.py`` checks, where the
real Bifrost has been built, that the real library reads these files as intended.
"""
from __future__ import annotations

import struct

import numpy as np

_M64 = (1 << 64) - 1
_WYP0, _WYP1 = 0xA0761D6478BD642F, 0xE7037ED1A0B428DB

ENCODINGS = ("auto", "bitvector", "single", "tiny_bmp", "tiny_list", "tiny_rle", "roaring_array", "roaring_bitset",
             "roaring_run", "pair")


def _wymix(a: int, b: int) -> int:
    r = (a & _M64) * (b & _M64)
    return (r & _M64) ^ (r >> 64)


def kmer_hash(left_aligned_kmer: int, seed: int) -> int:
    """Kmer::hash(seed) for MAX_KMER_SIZE=32: wyhash final v3 of the 8 little-endian bytes (bifrost/src/Kmer.hpp:120)."""
    x = int(left_aligned_kmer) & _M64
    lo, hi = x & 0xFFFFFFFF, x >> 32
    a, b = (lo << 32) | hi, (hi << 32) | lo
    return _wymix(_WYP1 ^ 8, _wymix(a ^ _WYP1, b ^ ((seed ^ _WYP0) & _M64)))


def _mul128(a: np.ndarray, b: np.ndarray) -> tuple[np.ndarray, np.ndarray]:
    """(lo, hi) of the 128-bit product of two uint64 arrays."""
    m32 = np.uint64(0xFFFFFFFF)
    s32 = np.uint64(32)
    al, ah, bl, bh = a & m32, a >> s32, b & m32, b >> s32
    with np.errstate(over="ignore"):
        ll, lh, hl, hh = al * bl, al * bh, ah * bl, ah * bh
        mid = (ll >> s32) + (lh & m32) + (hl & m32)
        lo = (ll & m32) | (mid << s32)
        hi = hh + (lh >> s32) + (hl >> s32) + (mid >> s32)
    return lo, hi


def kmer_hash_np(x: np.ndarray, seed: int) -> np.ndarray:
    x = x.astype(np.uint64)
    lo, hi = x & np.uint64(0xFFFFFFFF), x >> np.uint64(32)
    a, b = (lo << np.uint64(32)) | hi, (hi << np.uint64(32)) | lo
    l1, h1 = _mul128(a ^ np.uint64(_WYP1), b ^ np.uint64((seed ^ _WYP0) & _M64))
    l2, h2 = _mul128(np.full_like(x, _WYP1 ^ 8), l1 ^ h1)
    return l2 ^ h2


def left_align(kmers: np.ndarray, k: int) -> np.ndarray:
    """right-aligned 2-bit k-mers (synth.kmers_u64) -> Bifrost's Kmer layout (first base in the top bits)."""
    return kmers.astype(np.uint64) << np.uint64(64 - 2 * k)


# ---- colour-set encoders: ids = sorted list of (colour, k-mer) pair ids colour * n_kmers + pos ----------
def _runs(ids):
    out = []
    for v in ids:
        if out and out[-1][1] + 1 == v:
            out[-1][1] = v
        else:
            out.append([v, v])
    return out


def _enc_bitvector(ids):
    bits = 0
    for v in ids:
        bits |= 1 << v
    return struct.pack("<Q", (bits << 3) | 0x1)


def _enc_single(ids):
    return struct.pack("<Q", (ids[0] << 3) | 0x2)


def _enc_tiny(ids, mode):
    if not ids:
        return struct.pack("<QH", 0x0, 0)  # header only, size 0 (TinyBitmap.cpp:831-835)
    offset = ids[0] >> 16
    low = [v & 0xFFFF for v in ids]
    if mode == "tiny_bmp":
        nw = (low[-1] >> 4) + 1
        words = [0] * nw
        for v in low:
            words[v >> 4] |= 1 << (v & 15)
        body, card, m = words, len(ids), 0x0
    elif mode == "tiny_list":
        body, card, m = low, len(ids), 0x2
    else:
        body = [x & 0xFFFF for r in _runs(ids) for x in r]
        card, m = len(body), 0x4
    sz = len(body) + 3
    return struct.pack("<Q", 0x0) + struct.pack("<%dH" % sz, (sz << 3) | m, card, offset, *body)


def _enc_roaring(ids, kind):
    """Roaring portable serialisation; ``kind`` picks the container type used for every 2^16 block."""
    blocks = {}
    for v in ids:
        blocks.setdefault(v >> 16, []).append(v & 0xFFFF)
    keys = sorted(blocks)
    n = len(keys)
    run_flags = bytearray((n + 7) // 8)
    payload = []
    for i, key in enumerate(keys):
        vals = blocks[key]
        if kind == "roaring_run":
            run_flags[i // 8] |= 1 << (i % 8)
            rr = _runs(vals)
            payload.append(struct.pack("<H", len(rr)) + b"".join(struct.pack("<HH", a, b - a) for a, b in rr))
        elif kind == "roaring_bitset" or len(vals) > 4096:
            words = np.zeros(1024, dtype=np.uint64)
            for v in vals:
                words[v >> 6] |= np.uint64(1) << np.uint64(v & 63)
            payload.append(words.tobytes())
        else:
            payload.append(struct.pack("<%dH" % len(vals), *vals))
    hasrun = kind == "roaring_run"
    if hasrun:
        head = struct.pack("<I", 12347 | ((n - 1) << 16)) + bytes(run_flags)
    else:
        head = struct.pack("<II", 12346, n)
    desc = b"".join(struct.pack("<HH", key, len(blocks[key]) - 1) for key in keys)
    # a bitset container is only legal above 4096 values: pad the request down to an array otherwise
    offsets = b""
    if not hasrun or n >= 4:
        pos = len(head) + len(desc) + 4 * n
        offs = []
        for pl in payload:
            offs.append(pos)
            pos += len(pl)
        offsets = struct.pack("<%dI" % n, *offs)
    blob = head + desc + offsets + b"".join(payload)
    return struct.pack("<Q", (len(blob) << 3) | 0x3) + blob


def encode_set(ids: list[int], encoding: str = "auto", *, n_kmers: int = 0, n_colors: int = 0) -> bytes:
    """One UnitigColors as ``UnitigColors::write`` lays it out.  Falls back to a legal encoding when the requested
    one cannot hold ``ids`` (e.g. a 61-bit vector for ids >= 61)."""
    if encoding == "pair":
        per = {}
        for v in ids:
            per[v // n_kmers] = per.get(v // n_kmers, 0) + 1
        full = sorted(c for c, cnt in per.items() if cnt == n_kmers)
        fs = set(full)
        rest = [v for v in ids if (v // n_kmers) not in fs]
        return struct.pack("<Q", 0x4) + encode_set(full, "auto") + encode_set(rest, "auto")
    one_block = bool(ids) and (ids[0] >> 16) == (ids[-1] >> 16)
    if encoding == "auto":
        if ids and ids[-1] < 61:
            encoding = "single" if len(ids) == 1 else "bitvector"
        elif not ids:
            encoding = "bitvector"
        elif one_block and len(_runs(ids)) * 2 + 3 <= 4096:
            encoding = "tiny_rle"
        else:
            encoding = "roaring_run"
    if encoding == "single" and len(ids) != 1:
        encoding = "bitvector"
    if encoding == "bitvector" and ids and ids[-1] >= 61:
        encoding = "tiny_rle"
    if encoding.startswith("tiny"):
        body = {"tiny_bmp": ((ids[-1] & 0xFFFF) >> 4) + 1 if ids else 0, "tiny_list": len(ids),
                "tiny_rle": 2 * len(_runs(ids))}[encoding]
        if ids and (not one_block or body + 3 > 4096):
            encoding = "roaring_run"
    if encoding == "roaring_bitset" and ids:
        # legal only for blocks above 4096 values
        blocks = {}
        for v in ids:
            blocks[v >> 16] = blocks.get(v >> 16, 0) + 1
        if min(blocks.values()) <= 4096:
            encoding = "roaring_array"
    if encoding == "bitvector":
        return _enc_bitvector(ids)
    if encoding == "single":
        return _enc_single(ids)
    if encoding.startswith("tiny"):
        return _enc_tiny(ids, encoding)
    if not ids:
        return _enc_bitvector(ids)
    return _enc_roaring(ids, encoding)


def pair_ids(n_kmers: int, full_colors, partial: dict | None = None) -> list[int]:
    """ids of a unitig whose colours in ``full_colors`` cover every k-mer; ``partial`` maps colour -> 0/1 array."""
    ids = []
    partial = partial or {}
    for c in sorted(set(full_colors) | set(partial)):
        if c in partial:
            ids.extend(int(c * n_kmers + p) for p in np.nonzero(np.asarray(partial[c]))[0])
        else:
            ids.extend(range(c * n_kmers, (c + 1) * n_kmers))
    return ids


def write_bfg_colors(path: str, heads: np.ndarray, sizes_bp: np.ndarray, k: int, names: list[str], sets: list[bytes] | None = None, *,
                     full_mask: np.ndarray | None = None, partial_ids: dict | None = None, nb_seeds: int = 31, seed: int = 12345, overflow_every: int = 0,
                     slack: float = 1.0, shared_sets: list[tuple[bytes, int]] | None = None, shared_refs: set | None = None) -> np.ndarray:
    """Write the colour file of a graph whose unitig u has head k-mer ``heads[u]`` (Bifrost layout, see
    ``left_align``) and ``sizes_bp[u]`` bases.  Colour sets come either encoded (``sets[u]`` from ``encode_set``) or, for
    large graphs, as ``full_mask[u]`` (bit c = colour c on every k-mer; encoded with the natural encoding) plus
    ``partial_ids[u]`` = the complete sorted id list of the few unitigs that carry a colour on part of their k-mers.
    ``shared_sets``: (encoded set, reference count) pairs written as the file's SharedUnitigColors section
    (DataStorage.tcc:619-627, 747-755: between the link bits and the colour sets; block positions for them come first);
    ``shared_refs``: unitigs whose colour set is written the way UnitigColors::write writes a reference to a shared set -- the flag
    word 0x5 alone (ColorSet.cpp:1190-1194: which set it referred to is not in the file).
    Returns the ``DA:Z:`` tag of every unitig (0 = overflow table) for the GFA segment lines."""
    n = len(heads)
    shared_sets = shared_sets or []
    shared_refs = shared_refs or set()
    heads = heads.astype(np.uint64)
    sizes_bp = np.asarray(sizes_bp, dtype=np.uint64)
    rng = np.random.default_rng(seed)
    seeds = rng.integers(1, 1 << 63, size=nb_seeds, dtype=np.uint64)
    nb_cs = max(1, int(n * slack))
    slot = np.full(n, -1, dtype=np.int64)
    da = np.zeros(n, dtype=np.int16)
    taken = np.zeros(nb_cs, dtype=bool)
    todo = np.arange(n)
    if overflow_every:
        todo = todo[(todo % overflow_every) != overflow_every - 1]
    for si in range(nb_seeds):
        if len(todo) == 0:
            break
        h = (kmer_hash_np(heads[todo], int(seeds[si])) % np.uint64(nb_cs)).astype(np.int64)
        free = ~taken[h]
        # first claimant of every free slot wins
        order = np.argsort(h, kind="stable")
        hs = h[order]
        first = np.ones(len(hs), dtype=bool)
        first[1:] = hs[1:] != hs[:-1]
        win = np.zeros(len(h), dtype=bool)
        win[order[first]] = True
        win &= free
        slot[todo[win]] = h[win]
        da[todo[win]] = si + 1
        taken[h[win]] = True
        todo = todo[~win]
    over = np.nonzero(slot < 0)[0]
    sz_cs = nb_cs + len(over)
    slot[over] = nb_cs + np.arange(len(over))
    owner = np.full(sz_cs, -1, dtype=np.int64)
    owner[slot] = np.arange(n)
    C = len(names)
    if sets is None:
        km = (sizes_bp - np.uint64(k) + np.uint64(1)).astype(np.int64)
        cache = {}

        partial_ids = partial_ids or {}

        def natural(u):
            if u in partial_ids:
                return encode_set(partial_ids[u], "auto")
            key = (int(km[u]), int(full_mask[u]))
            b = cache.get(key)
            if b is None:
                b = encode_set(pair_ids(key[0], [c for c in range(C) if (key[1] >> c) & 1]), "auto")
                cache[key] = b
            return b
    empty = _enc_bitvector([])
    block_sz = 1024
    n_shared = len(shared_sets)
    n_pos = (n_shared // block_sz + (n_shared % block_sz != 0)) + (sz_cs // block_sz + (sz_cs % block_sz != 0))
    with open(path, "wb") as f:
        f.write(struct.pack("<7Q", 2, nb_seeds, C, nb_cs, sz_cs, n_shared, len(over)))
        f.write(seeds.tobytes())
        f.write(struct.pack("<Q", block_sz))
        pos_at = f.tell()
        f.write(b"\0" * (16 * n_pos))
        for nm in names:
            f.write(nm.encode() + b"\n")
        link = np.zeros((sz_cs >> 6) + ((sz_cs & 63) != 0), dtype=np.uint64)
        occ = np.nonzero(owner >= 0)[0]
        np.bitwise_or.at(link, occ >> 6, np.uint64(1) << (occ & 63).astype(np.uint64))
        f.write(link.tobytes())
        positions = []
        buf = []
        at = f.tell()
        for i, (enc, refs) in enumerate(shared_sets):
            if i % block_sz == 0:
                positions.append(at)
            b = enc + struct.pack("<Q", refs)
            buf.append(b)
            at += len(b)
        for i in range(sz_cs):
            if i % block_sz == 0:
                positions.append(at)
            u = owner[i]
            b = empty if u < 0 else struct.pack("<Q", 5) if u in shared_refs else (sets[u] if sets is not None else natural(u))
            buf.append(b)
            at += len(b)
            if len(buf) >= 65536:
                f.write(b"".join(buf))
                buf = []
        f.write(b"".join(buf))
        for u in over:
            f.write(struct.pack("<QQQ", int(heads[u]), int(sizes_bp[u]), int(slot[u])))
        f.seek(pos_at)
        for p in positions:
            f.write(struct.pack("<qq", p, 0))  # std::streampos = {off_t, mbstate_t}
    return da
