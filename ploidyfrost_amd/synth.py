"""Synthetic inputs for the PloidyFrost hot path.

Nothing here is product code: it only manufactures inputs -- polyploid
haplotypes, the KMC count database that ``kmc -k25 -ci1 -cs10000`` would
produce for them (KMC1 on-disk layout, the one parsed by
KMC/kmc_api/kmc_file.cpp:246-299 of the reference), and, through
``ploidyfrost_amd.cdbg_build``, a compacted de Bruijn graph in Bifrost's
GFA dialect -- for tests, fixtures and bench.py.

All generators are seeded and deterministic.
"""
from __future__ import annotations

import os
import struct
from dataclasses import dataclass

import numpy as np

BASES = np.frombuffer(b"ACGT", dtype=np.uint8)
_M64 = (1 << 64) - 1


# --------------------------------------------------------------------------
# haplotypes
# --------------------------------------------------------------------------
@dataclass
class HapSpec:
    genome_len: int
    ploidy: int = 2
    seed: int = 1
    gap_lo: int = 15
    gap_hi: int = 900
    p_snp: float = 0.75
    p_del: float = 0.13          # 1-bp deletion
    max_ins: int = 6             # 1..max_ins bp insertion for the remainder
    p_multi: float = 0.03        # SNP sites carrying two different ALT bases (ploidy >= 3)


def make_haplotypes(spec: HapSpec, base_edit=None) -> list[np.ndarray]:
    """Base genome + ``ploidy`` haplotypes as uint8 arrays of 2-bit codes (A0 C1 G2 T3).  ``base_edit(base) -> base``
    may rewrite the base genome (same length) before the variants are laid on it."""
    rng = np.random.default_rng(spec.seed)
    L = spec.genome_len
    base = rng.integers(0, 4, size=L, dtype=np.uint8)
    if base_edit is not None:
        base = base_edit(base)
    # variant positions
    n_max = L // spec.gap_lo + 2
    gaps = rng.integers(spec.gap_lo, spec.gap_hi + 1, size=n_max)
    pos = np.cumsum(gaps)
    pos = pos[pos < L - 64]
    pos = pos[pos > 64]
    nv = len(pos)
    kind_r = rng.random(nv)
    kind = np.where(kind_r < spec.p_snp, 0, np.where(kind_r < spec.p_snp + spec.p_del, 1, 2)).astype(np.int8)
    ins_len = rng.integers(1, spec.max_ins + 1, size=nv)
    ins_seq = rng.integers(0, 4, size=(nv, spec.max_ins), dtype=np.uint8)
    alt_shift = rng.integers(1, 4, size=nv, dtype=np.uint8)
    multi = (rng.random(nv) < spec.p_multi) & (spec.ploidy >= 3)
    # carrier mask: non-empty proper subset of haplotypes
    full = (1 << spec.ploidy) - 1
    carriers = rng.integers(1, full, size=nv) if full > 1 else np.ones(nv, dtype=np.int64)
    # multi-allelic SNP sites: every haplotype draws its own allele (up to 4 alleles)
    multi_shift = rng.integers(0, 4, size=(spec.ploidy, nv), dtype=np.uint8)

    haps = []
    for h in range(spec.ploidy):
        seq = base.copy()
        has = ((carriers >> h) & 1).astype(bool)
        # SNPs
        m = has & (kind == 0) & ~multi
        seq[pos[m]] = (base[pos[m]] + alt_shift[m]) & 3
        m2 = multi & (kind == 0)
        if m2.any():
            seq[pos[m2]] = (base[pos[m2]] + multi_shift[h][m2]) & 3
        keep = np.ones(L, dtype=bool)
        md = has & (kind == 1)
        keep[pos[md]] = False
        mi = np.nonzero(has & (kind == 2))[0]
        if len(mi):
            ins_at = np.repeat(pos[mi], ins_len[mi])
            ins_vals = np.concatenate([ins_seq[i, : ins_len[i]] for i in mi])
        else:
            ins_at = np.zeros(0, dtype=np.int64)
            ins_vals = np.zeros(0, dtype=np.uint8)
        # apply deletions then insertions (positions refer to the base genome)
        # map base coordinates -> coordinates after deletions
        newidx = np.cumsum(keep) - keep  # index of each kept base in the deleted sequence
        seq_d = seq[keep]
        if len(ins_at):
            seq_d = np.insert(seq_d, newidx[ins_at], ins_vals)
        haps.append(seq_d.astype(np.uint8))
    return haps


def write_fasta(path: str, haps: list[np.ndarray], width: int = 0) -> None:
    with open(path, "wb") as f:
        for i, h in enumerate(haps):
            f.write(b">hap%d\n" % i)
            f.write(BASES[h].tobytes())
            f.write(b"\n")


# --------------------------------------------------------------------------
# k-mers
# --------------------------------------------------------------------------
def kmers_u64(seq: np.ndarray, k: int) -> tuple[np.ndarray, np.ndarray]:
    """Forward and reverse-complement k-mers (2 bits/base, first base most significant)."""
    n = len(seq) - k + 1
    if n <= 0:
        z = np.zeros(0, dtype=np.uint64)
        return z, z
    s = seq.astype(np.uint64)
    fw = np.zeros(n, dtype=np.uint64)
    rc = np.zeros(n, dtype=np.uint64)
    for j in range(k):
        fw |= s[j : j + n] << np.uint64(2 * (k - 1 - j))
        rc |= (np.uint64(3) - s[j : j + n]) << np.uint64(2 * j)
    return fw, rc


def canonical_counts(haps: list[np.ndarray], k: int) -> tuple[np.ndarray, np.ndarray]:
    """Sorted distinct canonical k-mers and their multiplicity over all haplotypes."""
    parts = []
    for h in haps:
        fw, rc = kmers_u64(h, k)
        parts.append(np.minimum(fw, rc))
    allk = np.concatenate(parts)
    return np.unique(allk, return_counts=True)


def mix64(x: np.ndarray) -> np.ndarray:
    """splitmix64 finaliser -- the deterministic 'hash of the k-mer' used for count jitter."""
    x = x.astype(np.uint64).copy()
    with np.errstate(over="ignore"):
        x ^= x >> np.uint64(30)
        x *= np.uint64(0xBF58476D1CE4E5B9)
        x ^= x >> np.uint64(27)
        x *= np.uint64(0x94D049BB133111EB)
        x ^= x >> np.uint64(31)
    return x


def synth_counts(kmers: np.ndarray, mult: np.ndarray, depth: int = 20, jitter: int = 3) -> np.ndarray:
    """count = multiplicity*depth + jitter in [-jitter, +jitter] (hash of the k-mer), clamped to >= 1."""
    j = (mix64(kmers) % np.uint64(2 * jitter + 1)).astype(np.int64) - jitter
    c = mult.astype(np.int64) * depth + j
    return np.maximum(c, 1).astype(np.uint32)


def revcomp_u64(kmers: np.ndarray, k: int) -> np.ndarray:
    """reverse complement of 2-bit packed k-mers (first base most significant)"""
    x = ~kmers.astype(np.uint64) & np.uint64((1 << (2 * k)) - 1)
    out = np.zeros_like(x)
    for _ in range(k):
        out = (out << np.uint64(2)) | (x & np.uint64(3))
        x = x >> np.uint64(2)
    return out


def stranded_counts(kmers: np.ndarray, mult: np.ndarray, k: int, depth: int = 20, jitter: int = 3):
    """A database counted without canonical merging (kmc -b): every k-mer in both orientations, each with its own count
    (multiplicity * depth/2 plus a jitter that depends on the oriented k-mer).  Returns (sorted distinct k-mers, counts)."""
    both = np.concatenate([kmers, revcomp_u64(kmers, k)])
    m = np.concatenate([mult, mult])
    both, idx = np.unique(both, return_index=True)
    return both, synth_counts(both, m[idx], depth=max(1, depth // 2), jitter=jitter)


# --------------------------------------------------------------------------
# KMC1 database writer
# --------------------------------------------------------------------------
def lut_prefix_len(k: int) -> int:
    """A prefix length p with (k-p) % 4 == 0 (kmc_file.cpp:283 sufix_size = (k-p)/4)."""
    for p in (5, 6, 7, 4, 3, 2, 1, 8):
        if p < k and (k - p) % 4 == 0:
            return p
    raise ValueError(k)


def write_kmc1(prefix: str, kmers: np.ndarray, counts: np.ndarray, k: int, *, counter_size: int = 2,
               min_count: int = 1, max_count: int = 65535, both_strands: bool = True,
               p: int | None = None) -> None:
    """Write ``prefix.kmc_pre`` / ``prefix.kmc_suf`` in the KMC1 layout.

    ``kmers`` must be sorted ascending (2-bit, first base most significant) and distinct.
    Layout as parsed by the reference reader (KMC/kmc_api/kmc_file.cpp:140-179, 246-299):
      .kmc_pre = 'KMCP' | u64 LUT[4^p] | u64 sentinel(=total) | header(7 x u64) | u32 header_offset | 'KMCP'
      .kmc_suf = 'KMCS' | total x { (k-p)/4 suffix bytes MSB-first | counter_size bytes LE } | 'KMCS'
    The u64 after the LUT is the word the reader indexes as LUT[4^p]; emitting it keeps
    the last prefix range inside the suffix array.
    """
    if p is None:
        p = lut_prefix_len(k)
    assert (k - p) % 4 == 0
    total = len(kmers)
    kmers = kmers.astype(np.uint64)
    suf_bytes = (k - p) // 4
    pre = (kmers >> np.uint64(2 * (k - p))).astype(np.int64)
    lut = np.searchsorted(pre, np.arange(4 ** p, dtype=np.int64), side="left").astype(np.uint64)
    header = np.zeros(7, dtype=np.uint64)
    header[0] = k | (0 << 32)                       # kmer_length | mode<<32
    header[1] = counter_size | (p << 32)            # counter_size | lut_prefix_length<<32
    header[2] = min_count | ((max_count & 0xFFFFFFFF) << 32)
    header[3] = total
    header[4] = 0 if both_strands else 1            # low nibble 1 => "not both strands"
    header[5] = 0
    header[6] = 0                                    # trailing u32 = version 0 (KMC1)
    header_offset = 7 * 8
    with open(prefix + ".kmc_pre", "wb") as f:
        f.write(b"KMCP")
        f.write(lut.tobytes())
        f.write(struct.pack("<Q", total))
        # the header must sit at (size - header_offset)/8 where size excludes markers and the offset word
        hb = header.tobytes()
        # version word = the u32 at EOF-12, i.e. the last 4 bytes of the header block
        f.write(hb)
        f.write(struct.pack("<I", header_offset))
        f.write(b"KMCP")
    rec = np.zeros((total, suf_bytes + counter_size), dtype=np.uint8)
    suf = kmers & np.uint64((1 << (2 * (k - p))) - 1)
    for b in range(suf_bytes):
        rec[:, b] = ((suf >> np.uint64(8 * (suf_bytes - 1 - b))) & np.uint64(0xFF)).astype(np.uint8)
    c = counts.astype(np.uint64)
    for b in range(counter_size):
        rec[:, suf_bytes + b] = ((c >> np.uint64(8 * b)) & np.uint64(0xFF)).astype(np.uint8)
    with open(prefix + ".kmc_suf", "wb") as f:
        f.write(b"KMCS")
        f.write(rec.tobytes())
        f.write(b"KMCS")


def make_dataset(outdir: str, name: str, spec: HapSpec, k: int = 25, depth: int = 20):
    """haplotype FASTA + KMC1 database for ``spec``; returns (fasta_path, kmc_prefix, haps)."""
    os.makedirs(outdir, exist_ok=True)
    haps = make_haplotypes(spec)
    fa = os.path.join(outdir, name + ".fa")
    write_fasta(fa, haps)
    km, mult = canonical_counts(haps, k)
    cnt = synth_counts(km, mult, depth=depth)
    db = os.path.join(outdir, name + "_kmc")
    write_kmc1(db, km, cnt, k)
    return fa, db, haps


def read_kmc1(prefix: str):
    """Inverse of write_kmc1 (KMC1 layout only): returns (kmers u64 sorted, counts u32, meta dict)."""
    pre = np.fromfile(prefix + ".kmc_pre", dtype=np.uint8)
    suf = np.fromfile(prefix + ".kmc_suf", dtype=np.uint8)
    assert pre[:4].tobytes() == b"KMCP" and pre[-4:].tobytes() == b"KMCP" and suf[:4].tobytes() == b"KMCS"
    assert int(pre[-12:-8].view("<u4")[0]) == 0, "KMC1 layout expected"
    header_offset = int(pre[-8])
    body = pre[4:-8]
    words = body[: (len(body) // 8) * 8].view("<u8")
    hi = (len(body) - header_offset) // 8
    k = int(words[hi] & 0xFFFFFFFF)
    counter_size = int(words[hi + 1] & 0xFFFFFFFF)
    p = int(words[hi + 1] >> np.uint64(32))
    min_count = int(words[hi + 2] & 0xFFFFFFFF)
    max_count = int(words[hi + 2] >> np.uint64(32))
    total = int(words[hi + 3])
    both = (int(words[hi + 4]) & 0xF) != 1
    lut = words[: 4 ** p + 1].astype(np.int64).copy()
    lut[4 ** p] = total
    sb = (k - p) // 4
    rec = suf[4 : 4 + total * (sb + counter_size)].reshape(total, sb + counter_size)
    sfx = np.zeros(total, dtype=np.uint64)
    for b in range(sb):
        sfx = (sfx << np.uint64(8)) | rec[:, b].astype(np.uint64)
    cnt = np.zeros(total, dtype=np.uint64)
    for b in range(counter_size):
        cnt |= rec[:, sb + b].astype(np.uint64) << np.uint64(8 * b)
    prefix_of = np.repeat(np.arange(4 ** p, dtype=np.uint64), np.diff(lut))
    kmers = (prefix_of << np.uint64(2 * (k - p))) | sfx
    return kmers, cnt.astype(np.uint32), dict(k=k, p=p, min_count=min_count, max_count=max_count, both_strands=both,
                                              total=total)


# --------------------------------------------------------------------------
# KMC2 ("0x200") layout: signature-binned prefix table, what `kmc` >= 2 really writes
# --------------------------------------------------------------------------
def kmc_norm_table(sig_len: int) -> np.ndarray:
    """norm[m] = min(allowed(m), allowed(revcomp(m))), disallowed -> 4^len  (KMC/kmc_api/mmer.h:34-87)."""
    n = 1 << (2 * sig_len)
    m = np.arange(n, dtype=np.uint32)

    def allowed(x):
        ok = np.ones(n, dtype=bool)
        ok &= (x & 0x3F) != 0x3F          # TTT suffix
        ok &= (x & 0x3F) != 0x3B          # TGT suffix
        ok &= (x & 0x3C) != 0x3C          # TG* suffix
        y = x.copy()
        for _ in range(sig_len - 3):
            ok &= (y & 0xF) != 0          # AA inside
            y = y >> 2
        ok &= y != 0                      # AAA prefix
        ok &= y != 0x04                   # ACA prefix
        ok &= (y & 0xF) != 0              # *AA prefix
        return ok

    rev = np.zeros(n, dtype=np.uint32)
    t = m.copy()
    for i in range(sig_len):
        rev |= (np.uint32(3) - (t & np.uint32(3))) << np.uint32(2 * (sig_len - 1 - i))
        t = t >> np.uint32(2)
    special = np.uint32(n)
    a = np.where(allowed(m), m, special)
    b = np.where(allowed(rev), rev, special)
    return np.minimum(a, b).astype(np.uint32)


def kmc_signatures(kmers: np.ndarray, k: int, sig_len: int) -> np.ndarray:
    """CKmerAPI::get_signature (kmer_api.h:653-673): min over the k-sig_len+1 windows of norm[window]."""
    norm = kmc_norm_table(sig_len)
    mask = np.uint64((1 << (2 * sig_len)) - 1)
    best = np.full(len(kmers), 1 << (2 * sig_len), dtype=np.uint32)
    for i in range(k - sig_len + 1):
        w = ((kmers >> np.uint64(2 * (k - sig_len - i))) & mask).astype(np.int64)
        best = np.minimum(best, norm[w])
    return best


def write_kmc2(prefix: str, kmers: np.ndarray, counts: np.ndarray, k: int, *, counter_size: int = 2, sig_len: int = 9,
               n_bins: int = 37, min_count: int = 1, max_count: int = 65535, both_strands: bool = True,
               p: int | None = None) -> None:
    """Write the KMC2 layout parsed by kmc_file.cpp:196-245:
      .kmc_pre = 'KMCP' | u64 LUT[n_bins*4^p + 1] | u32 signature_map[4^sig_len + 1] | header(64 B) | u32 64 | 'KMCP'
      .kmc_suf = 'KMCS' | records grouped by bin, sorted inside each bin | 'KMCS'
    header = u32 x7 {k, mode, counter_size, p, sig_len, min_count, max_count}, u64 total, u8 !both_strands, padding,
    u32 version 0x200 as its last word."""
    if p is None:
        p = lut_prefix_len(k)
    total = len(kmers)
    kmers = kmers.astype(np.uint64)
    sig = kmc_signatures(kmers, k, sig_len)
    n_sig = (1 << (2 * sig_len)) + 1
    sig_map = (mix64(np.arange(n_sig, dtype=np.uint64)) % np.uint64(n_bins)).astype(np.uint32)
    bins = sig_map[sig]
    order = np.lexsort((kmers, bins))
    kb, cb, bb = kmers[order], counts[order], bins[order]
    pre = (kb >> np.uint64(2 * (k - p))).astype(np.int64)
    key = bb.astype(np.int64) * (4 ** p) + pre
    lut = np.searchsorted(key, np.arange(n_bins * 4 ** p + 1, dtype=np.int64), side="left").astype(np.uint64)
    header = struct.pack("<7I", k, 0, counter_size, p, sig_len, min_count, max_count & 0xFFFFFFFF) + struct.pack("<Q", total) + \
        struct.pack("<B", 0 if both_strands else 1)
    header += b"\0" * (60 - len(header)) + struct.pack("<I", 0x200)
    with open(prefix + ".kmc_pre", "wb") as f:
        f.write(b"KMCP")
        f.write(lut.tobytes())
        f.write(sig_map.tobytes())
        f.write(header)
        f.write(struct.pack("<I", 64))
        f.write(b"KMCP")
    suf_bytes = (k - p) // 4
    rec = np.zeros((total, suf_bytes + counter_size), dtype=np.uint8)
    suf = kb & np.uint64((1 << (2 * (k - p))) - 1)
    for b in range(suf_bytes):
        rec[:, b] = ((suf >> np.uint64(8 * (suf_bytes - 1 - b))) & np.uint64(0xFF)).astype(np.uint8)
    c = cb.astype(np.uint64)
    for b in range(counter_size):
        rec[:, suf_bytes + b] = ((c >> np.uint64(8 * b)) & np.uint64(0xFF)).astype(np.uint8)
    with open(prefix + ".kmc_suf", "wb") as f:
        f.write(b"KMCS")
        f.write(rec.tobytes())
        f.write(b"KMCS")


def read_kmc(prefix: str):
    """(kmers u64 in file order, counts u32, meta) for a KMC1- or KMC2-layout database."""
    pre = np.fromfile(prefix + ".kmc_pre", dtype=np.uint8)
    version = int(pre[-12:-8].view("<u4")[0])
    if version == 0:
        return read_kmc1(prefix)
    assert version == 0x200, "unknown KMC version"
    suf = np.fromfile(prefix + ".kmc_suf", dtype=np.uint8)
    header_offset = int(pre[-8])
    h = pre[len(pre) - header_offset - 8 :]
    k, mode, counter_size, p, sig_len, min_count, max_count = (int(x) for x in h[:28].view("<u4"))
    total = int(h[28:36].view("<u8")[0])
    both = not bool(h[36])
    size = len(pre) - 12
    sig_bytes = (4 ** sig_len + 1) * 4
    n_lut = (size - sig_bytes - header_offset) // 8
    lut = pre[4 : 4 + 8 * n_lut].view("<u8").astype(np.int64).copy()
    lut[-1] = total
    n_pref = 4 ** p
    sb = (k - p) // 4
    rec = suf[4 : 4 + total * (sb + counter_size)].reshape(total, sb + counter_size)
    sfx = np.zeros(total, dtype=np.uint64)
    for b in range(sb):
        sfx = (sfx << np.uint64(8)) | rec[:, b].astype(np.uint64)
    cnt = np.zeros(total, dtype=np.uint64)
    for b in range(counter_size):
        cnt |= rec[:, sb + b].astype(np.uint64) << np.uint64(8 * b)
    prefix_of = np.repeat((np.arange(n_lut - 1, dtype=np.uint64) % np.uint64(n_pref)), np.diff(lut))
    kmers = (prefix_of << np.uint64(2 * (k - p))) | sfx
    return kmers, cnt.astype(np.uint32), dict(k=k, p=p, min_count=min_count, max_count=max_count, both_strands=both,
                                              total=total, layout="kmc2", sig_len=sig_len)


def bifrost_minimizer_hash(s: bytes) -> int:
    """Bifrost's minimizer hash of a g-mer (bifrost/src/RepHash.hpp:24-103: two rolling words, one per strand, through
    wyhash) -- generators use it to plant a g-mer that wins the minimizer race in the k-mers around it."""
    M = (1 << 64) - 1
    hv = [2053695854357871005, 5073395517033431291, 10060236952204337488, 7783083932390163561]
    g, h, ht = len(s), 0, 0
    for i in range(g):
        h = (((h << 1) | (h >> 63)) & M) ^ hv[(s[i] & 6) >> 1]
        ht = (((ht << 1) | (ht >> 63)) & M) ^ hv[((s[g - 1 - i] ^ 4) & 6) >> 1]
    lo, hi = min(h, ht), max(h, ht)
    a = ((lo & 0xFFFFFFFF) << 32) | (hi & 0xFFFFFFFF)
    b = ((hi >> 32) << 32) | (lo >> 32)

    def mix(x, y):
        r = (x & M) * (y & M)
        return (r & M) ^ (r >> 64)
    return mix(0xE7037ED1A0B428DB ^ 16, mix(a ^ 0xE7037ED1A0B428DB, b ^ 0xA0761D6478BD642F))


def plant_crowded_minimizer(rng, base, g, copies, tries=3000):
    """Overwrites `copies` places of the genome `base` (uint8 codes) with one g-mer whose minimizer hash is the lowest of
    `tries` random ones: the k-mers around every copy share that minimizer, which crowds its bucket in Bifrost's index."""
    cands = [rng.integers(0, 4, size=g, dtype=np.uint8) for _ in range(tries)]
    core = min(cands, key=lambda c: bifrost_minimizer_hash(BASES[c].tobytes()))
    out = base.copy()
    step = len(out) // (copies + 1)
    for i in range(copies):
        at = (i + 1) * step + int(rng.integers(-step // 4, step // 4))
        out[at: at + g] = core
    return out


# --------------------------------------------------------------------------
# repeat-rich genomes (the stress workload of bench.py --workload repeats)
# --------------------------------------------------------------------------
def repeat_rich_edit(seed: int, frac: float = 0.08, families: int = 50, fam_len=(300, 3000), divergence=(0.01, 0.03),
                     p_inverted: float = 0.25, p_tandem: float = 0.15):
    """A ``base_edit`` for make_haplotypes: about ``frac`` of the genome is overwritten with copies of ``families`` repeat
    families (consensus of fam_len bases, every copy with its own 1-3 % of substitutions), a quarter of the copies reverse
    complemented (inverted repeats: hairpins when two lie close), some as tandem arrays of a 20-200 bp unit (cycles).  What a
    random genome lacks: traversals that do not close for thousands of unitigs, components of thousands of commit records."""
    def edit(base: np.ndarray) -> np.ndarray:
        rng = np.random.default_rng(seed)
        L = len(base)
        out = base.copy()
        cons = [rng.integers(0, 4, size=int(rng.integers(fam_len[0], fam_len[1] + 1)), dtype=np.uint8) for _ in range(families)]
        target, placed = int(frac * L), 0
        while placed < target:
            copy = cons[int(rng.integers(0, families))].copy()
            hit = rng.random(len(copy)) < rng.uniform(*divergence)
            copy[hit] = (copy[hit] + rng.integers(1, 4, size=int(hit.sum())).astype(np.uint8)) & 3
            kind = rng.random()
            if kind < p_inverted:
                copy = (3 - copy)[::-1].copy()
            elif kind < p_inverted + p_tandem:
                copy = np.tile(copy[: int(rng.integers(20, 201))], int(rng.integers(2, 7)))
            if len(copy) + 4000 >= L:
                break
            at = int(rng.integers(2000, L - len(copy) - 2000))
            out[at: at + len(copy)] = copy
            placed += len(copy)
        return out
    return edit
