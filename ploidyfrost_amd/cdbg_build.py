"""Compacted de Bruijn graph of a set of haplotypes, built with torch tensor ops (GPU when
present, CPU otherwise).  Input generator for bench.py and the tests -- it stands in for
`Bifrost build -r haps.fa` (which is not part of this repository) when a large graph is needed.

All k-mers of all haplotypes are taken (as `Bifrost build -r` does).  A unitig is a maximal
chain of oriented k-mers v -> w with out-degree(v) == 1, in-degree(w) == 1 and v, w different
k-mers; because every k-mer occurrence in a haplotype is followed by one of its graph
successors, every unitig is a substring of some haplotype, so the chains are found by cutting
the haplotypes at the non-linkable steps and de-duplicating the pieces.  The result is written
in Bifrost's GFA dialect (H line with KL/ML tags, S lines; L lines are ignored by readers of this
path and are not emitted).
"""
from __future__ import annotations

import numpy as np
import torch


def _device(device=None):
    if device is not None:
        return torch.device(device)
    return torch.device("cuda") if torch.cuda.is_available() else torch.device("cpu")


def _kmers(codes: torch.Tensor, k: int):
    """forward / reverse-complement k-mers (int64, first base most significant) of one sequence"""
    n = codes.numel() - k + 1
    c = codes.to(torch.int64)
    fw = torch.zeros(n, dtype=torch.int64, device=codes.device)
    rc = torch.zeros(n, dtype=torch.int64, device=codes.device)
    for j in range(k):
        s = c[j : j + n]
        fw |= s << (2 * (k - 1 - j))
        rc |= (3 - s) << (2 * j)
    return fw, rc


def _rc(x: torch.Tensor, k: int) -> torch.Tensor:
    r = torch.zeros_like(x)
    y = x.clone()
    for _ in range(k):
        r = (r << 2) | (3 - (y & 3))
        y >>= 2
    return r


def _member(K: torch.Tensor, q: torch.Tensor) -> torch.Tensor:
    i = torch.searchsorted(K, q).clamp_(max=K.numel() - 1)
    return K[i] == q


def build_cdbg(haps: list[np.ndarray], k: int, device=None) -> dict:
    """Returns dict(codes uint8[total_bp], off int64[N+1], kmers int64[n] sorted canonical,
    mult int64[n] multiplicity over the haplotypes, k)."""
    dev = _device(device)
    hs = [torch.from_numpy(np.ascontiguousarray(h)).to(dev) for h in haps]
    canon, is_fw, base_of_kmer = [], [], []
    base_off = 0
    last_of_hap = []
    for h in hs:
        fw, rc = _kmers(h, k)
        canon.append(torch.minimum(fw, rc))
        is_fw.append(fw <= rc)
        n = fw.numel()
        base_of_kmer.append(torch.arange(n, device=dev, dtype=torch.int64) + base_off)
        flag = torch.zeros(n, dtype=torch.bool, device=dev)
        flag[-1] = True
        last_of_hap.append(flag)
        base_off += h.numel()
        del fw, rc
    canon = torch.cat(canon)
    is_fw = torch.cat(is_fw)
    base_pos = torch.cat(base_of_kmer)
    last = torch.cat(last_of_hap)
    all_codes = torch.cat(hs)
    K, inv, mult = torch.unique(canon, return_inverse=True, return_counts=True)
    del canon
    n = K.numel()
    mask = (1 << (2 * k)) - 1
    # 4-bit successor / predecessor masks of every canonical k-mer, read forward
    out_deg = torch.zeros(n, dtype=torch.int8, device=dev)
    in_deg = torch.zeros(n, dtype=torch.int8, device=dev)
    for b in range(4):
        y = ((K << 2) | b) & mask
        out_deg += _member(K, torch.minimum(y, _rc(y, k))).to(torch.int8)
        y = (K >> 2) | (b << (2 * (k - 1)))
        in_deg += _member(K, torch.minimum(y, _rc(y, k))).to(torch.int8)
    # oriented occurrence p reads k-mer inv[p] forward (is_fw) or reversed
    o_out = torch.where(is_fw, out_deg[inv], in_deg[inv])
    o_in = torch.where(is_fw, in_deg[inv], out_deg[inv])
    P = inv.numel()
    link = torch.zeros(P, dtype=torch.bool, device=dev)  # step p -> p+1 stays inside one unitig
    link[:-1] = (o_out[:-1] == 1) & (o_in[1:] == 1) & (inv[:-1] != inv[1:]) & ~last[:-1]
    is_end = ~link
    is_start = torch.ones(P, dtype=torch.bool, device=dev)
    is_start[1:] = is_end[:-1]
    starts = torch.nonzero(is_start).squeeze(1)
    ends = torch.nonzero(is_end).squeeze(1)
    node = inv * 2 + (~is_fw).to(torch.int64)
    key = torch.minimum(node[starts], node[ends] ^ 1)
    ks, perm = torch.sort(key, stable=True)
    first = torch.ones(ks.numel(), dtype=torch.bool, device=dev)
    first[1:] = ks[1:] != ks[:-1]
    keep = perm[first]
    keep, _ = torch.sort(keep)  # haplotype order
    s_p, e_p = starts[keep], ends[keep]
    n_km = e_p - s_p + 1
    if int(n_km.sum()) != n:
        raise RuntimeError("cdbg_build: unitigs do not partition the k-mer set (%d vs %d): repeat structure the "
                           "cut-and-deduplicate construction cannot handle" % (int(n_km.sum()), n))
    lens = n_km + (k - 1)
    off = torch.zeros(lens.numel() + 1, dtype=torch.int64, device=dev)
    off[1:] = torch.cumsum(lens, 0)
    total = int(off[-1])
    # gather the bases
    seg = torch.repeat_interleave(torch.arange(lens.numel(), device=dev), lens)
    within = torch.arange(total, device=dev, dtype=torch.int64) - off[seg]
    codes = all_codes[base_pos[s_p][seg] + within]
    return dict(codes=codes.cpu().numpy(), off=off.cpu().numpy(), kmers=K.cpu().numpy().astype(np.uint64),
                mult=mult.cpu().numpy(), k=k)


def write_gfa(path: str, g: dict, min_len_tag: int = 17, da_tags=None) -> int:
    """Bifrost-style GFA 1.0: header with KL/ML tags, one S line per unitig (with its ``DA:Z:`` colour-set tag when
    ``da_tags`` is given, as ``Bifrost build -c`` writes them)."""
    lut = np.frombuffer(b"ACGT", dtype=np.uint8)
    text = lut[g["codes"]].tobytes()
    off = g["off"]
    n = len(off) - 1
    with open(path, "wb") as f:
        f.write(b"H\tVN:Z:1.0\tBV:Z:1.0.6\tKL:Z:%d\tML:Z:%d\n" % (g["k"], min_len_tag))
        chunk = []
        for u in range(n):
            if da_tags is None:
                chunk.append(b"S\t%d\t%s\n" % (u + 1, text[off[u] : off[u + 1]]))
            else:
                chunk.append(b"S\t%d\t%s\tDA:Z:%d\n" % (u + 1, text[off[u] : off[u + 1]], da_tags[u]))
            if len(chunk) >= 65536:
                f.write(b"".join(chunk))
                chunk = []
        f.write(b"".join(chunk))
    return n


def unitig_strings(g: dict) -> list[bytes]:
    lut = np.frombuffer(b"ACGT", dtype=np.uint8)
    text = lut[g["codes"]].tobytes()
    off = g["off"]
    return [text[off[u] : off[u + 1]] for u in range(len(off) - 1)]
