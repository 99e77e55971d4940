"""Multi-GPU plumbing: one process per GPU, torch.distributed ("nccl" = RCCL over xGMI on ROCm;
"gloo" in the CPU tests).  The hot path itself needs no data-path collective -- ranks own disjoint
unitig partitions -- so the only exchange is at the end of a pass: every rank contributes its site
counters and its ordered record slab (the text of <prefix>_allele_frequency.txt), and the
reference-order result is their rank-order concatenation.
"""
from __future__ import annotations

import os

import numpy as np
import torch
import torch.distributed as dist


def env_rank_world():
    return int(os.environ.get("RANK", "0")), int(os.environ.get("LOCAL_RANK", "0")), int(os.environ.get("WORLD_SIZE", "1"))


def init(backend: str | None = None, device: torch.device | None = None) -> None:
    """Rendezvous from RANK / WORLD_SIZE / MASTER_ADDR / MASTER_PORT (127.0.0.1 by default)."""
    rank, local_rank, world = env_rank_world()
    if world <= 1 or dist.is_initialized():
        return
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29500")
    if backend is None:
        backend = "nccl" if torch.cuda.is_available() else "gloo"
    kw = {}
    if backend == "nccl" and device is not None:
        kw["device_id"] = device
    dist.init_process_group(backend, rank=rank, world_size=world, **kw)


def shard_range(n_units: int, rank: int, world: int, weights: np.ndarray | None = None) -> tuple[int, int]:
    """Contiguous block [u0, u1) of unit ids for `rank`; with `weights` (e.g. k-mers per unitig)
    the blocks are balanced by weight.  Rank-order concatenation of blocks = id order."""
    if world <= 1:
        return 0, n_units
    if weights is None:
        base, rem = divmod(n_units, world)
        u0 = rank * base + min(rank, rem)
        return u0, u0 + base + (1 if rank < rem else 0)
    c = np.concatenate([[0], np.cumsum(weights, dtype=np.float64)])
    cuts = np.searchsorted(c, c[-1] * np.arange(world + 1) / world, side="left")
    cuts[0], cuts[-1] = 0, n_units
    cuts = np.maximum.accumulate(cuts)
    return int(cuts[rank]), int(cuts[rank + 1])


def all_gather_counters(values: list[int], device: torch.device) -> np.ndarray:
    """[world, len(values)] int64: every rank's counters."""
    t = torch.tensor(values, dtype=torch.int64, device=device)
    if not dist.is_initialized() or dist.get_world_size() == 1:
        return t.cpu().numpy()[None, :]
    out = [torch.zeros_like(t) for _ in range(dist.get_world_size())]
    dist.all_gather(out, t)
    return torch.stack(out).cpu().numpy()


def all_gather_slabs(slab: np.ndarray, device: torch.device) -> list[np.ndarray]:
    """Variable-length byte slabs of every rank, in rank order (sizes first, then one all-gather of
    max-padded payloads: a single direct exchange, no ring of small messages)."""
    slab = np.ascontiguousarray(slab, dtype=np.uint8)
    if not dist.is_initialized() or dist.get_world_size() == 1:
        return [slab]
    world = dist.get_world_size()
    sizes = all_gather_counters([slab.size], device)[:, 0]
    cap = int(sizes.max())
    buf = torch.zeros(max(cap, 1), dtype=torch.uint8, device=device)
    if slab.size:
        buf[: slab.size] = torch.from_numpy(slab).to(device)
    out = [torch.zeros_like(buf) for _ in range(world)]
    dist.all_gather(out, buf)
    return [o[: int(n)].cpu().numpy() for o, n in zip(out, sizes)]
