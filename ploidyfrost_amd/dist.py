"""Multi-GPU plumbing: one process per GPU, torch.distributed ("nccl" = RCCL over xGMI on ROCm; "gloo" in the CPU tests).

One graph, replicated in every GPU's HBM, is cut over the ranks by entrance vertex:

  findSuperBubble    rank r traverses the candidate entrances on its contiguous unitig range (K-BFS); the traversal records and
                     their vertex lists are all-gathered (sizes, then max-padded payloads); every rank replays all records in
                     entrance order -- the commit replay is sequential by construction and cheap -- so all ranks hold the
                     same MyUnitig state.  Rank 0 writes super_bubble.txt.
  PloidyEstimation   the owner scan and the sequential pass run on every rank and give the same bubble list in output order;
                     rank r aligns its contiguous slice.  Two small all-gathers follow: how many bubbles each rank called
                     (var_count numbers bubbles across the whole run, src/CDBG.cpp:1254-1258), then the byte sizes of each
                     rank's ten text slabs together with its allele histograms and coverage counters.  With the sizes every
                     rank knows its offsets and writes its slabs straight into the shared result files: rank-order concatenation
                     is the reference's `-t 1` output, and no payload crosses ranks.
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


def all_gather_slabs(slab: np.ndarray, device: torch.device, keep_device: bool = False):
    """Variable-length byte slabs of every rank, in rank order (sizes first, then one all-gather of
    max-padded payloads: a single direct exchange, no ring of small messages).  keep_device: also return the gathered
    device tensors (padded), for consumers that read the slabs on the device."""
    slab = np.ascontiguousarray(slab, dtype=np.uint8)
    if not dist.is_initialized() or dist.get_world_size() == 1:
        return ([slab], None) if keep_device else [slab]
    world = dist.get_world_size()
    sizes = all_gather_counters([slab.size], device)[:, 0]
    cap = int(sizes.max())
    buf = torch.zeros(max(cap, 1), dtype=torch.uint8, device=device)
    if slab.size:
        buf[: slab.size] = torch.from_numpy(slab).to(device)
    out = [torch.zeros_like(buf) for _ in range(world)]
    dist.all_gather(out, buf)
    if keep_device == "only":   # the consumer reads the slabs where they are: (byte sizes, padded device tensors)
        return [int(n) for n in sizes], out
    if keep_device:
        return [o[: int(n)].cpu().numpy() for o, n in zip(out, sizes)], out
    return [o[: int(n)].cpu().numpy() for o, n in zip(out, sizes)]


def broadcast_str(text: str | None, src: int = 0) -> str:
    """A path (e.g. rank 0's output directory) to every rank."""
    if not dist.is_initialized() or dist.get_world_size() == 1:
        return text
    box = [text]
    dist.broadcast_object_list(box, src=src)
    return box[0]


def sharded_find(run, outpre: str, device: torch.device, stats: dict | None = None) -> None:
    """findSuperBubble of one graph over all ranks (see module docstring); `run` = hostapi.Run of the same graph on every rank."""
    rank, world = (dist.get_rank(), dist.get_world_size()) if dist.is_initialized() else (0, 1)
    n = run.times()["unitigs"]
    u0, u1 = shard_range(n, rank, world)
    rec, pool = run.find_shard(u0, u1)
    on_gpu = device.type == "cuda"
    from . import hipapi
    if on_gpu and world > 1:
        # the gathered shards stay in device memory: K-CC reads them there, and the host layer brings its copy down into pinned
        # memory itself (57 GB/s) instead of through a pageable numpy array
        rs, d_recs = all_gather_slabs(rec.view(np.uint8).reshape(-1), device, keep_device="only")
        ps, d_pools = all_gather_slabs(pool.view(np.uint8).reshape(-1), device, keep_device="only")
        if stats is not None:
            stats["find_gathered_bytes"] = int(sum(rs) + sum(ps))
            stats["shard_unitigs"] = (u0, u1)
        run.find_replay(outpre, None, [(a // hipapi.BFS_RECORD.itemsize, b // 4) for a, b in zip(rs, ps)], write_file=rank == 0,
                        dev_records=[t.data_ptr() for t in d_recs], dev_pools=[t.data_ptr() for t in d_pools])
        return
    recs, d_recs = all_gather_slabs(rec.view(np.uint8).reshape(-1), device, keep_device=True)
    pools, d_pools = all_gather_slabs(pool.view(np.uint8).reshape(-1), device, keep_device=True)
    if stats is not None:
        stats["find_gathered_bytes"] = int(sum(x.size for x in recs) + sum(x.size for x in pools))
        stats["shard_unitigs"] = (u0, u1)
    # the gathered shards still lie in device memory: the components of the parallel replay are found there
    run.find_replay(outpre, [np.ascontiguousarray(r).view(hipapi.BFS_RECORD) for r in recs],
                    [np.ascontiguousarray(p).view(np.uint32) for p in pools], write_file=rank == 0,
                    dev_records=[t.data_ptr() for t in d_recs] if on_gpu and d_recs else None,
                    dev_pools=[t.data_ptr() for t in d_pools] if on_gpu and d_pools else None)


def sharded_ploidy(run, outpre: str, lower: int, upper: int, device: torch.device, stats: dict | None = None, cutoffs=None):
    """PloidyEstimation of one graph over all ranks; all ranks' `run` must share one output directory.  cutoffs: a colored run's
    (lower, upper) per colour (lower / upper are then ignored).
    Returns (totals of the ten files in bytes, counters summed over ranks: sites with 2..5 alleles, coreCov, coreNum, called, bubbles)."""
    rank, world = (dist.get_rank(), dist.get_world_size()) if dist.is_initialized() else (0, 1)
    n_bubbles = run.ploidy_select(cutoffs) if cutoffs is not None else run.ploidy_select(lower, upper)
    t0, t1 = shard_range(n_bubbles, rank, world)
    called = run.ploidy_align(t0, t1)
    base = int(all_gather_counters([called], device)[:rank, 0].sum())
    sizes, counters = run.ploidy_text(base)
    table = all_gather_counters([int(x) for x in sizes] + [int(x) for x in counters], device)
    offsets = table[:rank, :10].sum(axis=0).astype(np.uint64)
    totals = table[:, :10].sum(axis=0).astype(np.uint64)
    run.ploidy_write(outpre, offsets, totals, truncate=True)
    if stats is not None:
        stats["slice"] = (t0, t1)
        stats["n_bubbles"] = n_bubbles
    return totals, table[:, 10:].sum(axis=0)
