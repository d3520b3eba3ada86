"""One process per GPU: LAS tile sharding and cross-tile label reconciliation.

The reference is single-process and bounds memory by processing independent file-order
chunks whose clusters are only reconciled afterwards, by label offsetting
(utils/tower_extraction.py:113-116) and a 30 m centre-distance de-dup
(utils/tower_extraction.py:153-162).  A stream of LAS tiles shards the same way: every
rank runs the whole hot path on its own tiles (no collective on the data path), then ONE
exchange makes the result global:

  * all_gather of the per-rank cluster counts        -> label offsets (exclusive prefix)
  * all_gather of the per-rank cluster tables (padded) -> every rank holds the global table;
    the duplicate-tower rule is then applied in global label order on the gathered table.

Traffic is a few KB per rank (RCCL over xGMI when the tensors live on the GPUs, gloo on CPU
tensors in the tests): latency bound, one collective each, no bulk data ever moves.
"""
from __future__ import annotations

import os

import numpy as np
import torch
import torch.distributed as dist


def init_from_env(backend=None):
    """Initialises torch.distributed from RANK / WORLD_SIZE / MASTER_* when WORLD_SIZE > 1.
    Returns (rank, world, local_rank)."""
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world > 1 and not dist.is_initialized():
        if backend is None:          # RCCL over xGMI on GPUs; PCH_DIST_BACKEND=gloo for CPU rehearsals
            backend = os.environ.get("PCH_DIST_BACKEND") or ("nccl" if torch.cuda.is_available() else "gloo")
        if backend == "nccl":
            torch.cuda.set_device(local)
        dist.init_process_group(backend=backend, rank=rank, world_size=world)
    return rank, world, local


def tiles_of_rank(n_tiles, rank, world):
    """Contiguous block split of tile ids (corridor order) across ranks."""
    per, extra = divmod(int(n_tiles), int(world))
    start = rank * per + min(rank, extra)
    return list(range(start, start + per + (1 if rank < extra else 0)))


def reconcile(nclusters, table, group=None):
    """Makes per-rank cluster ids global.

    nclusters : int, clusters found on this rank
    table     : float tensor [nclusters, C] (any per-cluster record, e.g. box min/max or tower
                centre + size) on the device the backend communicates from
    Returns (label_offset int, total int, global_table [total, C] on the same device,
             owner int64 [total] = rank that produced each row).
    """
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size(group) == 1:
        owner = torch.zeros((int(nclusters),), dtype=torch.int64, device=table.device)
        return 0, int(nclusters), table[: int(nclusters)], owner
    world = dist.get_world_size(group)
    rank = dist.get_rank(group)
    out_dev = table.device
    if dist.get_backend(group) == "gloo" and table.is_cuda:     # gloo rehearsal: exchange on the host
        table = table.cpu()
    dev = table.device
    cnt = torch.tensor([int(nclusters)], dtype=torch.int64, device=dev)
    counts = [torch.zeros_like(cnt) for _ in range(world)]
    dist.all_gather(counts, cnt, group=group)
    counts = [int(c.item()) for c in counts]
    kmax = max(max(counts), 1)
    C = table.shape[1] if table.dim() == 2 else 1
    padded = torch.zeros((kmax, C), dtype=table.dtype, device=dev)
    if nclusters:
        padded[: int(nclusters)] = table[: int(nclusters)].reshape(int(nclusters), C)
    gathered = [torch.zeros_like(padded) for _ in range(world)]
    dist.all_gather(gathered, padded, group=group)
    parts = [g[:c] for g, c in zip(gathered, counts)]
    owner = torch.cat([torch.full((c,), r, dtype=torch.int64, device=dev) for r, c in enumerate(counts)])
    offset = int(sum(counts[:rank]))
    return offset, int(sum(counts)), torch.cat(parts, dim=0).to(out_dev), owner.to(out_dev)


def dedup_centres(centres, threshold=30.0):
    """The reference's duplicate rule (utils/tower_extraction.py:153-162) on a table of
    centres in global label order: a row is dropped when it lies closer than ``threshold`` to
    an already accepted row.  Returns the kept row indices."""
    c = np.asarray(centres, dtype=np.float64).reshape(-1, 3)
    kept = []
    for i in range(c.shape[0]):
        if all(np.linalg.norm(c[i] - c[j]) >= threshold for j in kept):
            kept.append(i)
    return kept
