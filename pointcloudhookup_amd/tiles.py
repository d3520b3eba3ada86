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

Second mode (BASELINE config 4): ONE cloud cut into x-tiles, one per
rank, every tile extended by a 2*eps halo, clustered globally (no 50 000-row chunks).  A tower cut
by a tile edge must come out as ONE cluster with the id a single-GPU run over the whole cloud
gives it: ``cluster_tiled`` below - local exact DBSCAN per tile, one all_gather of the
(global row, local cluster) pairs of the core points near the tile edges, the same union-find on
every rank, then a relabel pass on the device (``ops.dbscan_relabel``) that also re-decides the
border points under the new numbering.
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


RECONCILE_CAP = 1024          # clusters per rank that travel in the first (usually only) exchange


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
    k = int(nclusters)
    C = table.shape[1] if table.dim() == 2 else 1
    # ONE collective and ONE host read in the usual case: every rank sends a fixed-capacity block whose first row
    # carries its cluster count (a step of the tile stream is ~2.4 ms; two collectives with a host read per rank
    # between them cost a tenth of that).  Only if some rank holds more clusters than the block does a second,
    # exactly sized exchange follow - every rank sees the same counts, so all take the same branch.
    cap = RECONCILE_CAP
    if not table.dtype.is_floating_point or k >= 2 ** 24:
        cap = 0                                                 # the count must survive the table's dtype exactly
    counts = None
    if cap:
        block = torch.zeros((cap + 1, C), dtype=table.dtype, device=dev)
        block[0, 0] = k
        m = min(k, cap)
        if m:
            block[1:1 + m] = table[:m].reshape(m, C)
        out = torch.empty((world * (cap + 1), C), dtype=table.dtype, device=dev)
        try:
            dist.all_gather_into_tensor(out, block, group=group)
        except (RuntimeError, AttributeError, NotImplementedError):     # a backend without the flat form
            parts = [torch.empty_like(block) for _ in range(world)]
            dist.all_gather(parts, block, group=group)
            out = torch.cat(parts, dim=0)
        out = out.reshape(world, cap + 1, C)
        counts = [int(v) for v in out[:, 0, 0].to(torch.float64).tolist()]
        if max(counts) <= cap:
            parts = [out[r, 1:1 + c] for r, c in enumerate(counts)]
            owner = torch.cat([torch.full((c,), r, dtype=torch.int64, device=dev) for r, c in enumerate(counts)])
            return (int(sum(counts[:rank])), int(sum(counts)), torch.cat(parts, dim=0).to(out_dev),
                    owner.to(out_dev))
    if counts is None:
        cnt = torch.tensor([k], dtype=torch.int64, device=dev)
        gathered_cnt = [torch.zeros_like(cnt) for _ in range(world)]
        dist.all_gather(gathered_cnt, cnt, group=group)
        counts = [int(c.item()) for c in gathered_cnt]
    kmax = max(max(counts), 1)
    padded = torch.zeros((kmax, C), dtype=table.dtype, device=dev)
    if k:
        padded[:k] = table[:k].reshape(k, C)
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


# ------------------------------------------------------------------------------------------------
# x-tiles with a halo: one cloud, clustered as a whole across the ranks
def tile_edges(xmin, xmax, world):
    """world+1 edges of equal-width x-tiles; tile r owns [edges[r], edges[r+1]) (the last one also xmax)."""
    return np.linspace(float(xmin), float(xmax), int(world) + 1)


def tile_select(x, edges, rank, halo):
    """(rows in this rank's tile incl. halo, owned flag per selected row) for coordinates x (numpy or
    torch 1-D).  Works on any array type with boolean masks."""
    lo, hi = float(edges[rank]), float(edges[rank + 1])
    last = rank == len(edges) - 2
    take = (x >= lo - halo) & ((x < hi + halo) | (last & (x <= hi + halo)))
    own = (x >= lo) & ((x < hi) | (last & (x <= hi)))
    return take, own


class HipFit:
    """Local clustering on the GPU: exact global DBSCAN of one tile + relabelling on the retained grid."""

    def __init__(self, eps, min_samples):
        self.eps, self.min_samples = float(eps), int(min_samples)
        self.labels = None

    def fit(self, points):
        from . import ops
        self.labels, core, k = ops.dbscan(points, self.eps, self.min_samples, 0, want_core=True)
        return self.labels, core.bool(), k

    def first_core_rows(self, k):
        """local row of the first core point of every cluster (what numbers the clusters)"""
        from . import ops
        return ops.dbscan_first_core_rows(self.labels.numel(), k, self.labels.device).long()

    def relabel(self, cluster_map):
        from . import ops
        cmap = torch.as_tensor(cluster_map, dtype=torch.int32, device=self.labels.device)
        return ops.dbscan_relabel(self.labels, cmap)


def _gather_rows(t, group):
    """all_gather of a [k, C] int64 tensor whose k differs per rank.  Returns list of per-rank tensors."""
    world = dist.get_world_size(group)
    cnt = torch.tensor([t.shape[0]], dtype=torch.int64, device=t.device)
    counts = [torch.zeros_like(cnt) for _ in range(world)]
    dist.all_gather(counts, cnt, group=group)
    counts = [int(c.item()) for c in counts]
    kmax = max(max(counts), 1)
    padded = torch.zeros((kmax,) + tuple(t.shape[1:]), dtype=t.dtype, device=t.device)
    padded[: t.shape[0]] = t
    out = [torch.zeros_like(padded) for _ in range(world)]
    dist.all_gather(out, padded, group=group)
    return [o[:c] for o, c in zip(out, counts)]


def cluster_tiled(points, rows, own, x_lo, x_hi, eps, min_samples, halo=None, fit=None, group=None):
    """Global DBSCAN of a cloud that is spread over the ranks as x-tiles with a halo of at least 2*eps.

    points : [n,3] float32 points of THIS rank's tile, halo included (device tensor for the HIP fit)
    rows   : [n] int64 global row of every point (its index in the whole cloud; defines the cluster numbering),
             ASCENDING (the tile keeps the cloud's order)
    own    : [n] bool, True for the points this rank reports (x inside its own tile)
    x_lo, x_hi : this rank's own x-range [x_lo, x_hi)
    halo   : width of the overlap on either side (default and minimum 2*eps); every rank must use the same
    Returns (labels int32 [n] for ALL local points (valid where ``own``), number of global clusters).

    Result for the owned points = one DBSCAN(eps, min_samples) over the whole cloud, ids = rank of the
    cluster's smallest core row (sklearn's numbering).  Why it is exact: a point's core flag is exact when
    its eps-ball lies inside the tile, i.e. for x in [x_lo - eps, x_hi + eps) with a 2*eps halo; flags in the
    outer ring can only be false negatives.  So local clusters are sound pieces of the true clusters, every
    true core-core edge is seen whole by the tile that owns one endpoint, and pieces that share a core
    point (same global row, seen by two tiles) belong together.  Border points are re-decided after the
    renumbering on the device (their core neighbours all have exact flags)."""
    fit = fit or HipFit(eps, min_samples)
    labels, core, k = fit.fit(points)
    k = int(k)
    single = not (dist.is_available() and dist.is_initialized()) or dist.get_world_size(group) == 1
    # bookkeeping stays where the points are (device tensors with the HIP fit): only the per-cluster table and
    # the pairs of the core points near the tile edges ever leave the device
    pts_t = torch.as_tensor(points)
    wdev = pts_t.device
    rows_d = torch.as_tensor(rows).to(wdev, torch.int64)
    lab_d = torch.as_tensor(labels).to(wdev, torch.int64)
    core_d = torch.as_tensor(core).to(wdev).bool()
    x_d = pts_t[:, 0]
    # smallest global core row of every local cluster.  `rows` ascends with the local row, so it is the global
    # row of the cluster's first local core point - which the fit knows (it numbers the clusters by it)
    cm = core_d & (lab_d >= 0)
    if hasattr(fit, "first_core_rows"):
        minrow = rows_d[fit.first_core_rows(k)].cpu() if k else torch.zeros(0, dtype=torch.int64)
    else:
        minrow = torch.full((max(k, 1),), torch.iinfo(torch.int64).max, dtype=torch.int64, device=wdev)
        if k:
            minrow.scatter_reduce_(0, lab_d[cm], rows_d[cm], reduce="amin")
        minrow = minrow[:k].cpu()
    if single:
        order = torch.argsort(minrow)
        cmap = torch.empty(k, dtype=torch.int64)
        cmap[order] = torch.arange(k)
        return fit.relabel(cmap.to(torch.int32)), k
    world, rank = dist.get_world_size(group), dist.get_rank(group)
    comm_dev = torch.device("cpu")
    if dist.get_backend(group) == "nccl":                 # RCCL moves device buffers (a few KB..MB over xGMI)
        comm_dev = wdev
    halo = 2.0 * float(eps) if halo is None else float(halo)
    if halo < 2.0 * float(eps):
        raise ValueError("halo must be at least 2*eps")
    if float(x_hi) - float(x_lo) < halo:
        raise ValueError("a tile must be at least one halo wide (only neighbouring tiles are matched)")
    # Which pieces belong together?  Every true core-core edge that crosses the tile edge e has an endpoint within
    # eps of e, and such a point has its exact core flag in BOTH tiles.  So it is enough that the left tile of every
    # edge publishes the (global row, local cluster) pairs of its core points with x in [e - eps, e + eps); the right
    # tile looks those rows up in its own labels (on the device) and reports the distinct (left piece, right piece)
    # links - a handful of pairs, whatever the number of shared points.
    e_hi = float(x_hi)
    strip = cm & (x_d >= e_hi - float(eps)) & (x_d < e_hi + float(eps))
    if rank == world - 1:
        strip = strip & False                              # no tile to the right
    pairs = torch.stack([rows_d[strip], lab_d[strip]], dim=1)        # (global row, LOCAL cluster id)
    all_minrow = _gather_rows(minrow.reshape(-1, 1).to(comm_dev), group)
    all_pairs = _gather_rows(pairs.to(comm_dev), group)
    counts = [int(m.shape[0]) for m in all_minrow]
    offs = np.concatenate([[0], np.cumsum(counts)])
    total = int(offs[-1])
    links = torch.zeros((0, 2), dtype=torch.int64, device=wdev)
    if rank > 0 and all_pairs[rank - 1].shape[0] and rows_d.numel():
        theirs = all_pairs[rank - 1].to(wdev)             # the left neighbour's strip at my lower edge
        at = torch.searchsorted(rows_d, theirs[:, 0]).clamp(max=rows_d.numel() - 1)      # rows ascend
        hit = (rows_d[at] == theirs[:, 0]) & cm[at]
        if hit.any():
            links = torch.unique(torch.stack([theirs[hit, 1] + int(offs[rank - 1]), lab_d[at[hit]] + int(offs[rank])],
                                             dim=1), dim=0)
    all_links = _gather_rows(links.to(comm_dev), group)
    if total == 0:
        return fit.relabel(torch.zeros(0, dtype=torch.int32)), 0
    # union-find over all local clusters (uid = rank offset + local id)
    parent = np.arange(total)

    def find(a):
        while parent[a] != a:
            parent[a] = parent[parent[a]]
            a = parent[a]
        return a

    for u, v in torch.cat(all_links).cpu().numpy().reshape(-1, 2):
        a, b2 = find(int(u)), find(int(v))
        if a != b2:
            parent[max(a, b2)] = min(a, b2)
    root = np.array([find(i) for i in range(total)])
    mr = torch.cat([m.reshape(-1) for m in all_minrow]).cpu().numpy()
    comp_min = np.full(total, np.iinfo(np.int64).max)
    np.minimum.at(comp_min, root, mr)
    roots = np.flatnonzero(root == np.arange(total))
    roots = roots[np.argsort(comp_min[roots], kind="stable")]          # numbered by smallest core row
    gid = np.full(total, -1, np.int64)
    gid[roots] = np.arange(len(roots))
    cmap = gid[root[offs[rank]:offs[rank] + k]]
    return fit.relabel(torch.as_tensor(cmap, dtype=torch.int32)), int(len(roots))


# ------------------------------------------------------------------------------------------------
# np.percentile over values that are spread over the ranks (the shared threshold of a tiled run)
def _key_to_f32(key):
    key = int(key) & 0xFFFFFFFF
    if key == 0xFFFFFFFF:
        return np.float32(np.nan)
    u = (key & 0x7FFFFFFF) if key & 0x80000000 else (~key & 0xFFFFFFFF)
    return np.array([u], dtype="<u4").view("<f4")[0]


class HipSelect:
    """the per-rank passes of the radix select on the GPU (pch_select_hist_f32 / pch_select_min_above_f32)"""

    def hist(self, values, pass_no, prefix):
        from . import ops
        return ops.select_hist(values, pass_no, prefix)

    def min_above(self, values, key):
        from . import ops
        return ops.select_min_above(values, key)


def shared_percentile(values, q_percent, sub=None, select=None, group=None):
    """np.percentile(concatenation of every rank's values - sub, q) in numpy >= 2 float32 semantics, bit for bit
    what pch_percentile_f32 gives for the concatenation: three all-reduced 4096-bin histogram passes locate the
    order statistic floor((N-1) q), one all-reduced minimum gives the next one, the lerp is numpy's.
    values: this rank's 1-D float32 values (device tensor for the HIP passes); sub: float32 subtracted from the
    two order statistics before the lerp (the centroid's z: x -> fl(x - sub) is monotone, so the select runs on
    the raw values).  Returns np.float32."""
    select = select or HipSelect()
    multi = dist.is_available() and dist.is_initialized() and dist.get_world_size(group) > 1

    def allsum(a):
        t = torch.as_tensor(np.asarray(a, dtype=np.int64))
        if multi:
            if dist.get_backend(group) == "nccl":
                t = t.to(torch.as_tensor(values).device)
            dist.all_reduce(t, op=dist.ReduceOp.SUM, group=group)
        return t.cpu().numpy()

    def allmin(v):
        t = torch.tensor([int(v)], dtype=torch.int64)
        if multi:
            if dist.get_backend(group) == "nccl":
                t = t.to(torch.as_tensor(values).device)
            dist.all_reduce(t, op=dist.ReduceOp.MIN, group=group)
        return int(t.item())

    n_local = int(torch.as_tensor(values).shape[0])
    N = int(allsum([n_local])[0])
    if N == 0:
        raise IndexError("index -1 is out of bounds for axis 0 with size 0")     # what np.percentile raises
    # numpy's index arithmetic in float32 (numpy/lib/_function_base_impl.py, method 'linear' under NEP 50)
    qf = np.float32(q_percent) / np.float32(100.0)
    vi = np.float32(N - 1) * qf
    prev = np.floor(vi)
    gamma = np.float32(vi - prev)
    same = False
    if vi >= np.float32(N - 1):
        prev, same = np.float32(N - 1), True
    if vi < 0:
        prev, same = np.float32(0), True
    k0 = min(max(int(prev), 0), N - 1)
    same = same or k0 == N - 1
    rank, prefix, nan_total = k0, 0, 0
    cnt_final = 0
    for p in range(3):
        h, nan = select.hist(values, p, prefix)
        h = allsum(h)
        if p == 0:
            nan_total = int(allsum([nan])[0])
        nb = 256 if p == 2 else 4096
        cum = np.cumsum(h[:nb])
        b = int(np.searchsorted(cum, rank, side="right"))
        below = int(cum[b - 1]) if b else 0
        rank -= below
        cnt_final = int(h[b])
        prefix = b if p == 0 else ((prefix << 12) | b if p == 1 else (prefix << 8) | b)
    v0key = prefix
    a = _key_to_f32(v0key)
    b_val = a
    if not same and rank + 1 >= cnt_final:               # the next order statistic is not in v0's bin: smallest key above
        b_val = _key_to_f32(allmin(select.min_above(values, v0key)))
    c = np.float32(0.0) if sub is None else np.float32(sub)
    a, b_val = np.float32(a - c), np.float32(b_val - c)
    diff = np.float32(b_val - a)
    r = np.float32(a + diff * gamma)
    if gamma >= np.float32(0.5):
        r = np.float32(b_val - diff * np.float32(np.float32(1.0) - gamma))
    if nan_total:
        r = np.float32(np.nan)
    return r
