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
every rank, then a relabel pass on the device (``ops.DbscanFit.relabel``) that also re-decides the
border points under the new numbering.
"""
from __future__ import annotations

import os

import numpy as np
import torch
import torch.distributed as dist


# The group the exchanges of this module use when the caller passes group=None: None = the default group.  With
# automatic backend choice on GPUs the DEFAULT group is gloo (it always comes up) and the exchanges run on an RCCL
# group that was probed first (init_from_env) - a node whose RCCL does not initialise degrades to gloo exchanges
# (a few KB each) instead of losing the run.
_EXCHANGE_GROUP = None

# PCH_TILES_FORCE_COLLECTIVES=1: a world of ONE rank still takes the multi-rank branch of every function below -
# every all_gather / all_reduce / broadcast is really issued (on RCCL when the group is "nccl": init_from_env then
# brings a one-rank nccl group up) instead of being short-circuited.  The only calls a single rank cannot make are the
# point-to-point hops of the centroid chain.  A test / rehearsal switch: it makes the RCCL code paths executable on a
# one-GPU box (tests/test_gpu_e2e.py), results are unchanged.
_FORCE = os.environ.get("PCH_TILES_FORCE_COLLECTIVES", "0") == "1"


def _pg(group):
    return group if group is not None else _EXCHANGE_GROUP


COLLECTIVES = {}      # name -> calls issued by this module in this process (tests / bench: how many per step?)


def _count(name):
    COLLECTIVES[name] = COLLECTIVES.get(name, 0) + 1


def _multi(group):
    """do the exchanges of this module really run (more than one rank, or forced)?"""
    if not (dist.is_available() and dist.is_initialized()):
        return False
    return dist.get_world_size(group) > 1 or _FORCE


def exchange_backend():
    """'nccl' (= RCCL on ROCm), 'gloo' or 'none': what carries the exchanges of this module by default."""
    if not _multi(_EXCHANGE_GROUP):
        return "none"
    return dist.get_backend(_EXCHANGE_GROUP)


def init_from_env(backend=None, timeout_s=None, single_device=False):
    """Initialises torch.distributed from RANK / WORLD_SIZE / MASTER_* when WORLD_SIZE > 1 (or when
    PCH_TILES_FORCE_COLLECTIVES=1 asks for a one-rank group).
    backend: "nccl" / "gloo" (also env PCH_DIST_BACKEND) = that backend for everything.  None on a GPU node: the
    default group is gloo and an RCCL group over all ranks is PROBED - created (outcome agreed over gloo BEFORE
    anybody enters a device collective), then one small all-reduce on the device with a short wait of its own, the
    outcome agreed over gloo again; if every rank succeeded the exchanges of this module use it
    (exchange_backend() == 'nccl'), otherwise they stay on gloo and rank 0 says so.  A rank that fails asymmetrically
    INSIDE the device all-reduce cannot be recovered from (the others sit in the collective until its timeout and the
    watchdog then aborts the process): the probe only removes the failures that show before that point.
    ``timeout_s`` bounds every collective (a rank that died leaves the others waiting at most that long).
    ``single_device``: all ranks use device 0 (a rehearsal on a one-GPU box; RCCL then refuses and gloo carries on).
    Returns (rank, world, local_rank)."""
    global _EXCHANGE_GROUP
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world == 1 and _FORCE and not dist.is_initialized():
        import socket
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if "MASTER_PORT" not in os.environ:
            with socket.socket() as so:
                so.bind(("127.0.0.1", 0))
                os.environ["MASTER_PORT"] = str(so.getsockname()[1])
        if backend is None:
            backend = os.environ.get("PCH_DIST_BACKEND") or ("nccl" if torch.cuda.is_available() else "gloo")
        if backend == "nccl":
            torch.cuda.set_device(local)
        dist.init_process_group(backend=backend, rank=0, world_size=1)
        _EXCHANGE_GROUP = None
        return rank, world, local
    if world > 1 and not dist.is_initialized():
        kw = {}
        if timeout_s:
            import datetime
            kw["timeout"] = datetime.timedelta(seconds=float(timeout_s))
        if backend is None:
            backend = os.environ.get("PCH_DIST_BACKEND")
        if backend is not None or not torch.cuda.is_available():
            backend = backend or "gloo"
            if backend == "nccl":
                torch.cuda.set_device(local)
            dist.init_process_group(backend=backend, rank=rank, world_size=world, **kw)
            _EXCHANGE_GROUP = None
            return rank, world, local
        devi = 0 if single_device else local                # rehearsal: all ranks on device 0
        torch.cuda.set_device(devi)
        dist.init_process_group(backend="gloo", rank=rank, world_size=world, **kw)

        def agreed(ok):                                     # over gloo: every rank takes the same branch
            flag = torch.tensor([1 if ok else 0], dtype=torch.int64)
            dist.all_reduce(flag, op=dist.ReduceOp.MIN)
            return int(flag.item()) == 1

        ok, why, g = True, "", None
        try:
            g = dist.new_group(ranks=list(range(world)), backend="nccl", **kw)
        except Exception as e:
            ok, why = False, f"new_group: {type(e).__name__}: {e}"
        if agreed(ok):                                      # nobody enters the device collective unless all have a group
            try:
                import datetime
                t = torch.ones(1, device=torch.device("cuda", devi))
                work = dist.all_reduce(t, group=g, async_op=True)
                work.wait(timeout=datetime.timedelta(seconds=min(60.0, float(timeout_s or 60.0))))
                torch.cuda.synchronize()
                if int(t.item()) != world:
                    ok, why = False, f"all_reduce gave {t.item()} for {world} ranks"
            except Exception as e:                          # e.g. two ranks on one device, IPC not available
                ok, why = False, f"{type(e).__name__}: {e}"
            ok = agreed(ok)
        else:
            ok = False
        if ok:
            _EXCHANGE_GROUP = g
        else:
            _EXCHANGE_GROUP = None
            if why:
                print(f"[pch tiles] rank {rank}: RCCL probe failed ({why.splitlines()[0][:200]}); exchanges stay on gloo",
                      flush=True)
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
    group = _pg(group)
    if not _multi(group):
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
        _count("all_gather")
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
    """Local clustering on the GPU: exact global DBSCAN of one tile + relabelling on the retained grid
    (ops.DbscanFit: the grid lives in a workspace of its own, whatever runs between fit and relabel)."""

    def __init__(self, eps, min_samples):
        self.eps, self.min_samples = float(eps), int(min_samples)
        self._fit = None

    def fit(self, points):
        from . import ops
        self._fit = ops.DbscanFit(points, self.eps, self.min_samples, 0)
        return self._fit.labels, self._fit.core.bool(), self._fit.nclusters

    def first_core_rows(self, k):
        """local row of the first core point of every cluster (what numbers the clusters)"""
        return self._fit.first_core_rows().long()

    def strip_pairs(self, x_lo, x_hi, cap):
        """((local row, local cluster) int32 [cap,2], count int32 [1]) - one pair per grid cell with a core point
        in the strip (pch_dbscan_strip_pairs_i32)"""
        return self._fit.strip_pairs(x_lo, x_hi, cap)

    def relabel(self, cluster_map):
        cmap = torch.as_tensor(cluster_map, dtype=torch.int32, device=self._fit.device)
        return self._fit.relabel(cmap)


def _mark(timings, label):
    """timings["trace"] (a list, if present) receives (label, perf_counter()) - host timestamps only, no
    synchronisation is added (tools: where does a tiled step spend its wall clock?)"""
    if timings is not None and "trace" in timings:
        import time
        timings["trace"].append((label, time.perf_counter()))


# capacities of the fixed-size exchange of cluster_tiled (per rank): clusters, strip pairs per edge.  A rank that
# holds more is seen by everybody in the block's header and ONE exactly sized exchange follows (same branch on all)
TILED_KCAP, TILED_PCAP = 2048, 2048           # clusters per rank; strip cells per edge


def _all_gather_block(block, group):
    """ONE collective: every rank's equally sized 1-D block -> [world, len] tensor on the block's device."""
    world = dist.get_world_size(group)
    out = torch.empty((world * block.numel(),), dtype=block.dtype, device=block.device)
    _count("all_gather")
    try:
        dist.all_gather_into_tensor(out, block, group=group)
    except (RuntimeError, AttributeError, NotImplementedError):     # a backend without the flat form
        parts = [torch.empty_like(block) for _ in range(world)]
        dist.all_gather(parts, block, group=group)
        out = torch.cat(parts)
    return out.reshape(world, block.numel())


def _exchange(header, payloads, caps, comm_dev, group, extra=()):
    """All-gathers, in ONE collective, a block per rank = [len(payloads) counts | extra words | payload 0 padded to
    caps[0] | ...].  payloads: 1-D int64 tensors; header: their true lengths (ints, or 0-d device tensors that are
    never read on this side); extra: a few more int64 words per rank that ride along (ints).  If some rank's length
    exceeds its capacity, one more exchange sized by the largest lengths follows.
    Returns (per-rank lists of numpy arrays, largest length of every payload over the ranks, extra words [world, n])."""
    nh = len(payloads) + len(extra)

    def pack(caps_now):
        blk = torch.zeros((nh + sum(caps_now),), dtype=torch.int64, device=payloads[0].device)
        at = nh
        for j, (p, c) in enumerate(zip(payloads, caps_now)):
            blk[j] = header[j] if torch.is_tensor(header[j]) else int(header[j])
            m = min(int(p.numel()), c)
            if m:
                blk[at:at + m] = p[:m]
            at += c
        for j, v in enumerate(extra):
            blk[len(payloads) + j] = int(v)
        return blk

    def unpack(out, caps_now):
        res = []
        for r in range(out.shape[0]):
            row, at, parts = out[r], nh, []
            for j, c in enumerate(caps_now):
                parts.append(row[at:at + min(int(row[j]), c)])
                at += c
            res.append(parts)
        return res

    caps = list(caps)
    out = _all_gather_block(pack(caps).to(comm_dev), group).cpu().numpy()
    need = out[:, :len(payloads)].max(axis=0)
    if (need > np.asarray(caps)).any():                  # rare: a rank overflowed its block; everybody sees it
        caps = [int(max(c, n)) for c, n in zip(caps, need)]
        out = _all_gather_block(pack(caps).to(comm_dev), group).cpu().numpy()
    return unpack(out, caps), [int(v) for v in need], out[:, len(payloads):nh]


def strip_representatives(points, rows, labels, core, strips, eps):
    """For every strip (x_from, x_to) of ``strips``: one (global row, local cluster) pair per LATTICE cell that holds a
    core point with x in [x_from, x_to).  The lattice is the same on every rank: cell = floor(double(coordinate) / side)
    per axis, side = eps/sqrt(3) (1 - 2^-16), anchored at the origin of the (shared, centred) frame - so two tiles that
    both hold the points of such a strip, with the same core flags, name the SAME rows: the smallest global core row of
    every cell.  Two points of one cell are closer than eps, so the core points of a cell are one cluster in either
    tile.  The strips must not overlap (a tile is at least one halo wide).
    points [n,3] float32, rows [n] int64 ascending, labels [n] int32 / int64, core [n] bool (same device).
    Returns a list of int64 [m,2] tensors sorted by row, one per strip (all strips in ONE pass: three host reads)."""
    none = torch.zeros((0, 2), dtype=torch.int64, device=points.device)
    if not strips:
        return []
    if len(strips) > 2:
        raise ValueError("strip_representatives: at most two strips per call")
    if points.is_cuda:                                    # the product path: one table-building kernel + one that emits
        from . import ops
        return ops.strip_lattice_reps(points, rows, labels.to(torch.int32), core, strips, eps)
    # CPU tensors (the gloo tests with a CPU stand-in for the fit): the same cells with torch operators
    labels = labels.to(torch.int64)
    x = points[:, 0]
    sid = torch.full((points.shape[0],), -1, dtype=torch.int64, device=points.device)
    for k, (a, b) in enumerate(strips):
        sid = torch.where((x >= float(a)) & (x < float(b)), torch.full_like(sid, k), sid)
    pick = core & (labels >= 0) & (sid >= 0)
    idx = torch.nonzero(pick).squeeze(1)                                  # host read 1
    if idx.numel() == 0:
        return [none for _ in strips]
    side = float(eps) / 3 ** 0.5 * (1.0 - 2.0 ** -16)
    # 21 bits per axis: +-2^20 cells = +-4.8e6 m at eps = 8 m (the centred frame of an EPSG-scale cloud lies up to
    # ~2.5e6 m from its origin: the float32 sequential "centroid" is that far off, utils/tower_extraction.py:63)
    c = torch.floor(points.index_select(0, idx).to(torch.float64) * (1.0 / side)).to(torch.int64) + (1 << 20)
    if bool(((c < 0) | (c >= (1 << 21))).any()):                          # host read 2
        raise ValueError("strip_representatives: coordinates beyond 2^20 lattice cells from the origin")
    sk = sid.index_select(0, idx)
    key = (sk << 63) | (c[:, 0] << 42) | (c[:, 1] << 21) | c[:, 2]
    r = rows.index_select(0, idx)
    uniq, inv = torch.unique(key, return_inverse=True)                    # host read 3 (its output size)
    rep = torch.full((uniq.numel(),), torch.iinfo(torch.int64).max, dtype=torch.int64, device=points.device)
    rep.scatter_reduce_(0, inv, r, reduce="amin")
    # the representative of cell u is the row rep[u]; its label: rows ascend, so look the row up again
    at = torch.searchsorted(rows, rep)
    lab = labels.index_select(0, at)
    strip_of = (uniq >> 63) & 1
    out = []
    for k in range(len(strips)):
        m = strip_of == k
        pairs = torch.stack([rep[m], lab[m]], dim=1)
        out.append(pairs[torch.argsort(pairs[:, 0])] if pairs.shape[0] else none)
    return out


def union_links(total, links, minrow):
    """Components of the graph (clusters 0..total-1, undirected links [m,2]) numbered by the smallest entry of
    ``minrow`` (int64 [total]) inside each - sklearn's numbering of the whole cloud.  A union-find over the (few)
    clusters that appear in a link; everything over the (many) clusters is vectorised, and the final ranking is a
    stable sort of what is a concatenation of ascending runs (every rank numbers its clusters by first core row).
    Returns (gid int64 [total], number of components)."""
    total = int(total)
    mr = np.asarray(minrow, dtype=np.int64)
    root = np.arange(total, dtype=np.int64)
    comp_min = mr
    lk = np.asarray(links, dtype=np.int64).reshape(-1, 2)
    if len(lk):
        nodes, inv = np.unique(lk.reshape(-1), return_inverse=True)
        parent = list(range(len(nodes)))

        def find(a):
            while parent[a] != a:
                parent[a] = parent[parent[a]]
                a = parent[a]
            return a

        for a, b in inv.reshape(-1, 2).tolist():
            ra, rb = find(a), find(b)
            if ra != rb:
                parent[max(ra, rb)] = min(ra, rb)            # nodes ascend: the root is the smallest cluster index
        loc = np.fromiter((find(i) for i in range(len(nodes))), dtype=np.int64, count=len(nodes))
        root[nodes] = nodes[loc]
        comp_min = mr.copy()
        np.minimum.at(comp_min, nodes[loc], mr[nodes])
    is_root = root == np.arange(total, dtype=np.int64)
    roots = np.flatnonzero(is_root)
    order = roots[np.argsort(comp_min[roots], kind="stable")]
    gid_of_root = np.empty(total, dtype=np.int64)
    gid_of_root[order] = np.arange(len(order), dtype=np.int64)
    return gid_of_root[root], int(len(order))


def cluster_tiled(points, rows, own, x_lo, x_hi, eps, min_samples, halo=None, fit=None, group=None, timings=None,
                  extra=()):
    """Global DBSCAN of a cloud that is spread over the ranks as x-tiles with a halo of at least 2*eps.

    points : [n,3] float32 points of THIS rank's tile, halo included (device tensor for the HIP fit)
    rows   : [n] int64 global row of every point (its index in the whole cloud; defines the cluster numbering),
             ASCENDING (the tile keeps the cloud's order)
    own    : [n] bool, True for the points this rank reports (x inside its own tile)
    x_lo, x_hi : this rank's own x-range [x_lo, x_hi); neighbouring ranks must pass the SAME number for the edge
             they share (the strips on either side of it are cut with it)
    halo   : width of the overlap on either side (default and minimum 2*eps); every rank must use the same
    timings: optional dict; "fit_ms" / "reconcile_ms" are ADDED to it (wall clock, the device drained in between)
    extra  : a few ints that ride along in the exchange; their per-rank values come back as the third result
    Returns (labels int32 [n] for ALL local points (valid where ``own``), number of global clusters,
             extra words of every rank: int64 [world, len(extra)]).

    Result for the owned points = one DBSCAN(eps, min_samples) over the whole cloud, ids = rank of the
    cluster's smallest core row (sklearn's numbering).  Why it is exact: a point's core flag is exact when
    its eps-ball lies inside the tile, i.e. for x in [x_lo - eps, x_hi + eps) with a 2*eps halo; flags in the
    outer ring can only be false negatives.  So local clusters are sound pieces of the true clusters, every
    true core-core edge is seen whole by the tile that owns one endpoint, and pieces that share a core
    point (same global row, seen by two tiles) belong together.  Border points are re-decided after the
    renumbering on the device (their core neighbours all have exact flags).

    The exchange: ONE collective per call - one all_gather of a fixed-capacity block per rank whose
    header carries the true counts: the smallest global core row of every local cluster, and for BOTH edges of the
    tile the (global row, local cluster) pairs of the strip [e - eps, e + eps) around the edge, one pair per cell of a
    lattice every rank shares (strip_representatives: both neighbours hold the strip's points with exact core
    flags, so both name the same rows).  Every rank then joins rank r's upper strip with rank r+1's lower strip on
    the global row - the links between pieces - and runs the same union (union_links); no second exchange."""
    group = _pg(group)
    import time
    t_start = time.perf_counter()
    fit = fit or HipFit(eps, min_samples)
    labels, core, k = fit.fit(points)
    k = int(k)
    t_fit = time.perf_counter()                            # fit() has read the cluster count: the device is drained
    _mark(timings, "fit")
    single = not _multi(group)
    # bookkeeping stays where the points are (device tensors with the HIP fit): only the per-cluster table and
    # the pairs of the strip cells ever leave the device
    pts_t = torch.as_tensor(points)
    wdev = pts_t.device
    rows_d = torch.as_tensor(rows).to(wdev, torch.int64)
    lab_d = torch.as_tensor(labels).to(wdev, torch.int64)
    core_d = torch.as_tensor(core).to(wdev).bool()
    # smallest global core row of every local cluster.  `rows` ascends with the local row, so it is the global
    # row of the cluster's first local core point - which the fit knows (it numbers the clusters by it)
    cm = core_d & (lab_d >= 0)
    if hasattr(fit, "first_core_rows"):
        minrow = rows_d[fit.first_core_rows(k)] if k else torch.zeros(0, dtype=torch.int64, device=wdev)
    else:
        minrow = torch.full((max(k, 1),), torch.iinfo(torch.int64).max, dtype=torch.int64, device=wdev)
        if k:
            minrow.scatter_reduce_(0, lab_d[cm], rows_d[cm], reduce="amin")
        minrow = minrow[:k]

    def done(labels_out, total, words):
        _mark(timings, "relabel_enqueued")
        if timings is not None and "trace" not in timings:
            if torch.as_tensor(labels_out).is_cuda:
                torch.cuda.synchronize(wdev)
            t_end = time.perf_counter()
            timings["fit_ms"] = timings.get("fit_ms", 0.0) + 1e3 * (t_fit - t_start)
            timings["reconcile_ms"] = timings.get("reconcile_ms", 0.0) + 1e3 * (t_end - t_fit)
        return labels_out, total, words

    if single:
        order = torch.argsort(minrow)
        cmap = torch.empty(k, dtype=torch.int64, device=wdev)
        cmap[order] = torch.arange(k, device=wdev)
        return done(fit.relabel(cmap.to(torch.int32)), k, np.asarray([list(extra)], dtype=np.int64))
    world, rank = dist.get_world_size(group), dist.get_rank(group)
    comm_dev = torch.device("cpu")
    if dist.get_backend(group) == "nccl":                 # RCCL moves device buffers (a few KB..MB over xGMI)
        comm_dev = wdev
    halo = 2.0 * float(eps) if halo is None else float(halo)
    if halo < 2.0 * float(eps):
        raise ValueError("halo must be at least 2*eps")
    if world > 1 and float(x_hi) - float(x_lo) < halo:
        raise ValueError("a tile must be at least one halo wide (only neighbouring tiles are matched)")
    # Which pieces belong together?  Every true core-core edge that crosses the tile edge e has an endpoint within
    # eps of e, and such a point has its exact core flag in BOTH tiles.  So both tiles publish (global row, local
    # cluster) for the core points with x in [e - eps, e + eps) - one per lattice cell, the same rows on either side.
    none = torch.zeros((0, 2), dtype=torch.int64, device=wdev)
    strips, which = [], []
    if rank > 0:
        strips.append((float(x_lo) - float(eps), float(x_lo) + float(eps)))
        which.append("lower")
    if rank < world - 1:
        strips.append((float(x_hi) - float(eps), float(x_hi) + float(eps)))
        which.append("upper")
    reps = dict(zip(which, strip_representatives(pts_t, rows_d, torch.as_tensor(labels).to(wdev), core_d, strips, eps)))
    lower, upper = reps.get("lower", none), reps.get("upper", none)
    _mark(timings, "pairs_built")
    got, _, words = _exchange([k, 2 * int(lower.shape[0]), 2 * int(upper.shape[0])],
                              [minrow, lower.reshape(-1), upper.reshape(-1)],
                              [TILED_KCAP, 2 * TILED_PCAP, 2 * TILED_PCAP], comm_dev, group, extra=extra)
    _mark(timings, "exchange1")
    counts = [int(g[0].shape[0]) for g in got]
    offs = np.concatenate([[0], np.cumsum(counts)])
    total = int(offs[-1])
    if total == 0:
        return done(fit.relabel(torch.zeros(0, dtype=torch.int32)), 0, words)
    # links: rank r's upper strip joined with rank r+1's lower strip on the global row (both sorted by row)
    links = []
    for r in range(world - 1):
        up, lo = got[r][2].reshape(-1, 2), got[r + 1][1].reshape(-1, 2)
        if len(up) and len(lo):
            _, iu, il = np.intersect1d(up[:, 0], lo[:, 0], assume_unique=True, return_indices=True)
            if len(iu):
                links.append(np.stack([up[iu, 1] + offs[r], lo[il, 1] + offs[r + 1]], axis=1))
    links = np.unique(np.concatenate(links), axis=0) if links else np.zeros((0, 2), np.int64)
    _mark(timings, "links_built")
    gid, ncomp = union_links(total, links, np.concatenate([g[0] for g in got]))
    cmap = gid[offs[rank]:offs[rank] + k]
    return done(fit.relabel(torch.as_tensor(cmap, dtype=torch.int32)), int(ncomp), words)


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


def shared_percentile(values, q_percent, sub=None, select=None, group=None, ride=None, first_hist=None):
    """np.percentile(concatenation of every rank's values - sub, q) in numpy >= 2 float32 semantics, bit for bit
    what pch_percentile_f32 gives for the concatenation: three all-reduced 4096-bin histogram passes locate the
    order statistic floor((N-1) q), one all-reduced minimum gives the next one, the lerp is numpy's.
    values: this rank's 1-D float32 values (device tensor for the HIP passes); sub: float32 subtracted from the
    two order statistics before the lerp (the centroid's z: x -> fl(x - sub) is monotone, so the select runs on
    the raw values) - or a callable that receives the summed ``ride`` words and returns it.
    ride: int64 words of this rank that are SUMMED over the ranks in the first exchange, beside the histogram - how
    tiled_step broadcasts the centroid (only the last rank contributes non-zero bits) without a collective of its
    own; first_hist: pass 0 of this rank, (hist, nan count), if the caller has already computed it.
    Returns np.float32."""
    group = _pg(group)
    select = select or HipSelect()
    multi = _multi(group)
    ride = np.zeros(0, dtype=np.int64) if ride is None else np.asarray(ride, dtype=np.int64).reshape(-1)

    def allsum(a):
        t = torch.as_tensor(np.asarray(a, dtype=np.int64))
        if multi:
            if dist.get_backend(group) == "nccl":
                t = t.to(torch.as_tensor(values).device)
            _count("all_reduce")
            dist.all_reduce(t, op=dist.ReduceOp.SUM, group=group)
        return t.cpu().numpy()

    def allmin(v):
        t = torch.tensor([int(v)], dtype=torch.int64)
        if multi:
            if dist.get_backend(group) == "nccl":
                t = t.to(torch.as_tensor(values).device)
            _count("all_reduce")
            dist.all_reduce(t, op=dist.ReduceOp.MIN, group=group)
        return int(t.item())

    n_local = int(torch.as_tensor(values).shape[0])
    rank = prefix = nan_total = cnt_final = N = 0
    same = False
    for p in range(3):
        h, nan = first_hist if (p == 0 and first_hist is not None) else select.hist(values, p, prefix)
        if p == 0:
            # the first exchange also carries every rank's value count and NaN count (and the caller's words): one
            # all-reduce, not three
            hx = allsum(np.concatenate([np.asarray(h, dtype=np.int64), [int(nan), n_local], ride]))
            nh = len(hx) - len(ride)
            ride_sum, hx = hx[nh:], hx[:nh]
            if callable(sub):
                sub = sub(ride_sum)
            h, nan_total, N = hx[:-2], int(hx[-2]), int(hx[-1])
            if N == 0:
                raise IndexError("index -1 is out of bounds for axis 0 with size 0")     # what np.percentile raises
            # numpy's index arithmetic in float32 (numpy/lib/_function_base_impl.py, method 'linear' under NEP 50)
            qf = np.float32(q_percent) / np.float32(100.0)
            vi = np.float32(N - 1) * qf
            prev = np.floor(vi)
            gamma = np.float32(vi - prev)
            if vi >= np.float32(N - 1):
                prev, same = np.float32(N - 1), True
            if vi < 0:
                prev, same = np.float32(0), True
            k0 = min(max(int(prev), 0), N - 1)
            same = same or k0 == N - 1
            rank = k0
        else:
            h = allsum(h)
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


# ------------------------------------------------------------------------------------------------
# np.mean(raw, axis=0) over rows that are spread over the ranks as consecutive file-order shards
class HipMeanShard:
    """the per-rank part of the chained float32 column sums on the GPU (ops.MeanShard: tables first, walk later)"""

    def __init__(self, rows, want_zcol=False):
        from . import ops
        self._sh = ops.MeanShard(rows, want_zcol=want_zcol)
        self.device = self._sh.device
        self.zcol = self._sh.zcol

    def walk(self, sum_in, total_n):
        return self._sh.walk(sum_in, total_n)


def sharded_centroid(own_rows, total_n, shard=None, group=None, timings=None, broadcast=True):
    """np.mean(concatenation of every rank's rows in RANK order, axis=0) of float32 [n_r,3] shards, bit for bit:
    numpy's sum is sequential per column, so the three running float32 sums travel down the line of ranks
    (12 bytes per hop, one send/recv each) and the last rank divides by float32(total_n) and broadcasts.  Every rank
    builds the summary tables of its own rows first and at the same time (they do not depend on the incoming
    sums); only the short serial walks are chained.  Replaces utils/tower_extraction.py:63 for a cloud whose
    file-order shards live on different GPUs.  Returns np.float32 [3] (host)."""
    group = _pg(group)
    if int(total_n) == 0:                                  # np.mean of an empty array: 0 / 0 (every rank knows total_n)
        return np.full(3, np.nan, dtype=np.float32)
    shard = shard or HipMeanShard(own_rows)
    _mark(timings, "c.tables_enqueued")
    if not _multi(group):
        return torch.as_tensor(shard.walk(None, int(total_n))).cpu().numpy().astype(np.float32)
    world, rank = dist.get_world_size(group), dist.get_rank(group)
    wdev = getattr(shard, "device", torch.device("cpu"))
    comm_dev = wdev if dist.get_backend(group) == "nccl" else torch.device("cpu")
    sum_in = None
    if rank > 0:
        buf = torch.empty(3, dtype=torch.float32, device=comm_dev)
        _count("recv")
        dist.recv(buf, src=dist.get_global_rank(group, rank - 1) if group is not None else rank - 1, group=group)
        _mark(timings, "c.recv")
        sum_in = buf.to(wdev)
    last = rank == world - 1
    out = torch.as_tensor(shard.walk(sum_in, int(total_n) if last else 0)).to(comm_dev, torch.float32)
    _mark(timings, "c.walk_done")
    if not last:
        _count("send")
        dist.send(out, dst=dist.get_global_rank(group, rank + 1) if group is not None else rank + 1, group=group)
        out = torch.empty(3, dtype=torch.float32, device=comm_dev)
    if not broadcast:
        return out.cpu().numpy().astype(np.float32) if last else None
    src = dist.get_global_rank(group, world - 1) if group is not None else world - 1
    _count("broadcast")
    dist.broadcast(out, src=src, group=group)
    _mark(timings, "c.bcast")
    return out.cpu().numpy().astype(np.float32)


def global_rows(local_row, n_own, group=None):
    """Global row numbers of a tile whose rows are numbered relative to its first OWNED row: the owned blocks follow
    each other in rank order, so the base is the exclusive prefix of n_own over the ranks (one all_gather of one
    integer per rank - what reading the LAS headers of a tile stream gives).  Returns (rows int64, total rows)."""
    group = _pg(group)
    if not _multi(group):
        return local_row, int(n_own)
    world, rank = dist.get_world_size(group), dist.get_rank(group)
    dev = local_row.device if dist.get_backend(group) == "nccl" else torch.device("cpu")
    mine = torch.tensor([int(n_own)], dtype=torch.int64, device=dev)
    out = _all_gather_block(mine, group).reshape(-1).cpu().tolist()
    return local_row + int(sum(out[:rank])), int(sum(out))


def tiled_step(tile, tile_rows, own, total_n, x_lo, x_hi, eps=8.0, min_samples=80, pct=25.0, offset=3.0,
               fallback_offset=1.0, min_keep=1000, halo=None, group=None, timings=None):
    """Stages B + C of the hot path over ONE cloud that is spread over the ranks (BASELINE config 4): every rank
    holds a consecutive file-order shard that is also an x-tile, extended by a halo of at least 2*eps.

    tile      : float32 [n_t,3] device tensor, this rank's rows (halo included) in file order
    tile_rows : int64 [n_t] global row of every tile row, ascending
    own       : (a, b): the tile rows [a, b) are the ones this rank owns (left halo | own | right halo); over the
                ranks the owned rows are the consecutive blocks [base_r, base_r + n_r) of the cloud, in rank order
    total_n   : rows of the whole cloud
    x_lo, x_hi: this rank's own x-range in the RAW frame (the tile edges)
    Returns dict(centroid f32[3], threshold f32, used_fallback, points (kept, centred, halo included), rows (global),
    own (bool per kept point), labels int32 (valid where own), nclusters (global)).

      centroid   tiles.sharded_centroid over the owned rows       utils/tower_extraction.py:63
      threshold  tiles.shared_percentile over the owned z + 3.0   :82-83 (fallback + 1.0 below 1000 survivors, :87-89)
      filter     ops.filter_gt on tile + halo with those values    :64,84
      cluster    tiles.cluster_tiled (global DBSCAN, no 50 000-row chunks: north star / SURVEY 8e-ii)
    """
    group = _pg(group)
    import time
    from . import ops
    t0 = time.perf_counter()
    a, b = int(own[0]), int(own[1])
    own_rows = tile[a:b]                                   # a view: the owned rows are one slice of the tile
    _mark(timings, "start")
    shard = HipMeanShard(own_rows, want_zcol=True)         # tables + a contiguous z column in one pass over the rows
    multi = _multi(group)
    # Collectives of one step (one rank per GPU): the two hops of the centroid chain (recv, send), ONE all-reduce
    # that carries the first histogram of the percentile, the value / NaN counts AND the centroid (the last rank's
    # float32 bits, zeros elsewhere: a broadcast for free), the two further histogram passes, and ONE all_gather
    # for the cluster tables, both strips and the survivor counts.  The fallback decision (< min_keep survivors at
    # the ordinary offset) is taken from the counts that ride in that last exchange: the filter and the fit run at
    # the ordinary offset first and are repeated at the fallback offset only in that (rare) case.
    select = HipSelect()
    hist0 = select.hist(shard.zcol, 0, 0)                  # does not depend on the centroid: computed beside the chain
    mine = sharded_centroid(own_rows, total_n, shard=shard, group=group, timings=timings, broadcast=not multi)
    _mark(timings, "centroid")
    cen = {}

    def take_centroid(words):                              # the summed ride-along words: the last rank's centroid bits
        cen["c"] = np.asarray(words, dtype=np.int64).astype(np.uint32).view(np.float32).copy()
        return cen["c"][2]

    if multi:
        bits = (np.zeros(3, dtype=np.int64) if mine is None
                else np.asarray(mine, dtype=np.float32).view(np.uint32).astype(np.int64))
        base = shared_percentile(shard.zcol, pct, sub=take_centroid, select=select, group=group, ride=bits,
                                 first_hist=hist0)
        centroid = cen["c"]
    else:
        centroid = mine
        base = shared_percentile(shard.zcol, pct, sub=centroid[2], select=select, group=group, first_hist=hist0)
    _mark(timings, "percentile")
    cx = float(centroid[0])                                # the kept points are centred: so are the edges

    def attempt(off):
        thr = np.float32(base + np.float32(off))
        kept = ops.filter_gt(tile, centroid, thr, want_index=True)
        loc = kept["index"].long()
        own_k = (loc >= a) & (loc < b)
        mine_kept = int(own_k.sum())
        t1 = time.perf_counter()
        _mark(timings, "filter")
        labels, K, words = cluster_tiled(kept["points"], tile_rows[loc], own_k, float(x_lo) - cx, float(x_hi) - cx,
                                         eps, min_samples, halo=halo, group=group, timings=timings,
                                         extra=(mine_kept,))
        return thr, kept, loc, own_k, int(np.asarray(words)[:, 0].sum()), labels, K, t1

    thr, kept, loc, own_k, survivors, labels, K, t1 = attempt(offset)
    used_fallback = False
    if survivors < int(min_keep):                          # utils/tower_extraction.py:87-89, on the WHOLE cloud's count
        thr, kept, loc, own_k, survivors, labels, K, t1 = attempt(fallback_offset)
        used_fallback = True
    if timings is not None and "trace" not in timings:
        timings["filter_ms"] = timings.get("filter_ms", 0.0) + 1e3 * (t1 - t0)
    return dict(centroid=centroid, threshold=thr, used_fallback=used_fallback, points=kept["points"],
                rows=tile_rows[loc], own=own_k, labels=labels, nclusters=int(K), survivors=survivors)
