"""Tensor-level operators: PyTorch-ROCm tensors in, C-ABI calls underneath.

torch is used for device memory, streams and (in tiles.py) torch.distributed only; all
arithmetic happens in libpch_hip.so.  Every function requires CUDA(HIP) tensors and raises
if the extension or a GPU is missing - there is no CPU fallback here by design.

One function per reference library call:
  voxel_downsample  <- open3d voxel_down_sample per chunk   ui/import_PC.py:8-13,45-58
  mean_seq_f32      <- np.mean(raw, axis=0)                  utils/tower_extraction.py:63
  percentile_f32    <- np.percentile(z, 25)                  utils/tower_extraction.py:83
  ground_filter     <- centring + percentile filter          utils/tower_extraction.py:63-64,82-89
  dbscan            <- chunked sklearn DBSCAN + label offset utils/tower_extraction.py:96-117
  segment_by_label  <- per-label boolean masks               utils/tower_extraction.py:125,131-134
"""
from __future__ import annotations

import threading

import torch

from . import _lib

_tls = threading.local()


def _stream():
    return torch.cuda.current_stream().cuda_stream


def _ptr(t):
    return 0 if t is None else t.data_ptr()


def _need_cuda(t, dtype, name):
    if not isinstance(t, torch.Tensor) or not t.is_cuda:
        raise TypeError(f"{name} must be a CUDA/HIP tensor (no CPU fallback in the product path)")
    if t.dtype != dtype:
        raise TypeError(f"{name} must be {dtype}, got {t.dtype}")
    return t.contiguous()


def _workspace(nbytes, device):
    """Grow-only per-thread, per-device scratch buffer."""
    cache = getattr(_tls, "ws", None)
    if cache is None:
        cache = _tls.ws = {}
    key = device.index if device.index is not None else torch.cuda.current_device()
    buf = cache.get(key)
    if buf is None or buf.numel() < nbytes:
        cache[key] = None
        buf = None
        buf = torch.empty(int(nbytes) + 256, dtype=torch.uint8, device=device)
        cache[key] = buf
    return buf


def release_workspace():
    _tls.ws = {}


def set_profiling(enable, only=None):
    """Enables per-kernel hipEvent timing inside the library; ``only`` (iterable of kernel names)
    restricts the recording so that the host cost of the event records stays negligible."""
    L = _lib.lib()
    L.pch_set_profiling_filter(",".join(only).encode() if only else None)
    L.pch_set_profiling(1 if enable else 0)


def get_profile():
    return _lib.get_profile()


# ---------------------------------------------------------------------------- stage A
def voxel_downsample(xyz, voxel_size, chunk_size=0):
    """Per-chunk voxel-grid downsample.  Returns (idx int32 [m,3], mean f64 [m,3],
    count int32 [m], chunk_offsets int64 [nchunks+1]); rows grouped by chunk; the order inside a
    chunk is deterministic but not sorted (include/pch_hip.h; Open3D's own is unspecified).  Synchronises (reads m)."""
    L = _lib.lib()
    xyz = _need_cuda(xyz, torch.float64, "xyz").reshape(-1, 3)
    n = xyz.shape[0]
    dev = xyz.device
    cs = int(chunk_size) if chunk_size and chunk_size > 0 else max(n, 1)
    nchunks = max(1, -(-n // cs))
    with torch.cuda.device(dev):
        idx = torch.empty((n, 3), dtype=torch.int32, device=dev)
        mean = torch.empty((n, 3), dtype=torch.float64, device=dev)
        count = torch.empty((n,), dtype=torch.int32, device=dev)
        offs = torch.zeros((nchunks + 1,), dtype=torch.int64, device=dev)
        m_dev = torch.zeros((1,), dtype=torch.int64, device=dev)
        nb = L.pch_voxel_downsample_ws_bytes(n, cs)
        ws = _workspace(nb, dev)
        _lib.check(L.pch_voxel_downsample_f64(_ptr(xyz), n, float(voxel_size), cs, _ptr(idx), _ptr(mean),
                                              _ptr(count), _ptr(offs), _ptr(m_dev), _ptr(ws), ws.numel(),
                                              _stream()))
        m = _lib.check_count(m_dev.item(), "voxel_downsample")
    return idx[:m], mean[:m], count[:m], offs


def las_records_xyz(records_u8, n, record_len):
    """Raw LAS point records (uint8 device tensor) -> int32 [n,3] X,Y,Z."""
    L = _lib.lib()
    rec = _need_cuda(records_u8, torch.uint8, "records")
    if rec.numel() < int(n) * int(record_len):
        raise ValueError("record buffer shorter than n * record_len")
    out = torch.empty((int(n), 3), dtype=torch.int32, device=rec.device)
    with torch.cuda.device(rec.device):
        _lib.check(L.pch_las_records_xyz_i32(_ptr(rec), int(n), int(record_len), _ptr(out), _stream()))
    return out


def las_scale(XYZ_i32, scales, offsets):
    """laspy scaled view: int32 [n,3] -> float64 [n,3] (X*scale+offset)."""
    import ctypes as C
    L = _lib.lib()
    X = _need_cuda(XYZ_i32, torch.int32, "XYZ").reshape(-1, 3)
    out = torch.empty(X.shape, dtype=torch.float64, device=X.device)
    sc = (C.c_double * 3)(*[float(v) for v in scales])
    of = (C.c_double * 3)(*[float(v) for v in offsets])
    with torch.cuda.device(X.device):
        _lib.check(L.pch_las_scale_i32_f64(_ptr(X), X.shape[0], C.cast(sc, C.c_void_p),
                                           C.cast(of, C.c_void_p), _ptr(out), _stream()))
    return out


def las_unscale(xyz_f64, scales, offsets):
    """laspy coordinate setter: float64 [n,3] -> int32 [n,3] (rint((v-offset)/scale))."""
    import ctypes as C
    L = _lib.lib()
    x = _need_cuda(xyz_f64, torch.float64, "xyz").reshape(-1, 3)
    out = torch.empty(x.shape, dtype=torch.int32, device=x.device)
    sc = (C.c_double * 3)(*[float(v) for v in scales])
    of = (C.c_double * 3)(*[float(v) for v in offsets])
    with torch.cuda.device(x.device):
        _lib.check(L.pch_las_unscale_f64_i32(_ptr(x), x.shape[0], C.cast(sc, C.c_void_p),
                                             C.cast(of, C.c_void_p), _ptr(out), _stream()))
    return out


# ---------------------------------------------------------------------------- stage B
def cast_f32(x_f64):
    L = _lib.lib()
    x = _need_cuda(x_f64, torch.float64, "x")
    out = torch.empty(x.shape, dtype=torch.float32, device=x.device)
    with torch.cuda.device(x.device):
        _lib.check(L.pch_cast_f64_f32(_ptr(x), x.numel(), _ptr(out), _stream()))
    return out


def mean_seq_f32(xyz, serial=False):
    """np.mean(xyz, axis=0) for C-order float32 [n,3], bit exact.  Returns float32 [3].
    ``serial`` selects the one-workgroup element-by-element variant (cross-check only)."""
    L = _lib.lib()
    xyz = _need_cuda(xyz, torch.float32, "xyz").reshape(-1, 3)
    out = torch.empty((3,), dtype=torch.float32, device=xyz.device)
    with torch.cuda.device(xyz.device):
        if serial:
            _lib.check(L.pch_mean_seq_serial_f32(_ptr(xyz), xyz.shape[0], _ptr(out), _stream()))
            return out
        nb = L.pch_mean_seq_f32_ws_bytes(xyz.shape[0])
        ws = _workspace(nb, xyz.device)
        _lib.check(L.pch_mean_seq_f32(_ptr(xyz), xyz.shape[0], _ptr(out), _ptr(ws), ws.numel(), _stream()))
    return out


def mean_seq_partial_f32(xyz, sum_in=None, total_n=0):
    """One file-order shard of np.mean(., axis=0): continues the float32 running sums ``sum_in`` (float32 [3] device
    tensor, None = +0.0) over the rows of ``xyz``.  total_n == 0: returns the running sums after the rows (hand them
    to the next shard); total_n > 0: returns sums / float32(total_n), the centroid (last shard).  float32 [3]."""
    sh = MeanShard(xyz)
    return sh.walk(sum_in, total_n)


class MeanShard:
    """A file-order shard of the sequential float32 column sums, in two phases (pch_mean_seq_partial_f32): the
    constructor builds the shard's summary tables (the passes over the rows; independent of what comes before the
    shard, so every rank does this at once), ``walk`` then continues GIVEN running sums over them (the short serial
    part that is chained from rank to rank).  The tables live in a workspace that belongs to this object."""

    def __init__(self, xyz, want_zcol=False):
        L = _lib.lib()
        self.xyz = _need_cuda(xyz, torch.float32, "xyz").reshape(-1, 3)
        self.n, self.device = self.xyz.shape[0], self.xyz.device
        with torch.cuda.device(self.device):
            self.workspace = torch.empty(int(L.pch_mean_seq_f32_ws_bytes(self.n)) + 256, dtype=torch.uint8,
                                         device=self.device)
            # contiguous copy of the z column, written by the pass that reads the rows anyway
            self.zcol = torch.empty((self.n,), dtype=torch.float32, device=self.device) if want_zcol else None
            _lib.check(L.pch_mean_seq_partial_f32(_ptr(self.xyz), self.n, 0, 0, 0, _ptr(self.zcol), 1,
                                                  _ptr(self.workspace), self.workspace.numel(), _stream()))

    def walk(self, sum_in=None, total_n=0):
        L = _lib.lib()
        if sum_in is not None:
            sum_in = _need_cuda(sum_in, torch.float32, "sum_in").reshape(3)
        out = torch.empty((3,), dtype=torch.float32, device=self.device)
        with torch.cuda.device(self.device):
            _lib.check(L.pch_mean_seq_partial_f32(_ptr(self.xyz), self.n, _ptr(sum_in), int(total_n), _ptr(out), 0, 2,
                                                  _ptr(self.workspace), self.workspace.numel(), _stream()))
        return out


def percentile_f32(values, q_percent, sub=None):
    """np.percentile(values - sub, q) for a float32 vector (any stride).  Returns float32 [1]."""
    L = _lib.lib()
    if not values.is_cuda or values.dtype != torch.float32 or values.dim() != 1:
        raise TypeError("values must be a 1-D float32 CUDA tensor")
    n = values.shape[0]
    stride = values.stride(0) if n > 1 else 1
    out = torch.empty((1,), dtype=torch.float32, device=values.device)
    with torch.cuda.device(values.device):
        nb = L.pch_percentile_f32_ws_bytes(n)
        ws = _workspace(nb, values.device)
        _lib.check(L.pch_percentile_f32(values.data_ptr(), n, stride, _ptr(sub), float(q_percent),
                                        _ptr(out), _ptr(ws), ws.numel(), _stream()))
    return out


def _strided_1d(values):
    if not values.is_cuda or values.dtype != torch.float32 or values.dim() != 1:
        raise TypeError("values must be a 1-D float32 CUDA tensor")
    n = values.shape[0]
    return n, (values.stride(0) if n > 1 else 1)


def select_hist(values, pass_no, prefix=0):
    """One histogram pass of the radix select over this tensor's values (see pch_select_hist_f32).
    Returns (hist int64 [4096] on the host, NaN count)."""
    L = _lib.lib()
    n, stride = _strided_1d(values)
    dev = values.device
    with torch.cuda.device(dev):
        hist = torch.empty((4096,), dtype=torch.int32, device=dev)
        nan = torch.zeros((1,), dtype=torch.int64, device=dev)
        ws = _workspace(L.pch_percentile_f32_ws_bytes(n), dev)
        _lib.check(L.pch_select_hist_f32(values.data_ptr() if n else 0, n, stride, int(pass_no), int(prefix) & 0xFFFFFFFF,
                                         _ptr(hist), _ptr(nan), _ptr(ws), ws.numel(), _stream()))
    return hist.cpu().numpy().view("<u4").astype("int64"), int(nan.item())


def select_min_above(values, key):
    """Smallest order-preserving key of this tensor's values that is > key (0xFFFFFFFF if none)."""
    L = _lib.lib()
    n, stride = _strided_1d(values)
    dev = values.device
    with torch.cuda.device(dev):
        out = torch.empty((1,), dtype=torch.int32, device=dev)
        ws = _workspace(L.pch_percentile_f32_ws_bytes(n), dev)
        _lib.check(L.pch_select_min_above_f32(values.data_ptr() if n else 0, n, stride, int(key) & 0xFFFFFFFF, _ptr(out),
                                              _ptr(ws), ws.numel(), _stream()))
    return int(out.cpu().numpy().view("<u4")[0])


def filter_gt(raw, centroid, threshold, want_index=True):
    """points = raw - centroid; keep z > threshold, order preserving, with GIVEN float32 centroid and threshold
    (utils/tower_extraction.py:64,84).  Returns dict(points, index | None, count, aabb).  Synchronises."""
    import ctypes as C
    import numpy as np
    L = _lib.lib()
    raw = _need_cuda(raw, torch.float32, "raw").reshape(-1, 3)
    n = raw.shape[0]
    dev = raw.device
    cen = (C.c_float * 3)(*[float(np.float32(v)) for v in centroid])
    with torch.cuda.device(dev):
        out_points = torch.empty((n, 3), dtype=torch.float32, device=dev)
        out_index = torch.empty((n,), dtype=torch.int32, device=dev) if want_index else None
        cnt = torch.zeros((1,), dtype=torch.int64, device=dev)
        aabb = torch.zeros((6,), dtype=torch.float32, device=dev)
        ws = _workspace(L.pch_filter_gt_ws_bytes(n), dev)
        _lib.check(L.pch_filter_gt_f32(_ptr(raw), n, C.cast(cen, C.c_void_p), float(np.float32(threshold)), _ptr(out_points),
                                       _ptr(out_index), _ptr(cnt), _ptr(aabb), _ptr(ws), ws.numel(), _stream()))
        m = _lib.check_count(cnt.item(), "filter_gt")
    return dict(points=out_points[:m], index=None if out_index is None else out_index[:m], count=m,
                aabb=aabb.cpu().numpy())


def ground_filter(raw, pct=25.0, offset=3.0, fallback_offset=1.0, min_keep=1000, want_index=True):
    """Fused stage B on float32 [n,3].  Returns dict(points [n_f,3] f32 (centred, file
    order), index int32 [n_f] | None, centroid f32[3] (host np), base, threshold,
    used_fallback, aabb (host, 6 floats), count).  Synchronises (reads n_f)."""
    L = _lib.lib()
    raw = _need_cuda(raw, torch.float32, "raw").reshape(-1, 3)
    n = raw.shape[0]
    dev = raw.device
    with torch.cuda.device(dev):
        out_points = torch.empty((n, 3), dtype=torch.float32, device=dev)
        out_index = torch.empty((n,), dtype=torch.int32, device=dev) if want_index else None
        scal = torch.zeros((18,), dtype=torch.float32, device=dev)     # [0:8] scalars, [8:14] aabb, [16:18] count (int64)
        nb = L.pch_ground_filter_ws_bytes(n)
        ws = _workspace(nb, dev)
        _lib.check(L.pch_ground_filter_f32(_ptr(raw), n, float(pct), float(offset), float(fallback_offset),
                                           int(min_keep), _ptr(out_points), _ptr(out_index), _ptr(scal),
                                           scal.data_ptr() + 64, scal.data_ptr() + 32, _ptr(ws), ws.numel(),
                                           _stream()))
        host = scal.cpu().numpy()           # one D2H copy, synchronises the stream
        nf = _lib.check_count(host[16:18].view("<i8")[0], "ground_filter")
    return dict(points=out_points[:nf], index=None if out_index is None else out_index[:nf],
                centroid=host[0:3].copy(), base=host[3], threshold=host[4],
                used_fallback=bool(host[5] != 0.0), count_at_offset=int(host[6:7].view("<u4")[0]),
                aabb=host[8:14].copy(), count=nf)


# ---------------------------------------------------------------------------- stage C
def dbscan(xyz, eps=8.0, min_samples=80, chunk_size=50000, aabb=None, want_core=False):
    """Chunked exact DBSCAN on float32 [n,3].  Returns (labels int32 [n], core uint8 [n] | None,
    nclusters int).  Synchronises."""
    import ctypes as C
    L = _lib.lib()
    xyz = _need_cuda(xyz, torch.float32, "xyz").reshape(-1, 3)
    n = xyz.shape[0]
    dev = xyz.device
    with torch.cuda.device(dev):
        labels = torch.empty((n,), dtype=torch.int32, device=dev)
        core = torch.empty((n,), dtype=torch.uint8, device=dev) if want_core else None
        ncl = torch.zeros((1,), dtype=torch.int32, device=dev)
        nb = L.pch_dbscan_ws_bytes(n)
        ws = _workspace(nb, dev)
        box = None
        if aabb is not None:
            box = (C.c_float * 6)(*[float(v) for v in aabb])
        _lib.check(L.pch_dbscan_f32(_ptr(xyz), n, float(eps), int(min_samples), int(chunk_size),
                                    None if box is None else C.cast(box, C.c_void_p), _ptr(labels),
                                    _ptr(core), _ptr(ncl), _ptr(ws), ws.numel(), _stream()))
        k = _lib.check_count(ncl.item(), "dbscan")
    return labels, core, k


class DbscanFit:
    """One exact DBSCAN whose grid stays alive: labels, core flags and cluster count of ``ops.dbscan`` plus a
    workspace that belongs to this fit alone, so that ``first_core_rows`` and ``relabel`` may follow after any
    number of other ops (the shared per-thread scratch buffer of ``_workspace`` is never used here).  The library
    remembers one fit per thread: a second fit on the same thread retires the first, and a retired fit's
    continuation calls raise (PCH_ERR_ARG) instead of reading another run's grid."""

    def __init__(self, xyz, eps=8.0, min_samples=80, chunk_size=0, aabb=None):
        import ctypes as C
        L = _lib.lib()
        xyz = _need_cuda(xyz, torch.float32, "xyz").reshape(-1, 3)
        n = xyz.shape[0]
        dev = xyz.device
        self.n, self.device = n, dev
        with torch.cuda.device(dev):
            self.labels = torch.empty((n,), dtype=torch.int32, device=dev)
            self.core = torch.empty((n,), dtype=torch.uint8, device=dev)
            ncl = torch.zeros((1,), dtype=torch.int32, device=dev)
            self.workspace = torch.empty(int(L.pch_dbscan_ws_bytes(n)) + 256, dtype=torch.uint8, device=dev)
            box = None if aabb is None else (C.c_float * 6)(*[float(v) for v in aabb])
            _lib.check(L.pch_dbscan_f32(_ptr(xyz), n, float(eps), int(min_samples), int(chunk_size),
                                        None if box is None else C.cast(box, C.c_void_p), _ptr(self.labels),
                                        _ptr(self.core), _ptr(ncl), _ptr(self.workspace), self.workspace.numel(),
                                        _stream()))
            self.nclusters = _lib.check_count(ncl.item(), "dbscan")

    def first_core_rows(self):
        """Smallest core row of every cluster (int32 [nclusters], ascending with the cluster id)."""
        L = _lib.lib()
        out = torch.empty((self.nclusters,), dtype=torch.int32, device=self.device)
        with torch.cuda.device(self.device):
            _lib.check(L.pch_dbscan_first_core_rows_i32(self.n, _ptr(out), _ptr(self.workspace),
                                                        self.workspace.numel(), _stream()))
        return out

    def pair_stats(self):
        """Tallies of the radius-count kernel of THIS fit, if it ran with ops.set_pair_counting(True): dict(pair_tests,
        lane_slots, cells_tested, tiles_staged) (pch_dbscan_pair_stats).  Synchronises."""
        import ctypes as C
        L = _lib.lib()
        out = (C.c_uint64 * 4)()
        with torch.cuda.device(self.device):
            _lib.check(L.pch_dbscan_pair_stats(self.n, C.cast(out, C.c_void_p), _ptr(self.workspace),
                                               self.workspace.numel(), _stream()))
        return dict(pair_tests=int(out[0]), lane_slots=int(out[1]), cells_tested=int(out[2]), tiles_staged=int(out[3]))

    def strip_pairs(self, x_lo, x_hi, cap=4096):
        """(int32 [cap,2] device buffer of (local row, cluster id) pairs, int32 [1] device count): one pair per grid
        cell that holds a core point with x_lo <= x < x_hi (pch_dbscan_strip_pairs_i32).  Asynchronous: the count
        is read by whoever needs it (it may exceed cap - then call again with a larger cap).  Before relabel."""
        import numpy as np
        L = _lib.lib()
        pairs = torch.empty((int(cap), 2), dtype=torch.int32, device=self.device)
        count = torch.empty((1,), dtype=torch.int32, device=self.device)
        with torch.cuda.device(self.device):
            _lib.check(L.pch_dbscan_strip_pairs_i32(self.n, float(np.float32(x_lo)), float(np.float32(x_hi)), int(cap),
                                                    _ptr(pairs), _ptr(count), _ptr(self.workspace),
                                                    self.workspace.numel(), _stream()))
        return pairs, count

    def relabel(self, cluster_map):
        """Core points take cluster_map[old id], border points are decided again as the smallest new id among
        their core neighbours.  ``labels`` is updated in place and returned."""
        L = _lib.lib()
        cmap = _need_cuda(cluster_map, torch.int32, "cluster_map")
        with torch.cuda.device(self.device):
            _lib.check(L.pch_dbscan_relabel_i32(_ptr(cmap), cmap.numel(), self.n, _ptr(self.labels),
                                                _ptr(self.workspace), self.workspace.numel(), _stream()))
        return self.labels


def set_pair_counting(enable):
    """Counting variant of the radius-count kernel for the fits this thread makes from now on (measurement only)."""
    _lib.lib().pch_dbscan_set_pair_counting(1 if enable else 0)


def set_dbscan_sort_mode(mode):
    """Cell sort of ops.dbscan: "auto" (by chunk count), "chunk" (one workgroup per chunk) or "global"
    (one radix sort).  Same results either way; tests compare them."""
    _lib.lib().pch_dbscan_set_sort_mode({"auto": 0, "chunk": 1, "global": 2}[mode])


def first_nonfinite_row(xyz):
    """Index of the first row of float32 [n,3] holding NaN/inf, -1 if none (what makes sklearn's
    DBSCAN.fit reject a chunk).  Synchronises."""
    L = _lib.lib()
    xyz = _need_cuda(xyz, torch.float32, "xyz").reshape(-1, 3)
    out = torch.empty((1,), dtype=torch.int64, device=xyz.device)
    with torch.cuda.device(xyz.device):
        _lib.check(L.pch_first_nonfinite_row_f32(_ptr(xyz), xyz.shape[0], _ptr(out), _stream()))
    return int(out.item())


# ---------------------------------------------------------------------- stages B + C + D0
_nf_hint = {}          # device index -> fraction of points the previous call kept (sizes the next workspace)


def tower_clusters(raw, eps=8.0, min_samples=80, chunk_size=50000, pct=25.0, offset=3.0,
                   fallback_offset=1.0, min_keep=1000, want_index=False, segment=True, k_cap=65536):
    """ground_filter + dbscan + segment_by_label behind one library call
    (pch_tower_clusters_f32): the host language is not visited between the stages.
    Returns (ground dict as ops.ground_filter, labels int32 [n_f], nclusters,
    perm | None, offsets | None, stats | None)."""
    import ctypes as C
    import numpy as np
    L = _lib.lib()
    raw = _need_cuda(raw, torch.float32, "raw").reshape(-1, 3)
    n = raw.shape[0]
    dev = raw.device
    key = dev.index if dev.index is not None else torch.cuda.current_device()
    hint = _nf_hint.get(key)
    # output / workspace capacity for the kept points: the previous call's kept fraction with headroom, a
    # quarter of the input on the first call; a call that keeps more is repeated once, sized for n
    guess = n // 4 if hint is None else int(hint * n * 1.25) + 1024     # hint: kept fraction of the last tile
    caps = [min(n, max(guess, 1 << 16)), n]
    with torch.cuda.device(dev):
        out_points = torch.empty((n, 3), dtype=torch.float32, device=dev)
        out_index = torch.empty((n,), dtype=torch.int32, device=dev) if want_index else None
        info = _lib.TowerClustersInfo()
        for nf_cap in caps:
            labels = torch.empty((nf_cap,), dtype=torch.int32, device=dev)
            perm = torch.empty((nf_cap,), dtype=torch.int32, device=dev) if segment else None
            offsets = torch.empty((k_cap + 1,), dtype=torch.int64, device=dev) if segment else None   # [0..k] are written
            stats = torch.empty((k_cap, 8), dtype=torch.float32, device=dev) if segment else None
            ws = _workspace(L.pch_tower_clusters_ws_bytes(n, nf_cap, k_cap), dev)
            rc = L.pch_tower_clusters_f32(_ptr(raw), n, float(pct), float(offset), float(fallback_offset),
                                          int(min_keep), float(eps), int(min_samples), int(chunk_size),
                                          _ptr(out_points), _ptr(out_index), _ptr(labels), _ptr(perm),
                                          _ptr(offsets), _ptr(stats), nf_cap, k_cap, C.addressof(info),
                                          _ptr(ws), ws.numel(), _stream())
            if rc == -2 and info.count > nf_cap and nf_cap < n:
                continue                                   # kept more than the hint allowed for: once more, sized n
            break
        nf, k = int(info.count), int(info.nclusters)
        if rc == -4 and segment and k > k_cap:             # more clusters than k_cap: group separately
            perm, offsets, stats = segment_by_label(labels[:nf], out_points[:nf], k)
            rc = 0
        _lib.check(rc)
    _nf_hint[key] = nf / max(n, 1)
    ground = dict(points=out_points[:nf], index=None if out_index is None else out_index[:nf],
                  centroid=np.array(info.centroid, dtype=np.float32), base=np.float32(info.base),
                  threshold=np.float32(info.threshold), used_fallback=bool(info.used_fallback),
                  count_at_offset=int(info.count_at_offset), aabb=np.array(info.aabb, dtype=np.float32), count=nf)
    if not segment:
        return ground, labels[:nf], k, None, None, None
    return ground, labels[:nf], k, perm[:nf], offsets[:k + 1], stats[:k]


def strip_lattice_reps(xyz, rows, labels, core, strips, eps, cap=4096):
    """One (global row, local cluster) pair per lattice cell of every strip (x_from, x_to) of ``strips`` (at most two):
    the smallest row among the cell's core points with x in the strip (pch_strip_lattice_reps_f32; the lattice is
    anchored at the frame's origin, so neighbouring tiles name the same rows).  xyz float32 [n,3], rows int64 [n]
    ascending, labels int32 [n], core bool / uint8 [n] - device tensors.  Returns a list of int64 [m,2] tensors (row,
    label), sorted by row, one per strip.  Synchronises (reads the counts)."""
    import ctypes as C
    import numpy as np
    L = _lib.lib()
    xyz = _need_cuda(xyz, torch.float32, "xyz").reshape(-1, 3)
    rows = _need_cuda(rows, torch.int64, "rows")
    labels = _need_cuda(labels, torch.int32, "labels")
    core = core.contiguous()
    if not core.is_cuda or core.dtype not in (torch.bool, torch.uint8):
        raise TypeError("core must be a bool / uint8 CUDA/HIP tensor (no CPU fallback in the product path)")
    core = core.view(torch.uint8)
    n, dev = xyz.shape[0], xyz.device
    if len(strips) > 2:
        raise ValueError("at most two strips per call")
    if not (rows.numel() == labels.numel() == core.numel() == n):
        raise ValueError("rows / labels / core must have one entry per point")
    flat = (C.c_float * (2 * max(len(strips), 1)))()
    for k, (a, b) in enumerate(strips):
        flat[2 * k], flat[2 * k + 1] = float(np.float32(a)), float(np.float32(b))
    cap = int(cap)
    with torch.cuda.device(dev):
        while True:
            out_rows = torch.empty((2, max(cap, 1)), dtype=torch.int64, device=dev)
            out_lab = torch.empty((2, max(cap, 1)), dtype=torch.int32, device=dev)
            cnt = torch.empty((4,), dtype=torch.int32, device=dev)
            ws = _workspace(L.pch_strip_lattice_reps_ws_bytes(cap), dev)
            _lib.check(L.pch_strip_lattice_reps_f32(_ptr(xyz), _ptr(rows), _ptr(labels), _ptr(core), n, len(strips),
                                                    C.cast(flat, C.c_void_p), float(eps), cap, _ptr(out_rows),
                                                    _ptr(out_lab), _ptr(cnt), _ptr(ws), ws.numel(), _stream()))
            c = cnt.cpu().tolist()
            if c[2] & 1:
                raise ValueError("strip_lattice_reps: coordinates beyond 2^20 lattice cells from the origin")
            need = max(c[0], c[1])
            if need <= cap and not (c[2] & 2):
                break
            cap = max(2 * cap, need)                       # a strip with more cells than the buffer: once more
    out = []
    for k in range(len(strips)):
        r, lab = out_rows[k, :c[k]], out_lab[k, :c[k]].to(torch.int64)
        order = torch.argsort(r)
        out.append(torch.stack([r[order], lab[order]], dim=1))
    return out


# ---------------------------------------------------------------------------- stage D0
def segment_by_label(labels, xyz, nclusters):
    """Returns (perm int32 [n], offsets int64 [K+1], stats f32 [K,8] (min xyz, max xyz, 0, 0))."""
    L = _lib.lib()
    labels = _need_cuda(labels, torch.int32, "labels")
    xyz = _need_cuda(xyz, torch.float32, "xyz").reshape(-1, 3)
    n = labels.shape[0]
    dev = labels.device
    K = int(nclusters)
    with torch.cuda.device(dev):
        perm = torch.empty((n,), dtype=torch.int32, device=dev)
        offsets = torch.zeros((K + 1,), dtype=torch.int64, device=dev)
        stats = torch.zeros((max(K, 1), 8), dtype=torch.float32, device=dev)
        nb = L.pch_segment_by_label_ws_bytes(n, K)
        ws = _workspace(nb, dev)
        _lib.check(L.pch_segment_by_label(_ptr(labels), _ptr(xyz), n, K, _ptr(perm), _ptr(offsets),
                                          _ptr(stats), _ptr(ws), ws.numel(), _stream()))
    return perm, offsets, stats[:K]


# ---------------------------------------------------------------------------- stage D1, fast mode
def obb_shell(xyz, perm, offsets, nclusters):
    """uint8 [offsets[K]] in grouped order: 1 for the points of every cluster that can be vertices of its
    convex hull (a superset of them), 0 for points strictly inside it (pch_obb_shell_f32)."""
    L = _lib.lib()
    xyz = _need_cuda(xyz, torch.float32, "xyz").reshape(-1, 3)
    perm = _need_cuda(perm, torch.int32, "perm")
    offsets = _need_cuda(offsets, torch.int64, "offsets")
    K = int(nclusters)
    if offsets.numel() != K + 1:
        raise ValueError("offsets must hold nclusters + 1 entries")
    dev = xyz.device
    with torch.cuda.device(dev):
        ng = int(offsets[K].item()) if K else 0
        keep = torch.empty((ng,), dtype=torch.uint8, device=dev)
        if ng:
            ws = _workspace(L.pch_obb_shell_ws_bytes(K), dev)
            _lib.check(L.pch_obb_shell_f32(_ptr(xyz), _ptr(perm), _ptr(offsets), K, ng, _ptr(keep), _ptr(ws),
                                           ws.numel(), _stream()))
    return keep


def obb_min_boxes(verts, vert_offsets, tris, tri_offsets, sorted_extents=False, nthreads=0):
    """Host arrays in, host arrays out: (to_origin [H,4,4], extents [H,3], status [H]) of H convex hulls
    (pch_obb_min_boxes_f64)."""
    import numpy as np
    L = _lib.lib()
    verts = np.ascontiguousarray(verts, dtype=np.float64).reshape(-1, 3)
    tris = np.ascontiguousarray(tris, dtype=np.int32).reshape(-1, 3)
    vo = np.ascontiguousarray(vert_offsets, dtype=np.int64)
    to = np.ascontiguousarray(tri_offsets, dtype=np.int64)
    H = len(vo) - 1
    if len(to) != H + 1 or (H >= 0 and (vo[-1] != len(verts) or to[-1] != len(tris))):
        raise ValueError("offset tables do not match the vertex / triangle arrays")
    T = np.zeros((max(H, 0), 4, 4), dtype=np.float64)
    E = np.zeros((max(H, 0), 3), dtype=np.float64)
    S = np.zeros((max(H, 0),), dtype=np.int32)
    if H > 0:
        _lib.check(L.pch_obb_min_boxes_f64(verts.ctypes.data, vo.ctypes.data, tris.ctypes.data, to.ctypes.data, H,
                                           int(bool(sorted_extents)), int(nthreads), T.ctypes.data, E.ctypes.data,
                                           S.ctypes.data))
    return T, E, S


def obb_search(verts, vert_offsets, angles, angle_offsets, nthreads=0):
    """Host arrays: (best int32 [H], volumes float64 [sum nc]) - index of the candidate direction of smallest
    box volume per hull, and the volume of every candidate (pch_obb_search_f64)."""
    import numpy as np
    L = _lib.lib()
    verts = np.ascontiguousarray(verts, dtype=np.float64).reshape(-1, 3)
    angles = np.ascontiguousarray(angles, dtype=np.float64).reshape(-1, 2)
    vo = np.ascontiguousarray(vert_offsets, dtype=np.int64)
    ao = np.ascontiguousarray(angle_offsets, dtype=np.int64)
    H = len(vo) - 1
    if len(ao) != H + 1 or vo[-1] != len(verts) or ao[-1] != len(angles):
        raise ValueError("offset tables do not match the vertex / angle arrays")
    best = np.full((max(H, 0),), -1, dtype=np.int32)
    vol = np.zeros((len(angles),), dtype=np.float64)
    if H > 0:
        _lib.check(L.pch_obb_search_f64(verts.ctypes.data, vo.ctypes.data, angles.ctypes.data, ao.ctypes.data, H,
                                        int(nthreads), best.ctypes.data, vol.ctypes.data))
    return best, vol


# ---------------------------------------------------------------------------- viewer helpers
def crop_aabb(xyz, lo, hi, want_index=False):
    """points[(p >= lo).all(1) & (p <= hi).all(1)] for float64 [n,3], order preserving (test/kuangxuan.py:69-79).
    Returns points [m,3] (and source rows int64 [m]).  Synchronises (reads m)."""
    import ctypes as C
    L = _lib.lib()
    xyz = _need_cuda(xyz, torch.float64, "xyz").reshape(-1, 3)
    n = xyz.shape[0]
    dev = xyz.device
    mn = (C.c_double * 3)(*[float(v) for v in lo])
    mx = (C.c_double * 3)(*[float(v) for v in hi])
    with torch.cuda.device(dev):
        out = torch.empty((n, 3), dtype=torch.float64, device=dev)
        idx = torch.empty((n,), dtype=torch.int64, device=dev) if want_index else None
        cnt = torch.zeros((1,), dtype=torch.int64, device=dev)
        ws = _workspace(L.pch_crop_aabb_ws_bytes(n), dev)
        _lib.check(L.pch_crop_aabb_f64(_ptr(xyz), n, C.cast(mn, C.c_void_p), C.cast(mx, C.c_void_p), _ptr(out),
                                       _ptr(idx), _ptr(cnt), _ptr(ws), ws.numel(), _stream()))
        m = _lib.check_count(cnt.item(), "crop_aabb")
    return (out[:m], idx[:m]) if want_index else out[:m]


def decimate(xyz, k, seed=0, want_index=False):
    """k distinct rows of float64 [n,3] (seeded; pyGUI_towers_test.py:174-177, ui/vtk_widget.py:115-118)."""
    L = _lib.lib()
    xyz = _need_cuda(xyz, torch.float64, "xyz").reshape(-1, 3)
    n, k = xyz.shape[0], int(k)
    if k > n:
        raise ValueError("Cannot take a larger sample than population when 'replace=False'")
    dev = xyz.device
    with torch.cuda.device(dev):
        out = torch.empty((k, 3), dtype=torch.float64, device=dev)
        idx = torch.empty((k,), dtype=torch.int64, device=dev) if want_index else None
        _lib.check(L.pch_decimate_f64(_ptr(xyz), n, k, int(seed) & (2**64 - 1), _ptr(out), _ptr(idx), _stream()))
    return (out, idx) if want_index else out
