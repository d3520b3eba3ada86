"""Oracle for stage A: per-chunk voxel-grid downsample (TEST INFRASTRUCTURE).

PARITY UNPINNED: the reference delegates to Open3D
(``/root/reference/ui/import_PC.py:8-13`` -> ``PointCloud.voxel_down_sample``),
which is not installed in this image, so this file restates Open3D's published
``PointCloud::VoxelDownSample`` algorithm (open3d/geometry/PointCloud.cpp,
unpinned version):

    voxel_min_bound = min_bound - voxel_size * 0.5
    idx_i           = floor((p_i - voxel_min_bound) / voxel_size)   (int, per axis)
    acc[idx_i]     += p_i                 (float64, in point order)
    out             = acc.sum / acc.count (one point per occupied voxel)

Open3D emits voxels in ``std::unordered_map`` iteration order, which is
platform defined; parity is therefore defined on the *set* of
(voxel index, mean, count) triples per chunk.  This oracle emits the voxels of a
chunk sorted by (ix, iy, iz); the HIP path emits them grouped by chunk in an order
of its own (``canonical`` below sorts any output the oracle's way for comparison).
"""
from __future__ import annotations

import numpy as np

INT_MAX = 2147483647


def voxel_down_sample(points, voxel_size):
    """One chunk.  ``points`` (n,3) float64 -> (idx int32 (m,3), mean f64 (m,3), count int32 (m,)).

    Follows ui/import_PC.py:8-13 (the ``astype(np.float64)`` + Open3D call).
    """
    pts = np.ascontiguousarray(np.asarray(points).astype(np.float64)).reshape(-1, 3)
    voxel_size = float(voxel_size)
    if not voxel_size > 0.0:
        raise ValueError("voxel_size <= 0")            # Open3D: LogError
    n = pts.shape[0]
    if n == 0:
        return (np.zeros((0, 3), np.int32), np.zeros((0, 3), np.float64),
                np.zeros((0,), np.int32))
    lo = pts.min(axis=0)
    hi = pts.max(axis=0)
    minb = lo - voxel_size * 0.5
    maxb = hi + voxel_size * 0.5
    if voxel_size * INT_MAX < float((maxb - minb).max()):
        raise ValueError("voxel_size is too small")    # Open3D: LogError
    ref = (pts - minb) / voxel_size                    # f64 sub then f64 div, no FMA
    idx = np.floor(ref).astype(np.int64)
    # order voxels by (ix, iy, iz); stable so that in-voxel order == point order
    order = np.lexsort((idx[:, 2], idx[:, 1], idx[:, 0]))
    sidx = idx[order]
    new = np.ones(n, dtype=bool)
    new[1:] = np.any(sidx[1:] != sidx[:-1], axis=1)
    seg = np.cumsum(new) - 1                           # voxel id of each sorted point
    m = int(seg[-1]) + 1
    inv = np.empty(n, dtype=np.int64)
    inv[order] = seg                                   # voxel id per ORIGINAL point
    sums = np.zeros((m, 3), dtype=np.float64)
    # np.add.at is unbuffered and applies the updates in index order, i.e. the
    # same sequential float64 accumulation as Open3D's AccumulatedPoint::AddPoint.
    np.add.at(sums, inv, pts)
    count = np.bincount(inv, minlength=m).astype(np.int32)
    mean = sums / count[:, None].astype(np.float64)
    return sidx[new].astype(np.int32), mean, count


def voxel_down_sample_chunked(points, voxel_size, chunk_size):
    """File-order chunk loop of ui/import_PC.py:45-60: every chunk gets its own
    grid origin, chunk outputs are stacked, cross-chunk duplicates are kept.

    Returns (idx, mean, count, chunk_offsets) with ``chunk_offsets`` int64
    (nchunks+1,) giving each chunk's slice of the stacked output.
    """
    pts = np.asarray(points)
    n = pts.shape[0]
    idxs, means, counts, offs = [], [], [], [0]
    for start in range(0, n, int(chunk_size)):
        end = min(start + int(chunk_size), n)
        i, m, c = voxel_down_sample(pts[start:end], voxel_size)
        idxs.append(i); means.append(m); counts.append(c)
        offs.append(offs[-1] + len(c))
    if not idxs:
        return (np.zeros((0, 3), np.int32), np.zeros((0, 3)), np.zeros((0,), np.int32),
                np.zeros((1,), np.int64))
    return (np.vstack(idxs), np.vstack(means), np.concatenate(counts),
            np.asarray(offs, dtype=np.int64))


def canonical(idx, mean, count, chunk_offsets):
    """(idx, mean, count) with every chunk's voxels sorted by (ix, iy, iz) - the order this oracle emits.
    Parity of stage A is set equality per chunk: comparing canonical forms is exactly that,
    provided no voxel index repeats inside a chunk (checked)."""
    idx = np.asarray(idx)
    mean = np.asarray(mean)
    count = np.asarray(count)
    offs = np.asarray(chunk_offsets, dtype=np.int64)
    order = np.empty(len(count), dtype=np.int64)
    for c in range(len(offs) - 1):
        a, b = int(offs[c]), int(offs[c + 1])
        o = np.lexsort((idx[a:b, 2], idx[a:b, 1], idx[a:b, 0]))
        order[a:b] = a + o
        si = idx[a:b][o]
        if len(si) > 1:
            assert np.any(si[1:] != si[:-1], axis=1).all(), "a voxel index repeats inside a chunk"
    return idx[order], mean[order], count[order]


def las_scaled(X, scale, offset):
    """laspy's scaled view (stage A0): ``X * scale + offset`` in float64
    (laspy ScaledArrayView; ui/import_PC.py:47-48 reads chunk.x/.y/.z)."""
    return np.asarray(X, dtype=np.float64) * float(scale) + float(offset)


def las_unscale(v, scale, offset):
    """laspy's setter for .x/.y/.z (stage A3, ui/import_PC.py:61-63):
    ``np.round((v - offset) / scale)`` cast to int32."""
    return np.round((np.asarray(v, dtype=np.float64) - float(offset)) / float(scale)).astype(np.int32)
