"""Host side of stage D1: oriented bounding box of one cluster
(reference: utils/tower_extraction.py:137-139, trimesh ``bounding_box_oriented``).

trimesh is not a dependency; this module implements the same published procedure on
scipy's qhull binding (the library trimesh itself calls): 3-D hull -> candidate normals
(hemisphere folded, de-duplicated at 0.1 rad in spherical coordinates) -> per candidate a
minimum-area rectangle of the projected hull -> smallest volume wins.

Exact mode (``boxes_of``, the default): qhull sees the FULL cluster, like trimesh's, in worker processes;
the ~80-110 candidate directions of every cluster are priced by one native call (pch_obb_search_f64) and
only the direction(s) that can win are evaluated by the python arithmetic below - bit for bit the result
of the python loop (tests/test_host.py), at ~0.6 ms instead of ~18 ms per cluster on top of the ~7-10 ms
hull.  Fast mode (``boxes_fast``, opt-in): the device drops the points strictly inside the hull first; qhull
then lists the same hull's facets in another order, trimesh's "first normal of every 0.1 rad bucket" rule
picks other candidates and in about 1 cluster of 6 another box of (almost) the same volume wins - centimetres
apart, more than the 1e-3 m the north star allows.

``extent_order='unsorted'`` (default) returns extents as [rect_long, rect_short,
normal_extent] - the behaviour the authors' recorded run shows
(test/kuangxuan.py:30: 17.4 m high, 20.1 m wide); ``'trimesh_sorted'`` sorts ascending and
permutes the axes like current trimesh releases.
"""
from __future__ import annotations

import numpy as np
from scipy.spatial import ConvexHull

_TOL = np.finfo(np.float64).resolution * 100.0
_EXTENT_ORDERS = ("unsorted", "trimesh_sorted")


def hull_vertices_normals(points):
    """qhull ('QbB Pp Qt') hull: vertices in ascending input order + unit triangle normals."""
    p = np.asarray(points, dtype=np.float64)
    hull = ConvexHull(p, qhull_options="QbB Pp Qt")
    keep = np.sort(hull.vertices)
    remap = np.zeros(len(p), dtype=np.int64)
    remap[keep] = np.arange(len(keep))
    v = p[keep]
    tri = v[remap[hull.simplices]]
    n = np.cross(tri[:, 1] - tri[:, 0], tri[:, 2] - tri[:, 0])
    ln = np.sqrt(np.einsum("ij,ij->i", n, n))
    ok = ln > _TOL
    return v, n[ok] / ln[ok, None]


def candidate_angles(normals, digits=1):
    """(theta, phi) of the distinct candidate directions, in trimesh's evaluation order."""
    neg = normals < -_TOL
    zero = ~(neg | (normals > _TOL))
    flip = neg[:, 2] | (zero[:, 2] & neg[:, 1]) | (zero[:, 2] & zero[:, 1] & neg[:, 0])
    v = np.where(flip[:, None], -normals, normals)
    ang = np.column_stack((np.arctan2(v[:, 1], v[:, 0]), np.arccos(np.clip(v[:, 2], -1.0, 1.0))))
    q = np.round(ang * 10 ** digits).astype(np.int64)
    code = q[:, 0] ^ (q[:, 1] << 32)
    _, first = np.unique(code, return_index=True)
    return ang[first]


def _frame_to_z(theta, phi):
    """inverse of Rz(theta)Ry(phi): the 4x4 that turns direction (theta,phi) onto +Z."""
    ct, st, cp, sp = np.cos(theta), np.sin(theta), np.cos(phi), np.sin(phi)
    m = np.eye(4)
    m[:3, :3] = [[cp * ct, -st, sp * ct], [cp * st, ct, sp * st], [-sp, 0.0, cp]]
    return np.linalg.inv(m)


def _planar(theta, offset=(0.0, 0.0)):
    c, s = np.cos(theta), np.sin(theta)
    return np.array([[c, s, offset[0]], [-s, c, offset[1]], [0.0, 0.0, 1.0]])


def min_area_rectangle(xy):
    """trimesh oriented_bounds_2D: (3x3 transform, [long, short])."""
    hull = ConvexHull(np.asarray(xy, dtype=np.float64), qhull_options="QbB")
    seg = hull.points[hull.simplices]
    hp = hull.points[hull.vertices]
    ev = seg[:, 1] - seg[:, 0]
    ln = np.sqrt((ev ** 2).sum(axis=1))
    good = ln > 1e-10
    ev = ev[good] / ln[good, None]
    pv = ev[:, ::-1] * [-1.0, 1.0]
    px, py = ev @ hp.T, pv @ hp.T
    lo = np.column_stack((px.min(axis=1), py.min(axis=1)))
    hi = np.column_stack((px.max(axis=1), py.max(axis=1)))
    ext = hi - lo
    k = int((ext[:, 0] * ext[:, 1]).argmin())
    rect = ext[k]
    t = _planar(np.arctan2(ev[k, 1], ev[k, 0]), -lo[k] - rect * 0.5)
    if rect[0] < rect[1]:
        t = _planar(np.pi / 2) @ t
        rect = rect[::-1].copy()
    return t, rect


def hull_candidates(points):
    """(hull vertices [nv,3], candidate (theta, phi) [nc,2] in evaluation order) of one cluster."""
    verts, normals = hull_vertices_normals(points)
    return verts, candidate_angles(normals)


def _candidate_box(hom, theta, phi):
    """one candidate direction: (volume, extents, to2d, planar transform)"""
    to2d = _frame_to_z(theta, phi)
    proj = (to2d @ hom.T).T[:, :3]
    h = np.ptp(proj[:, 2])
    t2, rect = min_area_rectangle(proj[:, :2])
    return rect[0] * rect[1] * h, np.array([rect[0], rect[1], h]), to2d, t2


def _finish(hom, best, extent_order):
    _, extents, to2d, t2 = best
    rz = np.eye(4)
    rz[:2, :2] = t2[:2, :2]
    to_origin = rz @ to2d
    moved = (to_origin @ hom.T).T[:, :3]
    to_origin[:3, 3] = -(moved.min(axis=0) + np.ptp(moved, axis=0) * 0.5)
    if extent_order == "trimesh_sorted":
        order = extents.argsort()
        flip = np.eye(4)
        flip[:3, :3] = -np.eye(3)[order]
        if np.isclose(np.trace(flip[:3, :3]), 0.0):
            flip[:3, :3] = flip[:3, :3] @ -np.eye(3)
        to_origin = flip @ to_origin
        extents = extents[order]
    return to_origin, extents


def bounds_from_candidates(verts, angles, extent_order="unsorted", shortlist=None):
    """The search over the candidate directions.  ``shortlist`` (indices into ``angles``, ascending; from the
    native search pch_obb_search_f64) restricts the loop to the candidates that can win: they are evaluated
    with this module's own arithmetic and compared like the full loop compares, so the result is the full
    loop's."""
    if extent_order not in _EXTENT_ORDERS:
        raise ValueError(f"extent_order must be one of {_EXTENT_ORDERS}")
    hom = np.column_stack((verts, np.ones(len(verts))))
    best = None
    for theta, phi in (angles if shortlist is None else angles[shortlist]):
        cand = _candidate_box(hom, theta, phi)
        if best is None or cand[0] < best[0]:
            best = cand
    return _finish(hom, best, extent_order)


def oriented_bounds(points, extent_order="unsorted"):
    """Returns (to_origin 4x4, extents[3])."""
    if extent_order not in _EXTENT_ORDERS:
        raise ValueError(f"extent_order must be one of {_EXTENT_ORDERS}")
    verts, angles = hull_candidates(points)
    return bounds_from_candidates(verts, angles, extent_order)


def bounding_box_oriented(points, extent_order="unsorted"):
    """(extents[3], transform 4x4 box->world): the two Box fields the reference reads
    (utils/tower_extraction.py:139,151,165)."""
    to_origin, extents = oriented_bounds(points, extent_order)
    return extents, np.linalg.inv(to_origin)


def _hull_triangles(points):
    """(hull vertices [nv,3] in ascending input order, triangles [nt,3] int32 into them, qhull's order)."""
    p = np.asarray(points, dtype=np.float64)
    hull = ConvexHull(p, qhull_options="QbB Pp Qt")
    ids = np.sort(hull.vertices)
    remap = np.empty(len(p), dtype=np.int32)
    remap[ids] = np.arange(len(ids), dtype=np.int32)
    return p[ids], remap[hull.simplices]


# ---- many clusters at once -------------------------------------------------------------
# The boxes of different clusters are independent, and qhull + the direction search hold the
# GIL, so threads do not help.  PCH_OBB_WORKERS > 1 spreads the clusters over worker PROCESSES:
# plain `python -m pointcloudhookup_amd.obb --worker` children that import only this module
# (never the host application's main script, never a GPU context) and answer pickled requests
# on their pipes.  Every cluster is still computed by bounding_box_oriented on the same points,
# so the results are identical to the serial loop; results come back in input order.
_WORKERS = []


def _boxed(args):
    points, extent_order = args
    try:
        if extent_order == "__hull__":        # first half only: the native search runs in the caller
            return hull_candidates(points), None
        if extent_order == "__hulltri__":     # fast mode: hull vertices + qhull's triangles for pch_obb_min_boxes_f64
            return _hull_triangles(points), None
        return bounding_box_oriented(points, extent_order), None
    except Exception as e:                    # reported per cluster, like the serial loop does
        return None, e


# candidates whose volume is within this (relative) of the smallest are decided by the python arithmetic
_TIE = 1e-9


class _Worker:
    def __init__(self):
        import os
        import subprocess
        import sys
        root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
        env = dict(os.environ)
        env["PYTHONPATH"] = root + os.pathsep + env.get("PYTHONPATH", "")
        self.proc = subprocess.Popen([sys.executable, "-m", "pointcloudhookup_amd.obb", "--worker"],
                                     stdin=subprocess.PIPE, stdout=subprocess.PIPE, env=env)

    def ask(self, args):
        import pickle
        pickle.dump(args, self.proc.stdin, protocol=pickle.HIGHEST_PROTOCOL)
        self.proc.stdin.flush()
        return pickle.load(self.proc.stdout)

    def close(self):
        try:
            self.proc.stdin.close()
            self.proc.wait(timeout=5)
        except Exception:
            self.proc.kill()


def _workers(n):
    import atexit
    if not _WORKERS:
        atexit.register(lambda: [w.close() for w in _WORKERS])
    while len(_WORKERS) < n:
        _WORKERS.append(_Worker())
    return _WORKERS[:n]


def boxes_of(clusters, extent_order="unsorted", workers=None, search=None):
    """Yields ((extents, transform), None) or (None, exception) for every (n_k,3) array in
    ``clusters``, in order.  workers: None -> PCH_OBB_WORKERS; unset: up to 8 worker processes once there are
    at least 16 clusters (a 100 M-point tile has hundreds, ~10 ms of qhull each), none for small jobs.
    search: "native" (default, env PCH_OBB_SEARCH) - qhull and the candidate directions per cluster as below,
    then ONE call of pch_obb_search_f64 prices every direction of every cluster and only the directions within
    1e-9 of the smallest volume (usually one) go through this module's python arithmetic; "python" - the loop
    over all ~80 directions in python.  The boxes are identical either way."""
    import os
    clusters = list(clusters)
    if extent_order not in _EXTENT_ORDERS:
        raise ValueError(f"extent_order must be one of {_EXTENT_ORDERS}")
    if search is None:
        search = os.environ.get("PCH_OBB_SEARCH", "native")
    if search not in ("native", "python"):
        raise ValueError("search must be 'native' or 'python'")
    if workers is None:
        env = os.environ.get("PCH_OBB_WORKERS")
        workers = int(env) if env else (min(8, os.cpu_count() or 1) if len(clusters) >= 16 else 1)
    first = _per_cluster(clusters, "__hull__" if search == "native" else extent_order, workers)
    if search == "python":
        yield from first
        return
    ok = [i for i, (h, e) in enumerate(first) if e is None]
    winners = {}
    if ok:
        from . import ops
        vo = np.cumsum([0] + [len(first[i][0][0]) for i in ok])
        ao = np.cumsum([0] + [len(first[i][0][1]) for i in ok])
        best, vol = ops.obb_search(np.concatenate([first[i][0][0] for i in ok]), vo,
                                   np.concatenate([first[i][0][1].reshape(-1, 2) for i in ok]), ao)
        for j, i in enumerate(ok):
            if best[j] >= 0:
                v = vol[ao[j]:ao[j + 1]]
                winners[i] = np.flatnonzero(v <= v[best[j]] * (1.0 + _TIE))
            else:
                winners[i] = None              # no rectangle anywhere: the python loop reports it its way
    for i, (h, e) in enumerate(first):
        if e is not None:
            yield None, e
            continue
        try:
            to_origin, extents = bounds_from_candidates(h[0], h[1], extent_order, winners[i])
            yield (extents, np.linalg.inv(to_origin)), None
        except Exception as e2:
            yield None, e2


def prestart(workers=None):
    """Starts the worker processes without waiting for them (their ~1 s of imports then runs beside the
    caller's own work).  The drop-in calls this before it touches the file."""
    import os
    if workers is None:
        env = os.environ.get("PCH_OBB_WORKERS")
        workers = int(env) if env else min(8, os.cpu_count() or 1)
    if workers > 1:
        _workers(int(workers))


def _per_cluster(clusters, what, workers):
    """[_boxed((c, what)) for c in clusters], spread over worker processes when asked to."""
    if workers <= 1 or len(clusters) < 2:
        return [_boxed((c, what)) for c in clusters]
    import queue
    import threading
    tasks = queue.SimpleQueue()
    for i, c in enumerate(clusters):
        tasks.put((i, np.ascontiguousarray(c)))
    results = [None] * len(clusters)

    def serve(w):                              # one thread per worker process: request, answer, next
        while True:
            try:
                i, c = tasks.get_nowait()
            except queue.Empty:
                return
            try:
                results[i] = w.ask((c, what))
            except Exception as e:            # a dead worker: compute here instead
                results[i] = _boxed((c, what)) if not isinstance(e, KeyboardInterrupt) else (None, e)

    threads = [threading.Thread(target=serve, args=(w,), daemon=True)
               for w in _workers(min(int(workers), len(clusters)))]
    for t in threads:
        t.start()
    for t in threads:
        t.join()
    return results


# ---- fast mode --------------------------------------------------------------------------------
# Same published procedure, but qhull sees only the points the device could not prove to be strictly
# inside the hull (pch_obb_shell_f32, about 1 % of a cluster) and the candidate search runs natively for
# all clusters at once (pch_obb_min_boxes_f64).  Extents and centres agree with the exact mode to rounding
# whenever qhull builds the same facets from the reduced input - mostly, not always,
# and the sign of the two rectangle axes follows our own edge orientation, not qhull's.  Opt-in.
def boxes_fast(points, perm, offsets, nclusters, extent_order="unsorted", nthreads=0, workers=None):
    """points float32 [N_f,3] (device), perm / offsets as returned by ops.segment_by_label.
    Returns a list of ((extents, transform), None) or (None, exception) per cluster, in label order.
    workers: processes for the qhull calls on the kept points; None: those that are already running (prestart)."""
    import torch
    from . import ops
    if extent_order not in _EXTENT_ORDERS:
        raise ValueError(f"extent_order must be one of {_EXTENT_ORDERS}")
    K = int(nclusters)
    if K == 0:
        return []
    keep = ops.obb_shell(points, perm, offsets, K)
    pos = keep.nonzero().squeeze(1)                              # grouped positions, ascending
    rows = perm.index_select(0, pos).long()
    kept = points.index_select(0, rows).cpu().numpy().astype(np.float64)
    bounds = torch.searchsorted(pos, offsets.to(pos.dtype)).cpu().numpy()
    verts, tris, vo, to = [], [], [0], [0]
    results = [None] * K
    hulls = []
    parts = [kept[bounds[k]:bounds[k + 1]] for k in range(K)]
    if workers is None:                       # worker processes that are already running take the hulls (qhull holds the GIL)
        workers = len(_WORKERS) if K >= 16 else 1
    for k, (h, e) in enumerate(_per_cluster(parts, "__hulltri__", workers)):
        if e is not None:                     # QhullError: too few / degenerate points
            results[k] = (None, e)
            continue
        verts.append(h[0])
        tris.append(h[1])
        vo.append(vo[-1] + len(h[0]))
        to.append(to[-1] + len(h[1]))
        hulls.append(k)
    if hulls:
        T, E, S = ops.obb_min_boxes(np.concatenate(verts), vo, np.concatenate(tris), to,
                                    extent_order == "trimesh_sorted", nthreads)
        inv = np.linalg.inv(T)
        for i, k in enumerate(hulls):
            results[k] = ((E[i], inv[i]), None) if S[i] == 0 else \
                (None, ValueError("degenerate hull: no candidate direction"))
    return results


if __name__ == "__main__":
    import pickle
    import sys
    if "--worker" in sys.argv:
        inp, out = sys.stdin.buffer, sys.stdout.buffer
        sys.stdout = sys.stderr                # stray prints must not corrupt the answer stream
        while True:
            try:
                req = pickle.load(inp)
            except EOFError:
                break
            pickle.dump(_boxed(req), out, protocol=pickle.HIGHEST_PROTOCOL)
            out.flush()
