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


# ---- what qhull is shown ---------------------------------------------------------------------------------
# qhull's cost is linear in the number of points it is SHOWN (~0.15 us per point: every point is partitioned against
# the facets of the initial simplex before anything else happens), and a third of a tower cluster lies strictly inside
# that very simplex.  Such a point is inert in qhull's run: it is below every initial facet, so it never enters an outside
# set, never becomes a (temporary) vertex, and leaves no trace in the facet list - the run on the remaining points is the
# same run, facet for facet, in the same order (which is what the candidate list of the box search depends on).
# The initial simplex is predicted from qhull's own rule (libqhull_r 2019.1, qh_maxmin + qh_maxsimplex, as bundled with
# scipy): the first minimum and maximum point of every coordinate are the candidates; the x-extremes come first, then the
# candidate with the largest |2x2 determinant| of the x,y differences, then the one with the largest |3x3 determinant|
# (the unit-box scaling of 'QbB' multiplies all determinants of a step by the same factor).  The prediction stands down
# - qhull then sees every point - whenever qhull's choice could hinge on rounding or on its 'search all points' rule:
# candidates whose determinants are within 1e-6 of the best, a determinant below 5 % of what the extents allow (qhull
# searches from 0.1 % down), fewer than 64 points, a zero extent.  Points count as inside only when all four barycentric
# coordinates exceed 1e-6 (qhull's own tolerances are ~1e-13 of the unit box).
# Checked against qhull itself: hull vertices, triangles and candidate directions of the full and the reduced input are
# identical on 48 400 random clusters of seven shapes (tools/prefilter_check.py; a deliberately wrong tetrahedron shows
# up there as mismatches), on every cluster of the test suite, and once per process on four synthetic towers
# (_prefilter_ok: a qhull build that chooses differently switches the reduction off).  PCH_OBB_PREFILTER=0 disables it.
_PREFILTER = None


def predicted_simplex(p):
    """indices of the four points qhull starts from, or None where the prediction stands down"""
    n = len(p)
    if n < 64:
        return None
    lo, hi = p.min(axis=0), p.max(axis=0)
    w = hi - lo
    if not (w > 0).all() or not np.isfinite(w).all():
        return None
    mp = []
    for k in range(3):                                   # first occurrence, like qhull's strict comparisons
        mp += [int(p[:, k].argmin()), int(p[:, k].argmax())]
    q = (p[mp] - lo) / w
    s = [mp[0], mp[1]]                                    # the extremes of x (set order: min first)
    if s[0] == s[1]:
        return None
    prev = float(q[1, 0] - q[0, 0])
    for k in (2, 3):
        cands = []
        for i in range(6):
            if mp[i] in s:
                continue
            rows = np.array([q[mp.index(sj)][:k] - q[i][:k] for sj in s[:k]])
            cands.append((abs(float(np.linalg.det(rows))), mp[i]))
        if not cands:
            return None
        cands.sort(key=lambda t: -t[0])
        best = cands[0]
        if any(c != best[1] and d > best[0] * (1.0 - 1e-6) for d, c in cands[1:]):
            return None                                   # qhull's own rounding would decide
        if not best[0] > 0.05 * prev:
            return None                                   # too close to qhull's 'search all points' regime
        prev = best[0]
        s.append(best[1])
    return s


def qhull_input(points):
    """float64 [m,3]: what qhull is shown of ``points`` ([n,3] float32 or float64) - all of them, or (natively,
    pch_obbhost_reduce_*) all but the rows strictly inside qhull's initial simplex"""
    global _PREFILTER
    if _PREFILTER is None:
        import os
        _PREFILTER = False                                # (the self-check below runs the unreduced path)
        _PREFILTER = (os.environ.get("PCH_OBB_PREFILTER", "1") != "0" and _hostlib() is not None
                      and _prefilter_ok())
    a = np.ascontiguousarray(points)
    if not _PREFILTER or a.ndim != 2 or a.shape[1] != 3 or a.dtype not in (np.float32, np.float64) or len(a) < 64:
        return np.asarray(points, dtype=np.float64)
    return _reduced_native(a)


def _reduced_native(a):
    import ctypes as C
    lib = _hostlib()
    out = np.empty((len(a), 3), dtype=np.float64)
    rows = C.c_int64(0)
    fn = lib.pch_obbhost_reduce_f32 if a.dtype == np.float32 else lib.pch_obbhost_reduce_f64
    rc = fn(a.ctypes.data, len(a), out.ctypes.data, C.addressof(rows))
    if rc < 0:
        return np.asarray(a, dtype=np.float64)
    return out[: rows.value]


def _reduced(p):
    s = predicted_simplex(p)
    if s is None:
        return p, None
    T = p[s]
    try:
        A = np.linalg.inv((T[1:] - T[0]).T)
    except np.linalg.LinAlgError:
        return p, None
    b = (p - T[0]) @ A.T
    inside = (b > 1e-6).all(axis=1) & (b.sum(axis=1) < 1.0 - 1e-6)
    if not inside.any():
        return p, None
    keep = np.flatnonzero(~inside)
    return p[keep], keep


def _prefilter_ok():
    """once per process: does THIS qhull build start from the predicted simplex?  Four synthetic towers, full against
    reduced input (the native reduction, as used): hull vertices, triangles in qhull's order and facet equations must
    be identical; any difference (or exception) switches the reduction off."""
    try:
        rng = np.random.default_rng(20261004)
        for i in range(4):
            c = (rng.normal(size=(6000, 3)) * [2.5 + i, 2.5, 9.0] + [30.0 * i, -7.0, 22.0]).astype(np.float32)
            if i == 3:
                c = (c.astype(np.float64) @ np.linalg.qr(rng.normal(size=(3, 3)))[0]).astype(np.float32)
            full = np.asarray(c, dtype=np.float64)
            sub = _reduced_native(np.ascontiguousarray(c))
            if len(sub) == len(full):
                continue
            a = ConvexHull(full, qhull_options="QbB Pp Qt")
            b = ConvexHull(sub, qhull_options="QbB Pp Qt")
            if not (np.array_equal(full[a.simplices], sub[b.simplices]) and np.array_equal(a.equations, b.equations)):
                return False
        return True
    except Exception:
        return False


# ---- the qhull call itself ------------------------------------------------------------------------------------
# scipy.spatial.ConvexHull spends a quarter of a 43 000-point call OUTSIDE qhull (neighbour / coplanar / good arrays,
# strided min/max bounds of the input, a second copy of the points): _hull_simplices drives the same compiled object
# (scipy.spatial._qhull._Qhull - the class ConvexHull itself instantiates, same options, same qhull run) and takes only
# the triangle list.  _Qhull is scipy-private, so the short cut is taken only where it is found AND has reproduced
# ConvexHull's triangles on a check cloud in this process; anything else (another scipy, an exception) and ConvexHull
# is called as before.  PCH_OBB_BARE_QHULL=0 switches it off.
_BARE = None
_OPTS = "QbB Pp Qt"


def _bare_simplices(p):
    from scipy.spatial import _qhull
    q = _qhull._Qhull(b"i", p, _OPTS.encode(), required_options=b"Qt", incremental=False)
    try:
        q.triangulate()
        return q.get_simplex_facet_array()[0]
    finally:
        q.close()


def _bare_ok():
    try:
        rng = np.random.default_rng(20261005)
        for n in (300, 20000):
            c = np.ascontiguousarray((rng.normal(size=(n, 3)) * [2.5, 3.5, 9.0]).astype(np.float32), dtype=np.float64)
            want = ConvexHull(c, qhull_options=_OPTS)
            got = _bare_simplices(c)
            if got.dtype != want.simplices.dtype or not np.array_equal(got, want.simplices):
                return False
        return True
    except Exception:
        return False


def _hull_simplices(p):
    """[nt,3] point indices of the hull triangles of float64 rows ``p``, qhull's order: ConvexHull(p, 'QbB Pp Qt')
    .simplices (errors included - they are raised by the same constructor)"""
    global _BARE
    if _BARE is None:
        import os
        _BARE = os.environ.get("PCH_OBB_BARE_QHULL", "1") != "0" and _bare_ok()
    if _BARE and p.ndim == 2 and p.shape[1] == 3 and p.dtype == np.float64 and p.flags.c_contiguous:
        return _bare_simplices(p)
    return ConvexHull(p, qhull_options=_OPTS).simplices


def hull_vertices_normals(points):
    """qhull ('QbB Pp Qt') hull: vertices in ascending input order + unit triangle normals."""
    p = qhull_input(points)
    simplices = _hull_simplices(p)
    keep = np.unique(simplices)                           # = sort(ConvexHull.vertices)
    remap = np.zeros(len(p), dtype=np.int64)
    remap[keep] = np.arange(len(keep))
    v = p[keep]
    tri = v[remap[simplices]]
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
    p = qhull_input(points)
    simplices = _hull_simplices(p)
    ids = np.unique(simplices)
    remap = np.empty(len(p), dtype=np.int32)
    remap[ids] = np.arange(len(ids), dtype=np.int32)
    return p[ids], remap[simplices]


# ---- many clusters at once -------------------------------------------------------------
# The boxes of different clusters are independent and cost ~0.15 us of qhull per point (7 ms for a 43 000-point
# tower), so they are spread over a pool of worker PROCESSES: plain `python -m pointcloudhookup_amd.obb --worker`
# children that import only this module (never the host application's main script, never the HIP runtime: the
# candidate search they run comes from the host-only libpch_obbhost.so).  Hand-off (round 4): the points of all
# clusters lie in ONE shared-memory buffer (a memfd the parent maps, registers with the HIP runtime so that the
# device copies straight into it, and the workers map through /proc/<parent>/fd/<n>); a task is a few dozen bytes
# (path, offset, rows), an answer is the finished box (19 doubles) - nothing large is pickled or piped.  One
# dispatcher thread in the parent feeds idle workers from a global task queue (largest clusters first) and collects
# the answers; several jobs may be in flight, so the boxes of tile k are computed while the device clusters tile k+1.
# Every cluster is still computed by the statements of bounding_box_oriented on the same points (hull ->
# candidates -> native pricing -> python arithmetic for the winner), so the results are the serial loop's.
_TIE = 1e-9          # candidates whose volume is within this (relative) of the smallest are decided by the python arithmetic
_HOSTLIB = None


def _hostlib():
    """libpch_obbhost.so (include/pch_obbhost.h), or None when it is not built (the python loop then prices)."""
    global _HOSTLIB
    if _HOSTLIB is None:
        import ctypes as C
        import os
        path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "libpch_obbhost.so")
        try:
            lib = C.CDLL(path)
            lib.pch_obbhost_search_f64.restype = C.c_int
            lib.pch_obbhost_search_f64.argtypes = [C.c_void_p, C.c_int64, C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p]
            for fn in (lib.pch_obbhost_reduce_f32, lib.pch_obbhost_reduce_f64):
                fn.restype = C.c_int
                fn.argtypes = [C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p]
            _HOSTLIB = lib
        except OSError:
            _HOSTLIB = False
    return _HOSTLIB or None


def _winners(verts, angles):
    """Indices of the candidate directions that can win (native pricing of all of them, then everything within
    _TIE of the smallest volume), or None = evaluate all of them in python."""
    lib = _hostlib()
    if lib is None or len(angles) == 0:
        return None
    import ctypes as C
    v = np.ascontiguousarray(verts, dtype=np.float64)
    a = np.ascontiguousarray(angles, dtype=np.float64).reshape(-1, 2)
    vol = np.empty(len(a), dtype=np.float64)
    best = C.c_int32(-1)
    rc = lib.pch_obbhost_search_f64(v.ctypes.data, len(v), a.ctypes.data, len(a), C.addressof(best), vol.ctypes.data)
    if rc != 0 or best.value < 0:
        return None                            # no rectangle anywhere: the python loop reports it its way
    return np.flatnonzero(vol <= vol[best.value] * (1.0 + _TIE))


def _boxed(args):
    points, what = args
    try:
        if what == "__hull__":                 # first half only: the search runs in the caller
            return hull_candidates(points), None
        if what == "__hulltri__":              # fast mode: hull vertices + qhull's triangles for pch_obb_min_boxes_f64
            return _hull_triangles(points), None
        if what.startswith("native:"):         # the whole exact box: hull, candidates, native pricing, python winner
            verts, angles = hull_candidates(points)
            to_origin, extents = bounds_from_candidates(verts, angles, what[7:], _winners(verts, angles))
            return (extents, np.linalg.inv(to_origin)), None
        return bounding_box_oriented(points, what), None
    except Exception as e:                     # reported per cluster, like the serial loop does
        return None, e


def _affinity_cpus():
    import os
    try:
        return max(1, len(os.sched_getaffinity(0)))
    except AttributeError:
        return max(1, os.cpu_count() or 1)


def cpu_quota():
    """(cores, period in s) of the cgroup CPU quota this process runs under, or None where none is readable.  A quota
    is a BUDGET - cores x period CPU-seconds per period, spent at any parallelism - not a core count: 16 cores /
    100 ms lets 48 processes run for 33 ms and then freezes every thread of the group until the period ends."""
    for path in ("/sys/fs/cgroup/cpu.max", "/sys/fs/cgroup/cpu/cpu.cfs_quota_us"):
        try:
            txt = open(path).read().split()
            if path.endswith("cpu.max"):
                quota, period = txt[0], float(txt[1])
            else:
                quota, period = txt[0], float(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
            if quota not in ("max", "-1") and float(quota) > 0 and period > 0:
                return float(quota) / period, period / 1e6
            return None
        except (OSError, ValueError, IndexError):
            continue
    return None


def usable_cpus():
    """cores this process may use in the long run: affinity mask, cut by a cgroup CPU quota where one is readable"""
    n = _affinity_cpus()
    q = cpu_quota()
    if q is not None:
        n = min(n, max(1, int(q[0] + 0.5)))
    return max(1, n)


POOL_CAP = 64         # worker processes at most (a 100 M-point tile has ~230 clusters; beyond ~4 per worker the tail idles)
BURST = 2             # under a CPU quota: worker processes per quota core (see Pool._may_start)
import os as _os
LEDGER = float(_os.environ.get("PCH_OBB_LEDGER", "0.8"))   # share of a period's CPU budget the pool lets itself spend


def default_workers():
    """PCH_OBB_WORKERS, else one worker per usable core up to POOL_CAP - and under a cgroup CPU quota BURST workers per
    quota core where the machine has the cores: the table of ONE tile (~0.8 CPU-seconds) fits the budget of one quota
    period, so it may be spent at twice the parallelism; the dispatcher keeps the pool inside the budget
    (Pool._may_start), so a stream of tables still runs at the quota's pace without the group being frozen."""
    import os
    env = os.environ.get("PCH_OBB_WORKERS")
    if env:
        return max(1, int(env))
    n = usable_cpus()
    if cpu_quota() is not None:
        n = min(_affinity_cpus(), BURST * n)
    return min(POOL_CAP, n)


def _send(fh, obj):
    import pickle
    import struct
    data = pickle.dumps(obj, protocol=pickle.HIGHEST_PROTOCOL)
    fh.write(struct.pack("<I", len(data)) + data)
    fh.flush()


class Buffer:
    """A shared-memory buffer the workers can map: a memfd (freed by the kernel when the last mapping goes - nothing
    can leak into /dev/shm), named /proc/<pid>/fd/<n> for the workers.  ``pin()`` registers it with the HIP runtime
    so that a device-to-host copy goes straight into it."""

    _count = 0

    def __init__(self, nbytes):
        import mmap
        import os
        self.nbytes = int(max(nbytes, 1 << 16))
        try:
            self.fd = os.memfd_create("pch_obb")
            self.path = f"/proc/{os.getpid()}/fd/{self.fd}"
        except (AttributeError, OSError):                        # no memfd: a named file under /dev/shm
            import tempfile
            self.fd, self.path = tempfile.mkstemp(prefix=f"pch_obb_{os.getpid()}_", dir="/dev/shm")
            self._unlink = self.path
        os.ftruncate(self.fd, self.nbytes)
        self.map = mmap.mmap(self.fd, self.nbytes)
        self.array = np.frombuffer(self.map, dtype=np.uint8)
        self.pinned = False
        self.busy = False
        Buffer._count += 1
        self.where = (self.path, self.nbytes, f"{os.getpid()}.{Buffer._count}")     # what a task carries

    def pin(self):
        if not self.pinned:
            try:
                import torch
                rc = torch.cuda.cudart().cudaHostRegister(self.array.ctypes.data, self.nbytes, 0)
                self.pinned = (rc is None) or (int(rc) == 0)
            except Exception:
                self.pinned = False
        return self.pinned

    def tensor(self, dtype, count):
        import torch
        return torch.frombuffer(self.map, dtype=dtype, count=int(count))

    def close(self):
        import os
        try:
            if self.pinned:
                import torch
                torch.cuda.cudart().cudaHostUnregister(self.array.ctypes.data)
        except Exception:
            pass
        self.array = None
        try:
            self.map.close()
        except (BufferError, ValueError):
            pass
        try:
            os.close(self.fd)
        except OSError:
            pass
        if getattr(self, "_unlink", None):
            try:
                os.unlink(self._unlink)
            except OSError:
                pass


class Job:
    """One batch of tasks in the pool.  ``wait()`` returns the results in task order: (value, exception | None)."""

    def __init__(self, ntasks):
        import threading
        self.results = [None] * ntasks
        self.left = ntasks
        self.done = threading.Event()
        self.worker_s = 0.0                    # wall seconds the workers reported for this job's tasks (sum)
        self.worker_cpu_s = 0.0                # ... and the CPU seconds of the same (less where a CPU quota throttles)
        self.t_submit = self.t_done = 0.0
        if ntasks == 0:
            self.done.set()

    def wait(self, timeout=None):
        if not self.done.wait(timeout):
            raise TimeoutError("box workers did not answer in time")
        return self.results


class _Proc:
    def __init__(self):
        import os
        import subprocess
        import sys
        root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
        env = dict(os.environ)
        env["PYTHONPATH"] = root + os.pathsep + env.get("PYTHONPATH", "")
        for var in ("OMP_NUM_THREADS", "OPENBLAS_NUM_THREADS", "MKL_NUM_THREADS"):
            env[var] = "1"                     # one core per worker: the pool is the parallelism
        self.proc = subprocess.Popen([sys.executable, "-m", "pointcloudhookup_amd.obb", "--worker"],
                                     stdin=subprocess.PIPE, stdout=subprocess.PIPE, env=env)
        self.fd = self.proc.stdout.fileno()
        os.set_blocking(self.fd, False)
        self.buf = bytearray()
        self.task = None                       # (job, index, request) in flight
        self.t_task = 0.0                      # when it was handed out
        self.ready = False                     # its imports and self-checks are done (it said so)
        self.alive = True


class Pool:
    """Worker processes + one dispatcher thread.  submit(requests) -> Job."""

    def __init__(self):
        import collections
        import os
        import threading
        self.procs = []
        self.queue = collections.deque()       # (job, index, request), in submission order
        self.lock = threading.Lock()
        self.wake_r, self.wake_w = os.pipe()
        os.set_blocking(self.wake_r, False)
        os.set_blocking(self.wake_w, False)
        self.stopping = False
        self.thread = None
        self.buffers = []
        self.quota = cpu_quota()               # (cores, period s) | None
        self.spent = collections.deque()       # (when, CPU seconds) of the tasks finished lately

    # ---- workers
    def grow(self, n):
        with self.lock:
            while sum(p.alive for p in self.procs) < n:        # workers that died are replaced
                self.procs.append(_Proc())
            if self.thread is None:
                import threading
                self.thread = threading.Thread(target=self._run, name="pch-obb-dispatch", daemon=True)
                self.thread.start()
        self._wake()

    def size(self):
        return sum(p.alive for p in self.procs)

    def _wake(self):
        import os
        try:
            os.write(self.wake_w, b"x")
        except (BlockingIOError, OSError):
            pass

    # ---- shared buffers (a small ring: one per job in flight)
    def buffer(self, nbytes, pin=False):
        with self.lock:
            best = None
            for b in self.buffers:
                if not b.busy and b.nbytes >= nbytes and (best is None or b.nbytes < best.nbytes):
                    best = b
            if best is None:
                for b in [b for b in self.buffers if not b.busy]:       # too small: replaced, not accumulated
                    self.buffers.remove(b)
                    b.close()
                best = Buffer(int(nbytes * 1.25) + (1 << 20))
                self.buffers.append(best)
            best.busy = True
        if pin:
            best.pin()
        return best

    def release(self, buf):
        with self.lock:
            buf.busy = False

    # ---- jobs
    def submit(self, requests):
        import time
        job = Job(len(requests))
        job.t_submit = time.perf_counter()
        if requests:
            with self.lock:
                for i, r in enumerate(requests):
                    self.queue.append((job, i, r))
            self._wake()
        return job

    def _finish(self, job, index, value):
        import time
        job.results[index] = value
        job.left -= 1
        if job.left == 0:
            job.t_done = time.perf_counter()
            job.done.set()

    # ---- the CPU budget (only under a cgroup quota)
    def _may_start(self, procs, now):
        """may another task start now?  Always while fewer workers are busy than the quota has cores.  Beyond that
        only while the CPU-seconds the pool has spent in the last quota period (finished tasks + the running ones so
        far) leave room in the period's budget: a group that overdraws it is frozen - every thread of it, this
        process's main thread and its HIP runtime threads included - until the period ends, which costs a stream of
        tiles more than the extra workers gain.  (The kernel's periods are fixed windows of unknown phase; the sliding
        window and the 0.8 allow for that and for what the parent itself burns.)"""
        if self.quota is None:
            return True
        cores, period = self.quota
        busy = [p for p in procs if p.alive and p.task is not None]
        if len(busy) < max(1, int(cores) - 1):
            return True
        while self.spent and self.spent[0][0] < now - period:
            self.spent.popleft()
        used = sum(c for _, c in self.spent) + sum(min(now - p.t_task, period) for p in busy)
        return used + 0.004 * (len(busy) + 1) < LEDGER * cores * period

    def _hand_out(self, sel, procs):
        """queued tasks to idle, ready workers; returns True when tasks are left over only for want of budget"""
        import time
        for p in procs:
            if not p.alive or not p.ready or p.task is not None:
                continue
            now = time.perf_counter()
            if not self._may_start(procs, now):
                return bool(self.queue)
            with self.lock:
                item = self.queue.popleft() if self.queue else None
            if item is None:
                return False
            try:
                _send(p.proc.stdin, item[2])
                p.task, p.t_task = item, now
            except (OSError, ValueError):
                self._dead(sel, p, item)
        return False

    def _run(self):
        import os
        import pickle
        import selectors
        import struct
        import time
        sel = selectors.DefaultSelector()
        sel.register(self.wake_r, selectors.EVENT_READ, None)
        known = set()
        while not self.stopping:
            with self.lock:
                procs = list(self.procs)
            for p in procs:
                if p.alive and id(p) not in known:
                    known.add(id(p))
                    sel.register(p.fd, selectors.EVENT_READ, p)
            waiting = self._hand_out(sel, procs)
            if not any(p.alive for p in procs):                  # nobody left: compute here
                while True:
                    with self.lock:
                        item = self.queue.popleft() if self.queue else None
                    if item is None:
                        break
                    self._finish(item[0], item[1], _answer(item[2])[0])
            for key, _ in sel.select(timeout=0.002 if waiting else 0.5):
                if key.data is None:
                    try:
                        os.read(self.wake_r, 4096)
                    except (BlockingIOError, OSError):
                        pass
                    continue
                p = key.data
                try:
                    chunk = os.read(p.fd, 1 << 16)
                except BlockingIOError:
                    continue
                except OSError:
                    chunk = b""
                if not chunk:
                    self._dead(sel, p, p.task)
                    continue
                p.buf += chunk
                while len(p.buf) >= 4:
                    (ln,) = struct.unpack_from("<I", p.buf, 0)
                    if len(p.buf) < 4 + ln:
                        break
                    msg = pickle.loads(bytes(p.buf[4:4 + ln]))
                    del p.buf[:4 + ln]
                    if isinstance(msg, str):                     # "ready": imports and self-checks done
                        p.ready = True
                        continue
                    value, secs, cpu = msg
                    job, index, _ = p.task
                    p.task = None
                    job.worker_s += secs
                    job.worker_cpu_s += cpu
                    self.spent.append((time.perf_counter(), cpu))
                    self._finish(job, index, value)
                self._hand_out(sel, procs)                       # next tasks at once, without another pass

    def _dead(self, sel, p, item):
        """a worker that went away: its task (if any) is computed in the dispatcher, the rest goes to the others"""
        if p.alive:
            p.alive = False
            try:
                sel.unregister(p.fd)
            except (KeyError, ValueError):
                pass
        p.task = None
        if item is not None:
            self._finish(item[0], item[1], _answer(item[2])[0])

    def close(self):
        self.stopping = True
        self._wake()
        for p in self.procs:
            try:
                p.proc.stdin.close()
                p.proc.wait(timeout=2)
            except Exception:
                p.proc.kill()
        for b in self.buffers:
            b.close()
        self.buffers = []


_POOL = None


def pool(workers=None):
    """the process-wide pool, grown to ``workers`` (default_workers()) processes"""
    global _POOL
    if _POOL is None:
        import atexit
        _POOL = Pool()
        atexit.register(_POOL.close)
    n = default_workers() if workers is None else int(workers)
    if n > _POOL.size():
        _POOL.grow(n)
    return _POOL


def prestart(workers=None):
    """Starts the worker processes without waiting for them (their ~1 s of imports then runs beside the
    caller's own work).  The drop-in calls this before it touches the file."""
    n = default_workers() if workers is None else int(workers)
    if n > 1:
        pool(n)


_MAPS = {}            # worker side: buffer id -> uint8 array over its mapping


def _mapped(path, nbytes, ident):
    """the parent's buffer `ident` mapped read-only (cached; the parent keeps a small ring of buffers, and a path
    /proc/<pid>/fd/<n> may name another buffer later - hence the id).  Old mappings are dropped, not closed: they
    go when the last array over them goes."""
    import mmap
    import os
    arr = _MAPS.get(ident)
    if arr is None:
        while len(_MAPS) >= 4:
            _MAPS.pop(next(iter(_MAPS)))
        fd = os.open(path, os.O_RDONLY)
        try:
            arr = np.frombuffer(mmap.mmap(fd, nbytes, prot=mmap.PROT_READ), dtype=np.uint8)
        finally:
            os.close(fd)
        _MAPS[ident] = arr
    return arr


def _answer(req):
    """one request -> ((value, exception | None), wall seconds, CPU seconds).  Runs in a worker (or in the parent's dispatcher when
    no worker is left).  Requests: ("shm", (path, nbytes, id), byte offset, rows, dtype, what) | ("inline", array,
    what)."""
    import time
    t0, c0 = time.perf_counter(), time.process_time()
    try:
        if req[0] == "shm":
            _, where, off, rows, dtype, what = req
            item = np.dtype(dtype).itemsize * 3
            pts = _mapped(*where)[off:off + rows * item].view(dtype).reshape(rows, 3)
        else:
            _, pts, what = req
        out = _boxed((pts, what))
    except Exception as e:
        out = (None, e)
    return out, time.perf_counter() - t0, time.process_time() - c0


def boxes_job(buf, offsets, dtype, extent_order="unsorted", workers=None, order=None):
    """Submits the exact boxes of the clusters laid out in ``buf`` (a Buffer): cluster k = rows
    [offsets[k], offsets[k+1]) of the [*,3] ``dtype`` array at the buffer's start.  Returns a Job whose results
    are ((extents, transform), None) | (None, exception) per cluster, in cluster order."""
    if extent_order not in _EXTENT_ORDERS:
        raise ValueError(f"extent_order must be one of {_EXTENT_ORDERS}")
    pl = pool(workers)
    item = np.dtype(dtype).itemsize * 3
    k = len(offsets) - 1
    reqs = [("shm", buf.where, int(offsets[i]) * item, int(offsets[i + 1] - offsets[i]),
             np.dtype(dtype).str, "native:" + extent_order) for i in range(k)]
    # largest first: the tail of the job is then made of small clusters.  The Job keeps cluster order.
    sizes = np.diff(np.asarray(offsets, dtype=np.int64))
    perm = np.argsort(-sizes, kind="stable")
    job = pl.submit([reqs[i] for i in perm])
    job.perm = perm
    return job


def job_results(job, timeout=None):
    """results of a boxes_job in cluster order"""
    res = job.wait(timeout)
    out = [None] * len(res)
    for slot, i in enumerate(job.perm):
        out[int(i)] = res[slot]
    return out


def boxes_of(clusters, extent_order="unsorted", workers=None, search=None):
    """Yields ((extents, transform), None) or (None, exception) for every (n_k,3) array in
    ``clusters``, in order.  workers: None -> PCH_OBB_WORKERS; unset: the pool (one worker per usable core, up to
    POOL_CAP) once there are at least 16 clusters (a 100 M-point tile has hundreds, ~7 ms of qhull each), none for
    small jobs.
    search: "native" (default, env PCH_OBB_SEARCH) - qhull and the candidate directions per cluster as below,
    then every direction of the cluster is priced natively (pch_obb_search_f64's code, in the worker) and only the
    directions within 1e-9 of the smallest volume (usually one) go through this module's python arithmetic;
    "python" - the loop over all ~80 directions in python.  The boxes are identical either way."""
    import os
    clusters = list(clusters)
    if extent_order not in _EXTENT_ORDERS:
        raise ValueError(f"extent_order must be one of {_EXTENT_ORDERS}")
    if search is None:
        search = os.environ.get("PCH_OBB_SEARCH", "native")
    if search not in ("native", "python"):
        raise ValueError("search must be 'native' or 'python'")
    if workers is None:
        workers = default_workers() if (len(clusters) >= 16 or os.environ.get("PCH_OBB_WORKERS")) else 1
    what = ("native:" + extent_order) if search == "native" else extent_order
    yield from _per_cluster(clusters, what, workers)


def _per_cluster(clusters, what, workers):
    """[_boxed((c, what)) for c in clusters], spread over the worker pool when asked to: the arrays are packed into
    one shared buffer, the tasks name their slices."""
    if workers <= 1 or len(clusters) < 2:
        return [_boxed((c, what)) for c in clusters]
    arrs = [np.ascontiguousarray(c) for c in clusters]
    pl = pool(workers)
    ok = all(a.ndim == 2 and a.shape[1] == 3 and a.dtype == arrs[0].dtype and a.dtype.kind == "f" for a in arrs)
    if not ok:                                 # odd shapes / dtypes: the arrays travel inline
        return pl.submit([("inline", a, what) for a in arrs]).wait()
    item = arrs[0].dtype.itemsize * 3
    offs = np.cumsum([0] + [len(a) for a in arrs])
    buf = pl.buffer(int(offs[-1]) * item)
    try:
        flat = buf.array[:int(offs[-1]) * item].view(arrs[0].dtype).reshape(-1, 3)
        for a, lo in zip(arrs, offs[:-1]):
            flat[lo:lo + len(a)] = a
        del flat
        reqs = [("shm", buf.where, int(offs[i]) * item, len(arrs[i]), arrs[0].dtype.str, what)
                for i in range(len(arrs))]
        return pl.submit(reqs).wait()
    finally:
        pl.release(buf)


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
    if workers is None:                       # worker processes that are already running take the hulls
        workers = (_POOL.size() if _POOL is not None else 0) if K >= 16 else 1
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
    import struct
    import sys
    if "--worker" in sys.argv:
        inp, out = sys.stdin.buffer, sys.stdout.buffer
        sys.stdout = sys.stderr                # stray prints must not corrupt the answer stream
        try:                                   # the once-per-process checks now, not inside the first task
            qhull_input(np.zeros((64, 3), dtype=np.float32))
            _hull_simplices(np.ascontiguousarray(np.random.default_rng(0).normal(size=(16, 3))))
        except Exception:
            pass
        _send(out, "ready")
        while True:
            head = inp.read(4)
            if len(head) < 4:
                break
            (ln,) = struct.unpack("<I", head)
            req = pickle.loads(inp.read(ln))
            _send(out, _answer(req))
