"""Oracle for stage C: chunked DBSCAN (TEST INFRASTRUCTURE).

The reference runs ``sklearn.cluster.DBSCAN(eps, min_samples, n_jobs=-1,
algorithm='ball_tree').fit(chunk)`` on consecutive 50 000-row chunks of the
height-filtered float32 points and offsets the labels per chunk
(``/root/reference/utils/tower_extraction.py:96-117``).  scikit-learn is a
third-party dependency of the reference (unpinned there; 1.7.2 in this image).

This file restates what that call computes:

* neighbourhoods: ``sum_j (x_j - y_j)^2 <= eps*eps`` with the coordinates
  promoted float32 -> float64 and the three squares accumulated in order, in
  float64 (sklearn/metrics/_dist_metrics.pxd.tp ``euclidean_rdist``;
  sklearn/neighbors/_binary_tree.pxi.tp:1953-1958); the point itself is its own
  neighbour (sklearn/cluster/_dbscan.py:399-401);
* core points: ``len(neighbourhood) >= min_samples`` (_dbscan.py:423-434);
* labels: the depth-first sweep of sklearn/cluster/_dbscan_inner.pyx:10-41.

It is pinned against real sklearn output: ``tests/golden/dbscan_*.npz`` hold
labels produced by sklearn 1.7.2 here (``tests/golden/gen_golden.py``), and the
tests also call sklearn directly whenever it is importable.

Known, documented divergence: sklearn's ball tree accepts or prunes whole
nodes from floating-point *bounds* (``dist_UB <= r`` / ``dist_LB > r``,
_binary_tree.pxi.tp:1935-1947), so for a pair whose distance equals eps to
within rounding the tree may disagree with the exact per-pair predicate above.
Such ties have measure zero for real coordinates; the golden sets contain none.
"""
from __future__ import annotations

import ctypes
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_CLIB = None


def radius_neighbors_bruteforce(X, eps, block=512):
    """All-pairs restatement of ``NearestNeighbors(radius=eps).radius_neighbors(X)``.
    Returns a list of int64 index arrays (ascending), self included."""
    X64 = np.ascontiguousarray(np.asarray(X), dtype=np.float64)   # _binary_tree.pxi.tp:851
    n = X64.shape[0]
    r2 = float(eps) * float(eps)                                   # _dist_to_rdist
    out = []
    for s in range(0, n, block):
        q = X64[s:s + block]
        d = np.zeros((q.shape[0], n), dtype=np.float64)
        for j in range(X64.shape[1]):                              # d += tmp*tmp, j ascending
            t = q[:, j:j + 1] - X64[None, :, j]
            d += t * t
        hit = d <= r2
        for row in hit:
            out.append(np.flatnonzero(row).astype(np.int64))
    return out


def dbscan_inner_literal(is_core, neighborhoods):
    """Line-by-line python form of sklearn/cluster/_dbscan_inner.pyx:10-41."""
    n = len(neighborhoods)
    labels = np.full(n, -1, dtype=np.int64)
    label_num = 0
    stack = []
    for i in range(n):
        if labels[i] != -1 or not is_core[i]:
            continue
        while True:
            if labels[i] == -1:
                labels[i] = label_num
                if is_core[i]:
                    for v in neighborhoods[i]:
                        if labels[v] == -1:
                            stack.append(int(v))
            if not stack:
                break
            i = stack.pop()
        label_num += 1
    return labels


def dbscan_fit_numpy(X, eps, min_samples):
    """Small-n literal DBSCAN.fit: returns (labels int64[n], is_core uint8[n])."""
    nb = radius_neighbors_bruteforce(X, eps)
    n_neighbors = np.array([len(v) for v in nb], dtype=np.int64)
    is_core = (n_neighbors >= int(min_samples)).astype(np.uint8)
    return dbscan_inner_literal(is_core, nb), is_core


def dbscan_rule(X, eps, min_samples):
    """The order-free statement the HIP kernels implement:
    (1) connected components of the core-core <=eps graph, (2) cluster id = rank
    of the component's smallest core index, (3) a non-core point takes the smallest
    cluster id among its core neighbours, else -1.  Small n only."""
    nb = radius_neighbors_bruteforce(X, eps)
    n = len(nb)
    core = np.array([len(v) >= int(min_samples) for v in nb], dtype=bool)
    parent = np.arange(n)

    def find(a):
        while parent[a] != a:
            parent[a] = parent[parent[a]]
            a = parent[a]
        return a

    for i in range(n):
        if core[i]:
            for v in nb[i]:
                if core[v]:
                    ra, rb = find(i), find(int(v))
                    if ra != rb:
                        parent[max(ra, rb)] = min(ra, rb)
    labels = np.full(n, -1, dtype=np.int64)
    roots = sorted({find(i) for i in range(n) if core[i]})        # root == min core index
    rank = {r: k for k, r in enumerate(roots)}
    for i in range(n):
        if core[i]:
            labels[i] = rank[find(i)]
    for i in range(n):
        if not core[i]:
            cand = [labels[v] for v in nb[i] if core[v]]
            if cand:
                labels[i] = min(cand)
    return labels, core.astype(np.uint8)


# ----------------------------------------------------------------------------
# C restatement (oracle/dbscan_c.c) for chunk-sized inputs
# ----------------------------------------------------------------------------
def _clib():
    global _CLIB
    if _CLIB is None:
        path = os.path.join(_HERE, "_build", "liboracle.so")
        if not os.path.exists(path):
            raise FileNotFoundError(
                f"{path} missing: run `make -C oracle` (or __graft_entry__.build())")
        lib = ctypes.CDLL(path)
        lib.oracle_dbscan_f32.restype = ctypes.c_int
        lib.oracle_dbscan_f32.argtypes = [ctypes.c_void_p, ctypes.c_int64, ctypes.c_double,
                                          ctypes.c_int32, ctypes.c_void_p, ctypes.c_void_p]
        _CLIB = lib
    return _CLIB


def dbscan_fit_c(X, eps, min_samples):
    """All-pairs C restatement; O(n^2) time, O(n) memory.  (labels int64, core uint8)."""
    X = np.ascontiguousarray(np.asarray(X, dtype=np.float32)).reshape(-1, 3)
    n = X.shape[0]
    labels = np.empty(n, dtype=np.int32)
    core = np.empty(n, dtype=np.uint8)
    rc = _clib().oracle_dbscan_f32(X.ctypes.data, n, float(eps), int(min_samples),
                                   labels.ctypes.data, core.ctypes.data)
    if rc != 0:
        raise RuntimeError(f"oracle_dbscan_f32 failed: {rc}")
    return labels.astype(np.int64), core


def dbscan_fit_sklearn(X, eps, min_samples):
    """The reference's literal call (utils/tower_extraction.py:107-112)."""
    from sklearn.cluster import DBSCAN
    cl = DBSCAN(eps=eps, min_samples=min_samples, n_jobs=-1, algorithm='ball_tree').fit(X)
    core = np.zeros(len(X), dtype=np.uint8)
    core[cl.core_sample_indices_] = 1
    return cl.labels_.astype(np.int64), core


_FITS = {"numpy": dbscan_fit_numpy, "c": dbscan_fit_c, "sklearn": dbscan_fit_sklearn,
         "rule": dbscan_rule}


def dbscan_chunked(filtered_points, eps=8.0, min_samples=80, chunk_size=50000, fit="c"):
    """utils/tower_extraction.py:96-117: consecutive ``chunk_size`` rows, one fit per
    chunk, non-noise labels offset by ``current_label``.  ``chunk_size <= 0`` means a
    single global fit.  Returns int32 ``all_labels``."""
    f = _FITS[fit] if isinstance(fit, str) else fit
    pts = np.asarray(filtered_points, dtype=np.float32).reshape(-1, 3)
    n = pts.shape[0]
    all_labels = np.full(n, -1, dtype=np.int32)                       # :98
    current_label = 0                                                 # :99
    cs = int(chunk_size) if int(chunk_size) > 0 else max(n, 1)
    for start in range(0, n, cs):
        chunk = pts[start:start + cs]
        if not np.isfinite(chunk).all():
            # DBSCAN.fit validates its input first (sklearn check_array, ensure_all_finite) and raises
            # ValueError for NaN/inf; the reference's except clause (:118-119) leaves the chunk at -1
            # and current_label untouched.  (Its finally clause then escapes with UnboundLocalError -
            # tests/golden/refrun_nonfinite.npz - so this is the labelling its except clause intends.)
            continue
        chunk_labels = np.asarray(f(chunk, eps, min_samples)[0]).copy()
        chunk_labels[chunk_labels != -1] += current_label             # :114
        all_labels[start:start + cs] = chunk_labels                   # :115
        if np.any(chunk_labels != -1):                                # :116
            current_label = int(np.max(chunk_labels)) + 1
    return all_labels
