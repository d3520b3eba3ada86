"""Oracle for stage B: float32 centring + percentile height filter
(TEST INFRASTRUCTURE).

The reference performs this stage with plain numpy calls
(``/root/reference/utils/tower_extraction.py:62-64,82-89``); the oracle makes
the *same calls* on the numpy of this image (2.2.6), so it is the reference
behaviour by construction.  ``mean_seq_f32`` / ``percentile_linear_f32`` are
explicit restatements of what those numpy calls do, used to pin the device
algorithm's arithmetic; tests assert they equal the numpy calls bit for bit.
"""
from __future__ import annotations

import numpy as np


def ground_filter(raw_points_f32, offset=3.0, fallback_offset=1.0, min_keep=1000,
                  pct=25):
    """utils/tower_extraction.py:62-64,82-89 verbatim in behaviour.

    ``raw_points_f32`` is the (N,3) float32 array of line 62.
    Returns dict(centroid f32[3], points f32[N,3], base f32, threshold f32,
    keep bool[N], filtered f32[N_f,3], used_fallback bool).
    """
    raw = np.asarray(raw_points_f32, dtype=np.float32)
    centroid = np.mean(raw, axis=0)                       # :63  (float32, sequential)
    points = raw - centroid                               # :64
    z_values = points[:, 2]                               # :82
    base_height = np.percentile(z_values, pct)            # :83  (np.float32 scalar)
    thr = base_height + offset                            # :84  (float32 under NEP 50)
    keep = z_values > thr
    filtered = points[keep]
    used_fallback = False
    if len(filtered) < min_keep:                          # :87-89
        thr = base_height + fallback_offset
        keep = z_values > thr
        filtered = points[keep]
        used_fallback = True
    return dict(centroid=centroid, points=points, base=np.float32(base_height),
                threshold=np.float32(thr), keep=keep, filtered=filtered,
                used_fallback=used_fallback)


def mean_seq_f32(a):
    """What ``np.mean(a, axis=0)`` does for a C-contiguous (N,3) float32 array:
    a *sequential* float32 running sum per column (the add-reduce inner loop runs
    over the 3 contiguous columns, so numpy's pairwise summation never engages),
    then one float32 division by float32(N)."""
    a = np.asarray(a, dtype=np.float32)
    n = a.shape[0]
    if n == 0:
        return np.full(a.shape[1:], np.nan, dtype=np.float32)
    s = np.cumsum(a, axis=0, dtype=np.float32)[-1]        # sequential f32 accumulation
    return (s / np.float32(n)).astype(np.float32)


def percentile_linear_f32(sorted_or_not, q_percent=25):
    """What ``np.percentile(z, q)`` (method='linear') does for 1-D float32 ``z`` in
    numpy 2.2.6 (numpy/lib/_function_base_impl.py:4255-4258, 106-109, 4736-4760,
    4639-4660):

    * ``q = q_percent / float32(100)``            -> float32
    * ``vi = (n - 1) * q``                        -> float32 (python int is weak)
    * ``prev = floor(vi)``, ``next = prev + 1``   (both -1 when ``vi >= n-1``)
    * ``gamma = vi - prev``                       -> float32
    * lerp in float32: ``a + (b-a)*g`` and, where ``g >= 0.5``, ``b - (b-a)*(1-g)``
    * any NaN in the data -> NaN
    Returns (value f32, prev index, next index, gamma f32).
    """
    z = np.asarray(sorted_or_not, dtype=np.float32).ravel()
    n = z.shape[0]
    if n == 0:
        raise IndexError("percentile of empty array")
    q = np.float32(np.true_divide(q_percent, np.float32(100)))
    vi = np.float32(np.float32(n - 1) * q)
    prev = int(np.floor(vi))
    nxt = prev + 1
    if vi >= n - 1:
        prev = nxt = n - 1
    if vi < 0:
        prev = nxt = 0
    gamma = np.float32(vi - np.float32(np.floor(vi)))
    srt = np.sort(z)                                      # NaNs sort last
    a = srt[prev]
    b = srt[nxt]
    diff = np.float32(b - a)
    if gamma >= np.float32(0.5):
        val = np.float32(b - np.float32(diff * np.float32(np.float32(1) - gamma)))
    else:
        val = np.float32(a + np.float32(diff * gamma))
    if np.isnan(srt[-1]):
        val = np.float32(np.nan)
    return val, prev, nxt, gamma
