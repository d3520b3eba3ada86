#!/usr/bin/env python3
"""Is qhull's run really unchanged when it is not shown the rows inside its initial simplex?  (pointcloudhookup_amd/obb.py,
"what qhull is shown").  Random clusters of seven shapes (towers, boxes, coarse grids with many duplicate points, flat
slabs, rotated towers, mast + cross arm) through obb.hull_candidates with the reduction off and on: hull vertices and
candidate directions must be identical, bit for bit.  `--sabotage` uses a deliberately wrong (second best) tetrahedron in
the python reference of the prediction and must report mismatches - the check has teeth.
  python tools/prefilter_check.py [seed] [trials] [--sabotage]        (CPU only, ~15 ms per trial)"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pointcloudhookup_amd import obb  # noqa: E402

args = [a for a in sys.argv[1:] if not a.startswith("--")]
seed = int(args[0]) if args else 0
trials = int(args[1]) if len(args) > 1 else 500
sabotage = "--sabotage" in sys.argv


def cluster(rng):
    n = int(rng.integers(100, 60000))
    kind = int(rng.integers(0, 7))
    if kind == 0:
        p = rng.normal(size=(n, 3)) * [2.5, 2.5, 9]
    elif kind == 1:
        p = rng.uniform(-1, 1, (n, 3)) * [4, 6, 20]
    elif kind == 2:
        p = np.round(rng.normal(size=(n, 3)) * [2, 3, 8], 2)
    elif kind == 3:
        p = np.round(rng.normal(size=(n, 3)) * [2, 3, 8], 1)          # many duplicates / ties
    elif kind == 4:
        p = rng.normal(size=(n, 3)) * [10, 0.3, 5]                    # a slab
    elif kind == 5:
        p = (rng.normal(size=(n, 3)) * [2.5, 2.5, 9]) @ np.linalg.qr(rng.normal(size=(3, 3)))[0]
    else:
        p = np.vstack([rng.normal(size=(n // 2, 3)) * [1, 1, 12], rng.uniform(-6, 6, (n - n // 2, 3)) * [1, 1, 0.2]])
    return (p + rng.uniform(-500, 500, 3)).astype(np.float32), kind


def wrong_reduction(points):
    """the python reference of the prediction with the SECOND best last vertex"""
    p = np.asarray(points, dtype=np.float64)
    s = obb.predicted_simplex(p)
    if s is None:
        return p
    mp = []
    for k in range(3):
        mp += [int(p[:, k].argmin()), int(p[:, k].argmax())]
    others = [c for c in dict.fromkeys(mp) if c not in s]
    if not others:
        return p
    T = p[s[:3] + [others[0]]]
    try:
        A = np.linalg.inv((T[1:] - T[0]).T)
    except np.linalg.LinAlgError:
        return p
    b = (p - T[0]) @ A.T
    inside = (b > 1e-6).all(axis=1) & (b.sum(axis=1) < 1.0 - 1e-6)
    return p[~inside]


rng = np.random.default_rng(seed)
assert obb._hostlib() is not None, "libpch_obbhost.so is not built"
obb.qhull_input(cluster(rng)[0])                      # runs the per-process self-check
assert obb._PREFILTER, "the reduction is switched off (self-check failed or PCH_OBB_PREFILTER=0)"
bad = removed = total = stood_down = 0
t0 = time.time()
for t in range(trials):
    pts, kind = cluster(rng)
    obb._PREFILTER = False
    try:
        v0, a0 = obb.hull_candidates(pts)
    except Exception:
        obb._PREFILTER = True
        continue
    obb._PREFILTER = True
    if sabotage:
        real = obb.qhull_input
        obb.qhull_input = wrong_reduction
        try:
            v1, a1 = obb.hull_candidates(pts)
        finally:
            obb.qhull_input = real
        shown = len(wrong_reduction(pts))
    else:
        v1, a1 = obb.hull_candidates(pts)
        shown = len(obb.qhull_input(pts))
    total += len(pts)
    removed += len(pts) - shown
    stood_down += shown == len(pts)
    if not (v0.shape == v1.shape and np.array_equal(v0, v1) and a0.shape == a1.shape and np.array_equal(a0, a1)):
        bad += 1
        print("MISMATCH trial", t, "kind", kind, "n", len(pts), flush=True)
print(f"seed {seed}: {trials} trials, {bad} mismatches, {removed / max(total, 1):.3f} of the points not shown, "
      f"{stood_down} trials stood down, {time.time() - t0:.0f} s" + ("  [sabotaged tetrahedron]" if sabotage else ""))
sys.exit(1 if (bad and not sabotage) else 0)
