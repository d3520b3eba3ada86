"""Oracle for stage E: tower dict -> axis-aligned "kuangxuan" wire box
(TEST INFRASTRUCTURE).  Restates ``/root/reference/ui/extract.py:28-38`` (bounds),
``:53-77`` (8 corners, 12 edges, 24 end points) and ``:123-132`` (width/height pick).
Pinned by the values recorded in SURVEY.md section 8c for the example tower of
``ui/extract.py:460-464`` (tests/golden/kuangxuan_boxes.json).
"""
from __future__ import annotations

import numpy as np

PRESETS = {  # ui/extract.py:261-298 (kuangxuan presets only)
    "kuangxuan_original": (1.0, 1.67, 0.5, 1.0, 1.0, 2.0),
    "kuangxuan_conservative": (0.8, 1.2, 0.4, 0.8, 0.5, 1.5),
    "kuangxuan_aggressive": (1.5, 2.0, 0.8, 1.5, 1.5, 3.0),
}

_EDGES = ((0, 1), (1, 2), (2, 3), (3, 0), (4, 5), (5, 6), (6, 7), (7, 4),
          (0, 4), (1, 5), (2, 6), (3, 7))


def kuangxuan_bounds(center, extent, preset="kuangxuan_original"):
    xl, xr, yd, yu, zd, zu = PRESETS[preset]
    cx, cy, cz = center
    width = max(extent[0], extent[1])
    height = extent[2]
    lo = np.array([cx - width * xl, cy - width * yd, cz - height * zd])
    hi = np.array([cx + width * xr, cy + width * yu, cz + height * zu])
    return lo, hi


def box_line_points(lo, hi):
    x0, y0, z0 = lo
    x1, y1, z1 = hi
    c = [[x0, y0, z0], [x1, y0, z0], [x1, y1, z0], [x0, y1, z0],
         [x0, y0, z1], [x1, y0, z1], [x1, y1, z1], [x0, y1, z1]]
    out = []
    for a, b in _EDGES:
        out.append(c[a])
        out.append(c[b])
    return np.array(out)
