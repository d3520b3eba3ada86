"""Oracle for stages D0, D2-D4 and the end-to-end ``extract_towers`` result
(TEST INFRASTRUCTURE).  Follows ``/root/reference/utils/tower_extraction.py:56-218``
statement by statement on arrays instead of a LAS file (file I/O is not numerics).
"""
from __future__ import annotations

import numpy as np

from . import dbscan as _dbscan
from . import ground_filter as _gf
from . import obb as _obb


def north_angle_deg(rotation_matrix):
    """utils/tower_extraction.py:165-177."""
    x_axis = rotation_matrix[:, 0]
    horizontal = np.array([x_axis[0], x_axis[1], 0.0])
    nrm = np.linalg.norm(horizontal)
    if nrm > 1e-6:
        horizontal = horizontal / nrm
    else:
        horizontal = np.array([1.0, 0.0, 0.0])
    a = np.degrees(np.arctan2(horizontal[1], horizontal[0]))
    if a < 0:
        a += 360
    return (90 - a) % 360


def towers_from_labels(filtered_points, all_labels, centroid,
                       aspect_ratio_threshold=0.8, min_height=15.0, max_width=50.0,
                       min_width=8, duplicate_threshold=30.0, extent_order="unsorted"):
    """utils/tower_extraction.py:125-218 (without the LAS/xlsx side effects).
    Returns (towers list[dict], n_candidates)."""
    unique_labels = sorted(set(int(v) for v in np.unique(all_labels)) - {-1})   # :125
    tower_centers, towers = [], []
    for label in unique_labels:                                                 # :131
        cluster_points = filtered_points[all_labels == label]                   # :133-134
        try:
            extents, transform = _obb.bounding_box_oriented(cluster_points, extent_order)
        except Exception:                                                       # :213-215
            continue
        height = extents[2]                                                     # :142
        width = max(extents[0], extents[1])                                     # :143
        aspect_ratio = height / width                                           # :144
        if not (height > min_height and min_width < width < max_width
                and aspect_ratio > aspect_ratio_threshold):                     # :146
            continue
        obb_center = transform[:3, 3] + centroid                                # :151
        if any(np.linalg.norm(obb_center - c) < duplicate_threshold for c in tower_centers):
            continue                                                            # :154-162
        rot = transform[:3, :3]
        towers.append(dict(label=label, center=obb_center, rotation=rot, extent=extents,
                           height=height, width=width, aspect_ratio=aspect_ratio,
                           north_angle=north_angle_deg(rot), points=cluster_points))
        tower_centers.append(obb_center)
    return towers, len(unique_labels)


def extract_towers_arrays(x, y, z, eps=8.0, min_points=80, aspect_ratio_threshold=0.8,
                          min_height=15.0, max_width=50.0, min_width=8,
                          duplicate_threshold=30.0, chunk_size=50000, fit="c",
                          extent_order="unsorted"):
    """End-to-end B0..D3 on the float64 coordinate columns laspy would hand over
    (``las.x, las.y, las.z``).  Returns dict with every intermediate."""
    raw = np.stack([x, y, z], axis=1).astype(np.float32)                        # :62
    gf = _gf.ground_filter(raw)
    labels = _dbscan.dbscan_chunked(gf["filtered"], eps, min_points, chunk_size, fit)
    towers, ncand = towers_from_labels(gf["filtered"], labels, gf["centroid"],
                                       aspect_ratio_threshold, min_height, max_width,
                                       min_width, duplicate_threshold, extent_order)
    return dict(raw=raw, ground=gf, labels=labels, towers=towers, n_candidates=ncand)
