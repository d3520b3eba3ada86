"""Device orchestration of the hot path (stages B, C, D0) and the host tail (D1-D3).

``cluster_points`` is the unit the benchmark times: float32 [N,3] resident in HBM in,
per-point cluster labels + per-cluster row groups out, all computed by libpch_hip.so.
``tower_table`` turns those clusters into the reference's list of tower dicts
(utils/tower_extraction.py:125-218).
"""
from __future__ import annotations

import numpy as np
import torch

from . import obb as _obb
from . import ops

REF_CHUNK = 50000          # utils/tower_extraction.py:96


def cluster_points(raw, eps=8.0, min_points=80, chunk_size=REF_CHUNK, pct=25.0, offset=3.0,
                   fallback_offset=1.0, min_keep=1000, want_index=False, segment=True):
    """Stages B + C + D0 on a float32 [N,3] device tensor.

    Returns dict: ground (ops.ground_filter result), labels int32 [N_f] (device), nclusters,
    perm / offsets / stats (ops.segment_by_label) when ``segment``.
    """
    gf, labels, k, perm, offsets, stats = ops.tower_clusters(
        raw, eps, min_points, chunk_size, pct, offset, fallback_offset, min_keep,
        want_index=want_index, segment=segment)
    out = dict(ground=gf, labels=labels, nclusters=k)
    if segment:
        out.update(perm=perm, offsets=offsets, stats=stats)
    return out


def north_angle_deg(rotation):
    """utils/tower_extraction.py:165-177."""
    hx, hy = float(rotation[0, 0]), float(rotation[1, 0])
    nrm = float(np.linalg.norm(np.array([hx, hy, 0.0])))
    if nrm > 1e-6:
        hx, hy = hx / nrm, hy / nrm
    else:
        hx, hy = 1.0, 0.0
    a = np.degrees(np.arctan2(hy, hx))
    if a < 0:
        a += 360
    return (90 - a) % 360


def tower_table(clusters, aspect_ratio_threshold=0.8, min_height=15.0, max_width=50.0, min_width=8,
                duplicate_threshold=30.0, extent_order="unsorted", log=None, on_accept=None, obb_mode="exact"):
    """Stage D1-D3 for every cluster, in ascending label order (the iteration order of the
    reference's ``set(all_labels) - {-1}``).  Returns list of dicts with the reference's keys
    (center, rotation, extent, height, width, north_angle, points) plus 'label' and
    'aspect_ratio'.  ``log`` receives the duplicate / failure messages, ``on_accept(tower)`` is
    called for every accepted tower in order (the drop-in writes the tower LAS there).
    ``obb_mode``: "exact" (qhull on every full cluster, on the host) or "fast" (obb.boxes_fast: device
    pre-filter + native candidate search; only the accepted towers' points are copied to the host)."""
    if obb_mode not in ("exact", "fast"):
        raise ValueError("obb_mode must be 'exact' or 'fast'")
    gf = clusters["ground"]
    k = int(clusters["nclusters"])
    towers, centers = [], []
    if k == 0:
        return towers
    centroid = gf["centroid"]                                   # float32[3]
    offsets = clusters["offsets"].cpu().numpy()
    perm = clusters["perm"]
    pts = gf["points"]
    if obb_mode == "exact":
        # one gather + one D2H copy for all clustered points (noise rows stay on the device)
        rows = perm[: int(offsets[k])].long()
        host_pts = pts.index_select(0, rows).cpu().numpy()
        parts = [host_pts[offsets[label]:offsets[label + 1]] for label in range(k)]
        # boxes of all clusters (PCH_OBB_WORKERS > 1: worker processes), consumed in label order
        boxes = _obb.boxes_of(parts, extent_order)
    else:
        parts = None
        boxes = _obb.boxes_fast(pts, perm, clusters["offsets"], k, extent_order)
    # the reference's loop body (:131-215) per label; what it logs and accepts is replayed in order below, once
    # the points of the accepted towers are on the host
    events = []
    for label, (box, err) in enumerate(boxes):
        try:
            if err is not None:
                raise err
            extents, transform = box
            height = extents[2]
            width = max(extents[0], extents[1])
            aspect_ratio = height / width
            if not (height > min_height and min_width < width < max_width
                    and aspect_ratio > aspect_ratio_threshold):
                continue
            obb_center = transform[:3, 3] + centroid
            dup = None
            for c in centers:
                d = np.linalg.norm(obb_center - c)
                if d < duplicate_threshold:
                    dup = d
                    break
            if dup is not None:
                events.append(("log", f"⚠️ 跳过重复杆塔{label} (中心距: {dup:.1f}m)"))
                continue
            rot = transform[:3, :3]
            tower = dict(label=label, center=obb_center, rotation=rot, extent=extents,
                         height=height, width=width, aspect_ratio=aspect_ratio,
                         north_angle=north_angle_deg(rot), points=None)
            centers.append(obb_center)
            events.append(("tower", tower))
        except Exception as e:                                  # utils/tower_extraction.py:213-215
            events.append(("log", f"⚠️ 簇{label} 处理失败: {str(e)}"))
            continue
    accepted = [ev[1] for ev in events if ev[0] == "tower"]
    if parts is not None:
        for t in accepted:
            t["points"] = parts[t["label"]]
    elif accepted:
        spans = [(int(offsets[t["label"]]), int(offsets[t["label"] + 1])) for t in accepted]
        pos = torch.cat([torch.arange(a, b, device=perm.device) for a, b in spans])
        host_pts = pts.index_select(0, perm.index_select(0, pos).long()).cpu().numpy()
        at = 0
        for t, (a, b) in zip(accepted, spans):
            t["points"] = host_pts[at:at + (b - a)]
            at += b - a
    for kind, item in events:
        if kind == "log":
            if log:
                log(item)
            continue
        towers.append(item)
        if on_accept:
            on_accept(item)
    return towers
