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


def _accept(boxes, centroid, aspect_ratio_threshold, min_height, max_width, min_width, duplicate_threshold):
    """The reference's loop body (utils/tower_extraction.py:131-215) over the boxes of all clusters in label order:
    what it logs and what it accepts, as a list of ("log", text) / ("tower", dict) events."""
    events, centers = [], []
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
    return events


class TowerTableJob:
    """Stage D1-D3 of one tile in flight (``tower_table_async``): the clustered points are on their way to (or in)
    a shared host buffer, the boxes are being computed by the worker pool.  ``result()`` waits and returns the tower
    list; ``timings`` (after result()) holds the wall-clock split in ms."""

    def __init__(self):
        import threading
        self._done = threading.Event()
        self._towers = None
        self._error = None
        self.timings = {}

    def result(self, timeout=None):
        if not self._done.wait(timeout):
            raise TimeoutError("tower table still in flight")
        if self._error is not None:
            raise self._error
        return self._towers


_SIDE = {}            # device index -> the stream the hand-off of the clustered points runs on


def _exact_start(clusters, extent_order, timings):
    """Device side of the exact mode: the clustered rows gathered into cluster order and copied into a shared,
    registered host buffer on a side stream; returns (buffer, offsets, event after the copy, host view)."""
    import time
    t0 = time.perf_counter()
    gf = clusters["ground"]
    k = int(clusters["nclusters"])
    perm, pts = clusters["perm"], gf["points"]
    offsets = clusters["offsets"].cpu().numpy()
    if offsets[0] < 0:
        from . import _lib
        _lib.check_count(int(offsets[0]), "segment_by_label")
    m = int(offsets[k])
    dev = pts.device
    pl = _obb.pool()
    buf = pl.buffer(12 * max(m, 1), pin=True)
    side = _SIDE.get(dev.index)
    if side is None:
        side = _SIDE[dev.index] = torch.cuda.Stream(device=dev)
    side.wait_stream(torch.cuda.current_stream(dev))
    with torch.cuda.stream(side):
        # one gather + one D2H copy for all clustered points (noise rows stay on the device)
        gathered = pts.index_select(0, perm[:m].long())
        host = buf.tensor(torch.float32, 3 * m).view(m, 3)
        host.copy_(gathered, non_blocking=True)
        ev = torch.cuda.Event()
        ev.record(side)
    timings["enqueue_gather_d2h_ms"] = 1e3 * (time.perf_counter() - t0)
    # the side stream reads `pts` / `perm` and writes `gathered`: whoever waits for `ev` keeps them alive until then
    # (the caching allocator knows them as the main stream's blocks)
    return pl, buf, offsets, ev, [gathered, pts, perm]


def _exact_finish(pl, buf, offsets, ev, k, centroid, extent_order, params, timings):
    import time
    t0 = time.perf_counter()
    ev.synchronize()
    t1 = time.perf_counter()
    job = _obb.boxes_job(buf, offsets[:k + 1], np.float32, extent_order)
    boxes = _obb.job_results(job)
    t2 = time.perf_counter()
    events = _accept(boxes, centroid, *params)
    m = int(offsets[k])
    host = buf.array[:12 * m].view(np.float32).reshape(m, 3)
    for kind, t in events:
        if kind == "tower":                                     # a copy: the buffer goes back to the pool
            t["points"] = host[offsets[t["label"]]:offsets[t["label"] + 1]].copy()
    del host
    pl.release(buf)
    t3 = time.perf_counter()
    nw = max(pl.size(), 1)
    timings.update({"wait_gather_d2h_ms": 1e3 * (t1 - t0), "boxes_wall_ms": 1e3 * (t2 - t1),
                    "boxes_worker_wall_ms": 1e3 * job.worker_s, "boxes_worker_cpu_ms": 1e3 * job.worker_cpu_s,
                    "workers": nw,
                    "worker_utilisation": round(job.worker_s / max((t2 - t1) * nw, 1e-9), 3),
                    "accept_and_copy_ms": 1e3 * (t3 - t2), "clustered_points": m,
                    "bytes_to_host": 12 * m, "buffer_pinned": bool(buf.pinned)})
    return events


def tower_table_async(clusters, aspect_ratio_threshold=0.8, min_height=15.0, max_width=50.0, min_width=8,
                      duplicate_threshold=30.0, extent_order="unsorted"):
    """``tower_table`` (exact mode) without waiting for it: the hand-off of the clustered points is queued on a side
    stream, a host thread waits for it, lets the worker pool box the clusters and applies the reference's accept /
    de-dup rule.  The caller goes on - typically with ``cluster_points`` of the next tile - and collects the towers
    with ``.result()``.  ``log`` / ``on_accept`` are not taken: replay them from the returned list."""
    import threading
    job = TowerTableJob()
    k = int(clusters["nclusters"])
    if k == 0:
        job._towers = []
        job._done.set()
        return job
    centroid = clusters["ground"]["centroid"]
    params = (aspect_ratio_threshold, min_height, max_width, min_width, duplicate_threshold)
    pl, buf, offsets, ev, keep = _exact_start(clusters, extent_order, job.timings)

    def run():
        try:
            events = _exact_finish(pl, buf, offsets, ev, k, centroid, extent_order, params, job.timings)
            job._towers = [t for kind, t in events if kind == "tower"]
            job.events = events
        except BaseException as e:                              # handed to result()
            job._error = e
        finally:
            del keep[:]
            job._done.set()

    threading.Thread(target=run, name="pch-tower-table", daemon=True).start()
    return job


def tower_table(clusters, aspect_ratio_threshold=0.8, min_height=15.0, max_width=50.0, min_width=8,
                duplicate_threshold=30.0, extent_order="unsorted", log=None, on_accept=None, obb_mode="exact",
                timings=None, prepare=None):
    """Stage D1-D3 for every cluster, in ascending label order (the iteration order of the
    reference's ``set(all_labels) - {-1}``).  Returns list of dicts with the reference's keys
    (center, rotation, extent, height, width, north_angle, points) plus 'label' and
    'aspect_ratio'.  ``log`` receives the duplicate / failure messages, ``on_accept(tower)`` is
    called for every accepted tower in order (the drop-in writes the tower LAS there).
    ``obb_mode``: "exact" (qhull on every full cluster, in the worker pool of obb.py: the clustered points go
    to a shared, registered host buffer in one copy and the workers map it) or "fast" (obb.boxes_fast: device
    pre-filter + native candidate search; only the accepted towers' points are copied to the host).
    ``timings``: optional dict that receives the wall-clock split of the exact mode in ms.
    ``prepare(accepted)``: called once with all accepted towers (points filled in) before the replay starts - the
    drop-in starts writing the per-tower files there."""
    if obb_mode not in ("exact", "fast"):
        raise ValueError("obb_mode must be 'exact' or 'fast'")
    gf = clusters["ground"]
    k = int(clusters["nclusters"])
    towers = []
    if k == 0:
        return towers
    centroid = gf["centroid"]                                   # float32[3]
    params = (aspect_ratio_threshold, min_height, max_width, min_width, duplicate_threshold)
    tm = timings if timings is not None else {}
    if obb_mode == "exact":
        pl, buf, offsets, ev, keep = _exact_start(clusters, extent_order, tm)
        events = _exact_finish(pl, buf, offsets, ev, k, centroid, extent_order, params, tm)
        del keep
    else:
        offsets = clusters["offsets"].cpu().numpy()
        perm, pts = clusters["perm"], gf["points"]
        boxes = _obb.boxes_fast(pts, perm, clusters["offsets"], k, extent_order)
        events = _accept(boxes, centroid, *params)
        accepted = [ev[1] for ev in events if ev[0] == "tower"]
        if accepted:
            spans = [(int(offsets[t["label"]]), int(offsets[t["label"] + 1])) for t in accepted]
            pos = torch.cat([torch.arange(a, b, device=perm.device) for a, b in spans])
            host_pts = pts.index_select(0, perm.index_select(0, pos).long()).cpu().numpy()
            at = 0
            for t, (a, b) in zip(accepted, spans):
                t["points"] = host_pts[at:at + (b - a)]
                at += b - a
    if prepare:
        prepare([item for kind, item in events if kind == "tower"])
    for kind, item in events:
        if kind == "log":
            if log:
                log(item)
            continue
        towers.append(item)
        if on_accept:
            on_accept(item)
    return towers
