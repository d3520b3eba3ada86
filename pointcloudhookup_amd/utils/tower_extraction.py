"""Drop-in for the reference's ``utils/tower_extraction.py`` (== ``ui/ui/tower_extraction.py``).

Same module-level names, keyword defaults, return value, log texts, progress milestones and
side-effect files as /root/reference/utils/tower_extraction.py:20-285; the arithmetic runs on
the GPU through libpch_hip.so (height filter, chunked DBSCAN, label grouping) with only the
per-cluster box fit, LAS/xlsx writing and the Python bookkeeping on the host.  Like the
reference, read / filter / per-cluster failures are logged and what exists is returned; the ONE
raising path is the reference's own (see ``CHUNK_FAILURE`` below): with the default
``CHUNK_FAILURE = "reference"`` a chunk that scikit-learn would reject makes extract_towers raise
UnboundLocalError exactly as the reference does; with "noise" the function never raises.

Module knobs (not in the reference): ``OBB_EXTENT_ORDER`` ("unsorted" | "trimesh_sorted",
env PCH_OBB_EXTENT_ORDER) selects the extent convention of the box fit, ``DEVICE`` the GPU,
``CHUNK_FAILURE`` ("reference" | "noise", env PCH_CHUNK_FAILURE) what a chunk that scikit-learn
would reject (NaN/inf coordinates) does.  "reference" is what the reference really does, recorded
in tests/golden/refrun_nonfinite.npz: it logs the failure of that chunk and then its own
``finally: del chunk, clustering, chunk_labels`` (reference :120-122) raises UnboundLocalError out
of extract_towers, because ``clustering`` was never bound.  "noise" is the behaviour its
``except`` clause evidently intended: the chunk stays unlabelled and the run continues.
``OBB_MODE`` ("exact" | "fast", env PCH_OBB_MODE): "exact" hands every full cluster to qhull on the host
(what trimesh does); "fast" filters the clusters on the device down to the points that can be hull vertices
and searches the box natively (pointcloudhookup_amd/obb.py:boxes_fast) - same procedure, but qhull may merge
facets differently on the reduced input, so a box can differ in the centimetres; see DESIGN.md section 11.
"""
from __future__ import annotations

import os
import sys
import time
from pathlib import Path

import numpy as np

OBB_EXTENT_ORDER = os.environ.get("PCH_OBB_EXTENT_ORDER", "unsorted")
DEVICE = os.environ.get("PCH_DEVICE", "cuda:0")
CHUNK_FAILURE = os.environ.get("PCH_CHUNK_FAILURE", "reference")
OBB_MODE = os.environ.get("PCH_OBB_MODE", "exact")
PRESTART_BYTES = 100 << 20                             # LAS files from here on get worker processes early
CHUNK_SIZE = 50000                                     # utils/tower_extraction.py:96


def extract_towers(
        input_las_path,
        progress_callback=None,
        log_callback=None,
        eps=8.0,
        min_points=80,
        aspect_ratio_threshold=0.8,
        min_height=15.0,
        max_width=50.0,
        min_width=8,
        duplicate_threshold=30.0
):
    """Tower extraction with the reference's call surface (utils/tower_extraction.py:20-240).

    When ``input_las_path`` is the file ``run_voxel_downsampling`` has just written in this process, its records
    are still on the device (pointcloudhookup_amd/resident.py) and are taken instead of reading the file back -
    provided the file on disk still is the one that was written; a background writer is joined before this
    function returns, and if the file turns out to have been changed meanwhile the extraction is redone from it."""
    from .. import resident
    args = (progress_callback, log_callback, eps, min_points, aspect_ratio_threshold, min_height, max_width,
            min_width, duplicate_threshold)
    entry = resident.take(input_las_path)       # (a background writer may not even have created the file yet)
    if entry is None:
        return _extract(input_las_path, None, *args)
    try:
        towers = _extract(input_las_path, entry, *args)
    finally:
        same = resident.settle(entry)
    return towers if same else _extract(input_las_path, None, *args)


def _extract(input_las_path, resident_entry, progress_callback, log_callback, eps, min_points,
             aspect_ratio_threshold, min_height, max_width, min_width, duplicate_threshold):
    tower_obbs = []
    tower_rows = []

    def log(msg):
        (log_callback or print)(msg)

    def progress(value):
        if progress_callback:
            progress_callback(value)

    output_dir = Path("output_towers")
    output_dir.mkdir(exist_ok=True)

    # ---- read + float32 cast (reference :57-76)
    try:
        log("📂 读取点云文件...")
        progress(5)
        import torch
        from .. import las as _las
        from .. import ops, pipeline, stages
        clock = stages.Clock("extract_towers")
        if os.path.getsize(input_las_path) >= PRESTART_BYTES:
            from .. import obb as _obb
            _obb.prestart()                # the box workers import scipy while the file is being read
        dev = torch.device(DEVICE)
        if resident_entry is not None and resident_entry.records is not None:
            hdr, XYZ = resident_entry.header, resident_entry.records   # written by this process a moment ago
            clock.mark("LAS records taken from the device (no file read)")
        else:
            hdr, XYZ = _las.read_device(input_las_path, dev)           # records decoded on the GPU
            clock.mark("read LAS -> device int32 (file, H2D, decode)")
        raw = ops.cast_f32(ops.las_scale(XYZ, hdr.scales, hdr.offsets))
        del XYZ
        clock.mark("int32 -> float64 -> float32")
        header_info = {"scales": hdr.scales, "offsets": hdr.offsets,
                       "point_format": hdr.point_format, "version": hdr.version}
        log(f"✅ 点云读取完成，总点数: {raw.shape[0]}")
    except Exception as e:
        log(f"⚠️ 文件读取失败: {str(e)}")
        return tower_obbs

    # ---- percentile height filter (reference :79-93)
    try:
        log("🔍 执行高度过滤...")
        progress(10)
        if raw.shape[0] == 0:
            raise IndexError("index -1 is out of bounds for axis 0 with size 0")
        gf = ops.ground_filter(raw, 25.0, 3.0, 1.0, 1000, want_index=False)
        clock.mark("centroid + percentile filter")
        header_info["centroid"] = gf["centroid"]
        if gf["used_fallback"]:
            log(f"✅ 高度过滤完成，保留点数: {gf['count_at_offset']}")
            log("⚠️ 过滤后点数太少，尝试降低过滤阈值")
        else:
            log(f"✅ 高度过滤完成，保留点数: {gf['count']}")
    except Exception as e:
        log(f"⚠️ 高度过滤失败: {str(e)}")
        return tower_obbs
    del raw

    # ---- chunked clustering (reference :96-122); all chunks run in one device pass
    filtered = gf["points"]
    n_f = int(filtered.shape[0])
    n_chunks = (n_f + CHUNK_SIZE - 1) // CHUNK_SIZE
    log("\n=== 开始聚类处理 ===")
    progress(20)
    clusters = dict(ground=gf, nclusters=0)
    bad_chunks = _rejected_chunks(filtered, ops) if n_f else {}
    try:
        if n_f:
            labels, _, k = ops.dbscan(filtered, eps, min_points, CHUNK_SIZE, aabb=gf["aabb"])
            clock.mark("chunked DBSCAN")
            perm, offsets, stats = ops.segment_by_label(labels, filtered, k)
            clusters.update(labels=labels, nclusters=k, perm=perm, offsets=offsets, stats=stats)
            clock.mark("label grouping")
    except Exception as e:
        # all chunks run in one device pass, so a library error concerns all of them
        log(f"⚠️ 分块聚类失败（块0-{max(n_chunks - 1, 0)}）: {str(e)}")
        bad_chunks = {}
    for i in range(n_chunks):
        log(f"处理分块 {i + 1}/{n_chunks} ({min(CHUNK_SIZE, n_f - i * CHUNK_SIZE)}点)")
        if i in bad_chunks:                                            # reference :118-122
            log(f"⚠️ 分块聚类失败（块{i}）: {bad_chunks[i]}")
            if CHUNK_FAILURE == "reference":
                raise UnboundLocalError(_UNBOUND_MSG)
            continue
        progress(20 + int(50 * (i + 1) / n_chunks))

    # ---- tower detection + de-dup (reference :125-218)
    k = int(clusters["nclusters"])
    log(f"\n=== 开始杆塔检测（候选簇：{k}个） ===")
    progress(75)
    centroid = gf["centroid"]

    las_seconds = [0.0]
    las_jobs = {}

    def prepare(accepted):
        """the per-tower LAS files (reference :205-207) are written by a few threads while the accept / de-dup log
        is replayed; each tower's own messages are logged where the reference logs them"""
        if len(accepted) > 1:
            from concurrent.futures import ThreadPoolExecutor
            ex = ThreadPoolExecutor(max_workers=min(8, len(accepted)), thread_name_prefix="pch-tower-las")
            for t in accepted:
                las_jobs[t["label"]] = ex.submit(_save_tower_las_msgs, t["points"] + centroid, header_info,
                                                 output_dir / f"tower_{t['label']}.las")
            ex.shutdown(wait=False)

    def accept(t):
        label = t["label"]
        tower_obbs.append({"center": t["center"], "rotation": t["rotation"], "extent": t["extent"],
                           "height": t["height"], "width": t["width"],
                           "north_angle": t["north_angle"], "points": t["points"]})
        tower_rows.append({"ID": f"tower_{label}", "经度": t["center"][0], "纬度": t["center"][1],
                           "海拔高度": t["center"][2], "杆塔高度": t["height"],
                           "北方向偏角": t["north_angle"], "宽度": t["width"],
                           "长宽比": t["aspect_ratio"]})
        t_las = time.perf_counter()
        if label in las_jobs:
            for msg in las_jobs.pop(label).result():
                log(msg)
        else:
            original_points = t["points"] + centroid                   # float32, reference :205
            _save_tower_las(original_points, None, header_info, output_dir / f"tower_{label}.las", log)
        las_seconds[0] += time.perf_counter() - t_las
        log(f"✅ 杆塔{label}: {t['height']:.1f}m高 | {t['width']:.1f}m宽 | 中心坐标{t['center']}")
        progress(75 + int(15 * (label + 1) / max(k, 1)))

    clock.mark("per-chunk callbacks")
    pipeline.tower_table(clusters, aspect_ratio_threshold, min_height, max_width, min_width,
                         duplicate_threshold, OBB_EXTENT_ORDER, log=log, on_accept=accept, obb_mode=OBB_MODE,
                         prepare=prepare)
    clock.mark("tower boxes + accept / de-dup (D1-D3)")
    clock.move("tower boxes + accept / de-dup (D1-D3)", "per-tower LAS files", las_seconds[0])

    # ---- xlsx (reference :221-231)
    if tower_rows:
        try:
            import pandas as pd
            output_excel_path = "towers_info.xlsx"
            pd.DataFrame(tower_rows).to_excel(output_excel_path, index=False)
            log(f"\n✅ 杆塔信息已保存到: {output_excel_path}")
            log(f"检测到杆塔数量: {len(tower_obbs)}个")
        except Exception as e:
            log(f"⚠️ 保存Excel失败: {str(e)}")
    else:
        log("\n⚠️ 未检测到任何杆塔，不生成Excel文件")

    clock.mark("xlsx", sync=False)
    log("\n=== 清理内存 ===")
    del filtered, clusters, gf
    ops.release_workspace()
    progress(100)
    log("✅ 杆塔提取完成")
    return tower_obbs


_UNBOUND_MSG = ("local variable 'clustering' referenced before assignment" if sys.version_info < (3, 11)
                else "cannot access local variable 'clustering' where it is not associated with a value")


def _rejected_chunks(filtered, ops):
    """{chunk index: first sentence of scikit-learn's ValueError} for the chunks DBSCAN.fit's input
    validation rejects (NaN/inf).  The library leaves such chunks unlabelled; the common case - no
    such row anywhere - costs one small kernel."""
    out = {}
    n = int(filtered.shape[0])
    row = ops.first_nonfinite_row(filtered)
    while row >= 0:
        c = row // CHUNK_SIZE
        chunk = filtered[c * CHUNK_SIZE:(c + 1) * CHUNK_SIZE].cpu().numpy()
        out[c] = ("Input X contains NaN." if np.isnan(chunk).any()
                  else "Input X contains infinity or a value too large for dtype('float32').")
        if CHUNK_FAILURE == "reference":
            break                                                      # nothing after the first failure is reached
        nxt = (c + 1) * CHUNK_SIZE
        if nxt >= n:
            break
        row = ops.first_nonfinite_row(filtered[nxt:])
        row = row + nxt if row >= 0 else -1
    return out


def _save_tower_las_msgs(points, header_info, output_path):
    """writes the file, returns the messages the reference logs for it"""
    try:
        from .. import las as _las
        pts = np.asarray(points)
        sc, of = np.asarray(header_info["scales"], float), np.asarray(header_info["offsets"], float)
        XYZ = np.round((pts.astype(np.float64) - of) / sc).astype(np.int32)
        hdr = _las.LasHeader(point_format=header_info["point_format"], version=header_info["version"],
                             scales=sc, offsets=of)
        _las.write(str(output_path), hdr, XYZ)
        return [f"保存成功：{output_path}"]
    except Exception as e:
        return [f"⚠️ 保存失败 {output_path}: {str(e)}"]


def _save_tower_las(points, colors, header_info, output_path, log_callback=None):
    """Per-tower LAS with the input's point format / version / scales / offsets
    (reference :243-262); coordinates are re-quantised like laspy's x/y/z setters."""
    for msg in _save_tower_las_msgs(points, header_info, output_path):
        if log_callback:
            log_callback(msg)


def create_obb_geometries(tower_obbs):
    """Open3D line sets of the tower boxes (reference :265-279); needs open3d, imported lazily."""
    import open3d as o3d
    geometries = []
    for tower in tower_obbs:
        try:
            box = o3d.geometry.OrientedBoundingBox()
            box.center = tower['center']
            box.extent = tower['extent']
            box.R = tower['rotation']
            mesh = o3d.geometry.LineSet.create_from_oriented_bounding_box(box)
            mesh.paint_uniform_color([1, 0, 0])
            geometries.append(mesh)
        except Exception:
            continue
    return geometries


def extract_towers_optimized(*args, **kwargs):
    """Alias kept for callers of the older name (reference :283-285)."""
    return extract_towers(*args, **kwargs)
