"""CPU suite, part 2: host logic of the product package (no compute calls into the HIP
library - there is no GPU here): LAS I/O, box fit vs the oracle, drop-in call surface, C-ABI
symbol export, loud failure without a GPU, and the world_size-2 gloo reconciliation."""
import inspect
import json
import os
import re
import subprocess
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLD = os.path.join(ROOT, "tests", "golden")


# ------------------------------------------------------------------ C ABI
def test_library_loads_and_exports_every_declared_symbol():
    from pointcloudhookup_amd import _lib
    L = _lib.lib()
    assert L.pch_version() >= 100
    header = open(os.path.join(ROOT, "include", "pch_hip.h")).read()
    declared = set(re.findall(r"\b(pch_[a-z0-9_]+)\s*\(", header))
    assert declared, "no declarations parsed from include/pch_hip.h"
    assert declared == set(_lib.exported_symbols())
    nm = subprocess.run(["nm", "-D", "--defined-only", _lib.LIB_PATH], capture_output=True, text=True).stdout
    exported = set(re.findall(r" T (pch_[a-z0-9_]+)", nm))
    assert declared <= exported, declared - exported
    # workspace sizing is pure host code and may be called without a device
    assert L.pch_dbscan_ws_bytes(1000) > 0 and L.pch_ground_filter_ws_bytes(1000) > 0
    assert L.pch_voxel_downsample_ws_bytes(1000, 100) > 0 and L.pch_segment_by_label_ws_bytes(1000, 3) > 0
    assert L.pch_mean_seq_f32_ws_bytes(10 ** 8) < 2e8          # summary tables: < 2 B/point


def test_product_path_fails_loudly_without_gpu_tensors():
    import torch
    from pointcloudhookup_amd import ops
    with pytest.raises(TypeError, match="no CPU fallback"):
        ops.ground_filter(torch.zeros((10, 3), dtype=torch.float32))
    with pytest.raises(TypeError, match="no CPU fallback"):
        ops.dbscan(torch.zeros((10, 3), dtype=torch.float32))
    with pytest.raises(TypeError, match="no CPU fallback"):
        ops.voxel_downsample(torch.zeros((10, 3), dtype=torch.float64), 0.1)


def test_product_package_never_imports_the_oracle():
    pkg = os.path.join(ROOT, "pointcloudhookup_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".h", ".cpp")):
                txt = open(os.path.join(dirpath, f)).read()
                assert not re.search(r"^\s*(from|import)\s+oracle\b", txt, re.M), f
                assert "oracle/" not in txt, f


# ------------------------------------------------------------------ LAS
@pytest.mark.parametrize("fmt,ver", [(0, (1, 2)), (1, (1, 2)), (3, (1, 2)), (2, (1, 3)), (6, (1, 4)), (7, (1, 4))])
def test_las_roundtrip(tmp_path, fmt, ver):
    from pointcloudhookup_amd import las
    rng = np.random.default_rng(fmt)
    XYZ = rng.integers(-10 ** 8, 10 ** 8, (1234, 3)).astype(np.int32)
    hdr = las.LasHeader(point_format=fmt, version=ver, scales=np.array([0.001, 0.001, 0.01]),
                        offsets=np.array([437000.0, 3139000.0, 0.0]))
    p = str(tmp_path / "t.las")
    las.write(p, hdr, XYZ)
    assert os.path.getsize(p) == las.HEADER_SIZE[ver] + 1234 * las.RECORD_LEN[fmt]
    d = las.read(p)
    np.testing.assert_array_equal(d.XYZ, XYZ)
    assert d.header.point_format == fmt and tuple(d.header.version) == ver
    np.testing.assert_array_equal(d.header.scales, hdr.scales)
    np.testing.assert_array_equal(d.header.offsets, hdr.offsets)
    np.testing.assert_array_equal(d.x, XYZ[:, 0] * 0.001 + 437000.0)          # laspy scaled view
    np.testing.assert_allclose(d.header.maxs, d.XYZ.max(0) * hdr.scales + hdr.offsets)


def test_las_errors(tmp_path):
    from pointcloudhookup_amd import las
    with pytest.raises(FileNotFoundError):
        las.read(str(tmp_path / "missing.las"))
    bad = tmp_path / "bad.las"
    bad.write_bytes(b"NOPE" + b"\0" * 400)
    with pytest.raises(ValueError):
        las.read(str(bad))
    empty = str(tmp_path / "e.las")
    las.write(empty, las.LasHeader(), np.zeros((0, 3), np.int32))
    assert len(las.read(empty)) == 0


# ------------------------------------------------------------------ stage D host code vs oracle
def test_product_obb_equals_oracle_obb():
    from oracle import obb as o
    from pointcloudhookup_amd import obb as p
    rng = np.random.default_rng(0)
    for t in range(4):
        P = rng.normal(0, [2.5, 2.5, 9], (6000, 3))
        P[:, 2] = np.clip(P[:, 2] + 22, 3, 45)
        R = np.linalg.qr(rng.normal(size=(3, 3)))[0] if t % 2 else np.eye(3)
        X = (P @ R.T).astype(np.float32)
        for order in ("unsorted", "trimesh_sorted"):
            e1, T1 = o.bounding_box_oriented(X, order)
            e2, T2 = p.bounding_box_oriented(X, order)
            np.testing.assert_array_equal(e1, e2)
            np.testing.assert_array_equal(T1, T2)
    with pytest.raises(ValueError):
        p.oriented_bounds(X, "sideways")


def test_north_angle_matches_oracle():
    from oracle import towers as ot
    from pointcloudhookup_amd import pipeline
    rng = np.random.default_rng(1)
    for _ in range(20):
        R = np.linalg.qr(rng.normal(size=(3, 3)))[0]
        assert pipeline.north_angle_deg(R) == ot.north_angle_deg(R)


def test_extract_boxes_match_reference_golden():
    from pointcloudhookup_amd.ui import extract as ex
    g = json.load(open(os.path.join(GOLD, "kuangxuan_boxes.json")))
    for preset, ref in g["presets"].items():
        method, params = ex.get_bbox_preset(preset)
        assert method == "kuangxuan"
        lo, hi = ex.create_bbox_using_kuangxuan_method(np.array(g["center"]), 20.1, 17.4, **params)
        np.testing.assert_allclose(lo, ref["min"], atol=5e-3)
        np.testing.assert_allclose(hi, ref["max"], atol=5e-3)
        pts, color = ex.create_bbox_lineset_from_bounds(lo, hi)
        assert pts.shape == (24, 3) and pts.dtype == np.float64 and color == (1.0, 0.0, 0.0)
        if "lines" in ref:
            np.testing.assert_array_equal(pts, np.array(ref["lines"]))
    assert ex.get_bbox_preset("no_such_preset") == ex.get_bbox_preset("kuangxuan_original")
    boxes = ex.create_enhanced_tower_boxes_kuangxuan(
        [dict(center=np.array(g["center"]), extent=np.array(g["extent"]), rotation=np.eye(3))])
    assert [b[0].shape for b in boxes] == [(24, 3), (24, 3), (2, 3)]
    assert [b[1] for b in boxes] == [(1.0, 0.0, 0.0), (1.0, 1.0, 0.0), (0.0, 1.0, 0.0)]
    with pytest.raises(FileNotFoundError):
        ex.extract_and_visualize_towers("/nonexistent/file.las", [])


# ------------------------------------------------------------------ drop-in call surface
REF_SIGS = {
    # reference file:line of every def the GUI imports
    ("ui.import_PC", "process_chunk"): ["points_chunk", "voxel_size"],
    ("ui.import_PC", "run_voxel_downsampling"): ["input_path", "output_path", "voxel_size", "chunk_size",
                                                 "progress_callback", "log_callback"],
    ("ui.Sampling", "process_chunk"): ["points_chunk", "las", "voxel_size"],
    ("ui.Sampling", "voxel_downsample_open3d"): ["input_path", "output_path", "voxel_size", "chunk_size"],
    ("utils.tower_extraction", "extract_towers"): ["input_las_path", "progress_callback", "log_callback", "eps",
                                                   "min_points", "aspect_ratio_threshold", "min_height",
                                                   "max_width", "min_width", "duplicate_threshold"],
    ("utils.tower_extraction", "_save_tower_las"): ["points", "colors", "header_info", "output_path",
                                                    "log_callback"],
    ("utils.tower_extraction", "create_obb_geometries"): ["tower_obbs"],
    ("ui.extract", "extract_and_visualize_towers"): ["las_path", "tower_obbs", "scale_factors", "line_color",
                                                     "adaptive_scaling", "use_kuangxuan_method",
                                                     "kuangxuan_preset"],
    ("ui.extract", "extract_and_visualize_towers_kuangxuan"): ["las_path", "tower_obbs", "bbox_method",
                                                               "bbox_params", "line_color"],
    ("ui.extract", "create_bbox_using_kuangxuan_method"): ["center", "width", "height", "x_left_factor",
                                                           "x_right_factor", "y_down_factor", "y_up_factor",
                                                           "z_down_factor", "z_up_factor"],
}
REF_DEFAULTS = {
    ("ui.import_PC", "run_voxel_downsampling"): dict(voxel_size=0.1, chunk_size=1000000),
    ("ui.Sampling", "voxel_downsample_open3d"): dict(chunk_size=1000000),
    ("utils.tower_extraction", "extract_towers"): dict(eps=8.0, min_points=80, aspect_ratio_threshold=0.8,
                                                       min_height=15.0, max_width=50.0, min_width=8,
                                                       duplicate_threshold=30.0),
    ("ui.extract", "extract_and_visualize_towers"): dict(use_kuangxuan_method=True,
                                                         kuangxuan_preset="kuangxuan_original"),
}


@pytest.mark.parametrize("mod,fn", sorted(REF_SIGS))
def test_dropin_signatures(mod, fn):
    import importlib
    m = importlib.import_module(f"pointcloudhookup_amd.{mod}")
    sig = inspect.signature(getattr(m, fn))
    assert list(sig.parameters) == REF_SIGS[(mod, fn)]
    for k, v in REF_DEFAULTS.get((mod, fn), {}).items():
        assert sig.parameters[k].default == v


def test_dropin_twin_module_and_alias():
    from pointcloudhookup_amd.ui.ui import tower_extraction as twin
    from pointcloudhookup_amd.utils import tower_extraction as te
    assert twin.extract_towers is te.extract_towers
    assert te.extract_towers_optimized.__doc__


def test_voxel_dropin_raises_only_for_missing_input(tmp_path):
    from pointcloudhookup_amd.ui import import_PC
    with pytest.raises(FileNotFoundError):
        import_PC.run_voxel_downsampling(str(tmp_path / "nope.las"), str(tmp_path / "out" / "o.las"))


def test_extract_towers_never_raises(tmp_path, monkeypatch):
    """Reference contract (utils/tower_extraction.py:74-76): read failure -> log + []."""
    from pointcloudhookup_amd.utils import tower_extraction as te
    monkeypatch.chdir(tmp_path)
    logs, prog = [], []
    out = te.extract_towers(str(tmp_path / "missing.las"), progress_callback=prog.append,
                            log_callback=logs.append)
    assert out == [] and prog == [5]
    assert logs[0] == "📂 读取点云文件..." and logs[1].startswith("⚠️ 文件读取失败")
    assert (tmp_path / "output_towers").is_dir()


# ------------------------------------------------------------------ synthetic generators
def test_synth_generator_shape_and_determinism():
    from pointcloudhookup_amd import synth
    a = synth.corridor_numpy(200000, seed=5, towers=3)
    b = synth.corridor_numpy(200000, seed=5, towers=3)
    np.testing.assert_array_equal(a, b)
    assert a.shape == (200000, 3) and a[:, 0].min() >= -15 and a[:, 1].max() < 115
    L = synth.corridor_length(200000)
    assert L == pytest.approx(20.0)
    tower = a[a[:, 2] > 0.45]
    assert 0.09 < len(tower) / len(a) < 0.11 and tower[:, 2].max() <= 45.0
    u = synth.corridor_numpy(50000, seed=1, kind="uniform", offset=True)
    assert u[:, 2].min() >= 80.0 and u[:, 2].max() <= 110.0


# ------------------------------------------------------------------ multi-rank reconciliation (gloo)
_WORKER = r'''
import os, sys
sys.path.insert(0, sys.argv[1])
import numpy as np, torch, torch.distributed as dist
from pointcloudhookup_amd import tiles
rank, world, local = tiles.init_from_env(backend="gloo")
assert world == 2 and dist.get_backend() == "gloo"
k = 3 if rank == 0 else 2                       # clusters found on this rank's tile
table = torch.arange(k * 4, dtype=torch.float32).reshape(k, 4) + 100 * rank
off, total, glob, owner = tiles.reconcile(k, table)
assert total == 5 and off == (0 if rank == 0 else 3), (off, total)
assert owner.tolist() == [0, 0, 0, 1, 1]
exp = torch.cat([torch.arange(12, dtype=torch.float32).reshape(3, 4),
                 torch.arange(8, dtype=torch.float32).reshape(2, 4) + 100])
assert torch.equal(glob, exp)
# a rank with zero clusters still takes part
off2, total2, glob2, owner2 = tiles.reconcile(0 if rank == 0 else 4, torch.ones((4, 2)) * rank)
assert total2 == 4 and off2 == 0 and owner2.tolist() == [1, 1, 1, 1]
assert tiles.tiles_of_rank(5, rank, world) == ([0, 1, 2] if rank == 0 else [3, 4])
# more clusters on one rank than the first exchange carries: the exactly sized second exchange, same result;
# and a table whose dtype cannot carry the count (integers): the two-step form from the start
tiles.RECONCILE_CAP = 2
off3, total3, glob3, owner3 = tiles.reconcile(k, table)
assert (off3, total3) == (off, total) and torch.equal(glob3, exp) and owner3.tolist() == [0, 0, 0, 1, 1]
tiles.RECONCILE_CAP = 1024
off4, total4, glob4, owner4 = tiles.reconcile(k, table.to(torch.int64))
assert (off4, total4) == (off, total) and torch.equal(glob4, exp.to(torch.int64))
dist.barrier()
dist.destroy_process_group()
print("rank", rank, "ok")
'''


def test_reconcile_world2_gloo(tmp_path):
    script = tmp_path / "worker.py"
    script.write_text(_WORKER)
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT="29731", WORLD_SIZE="2")
    procs = [subprocess.Popen([sys.executable, str(script), ROOT], env=dict(env, RANK=str(r), LOCAL_RANK=str(r)),
                              stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True) for r in range(2)]
    outs = [p.communicate(timeout=180)[0] for p in procs]
    for r, (p, o) in enumerate(zip(procs, outs)):
        assert p.returncode == 0, o
        assert f"rank {r} ok" in o


def _bench(*argv, **env):
    e = dict(os.environ)
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        e.pop(k, None)
    e.update(env)
    return subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), *argv], env=e, capture_output=True,
                          text=True, timeout=300)


def test_bench_launches_its_own_ranks():
    """`python bench.py --gpus 2` (no launcher around it) must start two ranks itself - as a CHILD
    `torch.distributed.run`, before the parent has imported torch - relay rank 0's single JSON line and pass the
    ranks' exit code on.  GPU-free through the dry-run hook; the real step is covered on the GPU box."""
    import json
    r = _bench("--gpus", "2", "--steps", "3", "--warmup", "1", "--mode", "tiled", PCH_BENCH_DRYRUN="1")
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1, r.stdout                                   # ONE line on stdout, everything else on stderr
    out = json.loads(lines[0])
    assert out["n_gpus"] == 2 and out["ranks_seen"] == 2 and out["mode"] == "tiled" and out["steps"] == 3
    assert out["parent_imported_torch"] == "False"
    # a rank that fails: non-zero exit, no result line
    r = _bench("--gpus", "2", PCH_BENCH_DRYRUN="fail:1")
    assert r.returncode != 0 and not [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    # one GPU: no launcher in between, the process itself is the rank
    r = _bench("--gpus", "1", PCH_BENCH_DRYRUN="1")
    assert r.returncode == 0 and json.loads(r.stdout)["ranks_seen"] == 1
    # --gpus that contradicts the launcher's world size is an error, not a silent single-rank run
    r = _bench("--gpus", "4", PCH_BENCH_DRYRUN="1", RANK="0", WORLD_SIZE="2", LOCAL_RANK="0")
    assert r.returncode != 0 and "WORLD_SIZE=2" in r.stderr
    # more than 3 ranks on one shared device are refused before any GPU work
    r = _bench("--gpus", "4", PCH_BENCH_SINGLE_DEVICE="1", RANK="0", WORLD_SIZE="4", LOCAL_RANK="0")
    assert r.returncode != 0 and "at most 3 ranks" in r.stderr


def test_dedup_centres_first_wins():
    from pointcloudhookup_amd import tiles
    c = np.array([[0, 0, 0], [10, 0, 0], [100, 0, 0], [29.9, 0, 0], [131, 0, 0]], float)
    assert tiles.dedup_centres(c, 30.0) == [0, 2, 4]          # 1 and 3 duplicate 0; 4 is 31 m from 2
    assert tiles.dedup_centres(np.zeros((0, 3))) == []
    assert tiles.reconcile(2, __import__("torch").zeros((2, 8)))[0:2] == (0, 2)   # single process


def test_obb_worker_processes_return_the_serial_results():
    """PCH_OBB_WORKERS > 1 (spawned worker processes) gives exactly the boxes of the serial loop,
    in order, and reports a failing cluster the same way."""
    from pointcloudhookup_amd import obb
    rng = np.random.default_rng(5)
    clusters = [(rng.normal(0, 1, (400 + 50 * i, 3)) * np.array([3.0, 2.0, 9.0 + i])).astype(np.float32)
                for i in range(5)]
    clusters.insert(2, np.zeros((3, 3), np.float32))            # degenerate: qhull raises
    serial = list(obb.boxes_of(clusters, "unsorted", workers=1))
    pooled = list(obb.boxes_of(clusters, "unsorted", workers=2))
    assert len(serial) == len(pooled) == len(clusters)
    for (b0, e0), (b1, e1) in zip(serial, pooled):
        assert (e0 is None) == (e1 is None)
        if e0 is None:
            np.testing.assert_array_equal(b0[0], b1[0])
            np.testing.assert_array_equal(b0[1], b1[1])
    assert serial[2][1] is not None


# ------------------------------------------------------------------ x-tiles with a halo (BASELINE config 4)
_TILED_WORKER = r'''
import os, sys
sys.path.insert(0, sys.argv[1])
import numpy as np, torch, torch.distributed as dist
from pointcloudhookup_amd import tiles
from oracle import dbscan as odb                      # the checker (tests only)
rank, world, local = tiles.init_from_env(backend="gloo")
use_gpu = sys.argv[2] == "gpu"
EPS, MS = 8.0, 40

def cloud():
    rng = np.random.default_rng(4242)
    L = 600.0
    edges = tiles.tile_edges(0.0, L, world)
    parts = []
    for cx in list(edges[1:-1]) + [edges[1] - 30.0, 37.0, L - 21.0]:      # towers ON every inner edge + elsewhere
        parts.append(rng.normal([cx, 50.0, 20.0], [2.5, 2.5, 8.0], (700, 3)))
    parts.append(np.column_stack([rng.uniform(0, L, 1500), rng.uniform(0, 100, 1500), rng.uniform(0, 40, 1500)]))
    # a thin bridge of core points along x that crosses an edge, and sparse border material around it
    bx = np.linspace(edges[1] - 40.0, edges[1] + 40.0, 900)
    parts.append(np.column_stack([bx, np.full_like(bx, 80.0), np.full_like(bx, 5.0)]) + rng.normal(0, 0.3, (900, 3)))
    X = np.vstack(parts).astype(np.float32)
    return X[rng.permutation(len(X))], edges

X, edges = cloud()

class OracleFit:                                       # CPU stand-in for tiles.HipFit (tests only)
    def fit(self, pts):
        self.pts = np.asarray(pts, np.float32)
        self.labels, self.core = odb.dbscan_fit_c(self.pts, EPS, MS)
        k = int(self.labels.max()) + 1 if (self.labels >= 0).any() else 0
        return torch.from_numpy(self.labels.astype(np.int32)), torch.from_numpy(self.core.astype(bool)), k
    def strip_pairs(self, x_lo, x_hi, cap):              # one (row, cluster) pair per grid cell, as the library does
        x_lo, x_hi = np.float32(x_lo), np.float32(x_hi)
        c = self.core.astype(bool) & (self.pts[:, 0] >= x_lo) & (self.pts[:, 0] < x_hi)
        idx = np.flatnonzero(c)
        side = EPS / np.sqrt(3.0) * (1.0 - 2.0 ** -16)
        cell = np.floor((self.pts[idx].astype(np.float64) - self.pts.min(0).astype(np.float64)) / side).astype(np.int64)
        _, first = np.unique(cell, axis=0, return_index=True)
        rep = idx[np.sort(first)]
        out = np.zeros((cap, 2), np.int32)
        m = min(len(rep), cap)
        out[:m, 0], out[:m, 1] = rep[:m], self.labels[rep[:m]]
        self.strip_cells = len(rep)
        return torch.from_numpy(out), torch.tensor([len(rep)], dtype=torch.int32)
    def relabel(self, cmap):
        cmap = np.asarray(cmap, np.int64)
        new = np.full(len(self.pts), -1, np.int64)
        c = self.core.astype(bool)
        new[c] = cmap[self.labels[c]]
        P = self.pts.astype(np.float64)
        for i in np.flatnonzero(~c):
            d = ((P[c] - P[i]) ** 2)
            near = (d[:, 0] + d[:, 1] + d[:, 2]) <= EPS * EPS
            if near.any():
                new[i] = new[c][near].min()
        return torch.from_numpy(new.astype(np.int32))

take, own = tiles.tile_select(X[:, 0], edges, rank, 2 * EPS)
rows = np.flatnonzero(take)
pts = X[rows]
tiles.TILED_PCAP = 2                                   # a pair buffer that overflows: the exchange must repeat once
tiles.TILED_KCAP = 4                                   # ... and so must the cluster table
if use_gpu:
    dev = torch.device("cuda:0")
    labels, K, words = tiles.cluster_tiled(torch.from_numpy(pts).to(dev), torch.from_numpy(rows), own[rows],
                                           edges[rank], edges[rank + 1], EPS, MS, extra=(rank + 5,))
    labels = labels.cpu().numpy()
else:
    labels, K, words = tiles.cluster_tiled(torch.from_numpy(pts), torch.from_numpy(rows), own[rows],
                                           edges[rank], edges[rank + 1], EPS, MS, fit=OracleFit(), extra=(rank + 5,))
    labels = labels.numpy()
assert np.asarray(words)[:, 0].tolist() == [r + 5 for r in range(world)]      # the words that rode along
want, _ = odb.dbscan_fit_c(X, EPS, MS)                 # one DBSCAN over the whole cloud
o = own[rows]
assert K == want.max() + 1, (K, want.max() + 1)
assert np.array_equal(labels[o], want[rows[o]]), np.flatnonzero(labels[o] != want[rows[o]])[:10]
# the towers on the inner edges really are cut by the tiling: some cluster has owned points on both sides
if world > 1 and rank == 0:
    side = np.searchsorted(edges, X[:, 0], side="right") - 1
    cut = [c for c in range(K) if len(set(side[want == c])) > 1]
    assert len(cut) >= world - 1, cut
got = torch.zeros(len(X), dtype=torch.int64) - 7
got[torch.from_numpy(rows[o])] = torch.from_numpy(labels[o].astype(np.int64))
if world > 1:
    parts = [torch.zeros_like(got) for _ in range(world)]
    dist.all_gather(parts, got)
    full = torch.stack(parts).max(dim=0).values.numpy()
    assert np.array_equal(full, want)                  # every point owned exactly once, labels global
    dist.barrier()
    dist.destroy_process_group()
print("rank", rank, "ok", K)
'''


def _run_tiled(tmp_path, world, mode, port):
    script = tmp_path / f"tiled_{world}.py"
    script.write_text(_TILED_WORKER)
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), WORLD_SIZE=str(world))
    procs = [subprocess.Popen([sys.executable, str(script), ROOT, mode], env=dict(env, RANK=str(r), LOCAL_RANK=str(r)),
                              stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True) for r in range(world)]
    outs = [p.communicate(timeout=300)[0] for p in procs]
    for r, (p, o) in enumerate(zip(procs, outs)):
        assert p.returncode == 0, o
        assert f"rank {r} ok" in o


@pytest.mark.parametrize("world", [1, 2, 3])
def test_cluster_tiled_equals_global_dbscan_gloo(tmp_path, oracle_clib, world):
    """x-tiles with a 2*eps halo, towers and a thin bridge cut by the tile edges: the labels every rank reports
    for its own points equal ONE DBSCAN over the whole cloud (exchange and union-find over gloo; the local
    fit is the CPU oracle here, the HIP kernels in tests/test_gpu_e2e.py)."""
    _run_tiled(tmp_path, world, "cpu", 29741 + world)


# ------------------------------------------------------------------ native LAS header parse (no GPU needed)
def test_native_las_header_on_byte_built_files(tmp_path):
    import las_bytes
    from pointcloudhookup_amd import las
    from pointcloudhookup_amd._lib import PchError
    rng = np.random.default_rng(3)
    XYZ = rng.integers(-2**31, 2**31 - 1, size=(1000, 3), dtype=np.int64).astype(np.int32)
    cases = [dict(point_format=3, version=(1, 2)),
             dict(point_format=2, version=(1, 2), vlr_payloads=(b"x" * 40, b"y" * 7), pad_before_points=13),
             dict(point_format=1, version=(1, 3), extra_bytes=6),
             dict(point_format=6, version=(1, 4), vlr_payloads=(b"z" * 100,), extra_bytes=3),
             dict(point_format=0, version=(1, 4), legacy_count_zero=True)]
    for i, kw in enumerate(cases):
        p = str(tmp_path / f"c{i}.las")
        want = las_bytes.build(p, XYZ, **kw)
        for h in (las.read_header_native(p), las.read_header(p)):       # the library and the python parser agree
            assert h.point_count == want["n"] and h.record_length == want["record_length"]
            assert h.offset_to_points == want["offset_to_points"] and h.header_size == want["header_size"]
            assert h.point_format == want["point_format"] and tuple(h.version) == want["version"]
            np.testing.assert_array_equal(h.scales, want["scales"])
            np.testing.assert_array_equal(h.offsets, want["offsets"])
            np.testing.assert_array_equal(h.mins, want["mins"])
            np.testing.assert_array_equal(h.maxs, want["maxs"])
    bad = tmp_path / "bad.las"
    bad.write_bytes(b"NOPE" + b"\0" * 400)
    with pytest.raises(PchError):
        las.read_header_native(str(bad))
    trunc = tmp_path / "trunc.las"
    trunc.write_bytes(open(str(tmp_path / "c0.las"), "rb").read()[:5000])
    with pytest.raises(PchError):
        las.read_header_native(str(trunc))
    with pytest.raises(FileNotFoundError):
        las.read_header_native(str(tmp_path / "missing.las"))
    # a forged LAS 1.4 64-bit point count whose product with the record length wraps around 2^64: the truncation
    # check must not be fooled into accepting it (the reader would then copy from beyond the mapping)
    import struct
    good = bytearray(open(str(tmp_path / "c3.las"), "rb").read())      # format 6, LAS 1.4, 33-byte records
    rl = struct.unpack_from("<H", good, 105)[0]
    for forged in ((2**64 // rl) + 2, 2**64 - 1, 2**63):
        b = bytearray(good)
        struct.pack_into("<Q", b, 247, forged)
        f = tmp_path / "forged.las"
        f.write_bytes(bytes(b))
        with pytest.raises(PchError, match="truncated"):
            las.read_header_native(str(f))
    # offset to the point data inside the header block
    b = bytearray(good)
    struct.pack_into("<I", b, 96, 100)
    f = tmp_path / "inside.las"
    f.write_bytes(bytes(b))
    with pytest.raises(PchError, match="inside"):
        las.read_header_native(str(f))


# ------------------------------------------------------------------ shared percentile / threshold across ranks
_PCT_WORKER = r'''
import os, sys
sys.path.insert(0, sys.argv[1])
import numpy as np, torch, torch.distributed as dist
from pointcloudhookup_amd import tiles
rank, world, local = tiles.init_from_env(backend="gloo")
use_gpu = sys.argv[2] == "gpu"

def keys(v):
    u = np.ascontiguousarray(v, dtype=np.float32).view(np.uint32).astype(np.uint64)
    k = np.where(u & 0x80000000, (~u) & 0xFFFFFFFF, u | 0x80000000)
    return np.where(np.isnan(v), 0xFFFFFFFF, k).astype(np.uint64)

class NumpySelect:                                     # CPU stand-in for tiles.HipSelect (tests only)
    def hist(self, values, p, prefix):
        v = np.asarray(values, np.float32)
        k = keys(v)
        if p == 0:
            b = k >> 20
        elif p == 1:
            b = ((k >> 8) & 0xFFF)[(k >> 20) == prefix]
        else:
            b = (k & 0xFF)[(k >> 8) == prefix]
        return np.bincount(b.astype(np.int64), minlength=4096)[:4096].astype(np.int64), int(np.isnan(v).sum())
    def min_above(self, values, key):
        k = keys(np.asarray(values, np.float32))
        k = k[k > key]
        return int(k.min()) if len(k) else 0xFFFFFFFF

rng = np.random.default_rng(99)
cases = []
for n, q in ((1, 25), (2, 25), (5, 50), (1000, 25), (100003, 25), (100003, 99.5), (4096, 0), (4096, 100), (70001, 73.0)):
    z = rng.normal(80.0, 7.0, n).astype(np.float32)
    z[rng.integers(0, n, max(1, n // 3))] = np.float32(79.5)          # heavy duplicates around the quantile
    cases.append((z, q, np.float32(81.25)))
cases.append((np.concatenate([rng.normal(0, 1, 5000), [np.nan]]).astype(np.float32), 25, None))
cases.append((np.full(3000, 2.5, np.float32), 25, np.float32(1.0)))
for z, q, sub in cases:
    want = np.float32(np.percentile(z - (sub if sub is not None else np.float32(0)), q))
    cut = np.linspace(0, len(z), world + 1).astype(int)
    mine = z[cut[rank]:cut[rank + 1]]                                   # ragged, possibly empty parts
    if use_gpu:
        got = tiles.shared_percentile(torch.from_numpy(mine).to("cuda:0"), q, sub=sub)
    else:
        got = tiles.shared_percentile(torch.from_numpy(mine), q, sub=sub, select=NumpySelect())
    assert got.dtype == np.float32
    assert (np.isnan(got) and np.isnan(want)) or got.view(np.uint32) == want.view(np.uint32), (len(z), q, got, want)
if world > 1:
    dist.barrier()
    dist.destroy_process_group()
print("rank", rank, "ok")
'''


_MEAN_WORKER = r'''
import os, sys
sys.path.insert(0, sys.argv[1])
import numpy as np, torch, torch.distributed as dist
from pointcloudhookup_amd import tiles
rank, world, local = tiles.init_from_env(backend="gloo")
use_gpu = sys.argv[2] == "gpu"

class NumpyShard:                                      # CPU stand-in for tiles.HipMeanShard (tests only)
    device = torch.device("cpu")
    def __init__(self, rows):
        self.rows = np.ascontiguousarray(rows, np.float32).reshape(-1, 3)
    def walk(self, sum_in, total_n):
        a = self.rows if sum_in is None else np.vstack([np.asarray(sum_in, np.float32).reshape(1, 3), self.rows])
        with np.errstate(all="ignore"):                # numpy's own sequential float32 column loop
            s = np.add.reduce(a, axis=0) if len(a) else np.zeros(3, np.float32)
            if total_n:
                s = s / np.float32(total_n)
        return torch.from_numpy(np.asarray(s, np.float32))

rng = np.random.default_rng(5)
n = 400_003
cases = [(rng.random((n, 3)) * [1000, 100, 30] + [437000.0, 3139000.0, 80.0]).astype(np.float32),
         rng.normal(0, 100, (n, 3)).astype(np.float32),
         (rng.integers(0, 4000, (n, 3)) * 0.5).astype(np.float32),
         np.zeros((0, 3), np.float32) if world == 1 else rng.random((world - 1, 3)).astype(np.float32)]  # empty shards
for raw in cases:
    with np.errstate(all="ignore"):
        want = np.mean(raw, axis=0) if len(raw) else None
    cut = np.linspace(0, len(raw), world + 1).astype(int)
    cut[1:-1] += 3                                     # ragged, not on block edges
    cut = np.clip(cut, 0, len(raw))
    mine = raw[cut[rank]:cut[rank + 1]]
    if use_gpu:
        got = tiles.sharded_centroid(torch.from_numpy(mine).to("cuda:0"), len(raw))
    else:
        got = tiles.sharded_centroid(mine, len(raw), shard=NumpyShard(mine))
    assert got.dtype == np.float32 and got.shape == (3,)
    if want is None:
        assert np.isnan(got).all()                     # 0 / 0, like numpy
    else:
        assert np.array_equal(got.view(np.uint32), want.view(np.uint32)), (got, want)
if world > 1:
    dist.barrier()
    dist.destroy_process_group()
print("rank", rank, "ok")
'''


def _run_mean(tmp_path, world, mode, port):
    script = tmp_path / f"mean_{world}.py"
    script.write_text(_MEAN_WORKER)
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), WORLD_SIZE=str(world))
    procs = [subprocess.Popen([sys.executable, str(script), ROOT, mode], env=dict(env, RANK=str(r), LOCAL_RANK=str(r)),
                              stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True) for r in range(world)]
    outs = [p.communicate(timeout=300)[0] for p in procs]
    for r, (p, o) in enumerate(zip(procs, outs)):
        assert p.returncode == 0, o
        assert f"rank {r} ok" in o


@pytest.mark.parametrize("world", [1, 3])
def test_sharded_centroid_equals_numpy_on_the_concatenation(tmp_path, world):
    """utils/tower_extraction.py:63 on a cloud whose file-order shards live on different ranks: the three running
    float32 sums are handed down the line (send/recv of 12 bytes), the last rank divides and broadcasts - np.mean of
    the concatenation bit for bit, ragged and empty shards included.  The per-shard walk is numpy's own loop here and
    pch_mean_seq_partial_f32 in tests/test_gpu_e2e.py."""
    _run_mean(tmp_path, world, "cpu", 29791 + world)


def _run_pct(tmp_path, world, mode, port):
    script = tmp_path / f"pct_{world}.py"
    script.write_text(_PCT_WORKER)
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), WORLD_SIZE=str(world))
    procs = [subprocess.Popen([sys.executable, str(script), ROOT, mode], env=dict(env, RANK=str(r), LOCAL_RANK=str(r)),
                              stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True) for r in range(world)]
    outs = [p.communicate(timeout=300)[0] for p in procs]
    for r, (p, o) in enumerate(zip(procs, outs)):
        assert p.returncode == 0, o
        assert f"rank {r} ok" in o


@pytest.mark.parametrize("world", [1, 3])
def test_shared_percentile_equals_numpy_on_the_concatenation(tmp_path, world):
    """The percentile threshold of a tiled run: every rank histograms its own values, three all-reduced passes
    and one all-reduced minimum give np.percentile of the concatenation bit for bit (numpy >= 2 float32
    semantics), with duplicates around the quantile, NaN, ragged and empty parts."""
    _run_pct(tmp_path, world, "cpu", 29771 + world)


def test_native_box_search_equals_the_python_one():
    """pch_obb_min_boxes_f64 is host code: given qhull's hull it must find the box obb.oriented_bounds finds
    (extents and centre to rounding; the sign of the rectangle axes follows the edge orientation, which in
    the python form is qhull's and arbitrary)."""
    from scipy.spatial import ConvexHull
    from pointcloudhookup_amd import obb, ops
    rng = np.random.default_rng(8)
    verts, tris, vo, to, clouds = [], [], [0], [0], []
    for _ in range(24):
        n = int(rng.integers(20, 1500))
        q, _r = np.linalg.qr(rng.normal(size=(3, 3)))
        p = (rng.normal(size=(n, 3)) * rng.uniform(1, 12, 3)).astype(np.float32).astype(np.float64) @ q.T
        hull = ConvexHull(p, qhull_options="QbB Pp Qt")
        ids = np.sort(hull.vertices)
        remap = np.zeros(n, dtype=np.int64)
        remap[ids] = np.arange(len(ids))
        verts.append(p[ids]); tris.append(remap[hull.simplices])
        vo.append(vo[-1] + len(ids)); to.append(to[-1] + len(hull.simplices))
        clouds.append(p)
    for order in obb._EXTENT_ORDERS:
        for threads in (1, 4):
            T, E, S = ops.obb_min_boxes(np.concatenate(verts), vo, np.concatenate(tris), to,
                                        order == "trimesh_sorted", threads)
            assert (S == 0).all()
            for k, p in enumerate(clouds):
                t_ref, e_ref = obb.oriented_bounds(p, order)
                assert np.allclose(E[k], e_ref, atol=1e-10)
                assert np.allclose(np.linalg.inv(T[k])[:3, 3], np.linalg.inv(t_ref)[:3, 3], atol=1e-10)
                assert np.allclose(np.abs(T[k][:3, :3]), np.abs(t_ref[:3, :3]), atol=1e-10)
                assert abs(np.linalg.det(T[k][:3, :3])) == pytest.approx(1.0, abs=1e-12)
    # a flat "hull" has no volume: every candidate still gives a rectangle, the box has a zero extent
    with pytest.raises(RuntimeError):
        ops.obb_min_boxes(np.zeros((3, 3)), [0, 3], [[0, 1, 5]], [0, 1])


def test_exact_mode_with_native_search_is_the_python_loop_bit_for_bit():
    """boxes_of(search='native'): qhull sees the full cluster, pch_obb_search_f64 only names the winning
    direction, the winner is evaluated by the python arithmetic - so nothing may differ from the python loop,
    in either extent convention, with or without worker processes, including clouds whose best boxes tie
    (a cuboid: several facet normals give the same box; the python loop decides those)."""
    from pointcloudhookup_amd import obb
    rng = np.random.default_rng(12)
    clouds = []
    for i in range(18):
        n = int(rng.integers(300, 9000))
        if i % 3 == 0:
            p = rng.normal(size=(n, 3)) * [3, 3, 9]
        elif i % 3 == 1:
            p = rng.uniform(-1, 1, (n, 3)) * [4, 6, 20]
        else:
            p = np.round(rng.normal(size=(n, 3)) * [2, 3, 8], 2)          # 1 cm grid
        clouds.append(p.astype(np.float32))
    cuboid = np.array([[x, y, z] for x in (0.0, 4.0) for y in (0.0, 6.0) for z in (0.0, 20.0)])
    clouds.append(np.vstack([cuboid, rng.uniform(0.5, 3.5, (200, 3))]).astype(np.float32))
    clouds.append(np.zeros((10, 3), dtype=np.float32))                   # qhull refuses: same exception type
    for order in obb._EXTENT_ORDERS:
        want = list(obb.boxes_of(clouds, order, workers=1, search="python"))
        for workers in (1, 3):
            got = list(obb.boxes_of(clouds, order, workers=workers, search="native"))
            for (a, ea), (b, eb) in zip(want, got):
                assert type(ea) is type(eb)
                if ea is None:
                    assert np.array_equal(a[0], b[0]) and np.array_equal(a[1], b[1])
    with pytest.raises(ValueError):
        list(obb.boxes_of(clouds, "unsorted", search="fast"))


def test_native_search_edge_cases():
    from pointcloudhookup_amd import ops
    # no hulls
    best, vol = ops.obb_search(np.zeros((0, 3)), [0], np.zeros((0, 2)), [0])
    assert best.shape == (0,) and vol.shape == (0,)
    # a hull without candidates and a flat one (every projection has a rectangle, the volume is zero)
    sq = np.array([[0, 0, 0], [1, 0, 0], [1, 2, 0], [0, 2, 0]], dtype=np.float64)
    best, vol = ops.obb_search(np.vstack([sq, sq]), [0, 4, 8], np.array([[0.0, 0.0]]), [0, 0, 1])
    assert best.tolist() == [-1, 0] and vol[0] == 0.0
    with pytest.raises(ValueError):
        ops.obb_search(sq, [0, 3], np.zeros((1, 2)), [0, 1])            # offsets do not cover the arrays
    # three collinear vertices: no rectangle in that projection -> inf, never chosen over a real one
    line = np.array([[0, 0, 0], [1, 1, 0], [2, 2, 0]], dtype=np.float64)
    best, vol = ops.obb_search(line, [0, 3], np.array([[0.0, 0.0]]), [0, 1])
    assert best.tolist() == [-1] and np.isinf(vol[0])


# ------------------------------------------------------------------ stage D1: host-only library + worker pool
def test_host_only_obb_library_exports_its_header_and_links_no_hip_runtime():
    """libpch_obbhost.so (include/pch_obbhost.h) is what the box workers load: same search as pch_obb_search_f64,
    no HIP runtime behind it (a pool of dozens of workers must not open the GPU)."""
    import ctypes as C
    from pointcloudhookup_amd import obb, ops
    path = os.path.join(ROOT, "pointcloudhookup_amd", "libpch_obbhost.so")
    assert os.path.exists(path), "make -C pointcloudhookup_amd/csrc builds it"
    header = open(os.path.join(ROOT, "include", "pch_obbhost.h")).read()
    declared = set(re.findall(r"\b(pch_obbhost_[a-z0-9_]+)\s*\(", header))
    nm = subprocess.run(["nm", "-D", "--defined-only", path], capture_output=True, text=True).stdout
    assert declared and declared <= set(re.findall(r" T (pch_obbhost_[a-z0-9_]+)", nm))
    ldd = subprocess.run(["ldd", path], capture_output=True, text=True).stdout
    assert "amdhip" not in ldd and "hsa-runtime" not in ldd, ldd
    assert obb._hostlib() is not None
    rng = np.random.default_rng(3)
    pts = (rng.normal(size=(5000, 3)) * [3, 2, 9]).astype(np.float32)
    verts, angles = obb.hull_candidates(pts)
    best, vol = ops.obb_search(verts, [0, len(verts)], angles, [0, len(angles)])
    win = obb._winners(verts, angles)
    assert win is not None and int(best[0]) in win.tolist()
    assert np.array_equal(win, np.flatnonzero(vol <= vol[best[0]] * (1 + obb._TIE)))


def test_obb_pool_jobs_in_flight_shared_buffers_and_a_dead_worker():
    """The pool behind the exact mode: tasks name slices of a shared buffer (nothing large is pickled), several
    jobs may be in flight, buffers are recycled, and a worker that dies has its task computed in the parent."""
    from pointcloudhookup_amd import obb
    rng = np.random.default_rng(9)
    clouds = [(rng.normal(size=(int(rng.integers(500, 4000)), 3)) * [3, 2, 9 + i]).astype(np.float32)
              for i in range(12)]
    want = [obb._boxed((c, "native:unsorted")) for c in clouds]
    pl = obb.pool(3)
    assert pl.size() >= 3

    def job_of(arrs):
        offs = np.cumsum([0] + [len(a) for a in arrs])
        buf = pl.buffer(int(offs[-1]) * 12)
        buf.array[:int(offs[-1]) * 12].view(np.float32).reshape(-1, 3)[:] = np.concatenate(arrs)
        return buf, obb.boxes_job(buf, offs, np.float32, "unsorted")

    (b1, j1), (b2, j2) = job_of(clouds[:6]), job_of(clouds[6:])      # two jobs, two buffers, in flight together
    assert b1 is not b2
    got = obb.job_results(j1) + obb.job_results(j2)
    pl.release(b1)
    pl.release(b2)
    for (a, ea), (b, eb) in zip(want, got):
        assert ea is None and eb is None
        assert np.array_equal(a[0], b[0]) and np.array_equal(a[1], b[1])
    assert j1.worker_s > 0 and j2.worker_s > 0
    b3 = pl.buffer(1000)
    assert b3 in (b1, b2)                                             # recycled, not accumulated
    pl.release(b3)
    # kill one worker: everything is still answered, identically
    victim = next(p for p in pl.procs if p.alive)
    victim.proc.kill()
    victim.proc.wait()
    again = list(obb.boxes_of(clouds, "unsorted", workers=3))
    for (a, ea), (b, eb) in zip(want, again):
        assert eb is None and np.array_equal(a[0], b[0]) and np.array_equal(a[1], b[1])
    assert obb.usable_cpus() >= 1 and 1 <= obb.default_workers() <= max(obb.POOL_CAP, 1)


def test_obb_pool_keeps_inside_a_cpu_quota_budget(monkeypatch):
    """Under a cgroup CPU quota the pool holds more workers than the quota has cores (one tile's table fits one
    period's budget), and the dispatcher starts a task beyond the quota's core count only while the period's budget has
    room: the rule itself, on a pool without processes, and the quota parser."""
    import collections
    from pointcloudhookup_amd import obb

    class P:
        def __init__(self, t):
            self.alive, self.ready, self.task, self.t_task = True, True, (None, 0, None), t

    pl = obb.Pool()
    pl.quota = None
    assert pl._may_start([P(0.0)] * 100, 1.0)                         # no quota: no rule
    pl.quota = (4.0, 0.1)                                             # 4 cores / 100 ms: 0.4 CPU-s per period
    assert pl._may_start([P(0.99), P(0.99)], 1.0)                     # fewer busy than cores: always
    fresh = [P(0.999)] * 6                                            # six just started: 0.006 + 0.028 < 0.32
    assert pl._may_start(fresh, 1.0)
    old = [P(0.90)] * 6                                               # six that have run a whole period: 0.6 CPU-s
    assert not pl._may_start(old, 1.0)
    pl.spent = collections.deque([(0.95, 0.30)])                      # a finished burst inside the window
    assert not pl._may_start(fresh, 1.0)
    assert pl._may_start([P(1.199)] * 6, 1.2) and not pl.spent        # ... forgotten a period later
    # the parser: cgroup v2 text, "max" = no quota
    real_open = open

    def fake(text):
        def _open(path, *a, **k):
            if str(path) == "/sys/fs/cgroup/cpu.max":
                import io
                return io.StringIO(text)
            return real_open(path, *a, **k)
        return _open

    import builtins
    monkeypatch.delenv("PCH_OBB_WORKERS", raising=False)
    monkeypatch.setattr(builtins, "open", fake("1600000 100000\n"))
    assert obb.cpu_quota() == (16.0, 0.1)
    assert obb.usable_cpus() == min(16, obb._affinity_cpus())
    assert obb.default_workers() == min(obb.POOL_CAP, obb._affinity_cpus(), obb.BURST * obb.usable_cpus())
    monkeypatch.setattr(builtins, "open", fake("max 100000\n"))
    assert obb.cpu_quota() is None and obb.default_workers() == min(obb.POOL_CAP, obb._affinity_cpus())


def test_qhull_is_driven_without_the_wrapper_and_answers_the_same(monkeypatch):
    """obb._hull_simplices drives scipy's compiled qhull object directly (ConvexHull's own class, same options) and takes
    only the triangle list: triangles, in qhull's order, must be ConvexHull's on clouds of several shapes and sizes -
    degenerate input raises the same error - and PCH_OBB_BARE_QHULL=0 (or a failed self-check) goes back to ConvexHull."""
    from scipy.spatial import ConvexHull
    from pointcloudhookup_amd import obb
    rng = np.random.default_rng(77)
    monkeypatch.setattr(obb, "_BARE", None)
    monkeypatch.delenv("PCH_OBB_BARE_QHULL", raising=False)
    shapes = [rng.normal(size=(n, 3)) * sc for n, sc in ((4, [1, 1, 1]), (9, [3, 1, 9]), (500, [2.5, 2.5, 9]),
                                                         (20000, [3, 2, 11]))]
    shapes.append(np.round(rng.normal(size=(3000, 3)) * [2, 2, 6], 1))            # many coplanar / duplicate points
    shapes.append(rng.uniform(-1, 1, size=(2000, 3)) * [10, 10, 0.5] + [437000.0, 3139000.0, 80.0])
    for c in shapes:
        c = np.ascontiguousarray(c, dtype=np.float64)
        want = ConvexHull(c, qhull_options="QbB Pp Qt").simplices
        got = obb._hull_simplices(c)
        assert got.dtype == want.dtype and np.array_equal(got, want)
    assert obb._BARE is True                                                      # the short cut was really taken
    flat = np.ascontiguousarray(np.c_[rng.normal(size=(50, 2)), np.zeros(50)])    # coplanar: qhull refuses
    with pytest.raises(Exception) as e1:
        ConvexHull(flat, qhull_options="QbB Pp Qt")
    with pytest.raises(Exception) as e2:
        obb._hull_simplices(flat)
    assert type(e1.value) is type(e2.value)
    # switched off: the plain call, same answer
    monkeypatch.setattr(obb, "_BARE", None)
    monkeypatch.setenv("PCH_OBB_BARE_QHULL", "0")
    assert np.array_equal(obb._hull_simplices(shapes[2].astype(np.float64)), ConvexHull(shapes[2], qhull_options="QbB Pp Qt").simplices)
    assert obb._BARE is False
    monkeypatch.setattr(obb, "_BARE", None)


def test_qhull_is_shown_fewer_points_and_runs_the_same_run():
    """Rows strictly inside qhull's initial simplex are inert in its run; obb.qhull_input leaves them out (natively,
    pch_obbhost_reduce_*).  tools/prefilter_check.py compares hull vertices and candidate directions with the reduction
    off and on, bit for bit, on random clusters of seven shapes (48 400 clean trials when this was written); here 200 of
    them, the python statement of qhull's rule against the native one on what they decide, and the switches."""
    from pointcloudhookup_amd import obb
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "prefilter_check.py"), "7", "200"],
                       capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and " 0 mismatches" in r.stdout, r.stdout[-800:] + r.stderr[-800:]
    rng = np.random.default_rng(3)
    tower = (rng.normal(size=(20000, 3)) * [2.5, 2.5, 9]).astype(np.float32)
    shown = obb.qhull_input(tower)
    assert obb._PREFILTER is True and shown.dtype == np.float64 and 0.5 * len(tower) < len(shown) < len(tower)
    s = obb.predicted_simplex(tower.astype(np.float64))            # the python statement of the same rule
    T = tower[s].astype(np.float64)
    b = (tower.astype(np.float64) - T[0]) @ np.linalg.inv((T[1:] - T[0]).T).T
    inside = (b > 1e-6).all(axis=1) & (b.sum(axis=1) < 1 - 1e-6)
    assert abs(int((~inside).sum()) - len(shown)) <= 2             # (rows within rounding of the 1e-6 margin may differ)
    # stands down: few points, a zero extent, candidates that tie
    assert len(obb.qhull_input(tower[:50])) == 50
    flat = tower.copy(); flat[:, 1] = 3.0
    assert len(obb.qhull_input(flat)) == len(flat)
    cube = np.array([[x, y, z] for x in (0.0, 4.0) for y in (0.0, 6.0) for z in (0.0, 20.0)] * 10, dtype=np.float32)
    assert len(obb.qhull_input(cube)) == len(cube)
    keep = obb._PREFILTER
    try:
        obb._PREFILTER = False
        assert len(obb.qhull_input(tower)) == len(tower)
    finally:
        obb._PREFILTER = keep


def test_strip_representatives_agree_between_tiles_far_from_the_origin():
    """Both neighbours of a tile edge publish one (global row, local cluster) pair per lattice cell of the strip around
    the edge; because the lattice is anchored at the frame's origin - not at a tile's own box - the two name the same rows
    whatever else each tile holds, also 2.4e6 m from the origin (where the float32 'centroid' of a 100 M-point EPSG-scale
    cloud puts the centred frame), and both strips of a tile come out of one pass."""
    import torch
    from pointcloudhookup_amd import tiles
    rng = np.random.default_rng(8)
    far = np.array([260000.0, 2435000.0, 40.0])
    pts = (rng.uniform(0, 1, (6000, 3)) * [120.0, 60.0, 30.0] + far).astype(np.float32)
    rows = np.arange(len(pts)) * 3 + 7                                  # global rows, ascending
    core = rng.random(len(pts)) < 0.8
    x = pts[:, 0]
    e1, e2 = np.float32(far[0] + 40.0), np.float32(far[0] + 80.0)
    left = np.flatnonzero(x < e1 + 17)                                   # tile 0: up to e1 (+ halo)
    mid = np.flatnonzero((x >= e1 - 17) & (x < e2 + 17))                 # tile 1: [e1, e2) + halo

    def reps(sel, strips):
        lab = torch.from_numpy((rows[sel] % 5).astype(np.int64))
        return tiles.strip_representatives(torch.from_numpy(pts[sel]), torch.from_numpy(rows[sel]), lab,
                                           torch.from_numpy(core[sel]), strips, 8.0)

    up0, = reps(left, [(float(e1) - 8.0, float(e1) + 8.0)])
    lo1, up1 = reps(mid, [(float(e1) - 8.0, float(e1) + 8.0), (float(e2) - 8.0, float(e2) + 8.0)])
    assert up0.shape[0] > 20 and torch.equal(up0[:, 0], lo1[:, 0])      # the same rows on either side of the edge
    assert torch.equal(up0[:, 1], lo1[:, 1])                            # (labels here are a function of the row)
    strip = core & (x >= np.float32(float(e1) - 8.0)) & (x < np.float32(float(e1) + 8.0))
    side = 8.0 / 3 ** 0.5 * (1.0 - 2.0 ** -16)
    cells = np.floor(pts[strip].astype(np.float64) * (1.0 / side)).astype(np.int64)
    want = {}
    for r, c in zip(rows[strip], map(tuple, cells)):
        want[c] = min(want.get(c, r), r)
    assert sorted(want.values()) == up0[:, 0].tolist()                  # the smallest row of every cell, nothing else
    assert up1.shape[0] > 20 and set(up1[:, 0].tolist()).isdisjoint(lo1[:, 0].tolist())
    assert tiles.strip_representatives(torch.from_numpy(pts), torch.from_numpy(rows), torch.zeros(len(pts), dtype=torch.int64),
                                       torch.zeros(len(pts), dtype=torch.bool), [(0.0, 1e9)], 8.0)[0].shape == (0, 2)
    with pytest.raises(ValueError):
        reps(left, [(0.0, 1.0), (2.0, 3.0), (4.0, 5.0)])
