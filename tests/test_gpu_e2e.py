"""End-to-end parity through the drop-in call surface (LAS file in, reference-shaped results
out) and size-independent properties at BASELINE.json's full sizes."""
import os

import numpy as np
import pytest
import torch

from oracle import dbscan as odb
from oracle import ground_filter as ogf
from oracle import towers as otw
from oracle import voxel as ovx
from pointcloudhookup_amd import las, ops, pipeline, synth

pytestmark = pytest.mark.gpu

SCALES = np.array([0.001, 0.001, 0.001])
OFFSETS = np.array([437000.0, 3139000.0, 0.0])


@pytest.fixture(scope="module")
def config1_las(tmp_path_factory):
    """BASELINE config 1: 1 M-pt flat ground + 3 Gaussian towers, written as a LAS 1.2 file."""
    d = tmp_path_factory.mktemp("cfg1")
    pts = synth.corridor_numpy(1_000_000, seed=synth.SEED0, kind="corridor", offset=True, towers=3)
    XYZ = np.round((pts - OFFSETS) / SCALES).astype(np.int32)
    path = str(d / "cloud.las")
    las.write(path, las.LasHeader(point_format=3, version=(1, 2), scales=SCALES, offsets=OFFSETS), XYZ)
    xyz = np.stack([ovx.las_scaled(XYZ[:, a], SCALES[a], OFFSETS[a]) for a in range(3)], axis=1)
    return path, XYZ, xyz, d


def test_run_voxel_downsampling_dropin(cuda, config1_las):
    from pointcloudhookup_amd.ui.import_PC import process_chunk, run_voxel_downsampling
    path, XYZ, xyz, d = config1_las
    out = str(d / "output" / "point_2.las")
    logs, prog = [], []
    assert run_voxel_downsampling(path, out, voxel_size=0.1, chunk_size=500000,
                                  progress_callback=prog.append, log_callback=logs.append) is None
    ridx, rmean, rcnt, roffs = ovx.voxel_down_sample_chunked(xyz, 0.1, 500000)
    got = las.read(out)
    ref_XYZ = np.stack([ovx.las_unscale(rmean[:, a], SCALES[a], OFFSETS[a]) for a in range(3)], axis=1)
    assert len(got.XYZ) == len(ref_XYZ)
    for c in range(len(roffs) - 1):                                 # per chunk the same multiset of records (the order
        a, b = int(roffs[c]), int(roffs[c + 1])                     # inside a chunk is unspecified, as with Open3D)
        np.testing.assert_array_equal(got.XYZ[a:b][np.lexsort(got.XYZ[a:b].T[::-1])],
                                      ref_XYZ[a:b][np.lexsort(ref_XYZ[a:b].T[::-1])])
    assert got.header.point_format == 3 and tuple(got.header.version) == (1, 2)
    np.testing.assert_array_equal(got.header.scales, SCALES)
    assert prog == [50, 100]
    assert logs[0] == "📂 原始点数: 1000000" and logs[-2] == f"✅ 下采样完成，输出点数: {len(rcnt)}"
    assert logs[2] == "✅ 已完成第1块：500000 点"
    m = process_chunk(xyz[:20000], 0.5)
    want = ovx.voxel_down_sample(xyz[:20000], 0.5)[1]
    np.testing.assert_array_equal(m[np.lexsort(m.T[::-1])], want[np.lexsort(want.T[::-1])])


def test_sampling_cli_twin(cuda, config1_las, capsys):
    from pointcloudhookup_amd.ui.Sampling import voxel_downsample_open3d
    path, XYZ, xyz, d = config1_las
    out = str(d / "out2" / "p.las")
    voxel_downsample_open3d(path, out, 0.2, 250000)
    text = capsys.readouterr().out
    assert "成功生成下采样文件" in text and os.path.exists(out)
    ref = ovx.voxel_down_sample_chunked(xyz, 0.2, 250000)
    assert len(las.read(out)) == len(ref[2])
    voxel_downsample_open3d(str(d / "missing.las"), out, 0.2)       # swallowed like the reference
    assert "处理过程中发生错误" in capsys.readouterr().out


@pytest.mark.parametrize("order", ["unsorted", "trimesh_sorted"])
def test_extract_towers_dropin_matches_oracle(cuda, oracle_clib, config1_las, monkeypatch, order):
    from pointcloudhookup_amd.utils import tower_extraction as te
    path, XYZ, xyz, d = config1_las
    work = d / f"run_{order}"
    work.mkdir()
    monkeypatch.chdir(work)
    monkeypatch.setattr(te, "OBB_EXTENT_ORDER", order)
    logs, prog = [], []
    towers = te.extract_towers(path, progress_callback=prog.append, log_callback=logs.append)
    ref = otw.extract_towers_arrays(xyz[:, 0], xyz[:, 1], xyz[:, 2], fit="c", extent_order=order)
    assert len(towers) == len(ref["towers"])
    if order == "trimesh_sorted":
        assert len(towers) == 3                                     # BASELINE config 1: 3 towers
    for t, r in zip(towers, ref["towers"]):
        assert set(t) == {"center", "rotation", "extent", "height", "width", "north_angle", "points"}
        np.testing.assert_allclose(t["center"], r["center"], rtol=0, atol=1e-3)   # north_star tolerance
        np.testing.assert_array_equal(t["center"], r["center"])    # in fact bit identical
        np.testing.assert_array_equal(t["extent"], r["extent"])
        np.testing.assert_array_equal(t["rotation"], r["rotation"])
        assert t["north_angle"] == r["north_angle"]
        np.testing.assert_array_equal(t["points"].view(np.uint32), r["points"].view(np.uint32))
        f = work / "output_towers" / f"tower_{r['label']}.las"
        assert f.exists() and len(las.read(str(f))) == len(r["points"])
    assert prog[:3] == [5, 10, 20] and prog[-1] == 100 and 75 in prog and prog == sorted(prog)
    assert f"✅ 点云读取完成，总点数: {len(xyz)}" in logs
    assert f"✅ 高度过滤完成，保留点数: {len(ref['ground']['filtered'])}" in logs
    assert f"\n=== 开始杆塔检测（候选簇：{ref['n_candidates']}个） ===" in logs
    assert logs[-1] == "✅ 杆塔提取完成"
    if towers:                                                      # openpyxl is absent: logged, not raised
        assert any(m.startswith("⚠️ 保存Excel失败") or "杆塔信息已保存" in m for m in logs)


def test_extract_and_visualize_dropin(cuda, config1_las):
    from pointcloudhookup_amd.ui.extract import extract_and_visualize_towers
    path, XYZ, xyz, d = config1_las
    tower = dict(center=np.array([437050.0, 3139050.0, 100.0]), rotation=np.eye(3),
                 extent=np.array([20.1, 18.0, 17.4]))
    cloud, geoms = extract_and_visualize_towers(path, [tower])
    assert cloud.dtype == np.float64
    np.testing.assert_array_equal(cloud, xyz)                       # laspy scaled view, bit exact
    assert len(geoms) == 1 and geoms[0][0].shape == (24, 3) and geoms[0][1] == (1.0, 0.0, 0.0)
    lo, hi = geoms[0][0].min(0), geoms[0][0].max(0)
    np.testing.assert_allclose(lo, [437050 - 20.1, 3139050 - 10.05, 100 - 17.4])
    np.testing.assert_allclose(hi, [437050 + 20.1 * 1.67, 3139050 + 20.1, 100 + 34.8])
    cloud2, geoms2 = extract_and_visualize_towers(path, [tower], use_kuangxuan_method=False)
    assert geoms2[0][0].shape == (24, 3)


# ------------------------------------------------------------------ BASELINE config 2 (10 M)
def test_config2_voxel_then_cluster_properties(cuda, oracle_clib):
    n = 10_000_000
    xyz = synth.corridor_torch(n, seed=synth.SEED0 + 1, kind="corridor", offset=True, device=cuda)
    idx, mean, cnt, offs = ops.voxel_downsample(xyz, 0.2, 500000)
    assert int(cnt.sum()) == n and int(offs[-1]) == idx.shape[0] and offs.shape[0] == 21
    host = xyz[:500000].cpu().numpy()
    ridx, rmean, rcnt = ovx.voxel_down_sample(host, 0.2)            # first chunk against the oracle
    m0 = int(offs[1])
    gi, gm, gc = ovx.canonical(idx[:m0].cpu().numpy(), mean[:m0].cpu().numpy(), cnt[:m0].cpu().numpy(), [0, m0])
    np.testing.assert_array_equal(gi, ridx)                         # the same set of voxels (order inside a chunk
    np.testing.assert_array_equal(gm, rmean)                        # is the library's own, as it is Open3D's)
    np.testing.assert_array_equal(gc, rcnt)
    # idempotence on one chunk: each mean lies in its own voxel
    i2, m2, c2, _ = ops.voxel_downsample(mean[:m0].contiguous(), 0.2, 0)
    assert i2.shape[0] <= m0
    # filter + cluster the voxel output (the GUI's order of operations)
    raw = ops.cast_f32(mean)
    cl = pipeline.cluster_points(raw, 8.0, 80, 50000, want_index=True)
    ref = ogf.ground_filter(raw.cpu().numpy())
    np.testing.assert_array_equal(cl["ground"]["centroid"].view(np.uint32), ref["centroid"].view(np.uint32))
    np.testing.assert_array_equal(cl["ground"]["points"].cpu().numpy().view(np.uint32),
                                  ref["filtered"].view(np.uint32))
    labels = cl["labels"].cpu().numpy()
    k = cl["nclusters"]
    present = np.unique(labels[labels >= 0])
    np.testing.assert_array_equal(present, np.arange(k))            # ids dense, 0..K-1
    first = np.array([np.flatnonzero(labels == c)[0] // 50000 for c in range(k)])
    assert (np.diff(first) >= 0).all()                              # numbered chunk by chunk
    for ci in (0, len(labels) // 50000 // 2):                       # two chunks against the oracle
        chunk = ref["filtered"][ci * 50000:(ci + 1) * 50000]
        want, _ = odb.dbscan_fit_c(chunk, 8.0, 80)
        g = labels[ci * 50000:(ci + 1) * 50000].astype(np.int64)
        base = g[g >= 0].min() if (g >= 0).any() else 0
        np.testing.assert_array_equal(np.where(g >= 0, g - base, -1), want)
    again = pipeline.cluster_points(raw, 8.0, 80, 50000)
    assert torch.equal(again["labels"], cl["labels"])               # deterministic
    offsets = cl["offsets"].cpu().numpy()
    assert offsets[-1] == (labels >= 0).sum() and (np.diff(offsets) > 0).all()


# ------------------------------------------------------------------ BASELINE config 3 (100 M)
def test_config3_100m_properties(cuda):
    n = 100_000_000
    raw = synth.corridor_torch(n, seed=synth.SEED0 + 2, kind="corridor", offset=True, device=cuda,
                               dtype=torch.float32)
    cl = pipeline.cluster_points(raw, 8.0, 80, 50000, want_index=True)
    gf = cl["ground"]
    c = torch.tensor(gf["centroid"], device=cuda)
    z = raw[:, 2] - c[2]
    keep = z > float(gf["threshold"])
    assert int(keep.sum()) == gf["count"]
    idx = gf["index"].long()
    assert bool((idx[1:] > idx[:-1]).all())                         # order preserving compaction
    assert torch.equal(idx, torch.nonzero(keep).flatten())
    assert torch.equal(gf["points"], raw[idx] - c)                  # float32 centring, same op
    # the percentile base sits between the order statistics floor(k) and floor(k)+1, k = (n-1)/4
    k0 = int(np.floor(np.float32(n - 1) * np.float32(0.25)))
    base = float(gf["base"])
    assert int((z < base).sum()) <= k0 + 1 <= int((z <= base).sum()) + 1
    assert int((z <= base).sum()) >= k0 + 1
    labels = cl["labels"]
    k = cl["nclusters"]
    assert int(labels.max()) == k - 1 and int(labels.min()) >= -1
    counts = torch.bincount(labels[labels >= 0].long(), minlength=k)
    offsets = cl["offsets"]
    assert torch.equal(counts, offsets[1:] - offsets[:-1])          # checksum of the grouping
    perm = cl["perm"].long()
    assert torch.equal(labels[perm[: int(offsets[-1])]].long(),
                       torch.repeat_interleave(torch.arange(k, device=cuda), counts))
    assert int(counts.min()) >= 1
    # every cluster lives inside one 50k chunk (reference semantics) and is spatially compact
    lo = torch.div(perm[offsets[:-1]], 50000, rounding_mode="floor")
    hi = torch.div(perm[offsets[1:] - 1], 50000, rounding_mode="floor")
    assert torch.equal(lo, hi)
    stats = cl["stats"]
    assert float((stats[:, 3:6] - stats[:, 0:3]).max()) < 200.0


def test_tower_clusters_equals_the_three_stage_calls(cuda):
    """pch_tower_clusters_f32 (one library call) against ground_filter + dbscan +
    segment_by_label called one by one: every output identical, also when the kept-points
    hint is too small (second attempt) and when there are more clusters than k_cap."""
    raw = synth.corridor_torch(2_000_000, seed=synth.SEED0 + 7, kind="corridor", offset=True, device=cuda,
                               towers=6, dtype=torch.float32)
    gf = ops.ground_filter(raw, want_index=True)
    labels, _, k = ops.dbscan(gf["points"], 8.0, 80, 50000, aabb=gf["aabb"])
    perm, offs, stats = ops.segment_by_label(labels, gf["points"], k)
    assert k >= 2

    def check(res):
        g2, l2, k2, p2, o2, s2 = res
        assert k2 == k and g2["count"] == gf["count"]
        for name in ("centroid", "aabb"):
            np.testing.assert_array_equal(np.asarray(g2[name]), np.asarray(gf[name]))
        assert g2["base"] == gf["base"] and g2["threshold"] == gf["threshold"]
        assert g2["used_fallback"] == gf["used_fallback"] and g2["count_at_offset"] == gf["count_at_offset"]
        assert torch.equal(g2["points"], gf["points"]) and torch.equal(g2["index"], gf["index"])
        assert torch.equal(l2, labels) and torch.equal(p2, perm) and torch.equal(o2, offs)
        assert torch.equal(s2, stats)

    key = torch.device(cuda).index or 0
    ops._nf_hint.pop(key, None)
    check(ops.tower_clusters(raw, want_index=True))                 # first call: sized for n
    check(ops.tower_clusters(raw, want_index=True))                 # sized from the hint
    ops._nf_hint[key] = 1e-4                                        # hint far too small: retried with n
    check(ops.tower_clusters(raw, want_index=True))
    check(ops.tower_clusters(raw, want_index=True, k_cap=1))        # more clusters than k_cap
    g, l, kk, p, o, s = ops.tower_clusters(raw, segment=False)
    assert p is None and kk == k and torch.equal(l, labels)


# ------------------------------------------------------------------ BASELINE config 4: x-tiles + halo
@pytest.mark.parametrize("world", [1, 2, 3])
def test_cluster_tiled_hip_equals_global_dbscan(cuda, oracle_clib, tmp_path, world):
    """The tiled clustering of tests/test_host.py with the HIP kernels doing the local fit and the
    relabel pass (pch_dbscan_f32 + pch_dbscan_relabel_i32), `world` processes sharing this GPU, the
    exchange over gloo.  Towers and a thin bridge sit on the tile edges; every rank's labels must equal
    one DBSCAN over the whole cloud."""
    from test_host import _run_tiled
    _run_tiled(tmp_path, world, "gpu", 29751 + world)


@pytest.mark.parametrize("world", [1, 3])
def test_sharded_centroid_hip(cuda, tmp_path, world):
    """tests/test_host.py's sharded-centroid cases with pch_mean_seq_partial_f32 doing every rank's tables and walk
    (tiles.HipMeanShard), `world` processes sharing this GPU, the 12-byte hops over gloo."""
    from test_host import _run_mean
    _run_mean(tmp_path, world, "gpu", 29811 + world)


@pytest.mark.parametrize("world", [1, 2, 3])
def test_bench_tiled_mode_equals_the_single_gpu_run(cuda, world):
    """BASELINE config 4 end to end through bench.py's own launcher: `python bench.py --gpus W --mode tiled --verify`
    starts W ranks (sharing this one GPU, exchange over gloo - RCCL refuses two ranks on one device), every rank
    generates ONLY its x-tile + halo of a 12 M-point strip corridor at EPSG scale, and tiles.tiled_step (chained
    float32 centroid, percentile across the ranks, per-tile filter, global DBSCAN per tile, one fixed-capacity
    all_gather, relabel) must reproduce the single-GPU run of the whole cloud: centroid and threshold bit for bit,
    every owned label, every kept row owned exactly once.  Towers sit on the strip edges, hence on the tile edges."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, PCH_BENCH_SINGLE_DEVICE="1", PCH_DIST_BACKEND="gloo")
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", str(world), "--mode", "tiled",
                        "--points", "12000000", "--steps", "2", "--warmup", "1", "--verify"], env=env,
                       capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-1500:] + r.stderr[-3000:]
    out = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("{")][-1])
    t = out["tiled"]
    assert t["verified_against_single_gpu_run"] is True
    assert t["ranks_seen"] == world and out["n_gpus"] == world and t["points_total"] == 12_000_000
    assert t["clusters"] >= 3 and t["kept_points"] > 100_000
    assert set(t["phase_ms_max_over_ranks"]) == {"centroid_chain+threshold+filter", "local_fit", "reconciliation"}
    # rank 0 of W: no recv; one send unless it is also the last; three all_reduces, ONE all_gather: at most six
    calls = t["collectives_per_step_rank0"]
    if world > 1:
        assert calls == {"send": 1, "all_reduce": 3, "all_gather": 1}, calls


def test_tiled_step_over_rccl_with_forced_collectives(cuda):
    """PCH_TILES_FORCE_COLLECTIVES=1: a world of one rank brings a one-rank NCCL (= RCCL) group up on this GPU and
    every function of tiles.py takes its multi-rank branch - the all_reduce that carries the first histogram, the
    counts and the centroid bits, the two further histogram all_reduces and the all_gather of the int64 blocks
    (cluster table, both strips, survivor counts) really run on device tensors over RCCL; the point-to-point hops of
    the centroid chain are the only calls a single rank cannot make.  The result must still be the single-GPU run's
    (centroid and threshold bits, every label), and a step issues at most six collectives."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, PCH_TILES_FORCE_COLLECTIVES="1")
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT", "PCH_DIST_BACKEND",
              "PCH_BENCH_SINGLE_DEVICE"):
        env.pop(k, None)
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "1", "--mode", "tiled",
                        "--points", "12000000", "--steps", "3", "--warmup", "1", "--verify"], env=env,
                       capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-1500:] + r.stderr[-3000:]
    out = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("{")][-1])
    t = out["tiled"]
    assert t["verified_against_single_gpu_run"] is True
    assert t["backend"] == "nccl" and t["forced_collectives"] is True
    calls = t["collectives_per_step_rank0"]
    assert calls.get("all_reduce") == 3 and calls.get("all_gather") == 1, calls
    assert t["collectives_per_step_rank0_total"] <= 6
    print("[forced collectives over RCCL]", json.dumps({k: t[k] for k in (
        "backend", "collectives_per_step_rank0", "ms_per_step", "phase_ms_max_over_ranks", "clusters", "kept_points")}))


def test_rccl_probe_failure_degrades_to_gloo(cuda):
    """tiles.init_from_env with automatic backend choice: the default group is gloo and an RCCL group is PROBED.  Two
    ranks on this one GPU make RCCL refuse (duplicate device) - the run must go on with gloo exchanges and still
    reproduce the single-GPU result; the line says which backend carried the exchange."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, PCH_BENCH_SINGLE_DEVICE="1", PCH_BENCH_PROBE_RCCL="1")
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT", "PCH_DIST_BACKEND"):
        env.pop(k, None)
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--mode", "tiled",
                        "--points", "12000000", "--steps", "2", "--warmup", "1", "--verify"], env=env,
                       capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-1500:] + r.stderr[-3000:]
    out = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("{")][-1])
    assert out["tiled"]["verified_against_single_gpu_run"] is True
    assert out["tiled"]["backend"] == "gloo" and out["tiled"]["ranks_seen"] == 2
    assert "RCCL probe failed" in r.stdout + r.stderr


def test_dbscan_strip_pairs_cover_every_strip_core_point(cuda):
    """pch_dbscan_strip_pairs_i32: one (row, cluster) pair per grid cell with a core point in the strip.  Every core
    point of the strip must be within eps of the representative of ITS cluster's pairs (same cell => within eps), the
    representatives are core points of the strip carrying their own cluster id, and a short buffer reports the
    full count."""
    X = synth.corridor_torch(2_000_000, seed=synth.SEED0 + 41, kind="corridor", offset=False, device=cuda,
                             dtype=torch.float32)
    X = X[X[:, 2] > 3.0].contiguous()
    fit = ops.DbscanFit(X, 8.0, 80, 0)
    assert fit.nclusters >= 3
    cx = float(X[fit.labels == 0][:, 0].mean())                       # a strip through the middle of a tower
    lo, hi = np.float32(cx - 8.0), np.float32(cx + 8.0)
    pairs, cnt = fit.strip_pairs(lo, hi, 4096)
    m = int(cnt.item())
    assert 0 < m <= 4096
    pr = pairs[:m].long()
    core = fit.core.bool()
    instrip = core & (X[:, 0] >= float(lo)) & (X[:, 0] < float(hi))
    assert bool(instrip[pr[:, 0]].all())                              # representatives are strip core points
    assert torch.equal(fit.labels[pr[:, 0]].long(), pr[:, 1])         # ... with their own cluster id
    assert len(torch.unique(pr[:, 0])) == m
    P = X[instrip].double()
    R = X[pr[:, 0]].double()
    d2 = ((P[:, None, :] - R[None, :, :]) ** 2).sum(-1)
    same = fit.labels[instrip].long()[:, None] == pr[None, :, 1]
    near = ((d2 <= 64.0) & same).any(dim=1)
    assert bool(near.all())
    assert m < int(instrip.sum()) // 20                               # far fewer pairs than strip points
    short, cnt2 = fit.strip_pairs(lo, hi, 3)
    assert int(cnt2.item()) == m and short.shape == (3, 2)


def test_dbscan_relabel_redecides_border_points(cuda, oracle_clib):
    """Swapping the ids of two clusters must move a border point that touches both to the other one."""
    a = np.column_stack([np.linspace(0, 1, 30), np.zeros(30), np.zeros(30)])
    b = np.column_stack([np.linspace(3.2, 4.2, 30), np.zeros(30), np.zeros(30)])
    X = np.vstack([b, [[2.1, 0, 0]], a]).astype(np.float32)            # the tie fixture: point 30 touches both
    dev = torch.from_numpy(X).to(cuda)
    fit = ops.DbscanFit(dev, 1.15, 8, 0)
    assert fit.nclusters == 2 and int(fit.labels[30]) == 0 and not bool(fit.core[30])
    # any number of other ops may run between the fit and its continuation: the fit owns its workspace
    ops.percentile_f32(dev[:, 0].contiguous(), 25.0)
    ops.segment_by_label(fit.labels.clone(), dev, 2)
    assert fit.first_core_rows().tolist() == [0, 31]
    swapped = fit.relabel(torch.tensor([1, 0], dtype=torch.int32, device=cuda)).cpu().numpy()
    assert (swapped[:30] == 1).all() and (swapped[31:] == 0).all()
    assert swapped[30] == 0                                            # smallest NEW id among its core neighbours
    dropped = fit.relabel(torch.tensor([5, -1], dtype=torch.int32, device=cuda)).cpu().numpy()
    assert (dropped[:30] == -1).all() and (dropped[31:] == 5).all() and dropped[30] == 5


def test_dbscan_continuation_on_an_overwritten_workspace_is_refused(cuda):
    """pch_dbscan_relabel_i32 / pch_dbscan_first_core_rows_i32 continue the grid the last pch_dbscan_f32 left in
    its workspace.  A pch_* call that carves the same buffer in between must make them fail cleanly
    (PCH_ERR_ARG) - never follow overwritten cell tables on the device."""
    from pointcloudhookup_amd import _lib
    L = _lib.lib()
    X = synth.corridor_torch(20000, seed=synth.SEED0 + 31, kind="corridor", offset=False, device=cuda,
                             dtype=torch.float32)
    n = X.shape[0]
    ws = torch.empty(int(L.pch_dbscan_ws_bytes(n)) + 256, dtype=torch.uint8, device=cuda)
    labels = torch.empty(n, dtype=torch.int32, device=cuda)
    ncl = torch.zeros(1, dtype=torch.int32, device=cuda)
    st = torch.cuda.current_stream().cuda_stream

    def fit():
        _lib.check(L.pch_dbscan_f32(X.data_ptr(), n, 8.0, 20, 0, None, labels.data_ptr(), 0, ncl.data_ptr(),
                                    ws.data_ptr(), ws.numel(), st))
        return int(ncl.item())

    k = fit()
    assert k >= 1
    rows = torch.empty(k, dtype=torch.int32, device=cuda)
    assert L.pch_dbscan_first_core_rows_i32(n, rows.data_ptr(), ws.data_ptr(), ws.numel(), st) == 0
    # another op on the SAME buffer (here: the percentile select) overwrites the grid
    out = torch.empty(1, dtype=torch.float32, device=cuda)
    z = X[:, 2].contiguous()
    _lib.check(L.pch_percentile_f32(z.data_ptr(), n, 1, 0, 25.0, out.data_ptr(), ws.data_ptr(), ws.numel(), st))
    cmap = torch.zeros(k, dtype=torch.int32, device=cuda)
    before = labels.clone()
    assert L.pch_dbscan_first_core_rows_i32(n, rows.data_ptr(), ws.data_ptr(), ws.numel(), st) == -1
    assert b"untouched workspace" in L.pch_last_error()
    assert L.pch_dbscan_relabel_i32(cmap.data_ptr(), k, n, labels.data_ptr(), ws.data_ptr(), ws.numel(), st) == -1
    torch.cuda.synchronize()
    assert torch.equal(labels, before)                                 # nothing was launched
    # a sub-range of the buffer counts as well, a disjoint buffer does not
    k = fit()
    other = torch.empty(1 << 20, dtype=torch.uint8, device=cuda)
    _lib.check(L.pch_percentile_f32(z.data_ptr(), n, 1, 0, 25.0, out.data_ptr(), other.data_ptr(), other.numel(), st))
    assert L.pch_dbscan_first_core_rows_i32(n, rows.data_ptr(), ws.data_ptr(), ws.numel(), st) == 0
    _lib.check(L.pch_percentile_f32(z.data_ptr(), n, 1, 0, 25.0, out.data_ptr(), ws.data_ptr() + 4096,
                                    ws.numel() - 4096, st))
    assert L.pch_dbscan_first_core_rows_i32(n, rows.data_ptr(), ws.data_ptr(), ws.numel(), st) == -1


def test_lookback_wait_is_bounded(cuda):
    """The single-pass compactions chain their workgroups by a look-back poll (pch_lookback.h).  That wait has a
    wall-clock budget: with a tile that never publishes (the self-test kernel withholds ticket 1) the tiles behind it
    must give up, poison their status words and let the grid drain - the call comes back with PCH_ERR_TIMEOUT
    instead of leaving a spinning grid on the GPU.  The data-path kernels use the same function with a 4 s budget
    and report through a negative count (ops raise PchError)."""
    import time
    from pointcloudhookup_amd import _lib
    L = _lib.lib()
    scratch = torch.empty(256, dtype=torch.uint8, device=cuda)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    rc = L.pch_selftest_lookback_timeout(50, scratch.data_ptr(), scratch.numel(), torch.cuda.current_stream().cuda_stream)
    dt = time.perf_counter() - t0
    assert rc == _lib.PCH_ERR_TIMEOUT, L.pch_last_error()
    assert 0.04 < dt < 1.5, dt                                        # one waiter runs out its 50 ms, five more stop on
                                                                      # its poisoned word: not 6 x 50 ms, not 4 s
    # the self-test also checked the published count word: six failed tiles mark it with the (idempotent) sign bit,
    # so it reads negative however many tiles fail - a sum of -2^62 per tile wrapped to 0 at four
    with pytest.raises(_lib.PchError, match="PCH_ERR_TIMEOUT"):
        _lib.check_count(-(1 << 63) + 5, "x")
    assert _lib.check_count(7, "x") == 7
    # and the data path is unaffected: an ordinary filter still gives its count
    raw = synth.corridor_torch(300_000, seed=synth.SEED0 + 12, kind="corridor", offset=True, device=cuda,
                               dtype=torch.float32)
    assert ops.ground_filter(raw)["count"] > 0


def test_build_then_smoke_in_one_process(cuda):
    """__graft_entry__.build() loads libpch_hip.so before anything imports torch; PyTorch-ROCm brings its own
    HIP runtime, and with /opt/rocm's copy loaded first the library did not know torch's allocations
    (hipPointerGetAttributes failed in the device guard).  _lib.lib() therefore imports torch first."""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, "-c", "import __graft_entry__ as g; g.build(); g.smoke()"], cwd=root,
                       capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and "smoke ok" in r.stdout, r.stdout[-2000:] + r.stderr[-2000:]


@pytest.mark.parametrize("world", [1, 2])
def test_shared_percentile_hip(cuda, tmp_path, world):
    """tests/test_host.py's shared-percentile cases with the HIP select passes (pch_select_hist_f32,
    pch_select_min_above_f32) on every rank."""
    from test_host import _run_pct
    _run_pct(tmp_path, world, "gpu", 29781 + world)


def test_filter_gt_equals_the_fused_filter(cuda):
    """pch_filter_gt_f32 with the centroid and threshold the fused filter found gives the fused filter's output."""
    raw = synth.corridor_torch(3_000_001, seed=synth.SEED0 + 11, kind="corridor", offset=True, device=cuda,
                               dtype=torch.float32)
    gf = ops.ground_filter(raw, want_index=True)
    got = ops.filter_gt(raw, gf["centroid"], gf["threshold"], want_index=True)
    assert got["count"] == gf["count"]
    assert torch.equal(got["points"], gf["points"]) and torch.equal(got["index"], gf["index"])
    np.testing.assert_array_equal(got["aabb"], gf["aabb"])
    thr = tiles_shared_threshold(raw, gf)
    assert thr.view(np.uint32) == np.float32(gf["threshold"]).view(np.uint32)


def tiles_shared_threshold(raw, gf):
    from pointcloudhookup_amd import tiles
    base = tiles.shared_percentile(raw[:, 2], 25.0, sub=gf["centroid"][2])
    return np.float32(base + np.float32(3.0))


def _towers_equal(a, b):
    assert len(a) == len(b)
    for x, y in zip(a, b):
        assert set(x) == set(y)
        for k in x:
            assert np.array_equal(np.asarray(x[k]), np.asarray(y[k])), k


@pytest.mark.parametrize("async_write", [False, True])
def test_voxel_then_towers_hand_off_without_rereading_the_file(cuda, config1_las, tmp_path, monkeypatch, async_write):
    """run_voxel_downsampling leaves the records of its output file registered on the device; extract_towers on
    that path takes them instead of reading the file back (pointcloudhookup_amd/resident.py) - with the writer in
    the foreground (default) or in the background.  Either way the file on disk is byte for byte what the plain path
    writes, and the towers equal those of a run that reads the file: also after the file was touched (stamp
    mismatch -> file), and with the hand-off switched off."""
    import hashlib
    import time
    from pointcloudhookup_amd import las as _las, resident, stages
    from pointcloudhookup_amd.ui import import_PC
    from pointcloudhookup_amd.utils import tower_extraction as te
    path, XYZ, xyz, d = config1_las
    monkeypatch.chdir(tmp_path)
    monkeypatch.setattr(import_PC, "ASYNC_WRITE", async_write)
    sha = lambda p: hashlib.sha256(open(p, "rb").read()).hexdigest()
    # the plain path: hand-off off, file read back
    monkeypatch.setenv("PCH_RESIDENT_HANDOFF", "0")
    plain = str(tmp_path / "plain" / "point_2.las")
    import_PC.run_voxel_downsampling(path, plain, 0.1, 500000)
    want_sha = sha(plain)
    want = te.extract_towers(plain, log_callback=lambda m: None)
    assert len(want) >= 1
    monkeypatch.setenv("PCH_RESIDENT_HANDOFF", "1")
    reads = []
    real_read = _las.read_device
    monkeypatch.setattr(_las, "read_device", lambda p, dev: (reads.append(p), real_read(p, dev))[1])
    # 1. handed over on the device: no read of the intermediate file
    out = str(tmp_path / "output" / "point_2.las")
    import_PC.run_voxel_downsampling(path, out, 0.1, 500000)
    n_reads = len(reads)                                               # the input file was read, nothing else
    logs = []
    got = te.extract_towers(out, log_callback=logs.append)
    assert len(reads) == n_reads, "extract_towers read the file although its records were resident"
    assert sha(out) == want_sha
    _towers_equal(want, got)
    # 2. consumed: a second call reads the file
    _towers_equal(want, te.extract_towers(out, log_callback=lambda m: None))
    assert len(reads) == n_reads + 1
    # 3. touched between the two calls: the stamp no longer matches -> the file is read
    import_PC.run_voxel_downsampling(path, out, 0.1, 500000)
    resident.wait_for_writers()
    n_reads = len(reads)
    st = os.stat(out)
    os.utime(out, ns=(st.st_atime_ns, st.st_mtime_ns + 1_000_000))
    _towers_equal(want, te.extract_towers(out, log_callback=lambda m: None))
    assert len(reads) == n_reads + 1
    # 4. another path that merely holds the same bytes is not the registered file
    import_PC.run_voxel_downsampling(path, out, 0.1, 500000)
    resident.wait_for_writers()
    n_reads = len(reads)
    other = str(tmp_path / "copy.las")
    with open(other, "wb") as f:
        f.write(open(out, "rb").read())
    _towers_equal(want, te.extract_towers(other, log_callback=lambda m: None))
    assert len(reads) == n_reads + 1
    assert resident.take(out) is not None                              # still registered (and now consumed)


def test_strip_lattice_reps_kernel_equals_the_torch_statement(cuda):
    """pch_strip_lattice_reps_f32 (hash table of lattice cells, atomicMin of the row) against tiles.strip_representatives
    on CPU tensors (torch operators): the same (row, label) pairs for one and two strips, 2.4e6 m from the origin, with a
    pair buffer that is too small at first, and nothing for strips without core points."""
    from pointcloudhookup_amd import tiles
    rng = np.random.default_rng(18)
    far = np.array([260000.0, 2435000.0, 40.0])
    pts = np.vstack([rng.uniform(0, 1, (200_000, 3)) * [400.0, 100.0, 40.0],
                     rng.normal([150.0, 50.0, 20.0], [2.5, 2.5, 9.0], (60_000, 3))]) + far
    pts = pts[rng.permutation(len(pts))].astype(np.float32)
    rows = np.cumsum(rng.integers(1, 4, len(pts))).astype(np.int64)        # ascending global rows with gaps
    core = rng.random(len(pts)) < 0.7
    labels = np.where(rng.random(len(pts)) < 0.9, rng.integers(0, 50, len(pts)), -1).astype(np.int32)
    e1, e2 = float(np.float32(far[0] + 150.0)), float(np.float32(far[0] + 300.0))
    for strips in ([(e1 - 8.0, e1 + 8.0)], [(e1 - 8.0, e1 + 8.0), (e2 - 8.0, e2 + 8.0)], [(0.0, 1.0)]):
        want = tiles.strip_representatives(torch.from_numpy(pts), torch.from_numpy(rows),
                                           torch.from_numpy(labels.astype(np.int64)), torch.from_numpy(core), strips, 8.0)
        for cap in (4096, 16):                                           # 16: overflows, the call repeats itself
            got = ops.strip_lattice_reps(torch.from_numpy(pts).to(cuda), torch.from_numpy(rows).to(cuda),
                                         torch.from_numpy(labels).to(cuda), torch.from_numpy(core).to(cuda), strips, 8.0,
                                         cap=cap)
            assert len(got) == len(want)
            for g, w in zip(got, want):
                assert torch.equal(g.cpu(), w), (strips, cap, g.shape, w.shape)
    assert want[0].shape[0] == 0
    # through the dispatcher: device tensors take the kernel
    got = tiles.strip_representatives(torch.from_numpy(pts).to(cuda), torch.from_numpy(rows).to(cuda),
                                      torch.from_numpy(labels.astype(np.int64)).to(cuda), torch.from_numpy(core).to(cuda),
                                      [(e1 - 8.0, e1 + 8.0)], 8.0)
    want = tiles.strip_representatives(torch.from_numpy(pts), torch.from_numpy(rows),
                                       torch.from_numpy(labels.astype(np.int64)), torch.from_numpy(core),
                                       [(e1 - 8.0, e1 + 8.0)], 8.0)
    assert want[0].shape[0] > 100 and torch.equal(got[0].cpu(), want[0])
    with pytest.raises(ValueError):
        ops.strip_lattice_reps(torch.from_numpy(pts * 10).to(cuda), torch.from_numpy(rows).to(cuda),
                               torch.from_numpy(labels).to(cuda), torch.from_numpy(core).to(cuda),
                               [(0.0, 1e9)], 8.0)                       # 2.4e7 m from the origin: beyond the lattice
