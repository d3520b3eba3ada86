"""CPU suite, part 1: the oracle against its pins (golden vectors from the real third-party
calls, the live libraries when importable, and self-consistency of the restatements)."""
import json
import os

import numpy as np
import pytest

from oracle import boxes as obx
from oracle import dbscan as odb
from oracle import ground_filter as ogf
from oracle import obb as oobb
from oracle import towers as otw
from oracle import voxel as ovx

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
DB_CASES = ["blobs600", "towers5000", "towers30000_chunk10000", "all_noise", "border_tie"]


def _load(name):
    d = np.load(os.path.join(GOLD, f"dbscan_{name}.npz"))
    return d["X"], d["labels"], d["core"], float(d["eps"]), int(d["min_samples"]), int(d["chunk"])


# ------------------------------------------------------------------ stage C pins
@pytest.mark.parametrize("name", DB_CASES)
def test_dbscan_c_oracle_matches_sklearn_golden(oracle_clib, name):
    X, labels, core, eps, ms, chunk = _load(name)
    got = odb.dbscan_chunked(X, eps, ms, chunk, fit="c")
    np.testing.assert_array_equal(got, labels)


@pytest.mark.parametrize("name", ["blobs600", "border_tie", "all_noise"])
def test_dbscan_literal_and_rule_match_golden(name):
    X, labels, core, eps, ms, chunk = _load(name)
    lit, c1 = odb.dbscan_fit_numpy(X, eps, ms)
    rule, c2 = odb.dbscan_rule(X, eps, ms)
    np.testing.assert_array_equal(lit, labels)
    np.testing.assert_array_equal(rule, labels)          # the order-free rule the kernels implement
    np.testing.assert_array_equal(c1, core)
    np.testing.assert_array_equal(c2, core)


def test_dbscan_oracle_vs_live_sklearn(oracle_clib):
    pytest.importorskip("sklearn")
    rng = np.random.default_rng(11)
    for trial in range(4):
        X = np.vstack([rng.normal(rng.uniform(0, 60, 3), [2.5, 2.5, 9.0], (900, 3)) for _ in range(3)]
                      + [rng.uniform(0, 80, (500, 3))]).astype(np.float32)
        X = X[rng.permutation(len(X))]
        ref, rcore = odb.dbscan_fit_sklearn(X, 8.0, 80)
        got, gcore = odb.dbscan_fit_c(X, 8.0, 80)
        np.testing.assert_array_equal(got, ref)
        np.testing.assert_array_equal(gcore, rcore)


def test_dbscan_chunk_offsets_follow_reference_rule():
    # chunk 1 has no cluster: current_label must not advance (utils/tower_extraction.py:116)
    a = np.zeros((10, 3), np.float32)
    noise = (np.arange(30, dtype=np.float32).reshape(10, 3) * 1000 + 5000)
    X = np.vstack([a, noise, a + 1])
    got = odb.dbscan_chunked(X, 0.5, 5, 10, fit="numpy")
    assert got[:10].tolist() == [0] * 10 and got[10:20].tolist() == [-1] * 10 and got[20:].tolist() == [1] * 10


# ------------------------------------------------------------------ stage B pins
def test_numpy_stats_golden():
    gold = json.load(open(os.path.join(GOLD, "numpy_stats.json")))
    # the device percentile mirrors numpy >= 2 (float32 index arithmetic under NEP 50); numpy 1.x promotes
    # (n-1)*q to float64 and can pick another order statistic above 2^24 rows - a different reference
    assert int(np.__version__.split(".")[0]) == gold["_numpy"]["major"] == 2
    for n, g in gold.items():
        if n.startswith("_"):
            continue
        n = int(n)
        rng = np.random.default_rng(g["seed"])
        raw = (rng.random((n, 3)) * [1000.0, 100.0, 30.0] + [437000.0, 3139000.0, 80.0]).astype(np.float32)
        assert int(raw.view(np.uint32).sum(dtype=np.uint64)) == g["checksum"], "numpy random stream changed"
        c = np.mean(raw, axis=0)
        assert [int(v) for v in c.view(np.uint32)] == g["centroid_bits"]
        np.testing.assert_array_equal(ogf.mean_seq_f32(raw).view(np.uint32), c.view(np.uint32))
        z = raw[:, 2] - c[2]
        p = np.percentile(z, 25)
        assert int(np.float32(p).view(np.uint32)) == g["pct25_bits"]
        v, _, _, _ = ogf.percentile_linear_f32(z, 25)
        assert v.view(np.uint32) == np.float32(p).view(np.uint32)


def test_sequential_mean_is_far_from_true_mean():
    """the reference's float32 centroid is kilometres off."""
    gold = json.load(open(os.path.join(GOLD, "numpy_stats.json")))
    assert abs(gold["2000000"]["centroid"][1] - 3139050.0) > 1000.0


@pytest.mark.parametrize("n", [1, 2, 3, 4, 7, 100, 1001, 4097])
@pytest.mark.parametrize("q", [0, 25, 50, 99.5, 100])
def test_percentile_restatement_equals_numpy(n, q):
    rng = np.random.default_rng(n + int(q))
    z = rng.normal(0, 3, n).astype(np.float32)
    z[rng.integers(0, n, max(1, n // 3))] = np.float32(0.5)
    v, _, _, _ = ogf.percentile_linear_f32(z, q)
    ref = np.float32(np.percentile(z, q))
    assert v.view(np.uint32) == ref.view(np.uint32)


def test_ground_filter_fallback_and_shapes():
    rng = np.random.default_rng(0)
    raw = np.column_stack([rng.uniform(0, 50, 3000), rng.uniform(0, 50, 3000),
                           rng.normal(0, 0.7, 3000)]).astype(np.float32)
    raw[:40, 2] += 10
    r = ogf.ground_filter(raw)
    assert r["used_fallback"] and r["threshold"] == np.float32(r["base"] + 1.0)
    assert r["filtered"].dtype == np.float32 and r["filtered"].shape[1] == 3
    np.testing.assert_array_equal(r["filtered"], r["points"][r["keep"]])


# ------------------------------------------------------------------ stage A (parity unpinned)
def test_voxel_oracle_properties():
    rng = np.random.default_rng(5)
    pts = rng.random((5000, 3)) * [30, 20, 5] + [437000.0, 3139000.0, 80.0]
    idx, mean, cnt = ovx.voxel_down_sample(pts, 0.5)
    assert cnt.sum() == len(pts) and len(np.unique(idx, axis=0)) == len(idx)
    assert (idx >= 0).all()
    lo = pts.min(0) - 0.25
    np.testing.assert_array_equal(np.floor((mean - lo) / 0.5).astype(np.int32), idx)   # mean stays in its voxel
    # permutation changes neither the voxel set nor the counts (means only up to summation order)
    perm = rng.permutation(len(pts))
    idx2, mean2, cnt2 = ovx.voxel_down_sample(pts[perm], 0.5)
    np.testing.assert_array_equal(idx, idx2)
    np.testing.assert_array_equal(cnt, cnt2)
    np.testing.assert_allclose(mean, mean2, rtol=0, atol=1e-9)
    # one voxel holding everything returns the in-order float64 mean
    i1, m1, c1 = ovx.voxel_down_sample(pts[:100], 1e4)
    s = np.zeros(3)
    for p in pts[:100]:
        s = s + p
    np.testing.assert_array_equal(m1[0], s / 100.0)
    with pytest.raises(ValueError):
        ovx.voxel_down_sample(np.array([[0.0, 0, 0], [1e7, 0, 0]]), 1e-3)


def test_voxel_chunks_keep_cross_chunk_duplicates():
    pts = np.array([[0.01, 0.01, 0.01], [0.02, 0.02, 0.02], [0.01, 0.02, 0.01], [0.03, 0.01, 0.02]])
    idx, mean, cnt, offs = ovx.voxel_down_sample_chunked(pts, 1.0, 2)
    assert offs.tolist() == [0, 1, 2] and cnt.tolist() == [2, 2]      # one voxel per chunk, not merged


def test_las_scale_unscale_roundtrip():
    X = np.array([-2147483648, -1, 0, 1, 123456789, 2147483647], dtype=np.int64)
    v = ovx.las_scaled(X, 0.001, 437000.0)
    np.testing.assert_array_equal(ovx.las_unscale(v, 0.001, 437000.0), X.astype(np.int32))


# ------------------------------------------------------------------ stage D / E
def test_obb_oracle_recovers_rotated_box():
    rng = np.random.default_rng(2)
    P = (rng.random((4000, 3)) - 0.5) * [40.0, 12.0, 6.0]
    R = np.linalg.qr(rng.normal(size=(3, 3)))[0]
    if np.linalg.det(R) < 0:
        R[:, 0] *= -1
    X = P @ R.T + [5.0, -3.0, 2.0]
    ext, T = oobb.bounding_box_oriented(X, "trimesh_sorted")
    np.testing.assert_allclose(ext, [6.0, 12.0, 40.0], rtol=0.02)
    np.testing.assert_allclose(T[:3, 3], [5.0, -3.0, 2.0], atol=0.2)
    assert abs(np.linalg.det(T[:3, :3]) - 1.0) < 1e-9
    ext_u, T_u = oobb.bounding_box_oriented(X, "unsorted")
    assert ext_u[0] >= ext_u[1]                                   # rectangle long side first
    np.testing.assert_allclose(sorted(ext_u), sorted(ext), rtol=1e-9)
    # every point lies inside the box
    q = (np.linalg.inv(T_u) @ np.column_stack([X, np.ones(len(X))]).T).T[:, :3]
    assert (np.abs(q) <= ext_u / 2 + 1e-6).all()


def test_north_angle_convention():
    assert otw.north_angle_deg(np.eye(3)) == pytest.approx(90.0)       # box x-axis = east -> 90 deg
    Rz90 = np.array([[0.0, -1, 0], [1, 0, 0], [0, 0, 1]])
    assert otw.north_angle_deg(Rz90) == pytest.approx(0.0)             # box x-axis = north -> 0 deg
    vertical = np.array([[0.0, 0, 1], [0, 1, 0], [-1, 0, 0]])
    assert otw.north_angle_deg(vertical) == pytest.approx(90.0)        # degenerate -> [1,0,0]


def test_kuangxuan_boxes_golden():
    g = json.load(open(os.path.join(GOLD, "kuangxuan_boxes.json")))
    for preset, ref in g["presets"].items():
        lo, hi = obx.kuangxuan_bounds(g["center"], g["extent"], preset)
        np.testing.assert_allclose(lo, ref["min"], rtol=0, atol=5e-3)
        np.testing.assert_allclose(hi, ref["max"], rtol=0, atol=5e-3)
        if "lines" in ref:                                           # produced by the reference itself
            np.testing.assert_array_equal(lo, ref["min"])
            np.testing.assert_array_equal(hi, ref["max"])
            np.testing.assert_array_equal(obx.box_line_points(lo, hi), np.array(ref["lines"]))
    # the values SURVEY.md section 8c records for the reference's own example
    lo, hi = obx.kuangxuan_bounds(g["center"], g["extent"], "kuangxuan_original")
    np.testing.assert_allclose(lo, [437567.798, 3140681.53, 114.057], atol=5e-3)
    np.testing.assert_allclose(hi, [437621.465, 3140711.68, 166.257], atol=5e-3)


def test_e2e_config1_self_golden(oracle_clib):
    """BASELINE config 1 through the CPU oracle (plumbing, no GPU): reproduces the committed
    self-golden and finds the 3 synthetic towers under the sorted-extent convention."""
    from pointcloudhookup_amd import synth
    g = np.load(os.path.join(GOLD, "e2e_config1.npz"))
    pts = synth.corridor_numpy(int(g["n"]), seed=int(g["seed"]), kind="corridor", offset=True, towers=3)
    r = otw.extract_towers_arrays(pts[:, 0], pts[:, 1], pts[:, 2], fit="c", extent_order="trimesh_sorted")
    np.testing.assert_array_equal(r["ground"]["centroid"], g["centroid"])
    assert len(r["ground"]["filtered"]) == int(g["n_filtered"])
    np.testing.assert_array_equal(r["labels"], g["labels"].astype(np.int32))
    assert r["n_candidates"] == int(g["n_candidates"])
    centres = np.array([t["center"] for t in r["towers"]])
    assert len(centres) == 3
    np.testing.assert_allclose(centres, g["trimesh_sorted_center"], rtol=0, atol=1e-3)
    true_x = (np.arange(3) + 0.5) * 100.0 / 3 + 437000.0
    np.testing.assert_allclose(np.sort(centres[:, 0]), true_x, atol=3.0)


# ------------------------------------------------------------------ reference-run fixtures
# tests/golden/refrun_*.npz hold what the reference's OWN extract_towers produced in the build
# container (gen_golden.reference_runs).  The oracle must reproduce them: this is the pin that
# ties oracle stages B0-D0 and D2-D4 to the reference's code path (utils/tower_extraction.py:57-218).
import hashlib
import sys

sys.path.insert(0, GOLD)
import gen_golden as gg      # noqa: E402  (inputs are rebuilt from seeds; the fixture stores their checksum)

REFRUN = gg.REFRUN_CASES


def _sha(a):
    return hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()


def load_refrun(case):
    g = np.load(os.path.join(GOLD, f"refrun_{case}.npz"))
    x, y, z, XYZ, kwargs = gg.refrun_inputs(case)
    assert gg.input_checksum(x, y, z) == str(g["input_sha"]), "seeded input differs from the fixture's"
    assert kwargs == json.loads(str(g["kwargs_json"]))
    return g, x, y, z, XYZ, kwargs


@pytest.mark.parametrize("case", REFRUN)
def test_oracle_reproduces_reference_run(oracle_clib, case):
    g, x, y, z, XYZ, kwargs = load_refrun(case)
    logs = [str(s) for s in g["logs"]]
    if case == "empty":                                   # np.percentile of nothing: filter stage gives up (:91-93)
        assert logs[-1].startswith("⚠️ 高度过滤失败") and int(g["n_clusters"]) == 0
        with pytest.raises(IndexError):
            with np.errstate(all="ignore"):
                import warnings
                with warnings.catch_warnings():
                    warnings.simplefilter("ignore")
                    ogf.ground_filter(np.zeros((0, 3), np.float32))
        return
    raw = np.stack([x, y, z], axis=1).astype(np.float32)
    with np.errstate(all="ignore"):
        gf = ogf.ground_filter(raw)
    first = [ln for ln in logs if ln.startswith("✅ 高度过滤完成")][0]
    n_first = int(first.split(":")[-1])
    assert n_first == int(g["n_filtered_logged"])
    if gf["used_fallback"]:
        assert "⚠️ 过滤后点数太少，尝试降低过滤阈值" in logs and n_first < 1000
    else:
        assert len(gf["filtered"]) == n_first
    if case == "nonfinite":                               # every chunk holds NaN: sklearn raises (:118-119) and the
        assert str(g["raised"]).startswith("UnboundLocalError")   # reference's own finally clause then escapes
        labels = odb.dbscan_chunked(gf["filtered"], 8.0, 80, 50000, fit="c")
        assert (labels == -1).all()
        return
    eps, ms = kwargs.get("eps", 8.0), kwargs.get("min_points", 80)
    labels = odb.dbscan_chunked(gf["filtered"], eps, ms, 50000, fit="c")
    k = int(g["n_clusters"])
    assert labels.max() + 1 == k
    np.testing.assert_array_equal(np.bincount(labels[labels >= 0], minlength=k), g["cluster_sizes"])
    for c in range(k):                                    # byte-for-byte what the reference handed to trimesh
        assert _sha(gf["filtered"][labels == c]) == str(g["cluster_sha"][c])
    np.testing.assert_array_equal(labels, g["labels"].astype(np.int32))
    nlines = [ln for ln in logs if ln.startswith("处理分块")]
    assert len(nlines) == -(-len(gf["filtered"]) // 50000)
    assert f"\n=== 开始杆塔检测（候选簇：{k}个） ===" in logs
    targs = {a: kwargs[a] for a in ("aspect_ratio_threshold", "min_height", "max_width", "min_width",
                                    "duplicate_threshold") if a in kwargs}
    for order in ("unsorted", "trimesh_sorted"):          # D2-D4: the reference's own accept / de-dup / angle code
        towers, ncand = otw.towers_from_labels(gf["filtered"], labels, gf["centroid"], extent_order=order, **targs)
        assert ncand == k and len(towers) == len(g[f"{order}_center"])
        for i, t in enumerate(towers):
            np.testing.assert_array_equal(t["center"], g[f"{order}_center"][i])
            np.testing.assert_array_equal(t["extent"], g[f"{order}_extent"][i])
            np.testing.assert_array_equal(t["rotation"], g[f"{order}_rotation"][i])
            assert t["north_angle"] == g[f"{order}_north_angle"][i]
            assert t["height"] == g[f"{order}_height"][i] and t["width"] == g[f"{order}_width"][i]
            assert _sha(t["points"]) == str(g[f"{order}_points_sha"][i])
    towers, _ = otw.towers_from_labels(gf["filtered"], labels, gf["centroid"], extent_order="unsorted", **targs)
    assert [f"tower_{t['label']}.las" for t in towers] == [str(p) for p in g["las_paths"]]
    sc, of = g["scales"], g["offsets"]
    for t, want in zip(towers, g["las_XYZ_sha"]):         # coordinates the reference handed to laspy (:205,254-256)
        orig = (t["points"] + gf["centroid"]).astype(np.float64)
        XYZ_out = np.stack([ovx.las_unscale(orig[:, a], sc[a], of[a]) for a in range(3)], axis=1)
        assert _sha(XYZ_out) == str(want)


# ------------------------------------------------------------------ consumer contract
def _gim_transform(x, y):
    return 112.0 + (x - 437000.0) * 1e-5, 28.0 + (y - 3139000.0) * 9e-6


@pytest.mark.parametrize("case", ["config1_1m", "towers5x3"])
def test_gim_matching_consumer_contract(case):
    """utils/table_match_gim.py reads tower['center'] (x, y, z), ['height'] and ['north_angle'] of the dicts
    extract_towers returns.  tests/golden/gim_match.json holds what the reference's own match_towers made of the
    reference-run towers; the restatement in oracle/gim_match.py must reproduce it from the same dict fields."""
    from oracle import gim_match as ogm
    gold = json.load(open(os.path.join(GOLD, "gim_match.json")))["cases"][case]
    g = np.load(os.path.join(GOLD, f"refrun_{case}.npz"))
    towers = [dict(center=g["trimesh_sorted_center"][i], height=float(g["trimesh_sorted_height"][i]),
                   north_angle=float(g["trimesh_sorted_north_angle"][i])) for i in range(len(g["trimesh_sorted_center"]))]
    matched, conv = ogm.match_towers(gold["gim"], towers, _gim_transform)
    assert [list(m) for m in matched] == gold["matched"]
    for c, w in zip(conv, gold["converted"]):
        assert c["id"] == w["id"] and c["height"] == w["height"] and c["north_angle"] == w["north_angle"]
        np.testing.assert_array_equal(np.array(c["converted_center"]), np.array(w["converted_center"]))
        assert c["n_value"] == w["n_value"] == 25.0
    for t in towers:                                       # what the consumer relies on
        assert len(t["center"]) == 3 and 0.0 <= t["north_angle"] < 360.0 and t["height"] > 0.0
