"""The HIP path run DIRECTLY on the committed golden fixtures (no oracle in between):

* tests/golden/refrun_*.npz - what the reference's own utils/tower_extraction.py::extract_towers
  produced in the build container (logs, every cluster_points array it handed to trimesh, tower
  dicts, per-tower LAS coordinates); compared bit for bit with ops.tower_clusters and with the
  drop-in extract_towers on a LAS file holding the same integers.
* tests/golden/dbscan_*.npz - labels / core masks of the real sklearn call the reference makes.
"""
import hashlib
import json
import os
import sys

import numpy as np
import pytest
import torch

from pointcloudhookup_amd import las, ops

pytestmark = pytest.mark.gpu

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
sys.path.insert(0, GOLD)
import gen_golden as gg      # noqa: E402  (seeded input builders shared with the fixture generator)


def _sha(a):
    return hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()


def _load(case):
    g = np.load(os.path.join(GOLD, f"refrun_{case}.npz"))
    x, y, z, XYZ, kwargs = gg.refrun_inputs(case)
    assert gg.input_checksum(x, y, z) == str(g["input_sha"]), "seeded input differs from the fixture's"
    return g, x, y, z, XYZ, kwargs


# ------------------------------------------------------------------ stage C on the sklearn fixtures
@pytest.mark.parametrize("name", ["blobs600", "towers5000", "towers30000_chunk10000", "all_noise", "border_tie"])
def test_dbscan_on_sklearn_fixture(cuda, name):
    d = np.load(os.path.join(GOLD, f"dbscan_{name}.npz"))
    X = torch.from_numpy(d["X"]).to(cuda)
    labels, core, k = ops.dbscan(X, float(d["eps"]), int(d["min_samples"]), int(d["chunk"]), want_core=True)
    np.testing.assert_array_equal(labels.cpu().numpy(), d["labels"])
    np.testing.assert_array_equal(core.cpu().numpy(), d["core"])
    assert k == int(d["labels"].max()) + 1


# ------------------------------------------------------------------ stages B-D0 on the reference runs
@pytest.mark.parametrize("case", ["config1_1m", "towers5x3", "fallback", "nonfinite"])
def test_tower_clusters_on_reference_run(cuda, case):
    g, x, y, z, XYZ, kwargs = _load(case)
    logs = [str(s) for s in g["logs"]]
    raw_host = np.stack([x, y, z], axis=1).astype(np.float32)          # utils/tower_extraction.py:62
    raw = torch.from_numpy(raw_host).to(cuda)
    eps, ms = kwargs.get("eps", 8.0), kwargs.get("min_points", 80)
    gf, labels, k, perm, offsets, stats = ops.tower_clusters(raw, eps, ms, 50000)
    n_first = int(g["n_filtered_logged"])                              # the reference's own log line (:85)
    assert gf["count_at_offset"] == n_first
    assert gf["used_fallback"] == ("⚠️ 过滤后点数太少，尝试降低过滤阈值" in logs)
    if not gf["used_fallback"]:
        assert gf["count"] == n_first
    assert k == int(g["n_clusters"])
    if case == "nonfinite":                                            # every chunk holds NaN -> all noise
        assert int((labels != -1).sum()) == 0
        return
    np.testing.assert_array_equal(labels.cpu().numpy(), g["labels"].astype(np.int32))
    pts = gf["points"].cpu().numpy()
    perm, offsets = perm.cpu().numpy(), offsets.cpu().numpy()
    np.testing.assert_array_equal(np.diff(offsets), g["cluster_sizes"])
    for c in range(k):                                                 # byte-for-byte what the reference gave trimesh
        assert _sha(pts[perm[offsets[c]:offsets[c + 1]]]) == str(g["cluster_sha"][c])


@pytest.mark.parametrize("case", ["config1_1m", "towers5x3", "fallback"])
@pytest.mark.parametrize("order", ["unsorted", "trimesh_sorted"])
def test_dropin_extract_towers_on_reference_run(cuda, case, order, tmp_path, monkeypatch):
    from pointcloudhookup_amd.utils import tower_extraction as te
    g, x, y, z, XYZ, kwargs = _load(case)
    path = str(tmp_path / "cloud.las")
    las.write(path, las.LasHeader(point_format=3, version=(1, 2), scales=g["scales"], offsets=g["offsets"]), XYZ)
    monkeypatch.chdir(tmp_path)
    monkeypatch.setattr(te, "OBB_EXTENT_ORDER", order)
    logs, prog = [], []
    towers = te.extract_towers(path, progress_callback=prog.append, log_callback=logs.append, **kwargs)
    ref_logs = [str(s) for s in (g["logs"] if order == "unsorted" else g["trimesh_sorted_logs"])]
    # the xlsx writer is absent on the GPU box (openpyxl): the two lines around it may differ
    skip = ("\n✅ 杆塔信息已保存到", "检测到杆塔数量", "⚠️ 保存Excel失败")
    assert [m for m in logs if not m.startswith(skip)] == [m for m in ref_logs if not m.startswith(skip)]
    if order == "unsorted":
        assert prog == g["progress"].tolist()
    n = len(g[f"{order}_center"])
    assert len(towers) == n
    for i, t in enumerate(towers):
        assert set(t) == {"center", "rotation", "extent", "height", "width", "north_angle", "points"}
        np.testing.assert_allclose(t["center"], g[f"{order}_center"][i], rtol=0, atol=1e-3)   # north_star tolerance
        np.testing.assert_array_equal(t["center"], g[f"{order}_center"][i])
        np.testing.assert_array_equal(t["extent"], g[f"{order}_extent"][i])
        np.testing.assert_array_equal(t["rotation"], g[f"{order}_rotation"][i])
        assert t["north_angle"] == g[f"{order}_north_angle"][i]
        assert _sha(t["points"]) == str(g[f"{order}_points_sha"][i])
    if order == "trimesh_sorted" and case in ("config1_1m", "towers5x3"):
        # the consumer of these dicts (utils/table_match_gim.py reads center / height / north_angle): same matches
        # as the reference's own match_towers made of the reference-run towers (tests/golden/gim_match.json)
        from oracle import gim_match as ogm
        gold = json.load(open(os.path.join(GOLD, "gim_match.json")))["cases"][case]
        matched, conv = ogm.match_towers(gold["gim"], towers,
                                         lambda x, y: (112.0 + (x - 437000.0) * 1e-5, 28.0 + (y - 3139000.0) * 9e-6))
        assert [list(m) for m in matched] == gold["matched"]
        for c, w in zip(conv, gold["converted"]):
            np.testing.assert_array_equal(np.array(c["converted_center"]), np.array(w["converted_center"]))
            assert c["height"] == w["height"] and c["north_angle"] == w["north_angle"]
    if order == "unsorted":
        for name, want in zip(g["las_paths"], g["las_XYZ_sha"]):
            got = las.read(str(tmp_path / "output_towers" / str(name)))
            assert _sha(got.XYZ) == str(want)


def test_dropin_empty_cloud_like_reference(cuda, tmp_path, monkeypatch):
    from pointcloudhookup_amd.utils import tower_extraction as te
    g, x, y, z, XYZ, kwargs = _load("empty")
    path = str(tmp_path / "empty.las")
    las.write(path, las.LasHeader(point_format=3, version=(1, 2), scales=g["scales"], offsets=g["offsets"]), XYZ)
    monkeypatch.chdir(tmp_path)
    logs, prog = [], []
    assert te.extract_towers(path, progress_callback=prog.append, log_callback=logs.append) == []
    assert logs == [str(s) for s in g["logs"]] and prog == g["progress"].tolist()


def test_dropin_chunk_failure_like_reference(cuda, tmp_path, monkeypatch):
    """refrun_nonfinite: one NaN x makes the float32 centroid's x NaN, so every chunk holds NaN.
    scikit-learn rejects the first chunk, the reference logs that - and then its own `finally: del
    ... clustering ...` raises UnboundLocalError out of extract_towers (recorded in the fixture).
    LAS integers cannot hold a NaN, so it is injected behind the float32 cast."""
    from pointcloudhookup_amd.utils import tower_extraction as te
    g, x, y, z, _, kwargs = _load("nonfinite")
    assert str(g["raised"]).startswith("UnboundLocalError")
    sc, of = g["scales"], g["offsets"]
    xf = np.where(np.isnan(x), of[0], x)
    XYZ = np.stack([np.round((c - of[a]) / sc[a]) for a, c in enumerate((xf, y, z))], axis=1).astype(np.int32)
    path = str(tmp_path / "cloud.las")
    las.write(path, las.LasHeader(point_format=3, version=(1, 2), scales=sc, offsets=of), XYZ)
    bad = int(np.flatnonzero(np.isnan(x))[0])
    real_cast = ops.cast_f32

    def cast_with_nan(t):
        out = real_cast(t)
        out[bad, 0] = float("nan")
        return out

    monkeypatch.setattr(ops, "cast_f32", cast_with_nan)
    monkeypatch.chdir(tmp_path)
    ref_logs = [str(s) for s in g["logs"]]
    for mode in ("reference", "noise"):
        monkeypatch.setattr(te, "CHUNK_FAILURE", mode)
        logs, prog = [], []
        if mode == "reference":
            with pytest.raises(UnboundLocalError):
                te.extract_towers(path, progress_callback=prog.append, log_callback=logs.append)
            assert logs[:-1] == ref_logs[:-1] and prog == g["progress"].tolist()
            assert logs[-1] == ref_logs[-1].split("\n")[0]             # first sentence of sklearn's message
        else:
            assert te.extract_towers(path, progress_callback=prog.append, log_callback=logs.append) == []
            assert logs[:len(ref_logs) - 1] == ref_logs[:-1] and logs[-1] == "✅ 杆塔提取完成" and prog[-1] == 100


_FAST_KNOWN_MISS = ("towers5x3",)


@pytest.mark.parametrize("case", [
    "config1_1m",
    pytest.param("towers5x3", marks=pytest.mark.xfail(strict=True, reason=(
        "opt-in fast OBB mode: qhull lists the facets of the device-filtered cluster in another order, trimesh's "
        "'first normal per 0.1 rad bucket' rule then picks other candidates, one box comes out centimetres apart and "
        "flips the size filter - fast mode accepts 2 towers where the reference run accepts 1.  A parity failure of "
        "that mode (DESIGN.md section 11), which is why exact mode is the default"))),
    "fallback"])
def test_dropin_fast_obb_mode_on_reference_run(cuda, case, tmp_path, monkeypatch, capsys):
    """PCH_OBB_MODE=fast on the reference-run fixtures must give the reference run's towers: the same NUMBER of
    towers, the same points, centres within the north star's 1e-3 m.  Where it does not, the case is marked
    xfail(strict) with the reason - the assertion is not widened."""
    from pointcloudhookup_amd.utils import tower_extraction as te
    g, x, y, z, XYZ, kwargs = _load(case)
    path = str(tmp_path / "cloud.las")
    las.write(path, las.LasHeader(point_format=3, version=(1, 2), scales=g["scales"], offsets=g["offsets"]), XYZ)
    monkeypatch.chdir(tmp_path)
    monkeypatch.setattr(te, "OBB_MODE", "fast")
    towers = te.extract_towers(path, log_callback=lambda m: None, **kwargs)
    n = len(g["unsorted_center"])
    with capsys.disabled():
        print(f"\n[fast OBB vs reference run, {case}] towers: reference {n}, fast {len(towers)}")
    assert len(towers) == n
    for i, t in enumerate(towers):
        assert _sha(t["points"]) == str(g["unsorted_points_sha"][i])
        assert float(np.abs(t["center"] - g["unsorted_center"][i]).max()) <= 1e-3
        assert float(np.abs(np.asarray(t["extent"]) - g["unsorted_extent"][i]).max()) <= 1e-3
