"""Stage D1, fast mode (pch_obb_shell_f32 + pch_obb_min_boxes_f64) against the exact mode
(pointcloudhookup_amd/obb.py on the full cluster, pinned to oracle/obb.py in tests/test_oracle.py).

What can be asserted exactly: the device filter never drops a point that is on the hull, and whenever
qhull builds the same hull from the reduced input the boxes agree to rounding.  What cannot: qhull merges
near-coplanar facets depending on the points it was shown, so now and then the reduced input yields another
(equally small) box - that is why the fast mode is opt-in.  The tests bound that difference and print it."""
import numpy as np
import pytest
import torch
from scipy.spatial import ConvexHull

from pointcloudhookup_amd import obb, ops, pipeline

pytestmark = pytest.mark.gpu


def _cloud(rng, kind, n):
    if kind == "gauss":
        return rng.normal([0, 0, 25], [2.5, 2.5, 9], (n, 3))
    if kind == "box":
        return rng.uniform([-4, -6, 3], [4, 6, 45], (n, 3))
    if kind == "quantised":                                      # LAS-like 1 cm grid: exact coplanarity everywhere
        return np.round(rng.normal([0, 0, 25], [2, 3, 8], (n, 3)), 2)
    if kind == "cut":                                            # flat bottom, like the height filter leaves it
        p = rng.normal([0, 0, 20], [3, 3, 12], (3 * n, 3))
        return p[p[:, 2] > 12.0][:n]
    if kind == "lattice":                                        # a pylon-like frame: points on few planes
        t = rng.uniform(0, 1, (n, 1))
        corner = rng.integers(0, 4, (n, 1))
        sx, sy = np.where(corner & 1, 1.0, -1.0), np.where(corner & 2, 1.0, -1.0)
        w = 5.0 * (1 - 0.8 * t)
        return np.hstack([sx * w, sy * w, 40 * t]) + rng.normal(0, 0.02, (n, 3))
    raise ValueError(kind)


def _grouped(parts, rng, offset=(1000.0, -2000.0, 0.0), noise=500):
    """float32 points of several clusters shuffled together with noise rows; returns device tensors
    (points, perm, offsets) and the per-cluster float32 arrays in (label, row) order."""
    pts = [np.asarray(p, np.float64) + np.asarray(offset) + [60.0 * i, 0, 0] for i, p in enumerate(parts)]
    lab = [np.full(len(p), i, np.int32) for i, p in enumerate(pts)]
    pts.append(rng.uniform(-500, 500, (noise, 3)))
    lab.append(np.full(noise, -1, np.int32))
    P = np.vstack(pts).astype(np.float32)
    L = np.concatenate(lab)
    order = rng.permutation(len(P))
    P, L = P[order], L[order]
    dP, dL = torch.from_numpy(P).cuda(), torch.from_numpy(L).cuda()
    perm, offsets, _ = ops.segment_by_label(dL, dP, len(parts))
    clusters = [P[L == i] for i in range(len(parts))]
    return dP, perm, offsets, clusters


KINDS = ["gauss", "box", "quantised", "cut", "lattice"]


def test_shell_keeps_every_hull_point(cuda):
    rng = np.random.default_rng(11)
    parts = [_cloud(rng, k, n) for k, n in zip(KINDS * 2, (43000, 30000, 20000, 25000, 12000, 2048, 2047, 300, 5000, 9000))]
    dP, perm, offsets, clusters = _grouped(parts, rng)
    keep = ops.obb_shell(dP, perm, offsets, len(parts)).cpu().numpy().astype(bool)
    off = offsets.cpu().numpy()
    assert len(keep) == off[-1]
    for i, c in enumerate(clusters):
        kk = keep[off[i]:off[i + 1]]
        c64 = c.astype(np.float64)
        hull = ConvexHull(c64, qhull_options="QbB Pp Qt")
        assert kk[hull.vertices].all(), f"cluster {i}: a hull vertex was dropped"
        if len(c) < 2048:
            assert kk.all()
            continue
        # every dropped point is strictly inside the hull (all facet equations clearly negative)
        eq = ConvexHull(c64).equations                           # unscaled (QbB reports them in the unit cube)
        d = c64[~kk] @ eq[:, :3].T + eq[:, 3]
        assert d.max() < -5e-8
        assert kk.mean() < (0.3 if KINDS[i % len(KINDS)] == "lattice" else 0.12), (i, kk.mean())   # most is dropped
    # nothing outside the grouped range is touched, noise rows are not part of it
    assert off[-1] == sum(len(c) for c in clusters)


def test_fast_boxes_bound_their_clusters_and_match_exact_mode(cuda, capsys):
    rng = np.random.default_rng(5)
    parts = [_cloud(rng, KINDS[i % len(KINDS)], int(rng.integers(3000, 45000))) for i in range(40)]
    dP, perm, offsets, clusters = _grouped(parts, rng)
    for order in obb._EXTENT_ORDERS:
        fast = obb.boxes_fast(dP, perm, offsets, len(parts), order)
        dc, dv, same_hull = [], [], 0
        for (box, err), c in zip(fast, clusters):
            assert err is None
            ext, tr = box
            c64 = c.astype(np.float64)
            to = np.linalg.inv(tr)
            local = c64 @ to[:3, :3].T + to[:3, 3]
            assert (np.abs(local) <= ext / 2 + 1e-6).all()       # a bounding box of the WHOLE cluster
            assert np.allclose(to[:3, :3] @ to[:3, :3].T, np.eye(3), atol=1e-12)
            e_ext, e_tr = obb.bounding_box_oriented(c64, order)
            dv.append(abs(np.prod(ext) / np.prod(e_ext) - 1))
            dc.append(np.abs(tr[:3, 3] - e_tr[:3, 3]).max())
            if dc[-1] < 1e-9:
                same_hull += 1
                assert np.allclose(np.sort(ext), np.sort(e_ext), atol=1e-9)
                assert np.allclose(np.abs(tr[:3, :3]), np.abs(e_tr[:3, :3]), atol=1e-9)   # axes up to sign
        dc, dv = np.array(dc), np.array(dv)
        with capsys.disabled():
            print(f"\n[obb fast vs exact, {order}] centre delta: median {np.median(dc):.2e} m, max {dc.max():.2e} m, "
                  f"> 1e-3 m in {(dc > 1e-3).sum()} of {len(dc)}; volume ratio - 1: max {dv.max():.2e}")
        # the usual case is identical to rounding; in the rest qhull lists the facets of the reduced input in
        # another order, trimesh's rule "first normal of every 0.1 rad bucket" picks other representatives and
        # another box of about the same volume wins
        assert same_hull >= len(parts) * 2 // 3
        assert dv.max() < 2e-2 and dc.max() < 0.5


def test_fast_mode_degenerate_clusters_fail_like_exact_mode(cuda):
    rng = np.random.default_rng(3)
    flat = np.hstack([rng.uniform(-5, 5, (4000, 2)), np.zeros((4000, 1))])        # coplanar: qhull refuses
    line = np.hstack([rng.uniform(-5, 5, (300, 1)), np.zeros((300, 2))])
    good = _cloud(rng, "gauss", 6000)
    dP, perm, offsets, clusters = _grouped([flat, good, line], rng, offset=(0.0, 0.0, 0.0))
    fast = obb.boxes_fast(dP, perm, offsets, 3)
    exact = list(obb.boxes_of([c.astype(np.float64) for c in clusters], workers=1))
    assert [e is None for _, e in fast] == [e is None for _, e in exact] == [False, True, False]
    assert np.allclose(np.sort(fast[1][0][0]), np.sort(exact[1][0][0]), rtol=2e-2)


def test_tower_table_fast_mode_against_exact_mode(cuda):
    from pointcloudhookup_amd import synth
    tile = synth.corridor_torch(3_000_000, seed=synth.SEED0 + 4, kind="corridor", offset=True, towers=12,
                                dtype=torch.float32)
    cl = pipeline.cluster_points(tile)
    logs_e, logs_f = [], []
    exact = pipeline.tower_table(cl, log=logs_e.append)
    fast = pipeline.tower_table(cl, log=logs_f.append, obb_mode="fast")
    assert len(exact) >= 2
    assert [t["label"] for t in fast] == [t["label"] for t in exact]
    assert logs_e == logs_f
    for a, b in zip(exact, fast):
        assert np.array_equal(a["points"], b["points"])
        assert np.abs(a["center"] - b["center"]).max() < 0.05
        assert np.abs(np.asarray(a["extent"]) - np.asarray(b["extent"])).max() < 0.05
        d = abs(a["north_angle"] - b["north_angle"]) % 180.0     # axis sign is qhull's choice in exact mode
        assert min(d, 180.0 - d) < 2.0
    with pytest.raises(ValueError):
        pipeline.tower_table(cl, obb_mode="approximate")


def test_shell_argument_checks_and_small_inputs(cuda):
    rng = np.random.default_rng(2)
    # no clusters at all, and clusters that are all below the pre-filter's size: everything is kept
    pts = torch.from_numpy(rng.normal(size=(100, 3)).astype(np.float32)).cuda()
    perm = torch.arange(100, dtype=torch.int32, device="cuda")
    assert ops.obb_shell(pts, perm, torch.zeros(1, dtype=torch.int64, device="cuda"), 0).numel() == 0
    offs = torch.tensor([0, 40, 100], dtype=torch.int64, device="cuda")
    assert ops.obb_shell(pts, perm, offs, 2).cpu().numpy().all()
    with pytest.raises(ValueError):
        ops.obb_shell(pts, perm, offs, 3)                           # offsets do not match the cluster count
    with pytest.raises(TypeError):
        ops.obb_shell(pts.cpu(), perm, offs, 2)                     # no CPU fallback
    # a cluster made of one repeated point / a segment: nothing can be proven interior, nothing is dropped
    same = np.repeat(rng.normal(size=(1, 3)), 5000, axis=0)
    seg = np.outer(np.linspace(0, 1, 5000), [3.0, 4.0, 12.0])
    dP, perm2, offs2, clusters = _grouped([same, seg], rng, offset=(0.0, 0.0, 0.0), noise=10)
    keep = ops.obb_shell(dP, perm2, offs2, 2).cpu().numpy()
    assert keep.all()


def test_tower_table_through_the_pool_is_the_serial_loop_and_may_run_beside_the_next_tile(cuda):
    """Exact mode: the clustered points travel once into a shared registered buffer, worker processes box them.
    The result must be the serial loop's bit for bit (every cluster through obb.bounding_box_oriented in this
    process), for the blocking call and for tower_table_async with the next tile clustered meanwhile."""
    from pointcloudhookup_amd import synth
    tile = synth.corridor_torch(3_000_000, seed=synth.SEED0 + 4, kind="corridor", offset=True, towers=12,
                                dtype=torch.float32)
    cl = pipeline.cluster_points(tile)
    k = int(cl["nclusters"])
    offs = cl["offsets"].cpu().numpy()
    host = cl["ground"]["points"].index_select(0, cl["perm"][: int(offs[k])].long()).cpu().numpy()
    serial = [obb._boxed((host[offs[i]:offs[i + 1]], "unsorted")) for i in range(k)]
    want = [t for kind, t in pipeline._accept(serial, cl["ground"]["centroid"], 0.8, 15.0, 50.0, 8, 30.0)
            if kind == "tower"]
    assert len(want) >= 2
    tm, logs, seen = {}, [], []
    got = pipeline.tower_table(cl, log=logs.append, on_accept=seen.append, timings=tm,
                               prepare=lambda acc: seen.append(len(acc)))
    assert seen[0] == len(got) and seen[1:] == got                       # prepare first, then every tower in order
    assert tm["workers"] >= 1 and tm["clustered_points"] == int(offs[k]) and tm["boxes_worker_cpu_ms"] > 0

    def same(a, b):
        assert [t["label"] for t in a] == [t["label"] for t in b]
        for x, y in zip(a, b):
            for key in ("center", "rotation", "extent"):
                assert np.array_equal(np.asarray(x[key]), np.asarray(y[key])), key
            assert x["north_angle"] == y["north_angle"]
            assert np.array_equal(y["points"], host[offs[x["label"]]:offs[x["label"] + 1]])

    same(want, got)
    # in flight beside the next tile's device work; two tables at once use two buffers
    j1 = pipeline.tower_table_async(cl)
    cl2 = pipeline.cluster_points(tile)
    j2 = pipeline.tower_table_async(cl2)
    del cl2
    same(want, j1.result(120))
    same(want, j2.result(120))
    assert j1.timings["workers"] >= 1
