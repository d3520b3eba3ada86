"""Parity of every C-ABI operator against the CPU oracle on the same seeded inputs
(bit-exact for indices, counts, labels and for every float the reference computes)."""
import numpy as np
import pytest
import torch

from oracle import dbscan as odb
from oracle import ground_filter as ogf
from oracle import voxel as ovx
from pointcloudhookup_amd import ops, synth

pytestmark = pytest.mark.gpu

OFFSET = synth.GLOBAL_OFFSET


def _dev(a, cuda, dtype=None):
    t = torch.from_numpy(np.ascontiguousarray(a))
    if dtype is not None:
        t = t.to(dtype)
    return t.to(cuda)


# ------------------------------------------------------------------------------ stage A
@pytest.mark.parametrize("n,voxel,chunk", [(1, 0.1, 0), (5, 0.5, 2), (1000, 0.5, 0), (1000, 0.5, 300),
                                           (20000, 0.2, 7000), (200000, 0.1, 50000),
                                           (300000, 0.2, 0)])
def test_voxel_matches_oracle(cuda, n, voxel, chunk):
    rng = np.random.default_rng(n)
    pts = rng.random((n, 3)) * [60.0, 25.0, 8.0] + OFFSET
    pts[: n // 3] = np.round(pts[: n // 3], 1)              # many exact duplicates / shared voxels
    idx, mean, count, offs = ops.voxel_downsample(_dev(pts, cuda), voxel, chunk)
    ridx, rmean, rcount, roffs = ovx.voxel_down_sample_chunked(pts, voxel, chunk if chunk else n)
    assert idx.shape[0] == ridx.shape[0]
    _voxel_sets_equal(idx, mean, count, offs, ridx, rmean, rcount, roffs)


def _voxel_sets_equal(idx, mean, count, offs, ridx, rmean, rcount, roffs):
    """stage-A parity: per chunk the same SET of (voxel index, mean, count) - the library's order
    inside a chunk is its own (Open3D's is unordered_map order), so both sides are compared sorted by index"""
    offs = offs.cpu().numpy()
    np.testing.assert_array_equal(offs, roffs)
    gi, gm, gc = ovx.canonical(idx.cpu().numpy(), mean.cpu().numpy(), count.cpu().numpy(), offs)
    np.testing.assert_array_equal(gi, ridx)                          # voxel indices bit exact
    np.testing.assert_array_equal(gc, rcount)
    np.testing.assert_array_equal(gm.view(np.uint64), rmean.view(np.uint64))   # in-order f64 sums: bit exact


@pytest.mark.parametrize("case", ["dense_core", "one_voxel_20000", "wide_keys_u64", "huge_keys_general",
                                  "ragged_tiles", "tower_like", "all_equal_points"])
def test_voxel_paths_match_oracle(cuda, case):
    """Every path of the voxel finisher: units sorted inside LDS with 32-bit and 64-bit items, units too
    large for LDS (LSD passes in global memory), keys too wide for an LDS item, one voxel holding more rows
    than an LDS tile (in-order sum across tiles), ragged partition tiles."""
    rng = np.random.default_rng(hash(case) % 2**32)
    if case == "dense_core":              # 60 000 points inside 2 m: level-1 units far above the LDS capacity
        pts, voxel, chunk = rng.normal(0, 0.6, (60000, 3)) + OFFSET, 0.05, 0
    elif case == "one_voxel_20000":       # a single voxel with 20 000 rows + scattered others
        pts = np.vstack([rng.random((20000, 3)) * 0.09 + 5.0, rng.random((3000, 3)) * 40.0]) + OFFSET
        pts, voxel, chunk = pts[rng.permutation(len(pts))], 0.1, 0
    elif case == "wide_keys_u64":         # 2 km x 2 km x 100 m at 1 cm: 18+18+14 = 50 key bits -> 64-bit LDS items
        pts, voxel, chunk = rng.random((30000, 3)) * [2000.0, 2000.0, 100.0] + OFFSET, 0.01, 0
    elif case == "huge_keys_general":     # 21+21+21 = 63 key bits: too wide for an LDS item
        pts, voxel, chunk = rng.random((20000, 3)) * 2000.0 + OFFSET, 0.001, 6000
    elif case == "ragged_tiles":          # chunk sizes that are no multiple of the 4096-row partition tile
        pts, voxel, chunk = rng.random((50001, 3)) * [50.0, 20.0, 5.0] + OFFSET, 0.25, 12345
    elif case == "tower_like":            # ground plane + a dense column, two chunks
        g = np.column_stack([rng.uniform(0, 50, 150000), rng.uniform(0, 100, 150000), rng.normal(0, 0.05, 150000)])
        t = rng.normal([25, 50, 22], [2.5, 2.5, 9], (150000, 3))
        pts = np.vstack([g, t])
        pts, voxel, chunk = pts[rng.permutation(len(pts))] + OFFSET, 0.2, 200000
    else:                                 # all_equal_points: zero key bits
        pts, voxel, chunk = np.tile(OFFSET + [1.0, 2.0, 3.0], (9000, 1)), 0.1, 4000
    idx, mean, count, offs = ops.voxel_downsample(_dev(pts, cuda), voxel, chunk)
    ridx, rmean, rcount, roffs = ovx.voxel_down_sample_chunked(pts, voxel, chunk if chunk else len(pts))
    _voxel_sets_equal(idx, mean, count, offs, ridx, rmean, rcount, roffs)


def test_voxel_single_voxel_and_negative_coords(cuda):
    pts = np.array([[-1.0, -2.0, -3.0], [-1.01, -2.01, -3.01], [-0.99, -1.99, -2.99]])
    idx, mean, count, offs = ops.voxel_downsample(_dev(pts, cuda), 5.0, 0)
    ridx, rmean, rcount = ovx.voxel_down_sample(pts, 5.0)
    _voxel_sets_equal(idx, mean, count, offs, ridx, rmean, rcount, np.array([0, len(rcount)]))


def test_voxel_too_small_raises(cuda):
    from pointcloudhookup_amd._lib import PchError
    pts = np.array([[0.0, 0.0, 0.0], [1.0e7, 0.0, 0.0]])
    with pytest.raises(PchError):
        ops.voxel_downsample(_dev(pts, cuda), 1e-3, 0)          # Open3D: voxel_size is too small


def test_las_scale_roundtrip(cuda):
    rng = np.random.default_rng(3)
    X = rng.integers(-2**31, 2**31 - 1, size=(5000, 3), dtype=np.int64).astype(np.int32)
    sc, of = [0.001, 0.001, 0.01], [437000.0, 3139000.0, -12.5]
    out = ops.las_scale(_dev(X, cuda), sc, of).cpu().numpy()
    ref = np.stack([ovx.las_scaled(X[:, a], sc[a], of[a]) for a in range(3)], axis=1)
    np.testing.assert_array_equal(out, ref)
    back = ops.las_unscale(_dev(ref, cuda), sc, of).cpu().numpy()
    refb = np.stack([ovx.las_unscale(ref[:, a], sc[a], of[a]) for a in range(3)], axis=1)
    np.testing.assert_array_equal(back, refb)


@pytest.mark.parametrize("fmt,ver", [(0, (1, 2)), (2, (1, 2)), (3, (1, 2)), (7, (1, 4)), (8, (1, 4))])
def test_las_records_decoded_on_device(cuda, tmp_path, fmt, ver):
    """X,Y,Z gathered from the raw point records on the GPU == the host-side record parse
    (record lengths 20/26/34/36/38: aligned and misaligned int32 fields)."""
    from pointcloudhookup_amd import las
    rng = np.random.default_rng(fmt)
    XYZ = rng.integers(-2**31, 2**31 - 1, (10007, 3), dtype=np.int64).astype(np.int32)
    p = str(tmp_path / "t.las")
    las.write(p, las.LasHeader(point_format=fmt, version=ver, scales=np.array([0.001] * 3),
                               offsets=np.array([1.0, 2.0, 3.0])), XYZ)
    hdr, dev_XYZ = las.read_device(p, cuda)
    assert hdr.point_format == fmt and hdr.record_length == las.RECORD_LEN[fmt]
    np.testing.assert_array_equal(dev_XYZ.cpu().numpy(), XYZ)
    np.testing.assert_array_equal(las.read(p).XYZ, XYZ)


# ------------------------------------------------------------------------------ stage B
@pytest.mark.parametrize("n", [1, 2, 3, 7, 4096, 4097, 100000, 1500000])
def test_mean_seq_bit_exact(cuda, n):
    rng = np.random.default_rng(n)
    raw = (rng.random((n, 3)) * [1000, 100, 30] + OFFSET).astype(np.float32)
    got = ops.mean_seq_f32(_dev(raw, cuda)).cpu().numpy()
    np.testing.assert_array_equal(got.view(np.uint32), np.mean(raw, axis=0).view(np.uint32))


def _mean_cases():
    rng = np.random.default_rng(42)
    n = 700000
    yield "local", (rng.random((n, 3)) * [10000, 100, 30]).astype(np.float32)
    yield "zero_mean", rng.normal(0, 100, (n, 3)).astype(np.float32)
    yield "integers_ties", rng.integers(-50, 50, (n, 3)).astype(np.float32)
    yield "halves_ties", (rng.integers(0, 4000, (n, 3)) * 0.5).astype(np.float32)
    yield "quarter_lattice", (rng.integers(12556000, 12557000, (n, 3)) * 0.25).astype(np.float32)
    yield "constant", np.full((n, 3), 3.14e6, np.float32)
    yield "tiny", (rng.random((n, 3)) * 1e-40).astype(np.float32)
    yield "mixed_scale", (rng.random((n, 3)) * (10.0 ** rng.integers(-6, 7, (n, 1)))).astype(np.float32)
    a = (rng.random((n, 3)) * 100).astype(np.float32)
    a[123456, 1] = 1e30
    yield "outlier", a
    a = (rng.random((n, 3)) * 100).astype(np.float32)
    a[5000, 0] = np.inf
    a[600000, 0] = -np.inf
    a[77, 2] = np.nan
    a[300000, 1] = np.inf
    yield "nonfinite", a
    a = np.zeros((n, 3), np.float32)
    a[400000:] = (rng.random((n - 400000, 3)) - 0.3).astype(np.float32)
    yield "leading_zeros", a
    yield "negative", (-(rng.random((n, 3)) * [1000, 100, 30] + OFFSET)).astype(np.float32)
    yield "sign_flip_walk", (np.sin(np.arange(3 * n).reshape(n, 3) * 1e-3) * 50).astype(np.float32)


@pytest.mark.parametrize("name", [c[0] for c in _mean_cases()])
def test_mean_seq_distributions(cuda, name):
    raw = dict(_mean_cases())[name]
    t = _dev(raw, cuda)
    with np.errstate(all="ignore"):
        ref = np.mean(raw, axis=0)
    got = ops.mean_seq_f32(t).cpu().numpy()
    ser = ops.mean_seq_f32(t, serial=True).cpu().numpy()
    np.testing.assert_array_equal(ser.view(np.uint32) & 0x7FFFFFFF if np.isnan(ref).any() else ser.view(np.uint32),
                                  ref.view(np.uint32) & 0x7FFFFFFF if np.isnan(ref).any() else ref.view(np.uint32))
    if np.isnan(ref).any():
        assert (np.isnan(got) == np.isnan(ref)).all()
        np.testing.assert_array_equal(got[~np.isnan(ref)], ref[~np.isnan(ref)])
    else:
        np.testing.assert_array_equal(got.view(np.uint32), ref.view(np.uint32))


def test_mean_seq_hard_stretches_and_ragged_end(cuda):
    """The exact path of the centroid walk is the sequential chain itself (ms_blocks_exact: 64 broadcast rounds of 16
    register-held elements), and behind a failing block that left its binade the walk adds a doubling run of blocks
    without consulting the tables.  Columns built for that: a zero-mean stretch of 1.5 M rows that turns into a
    growing sum (the blind runs reach their cap of 32 and must be dropped again), a column that is hard, easy, hard
    again, and one that is hard up to its ragged last block (rows beyond the end count as -0.0, and the running sum
    IS -0.0 for the first rows of that column)."""
    rng = np.random.default_rng(77)
    n = 3_000_017                                              # 2929 full blocks + 721 rows
    a = rng.normal(0, 0.05, n)
    a[1_500_000:] = np.abs(rng.normal(20, 5, n - 1_500_000))
    b = rng.normal(0, 3.0, n)
    b[700_000:1_900_000] = rng.uniform(400, 500, 1_200_000)
    b[1_900_000:] = rng.normal(-450 * 1.2e6 / (n - 1_900_000), 3.0, n - 1_900_000)   # walks the sum back through zero
    c = rng.normal(0, 1e-3, n)
    c[:5000] = -0.0
    raw = np.column_stack([a, b, c]).astype(np.float32)
    got = ops.mean_seq_f32(_dev(raw, cuda)).cpu().numpy()
    np.testing.assert_array_equal(got.view(np.uint32), np.mean(raw, axis=0).view(np.uint32))
    # the same rows as a filter input: the percentile's sample half now reads the raw rows beside the summary
    ref = ogf.ground_filter(raw)
    flt = ops.ground_filter(_dev(raw, cuda))
    np.testing.assert_array_equal(flt["centroid"].view(np.uint32), ref["centroid"].view(np.uint32))
    assert np.float32(flt["threshold"]).view(np.uint32) == ref["threshold"].view(np.uint32)
    assert flt["count"] == len(ref["filtered"])
    np.testing.assert_array_equal(flt["index"].cpu().numpy(), np.flatnonzero(ref["keep"]))


def test_mean_seq_clouds_centred_on_themselves(cuda):
    """Zero-mean columns: the running sums wander about zero for the whole file, every block holds both signs, and the
    walk leans on the level-1 records' bounds of the RUNNING prefix (ms_summary_k, MS_TIGHT) instead of the order-free
    sum of the positive steps.  A corridor minus its own centroid; and columns whose blocks carry prefix bounds beyond
    32 bits (long same-sign plateaus of large values that cancel later, then noise of the same size) - a bound cut to
    32 bits and then scaled to a high candidate binade once certified blocks it must not."""
    from pointcloudhookup_amd import synth
    pts = synth.corridor_numpy(6_000_000, seed=synth.SEED0 + 9, kind="corridor", offset=False)
    raw = (pts - pts.mean(axis=0)).astype(np.float32)
    got = ops.mean_seq_f32(_dev(raw, cuda)).cpu().numpy()
    np.testing.assert_array_equal(got.view(np.uint32), np.mean(raw, axis=0).view(np.uint32))
    rng = np.random.default_rng(2026)
    n = 4_200_000
    a = np.concatenate([np.full(1_400_000, 3.1e6), np.full(1_400_000, -3.1e6), rng.normal(0, 1.0e6, 1_400_000)])
    b = np.concatenate([rng.normal(0, 2.0e6, 1_400_000), np.full(1_400_000, 7.7e5), rng.normal(-7.7e5, 3.0e5, 1_400_000)])
    c = rng.uniform(-50.0, 50.0, n) + np.where(np.arange(n) % 200_000 < 100_000, 2.2, -2.2)
    raw = np.column_stack([a, b, c]).astype(np.float32)
    got = ops.mean_seq_f32(_dev(raw, cuda)).cpu().numpy()
    np.testing.assert_array_equal(got.view(np.uint32), np.mean(raw, axis=0).view(np.uint32))
    ser = ops.mean_seq_f32(_dev(raw, cuda), serial=True).cpu().numpy()
    np.testing.assert_array_equal(ser.view(np.uint32), got.view(np.uint32))


def _chained(raw, cuts, cuda):
    """np.mean over shards chained in file order: running sums handed from shard to shard, the last one divides"""
    run = None
    bounds = [0] + list(cuts) + [len(raw)]
    for k in range(len(bounds) - 1):
        part = _dev(raw[bounds[k]:bounds[k + 1]], cuda)
        last = k == len(bounds) - 2
        run = ops.mean_seq_partial_f32(part, run, total_n=len(raw) if last else 0)
    return run.cpu().numpy()


@pytest.mark.parametrize("name", [c[0] for c in _mean_cases()])
def test_mean_seq_sharded_equals_numpy(cuda, name):
    """Three file-order shards chained through pch_mean_seq_partial_f32 (12 bytes per hop) give
    np.mean(concatenation, axis=0) bit for bit on the adversarial distributions of the unsharded test - the shard
    cuts fall inside 1024-row blocks, on a block edge and next to the array's ends."""
    raw = dict(_mean_cases())[name]
    n = len(raw)
    with np.errstate(all="ignore"):
        ref = np.mean(raw, axis=0)
    for cuts in ((n // 3 + 5, 2 * n // 3 + 1), (1024 * 200, 1024 * 200 + 1), (1, n - 1), (0, n), (n // 2, n // 2)):
        got = _chained(raw, cuts, cuda)
        if np.isnan(ref).any():
            assert (np.isnan(got) == np.isnan(ref)).all(), (name, cuts)
            np.testing.assert_array_equal(got[~np.isnan(ref)], ref[~np.isnan(ref)])
        else:
            np.testing.assert_array_equal(got.view(np.uint32), ref.view(np.uint32), err_msg=f"{name} {cuts}")


def test_mean_seq_sharded_corridor_8_shards(cuda):
    """Config 4's shape: 8 consecutive shards of an EPSG-scale corridor (12 M rows), sums passed down the line."""
    raw = synth.corridor_numpy(12_000_000, seed=synth.SEED0 + 9, kind="corridor", offset=True).astype(np.float32)
    n = len(raw)
    cuts = [n * k // 8 + (k % 3) for k in range(1, 8)]
    got = _chained(raw, cuts, cuda)
    np.testing.assert_array_equal(got.view(np.uint32), np.mean(raw, axis=0).view(np.uint32))
    one = ops.mean_seq_f32(_dev(raw, cuda)).cpu().numpy()
    np.testing.assert_array_equal(got.view(np.uint32), one.view(np.uint32))


def test_mean_seq_stagnation_40m(cuda):
    """float32 running sums that stop growing (ulp/2 > element): 40 M rows at EPSG:4547 scale."""
    n = 40_000_000
    g = torch.Generator(device=cuda)
    g.manual_seed(7)
    t = torch.rand((n, 3), generator=g, device=cuda, dtype=torch.float32)
    t = t * torch.tensor([1000.0, 100.0, 30.0], device=cuda) + torch.tensor(OFFSET, device=cuda,
                                                                          dtype=torch.float32)
    raw = t.cpu().numpy()
    ref = np.mean(raw, axis=0)
    got = ops.mean_seq_f32(t).cpu().numpy()
    np.testing.assert_array_equal(got.view(np.uint32), ref.view(np.uint32))
    assert abs(float(ref[1]) - 3139050.0) > 1e5        # the reference's centroid really is that far off


@pytest.mark.parametrize("n", [1, 2, 3, 4, 5, 9, 1000, 1001, 65537, 1000003])
@pytest.mark.parametrize("q", [25, 0, 100, 50, 73.5])
def test_percentile_bit_exact(cuda, n, q):
    rng = np.random.default_rng(n * 7 + int(q))
    z = rng.normal(0, 5, n).astype(np.float32)
    if n > 8:
        z[rng.integers(0, n, n // 4)] = np.float32(1.25)     # heavy duplicates around the rank
    pts = np.zeros((n, 3), np.float32)
    pts[:, 2] = z
    t = _dev(pts, cuda)
    got = ops.percentile_f32(t[:, 2], q).cpu().numpy()[0]
    ref = np.percentile(z, q)
    assert np.float32(got).view(np.uint32) == np.float32(ref).view(np.uint32), (got, ref)


def test_percentile_nan_and_sub(cuda):
    z = np.array([1.0, np.nan, 3.0, 2.0, 5.0], np.float32)
    got = ops.percentile_f32(_dev(z, cuda), 25).cpu().numpy()[0]
    assert np.isnan(got) and np.isnan(np.percentile(z, 25))
    z = np.random.default_rng(0).normal(80, 9, 5001).astype(np.float32)
    c = torch.tensor([77.125], dtype=torch.float32, device=cuda)
    got = ops.percentile_f32(_dev(z, cuda), 25, sub=c).cpu().numpy()[0]
    ref = np.percentile(z - np.float32(77.125), 25)
    assert np.float32(got).view(np.uint32) == np.float32(ref).view(np.uint32)


@pytest.mark.parametrize("n,kind,offset", [(50000, "corridor", True), (50000, "corridor", False),
                                           (200000, "uniform", True), (1000000, "corridor", True),
                                           (3000, "flat", False)])
def test_ground_filter_matches_numpy(cuda, n, kind, offset):
    if kind == "flat":       # fewer than 1000 survivors at +3.0 -> the +1.0 fallback
        rng = np.random.default_rng(5)
        pts = np.column_stack([rng.uniform(0, 50, n), rng.uniform(0, 50, n), rng.normal(0, 0.7, n)])
        pts[:40, 2] += 10.0
    else:
        pts = synth.corridor_numpy(n, seed=synth.SEED0 + 1, kind=kind, offset=offset, towers=3)
    raw = pts.astype(np.float32)
    ref = ogf.ground_filter(raw)
    got = ops.ground_filter(_dev(raw, cuda))
    np.testing.assert_array_equal(got["centroid"].view(np.uint32), ref["centroid"].view(np.uint32))
    assert np.float32(got["base"]).view(np.uint32) == ref["base"].view(np.uint32)
    assert np.float32(got["threshold"]).view(np.uint32) == ref["threshold"].view(np.uint32)
    assert got["used_fallback"] == ref["used_fallback"]
    assert got["count"] == len(ref["filtered"])
    np.testing.assert_array_equal(got["points"].cpu().numpy().view(np.uint32),
                                  ref["filtered"].view(np.uint32))
    np.testing.assert_array_equal(got["index"].cpu().numpy(), np.flatnonzero(ref["keep"]))
    if got["count"]:
        f = ref["filtered"]
        np.testing.assert_array_equal(got["aabb"], np.concatenate([f.min(0), f.max(0)]))


@pytest.mark.parametrize("case", ["slots", "slot_overflow", "estimate_too_high", "fallback_threshold", "nan_rows"])
def test_ground_filter_candidate_slots_and_their_fallbacks(cuda, case):
    """From 131 072 rows on, the centroid pass also emits the candidate rows of the filter (raw z above a LOW estimate
    of the threshold, 256 per 1024-row block) and the sweep reads those instead of the tile - but only when a
    device-side guard proves that no survivor can be missing (gf_cand_ok).  Every route must give numpy's result bit
    for bit: the slots; a block with more candidates than its slot holds (a tower in file order: overflow word ->
    the sweep over the tile); an estimate that is not low enough (threshold far below the sampled quantile + margin:
    guard fails -> the sweep over the tile); the `< 1000 survivors` fallback threshold; NaN / inf rows."""
    rng = np.random.default_rng(11)
    n = 400_003
    pts = np.column_stack([rng.uniform(0, 2000, n), rng.uniform(0, 100, n), rng.normal(0, 0.05, n)])
    tall = rng.random(n) < 0.08
    pts[tall, 2] = rng.uniform(4, 40, tall.sum())
    if case == "slot_overflow":
        pts[200_000:206_000, 2] = rng.uniform(10, 30, 6000)          # six whole blocks of candidates in file order
    if case == "estimate_too_high":
        # a quarter of the rows far below the rest: the 25th percentile sits on a cliff of the distribution, so the
        # exact value may lie well below the sampled estimate - whichever route is taken must be exact
        low = rng.random(n) < 0.2503
        pts[low, 2] = rng.uniform(-500, -400, low.sum())
    if case == "fallback_threshold":
        pts[:, 2] = rng.normal(0, 0.7, n)                             # fewer than 1000 rows above +3.0
        pts[:300, 2] += 10.0
    if case == "nan_rows":                                             # non-finite x / y on rows that survive
        pts[[12345, 222222, 333333], 2] = 20.0
        pts[12345, 0] = np.nan
        pts[222222, 1] = np.inf
        pts[333333, 1] = -np.inf
    raw = (pts + np.array([437000.0, 3139000.0, 80.0])).astype(np.float32)
    ref = ogf.ground_filter(raw)
    got = ops.ground_filter(_dev(raw, cuda))
    assert got["used_fallback"] == ref["used_fallback"]
    if case not in ("nan_rows",):
        assert ref["used_fallback"] == (case == "fallback_threshold")
    np.testing.assert_array_equal(got["centroid"].view(np.uint32), ref["centroid"].view(np.uint32))
    assert np.float32(got["threshold"]).view(np.uint32) == ref["threshold"].view(np.uint32)
    assert got["count"] == len(ref["filtered"])
    gp, f = got["points"].cpu().numpy(), ref["filtered"]
    nanpos = np.isnan(f)
    np.testing.assert_array_equal(np.isnan(gp), nanpos)               # NaN - c is NaN on both sides (payload bits aside)
    np.testing.assert_array_equal(gp.view(np.uint32)[~nanpos], f.view(np.uint32)[~nanpos])
    np.testing.assert_array_equal(got["index"].cpu().numpy(), np.flatnonzero(ref["keep"]))
    fin = np.isfinite(f).all(axis=1)
    if fin.any():                            # (a NaN / inf in a column makes that column's centroid, hence every row, non-finite)
        np.testing.assert_array_equal(got["aabb"], np.concatenate([f[fin].min(0), f[fin].max(0)]))


# ------------------------------------------------------------------------------ stage C
def _blobs(rng, n, k, spread, sigma, clutter):
    per = (n - clutter) // k
    parts = [rng.normal(rng.uniform(-spread, spread, 3), sigma, (per, 3)) for _ in range(k)]
    parts.append(rng.uniform(-spread * 1.3, spread * 1.3, (n - per * k, 3)))
    X = np.vstack(parts).astype(np.float32)
    return X[rng.permutation(len(X))]


@pytest.mark.parametrize("seed", range(6))
def test_dbscan_small_vs_sklearn_rule(cuda, seed):
    rng = np.random.default_rng(seed)
    X = _blobs(rng, 700, 3, 6.0, 1.0, 200)
    ref, rcore = odb.dbscan_fit_sklearn(X, 1.0, 8)
    labels, core, k = ops.dbscan(_dev(X, cuda), 1.0, 8, 0, want_core=True)
    np.testing.assert_array_equal(core.cpu().numpy(), rcore)
    np.testing.assert_array_equal(labels.cpu().numpy(), ref)
    assert k == ref.max() + 1


@pytest.mark.parametrize("n,eps,ms,chunk", [(5000, 8.0, 80, 0), (5000, 8.0, 80, 1200),
                                            (30000, 8.0, 80, 10000), (30000, 2.0, 10, 0),
                                            (12000, 0.5, 3, 5000), (4000, 8.0, 1, 0),
                                            (3000, 1e-3, 2, 0), (3000, 1e3, 5, 0)])
def test_dbscan_chunked_vs_oracle(cuda, oracle_clib, n, eps, ms, chunk):
    rng = np.random.default_rng(n + ms)
    k = 4
    per = n // (k + 1)
    parts = [rng.normal([rng.uniform(0, 300), rng.uniform(0, 100), 22.0], [2.5, 2.5, 9.0], (per, 3))
             for _ in range(k)]
    parts.append(np.column_stack([rng.uniform(0, 300, n - k * per), rng.uniform(0, 100, n - k * per),
                                  rng.uniform(0, 30, n - k * per)]))
    X = np.vstack(parts).astype(np.float32)
    X = X[rng.permutation(n)]
    ref = odb.dbscan_chunked(X, eps, ms, chunk, fit="c")
    labels, _, kk = ops.dbscan(_dev(X, cuda), eps, ms, chunk)
    np.testing.assert_array_equal(labels.cpu().numpy(), ref)
    assert kk == (ref.max() + 1 if (ref >= 0).any() else 0)


def test_dbscan_reference_chunk_vs_sklearn(cuda):
    """One full 50 000-row chunk shaped like the reference's input, checked against the real
    sklearn call of utils/tower_extraction.py:107-112."""
    rng = np.random.default_rng(1)
    X = np.vstack([rng.normal([c, 50, 22], [3, 3, 10], (15000, 3)) for c in (100, 400, 700)]
                  + [np.column_stack([rng.uniform(0, 1000, 5000), rng.uniform(0, 100, 5000),
                                      rng.uniform(0, 30, 5000)])]).astype(np.float32)
    X = X[rng.permutation(len(X))]
    ref, rcore = odb.dbscan_fit_sklearn(X, 8.0, 80)
    labels, core, k = ops.dbscan(_dev(X, cuda), 8.0, 80, 50000, want_core=True)
    np.testing.assert_array_equal(core.cpu().numpy(), rcore)
    np.testing.assert_array_equal(labels.cpu().numpy(), ref)


def test_dbscan_edge_cases(cuda):
    one = np.zeros((1, 3), np.float32)
    labels, core, k = ops.dbscan(_dev(one, cuda), 8.0, 1, 0, want_core=True)
    assert labels.cpu().tolist() == [0] and k == 1 and core.cpu().tolist() == [1]
    labels, _, k = ops.dbscan(_dev(one, cuda), 8.0, 2, 0)
    assert labels.cpu().tolist() == [-1] and k == 0
    same = np.ones((500, 3), np.float32) * 7.5                      # all coincident
    labels, _, k = ops.dbscan(_dev(same, cuda), 0.5, 80, 0)
    assert k == 1 and (labels.cpu().numpy() == 0).all()
    labels, _, k = ops.dbscan(_dev(same, cuda), 0.5, 80, 100)       # 5 chunks -> ids 0..4
    np.testing.assert_array_equal(labels.cpu().numpy(), np.repeat(np.arange(5), 100))
    empty = torch.zeros((0, 3), dtype=torch.float32, device=cuda)
    labels, _, k = ops.dbscan(empty, 8.0, 80, 0)
    assert labels.numel() == 0 and k == 0


def test_dbscan_nonfinite_chunk_stays_noise(cuda, oracle_clib):
    """sklearn raises ValueError for a chunk holding NaN/inf; the reference catches it and leaves
    that chunk at -1 while the other chunks (and the label counter) carry on
    (utils/tower_extraction.py:118-119)."""
    rng = np.random.default_rng(4)
    X = np.vstack([rng.normal([c, 50, 22], [2.5, 2.5, 9], (2000, 3)) for c in (100, 400, 700)]).astype(np.float32)
    X = X[rng.permutation(len(X))]
    bad = X.copy()
    bad[2500, 0] = np.nan                      # chunk 1 of 3 (chunk size 2000)
    bad[2600, 1] = np.inf
    want = np.full(len(X), -1, np.int64)
    cur = 0
    for s0 in (0, 4000):                       # chunks 0 and 2 are clustered, chunk 1 raises
        lab, _ = odb.dbscan_fit_c(X[s0:s0 + 2000], 8.0, 80)
        lab = lab.copy()
        lab[lab >= 0] += cur
        want[s0:s0 + 2000] = lab
        cur = lab.max() + 1 if (lab >= 0).any() else cur
    labels, core, k = ops.dbscan(_dev(bad, cuda), 8.0, 80, 2000, want_core=True)
    np.testing.assert_array_equal(labels.cpu().numpy(), want)
    assert k == cur and int(core.cpu().numpy()[2000:4000].sum()) == 0
    allbad = np.full((100, 3), np.nan, np.float32)
    labels, _, k = ops.dbscan(_dev(allbad, cuda), 8.0, 5, 0)
    assert k == 0 and (labels.cpu().numpy() == -1).all()


def test_dbscan_border_tie_takes_smallest_cluster(cuda):
    # two dense lines, one border point exactly between them within eps of both
    a = np.column_stack([np.linspace(0, 1, 30), np.zeros(30), np.zeros(30)])
    b = np.column_stack([np.linspace(3.2, 4.2, 30), np.zeros(30), np.zeros(30)])
    mid = np.array([[2.1, 0.0, 0.0]])
    X = np.vstack([b, mid, a]).astype(np.float32)                   # cluster 0 = b (seen first)
    ref, _ = odb.dbscan_fit_sklearn(X, 1.15, 8)
    labels, _, _ = ops.dbscan(_dev(X, cuda), 1.15, 8, 0)
    np.testing.assert_array_equal(labels.cpu().numpy(), ref)


# ------------------------------------------------------------------------------ stage D0
@pytest.mark.parametrize("n,K", [(100000, 37), (100000, 255), (70001, 256), (50000, 1000), (3000, 1)])
def test_segment_by_label(cuda, n, K):
    rng = np.random.default_rng(9)
    labels = rng.integers(-1, K, n).astype(np.int32)
    labels[labels == 5] = -1                                        # an empty cluster
    labels[1000:1800] = min(3, K - 1)                               # a run of one label (merged adds)
    xyz = rng.normal(0, 10, (n, 3)).astype(np.float32)
    perm, offs, stats = ops.segment_by_label(_dev(labels, cuda), _dev(xyz, cuda), K)
    perm, offs, stats = perm.cpu().numpy(), offs.cpu().numpy(), stats.cpu().numpy()
    for k in range(K):
        rows = perm[offs[k]:offs[k + 1]]
        np.testing.assert_array_equal(rows, np.flatnonzero(labels == k))
        if len(rows):
            np.testing.assert_array_equal(stats[k, :3], xyz[rows].min(0))
            np.testing.assert_array_equal(stats[k, 3:6], xyz[rows].max(0))
    np.testing.assert_array_equal(perm[offs[K]:], np.flatnonzero(labels == -1))


def test_dbscan_chunk_local_sort_equals_global_sort(cuda):
    """The chunk-local cell sort (one workgroup per chunk) and the general global radix sort feed
    the same clustering: labels and core masks must agree, with ragged last chunk, a NaN chunk and
    a chunk size that is not a multiple of anything."""
    rng = np.random.default_rng(11)
    n = 123457
    X = np.vstack([rng.normal([c, 50, 22], [3, 3, 10], (20000, 3)) for c in (100, 400, 700, 1000)]
                  + [np.column_stack([rng.uniform(0, 1200, n - 80000), rng.uniform(0, 100, n - 80000),
                                      rng.uniform(0, 30, n - 80000)])]).astype(np.float32)
    X = X[rng.permutation(n)]
    X[60001, 2] = np.nan
    dev = _dev(X, cuda)
    # chunk sizes around the sort kernel's 4096-row tile (its full and ragged tiles are separate code: exact
    # multiples, one row more, less than a tile, the largest chunk it takes) and cell sizes that make the chunk's
    # relative key need one, two and three passes (first / last pass are separate instantiations too)
    cases = [(8.0, 80, c) for c in (50000, 7777, 1000, 0, 4096, 8192, 4097, 100, 131072)]
    cases += [(60.0, 80, 50000), (60.0, 80, 4096), (0.7, 5, 50000), (0.7, 5, 8192), (0.2, 3, 20000)]
    for eps, ms, chunk in cases:
        try:
            ops.set_dbscan_sort_mode("chunk")
            la, ca, ka = ops.dbscan(dev, eps, ms, chunk, want_core=True)
            ops.set_dbscan_sort_mode("global")
            lb, cb, kb = ops.dbscan(dev, eps, ms, chunk, want_core=True)
        finally:
            ops.set_dbscan_sort_mode("auto")                       # the library's own choice (by chunk count)
        lc, cc, kc = ops.dbscan(dev, eps, ms, chunk, want_core=True)
        assert ka == kb == kc, (eps, ms, chunk)
        assert torch.equal(la, lb) and torch.equal(ca, cb) and torch.equal(la, lc) and torch.equal(ca, cc), (eps, ms, chunk)


@pytest.mark.parametrize("n", [1, 2, 63, 64, 65, 1023, 1024, 1025, 16383, 16384, 16385, 32769, 70001])
def test_ground_filter_sizes_around_tile_edges(cuda, n):
    """Block (1024), compaction tile (16384) and wave (64) boundaries, single rows included."""
    rng = np.random.default_rng(n)
    raw = np.column_stack([rng.uniform(437000, 437300, n), rng.uniform(3139000, 3139100, n),
                           80 + np.abs(rng.normal(0, 4.0, n))]).astype(np.float32)
    ref = ogf.ground_filter(raw)
    got = ops.ground_filter(_dev(raw, cuda))
    np.testing.assert_array_equal(got["centroid"].view(np.uint32), ref["centroid"].view(np.uint32))
    assert np.float32(got["base"]).view(np.uint32) == ref["base"].view(np.uint32)
    assert got["used_fallback"] == ref["used_fallback"] and got["count"] == len(ref["filtered"])
    np.testing.assert_array_equal(got["points"].cpu().numpy().view(np.uint32), ref["filtered"].view(np.uint32))
    np.testing.assert_array_equal(got["index"].cpu().numpy(), np.flatnonzero(ref["keep"]))


@pytest.mark.parametrize("mode", ["chunk", "global"])
def test_dbscan_nan_chunks_first_last_adjacent(cuda, oracle_clib, mode):
    """NaN/inf chunks at the front, at the end and side by side keep their own cell (never core),
    so the chunk -> cell table stays complete under both cell sorts; the other chunks are clustered
    as if alone and numbered like the reference (a failed chunk does not advance the counter)."""
    rng = np.random.default_rng(5)
    chunk, nch = 3000, 7
    X = np.vstack([rng.normal([c * 40.0, 0, 20], [2, 2, 6], (chunk, 3)) for c in range(nch)]).astype(np.float32)
    for c, col, val in ((0, 0, np.nan), (3, 1, np.inf), (4, 2, -np.inf), (6, 0, np.nan)):
        X[c * chunk + int(rng.integers(0, chunk)), col] = val
    want = odb.dbscan_chunked(X, 8.0, 80, chunk, fit="c")
    assert (want[:chunk] == -1).all() and (want[-chunk:] == -1).all() and want.max() >= 2
    try:
        ops.set_dbscan_sort_mode(mode)
        lab, core, k = ops.dbscan(_dev(X, cuda), 8.0, 80, chunk, want_core=True)
    finally:
        ops.set_dbscan_sort_mode("auto")
    np.testing.assert_array_equal(lab.cpu().numpy(), want)
    assert k == want.max() + 1
    assert ops.first_nonfinite_row(_dev(X, cuda)) == int(np.flatnonzero(~np.isfinite(X).all(1))[0])
    assert ops.first_nonfinite_row(_dev(X[chunk:3 * chunk], cuda)) == -1


def test_entry_points_reject_host_pointers_and_keep_the_callers_device(cuda):
    import ctypes as C
    from pointcloudhookup_amd import _lib
    L = _lib.lib()
    host = np.zeros((16, 3), np.float32)
    out = torch.zeros(3, dtype=torch.float32, device=cuda)
    ws = torch.zeros(1 << 20, dtype=torch.uint8, device=cuda)
    rc = L.pch_mean_seq_f32(host.ctypes.data_as(C.c_void_p), 16, out.data_ptr(), ws.data_ptr(), ws.numel(), None)
    assert rc == -1 and b"device pointer" in L.pch_last_error()
    before = torch.cuda.current_device()
    ops.mean_seq_f32(torch.ones((100, 3), dtype=torch.float32, device=cuda))
    assert torch.cuda.current_device() == before


def test_count_at_offset_is_an_integer_above_2_pow_24(cuda):
    """The count kept at the first threshold travels as an integer (2^24 + 1 is not a float32)."""
    n = (1 << 24) + 1 + 5000
    raw = torch.zeros((n, 3), dtype=torch.float32, device=cuda)
    raw[5000:, 2] = 10.0                      # percentile(25) = 10 - mean; z - mean > base + 3 never holds ...
    raw[:5000, 2] = -100.0                    # ... so shift: 5000 low rows, the rest 110 above them
    gf = ops.ground_filter(raw, 0.01, 3.0, 1.0, 1000, want_index=False)
    assert gf["count"] == (1 << 24) + 1 and gf["count_at_offset"] == (1 << 24) + 1
    g2 = ops.tower_clusters(raw, pct=0.01, segment=False)[0]
    assert g2["count_at_offset"] == (1 << 24) + 1


# ------------------------------------------------------------------------------ native LAS I/O
def test_native_las_reader_on_byte_built_files(cuda, tmp_path):
    """pch_las_read_xyz_i32 against files assembled from the LAS specification in tests/las_bytes.py (not by
    our own writer): VLRs, padding in front of the point data, extra bytes, odd record lengths (misaligned
    X/Y/Z), LAS 1.4 with the 64-bit count only, and a file of four 64 MiB hops (the ring of three pinned buffers
    wraps around; every hop is read by several threads with pread)."""
    import las_bytes
    from pointcloudhookup_amd import las
    rng = np.random.default_rng(5)
    cases = [dict(n=1000, point_format=3, version=(1, 2)),
             dict(n=777, point_format=2, version=(1, 2), vlr_payloads=(b"x" * 40, b"y" * 7), pad_before_points=13),
             dict(n=4097, point_format=1, version=(1, 3), extra_bytes=5),
             dict(n=5000, point_format=6, version=(1, 4), vlr_payloads=(b"z" * 100,), extra_bytes=3),
             dict(n=0, point_format=0, version=(1, 2)),
             dict(n=5_500_000, point_format=7, version=(1, 4), extra_bytes=1)]      # 37-byte records, 4 hops
    for i, kw in enumerate(cases):
        n = kw.pop("n")
        XYZ = rng.integers(-2**31, 2**31 - 1, size=(n, 3), dtype=np.int64).astype(np.int32)
        p = str(tmp_path / f"c{i}.las")
        las_bytes.build(p, XYZ, **kw)
        hdr, dev = las.read_device(p, cuda)
        assert dev.shape == (n, 3) and dev.dtype == torch.int32
        np.testing.assert_array_equal(dev.cpu().numpy(), XYZ)
        np.testing.assert_array_equal(las.read(p).XYZ, XYZ)          # the python reader agrees


def test_native_las_writer_roundtrip_and_layout(cuda, tmp_path):
    import las_bytes
    from pointcloudhookup_amd import las
    rng = np.random.default_rng(6)
    # 7.5 M x 28 B = 210 MB: four hops (records laid out on the device, pwrite by several threads); 26-byte records
    # (format 2) are not a multiple of 4: the last word of a hop is written byte by byte
    for n, fmt, ver in ((0, 3, (1, 2)), (1, 0, (1, 2)), (5000, 3, (1, 2)), (7_500_000, 1, (1, 3)), (1234, 6, (1, 4)),
                        (333_331, 2, (1, 2))):
        XYZ = rng.integers(-10**9, 10**9, size=(n, 3), dtype=np.int64).astype(np.int32)
        p = str(tmp_path / f"w{n}.las")
        hdr = las.LasHeader(point_format=fmt, version=ver, scales=np.array([0.001, 0.002, 0.01]),
                            offsets=np.array([437000.0, 3139000.0, -5.0]))
        back = las.write_device(p, hdr, torch.from_numpy(XYZ).to(cuda))
        got = las_bytes.parse_xyz(p)                                  # spec-level parse of what the library wrote
        assert got["n"] == n == back.point_count and got["point_format"] == fmt and got["version"] == ver
        assert got["offset_to_points"] == got["header_size"] and got["num_vlrs"] == 0
        assert got["record_length"] == las.RECORD_LEN[fmt]
        np.testing.assert_array_equal(got["XYZ"], XYZ)
        assert not got["other_bytes"].any()                           # every other record field is zero
        np.testing.assert_array_equal(got["scales"], hdr.scales)
        np.testing.assert_array_equal(got["offsets"], hdr.offsets)
        if n:
            np.testing.assert_array_equal(got["mins"], XYZ.min(0) * hdr.scales + hdr.offsets)
            np.testing.assert_array_equal(got["maxs"], XYZ.max(0) * hdr.scales + hdr.offsets)
        ref = str(tmp_path / f"r{n}.las")                             # same bytes as the python writer (drop-in files)
        las.write(ref, hdr, XYZ)
        assert open(ref, "rb").read() == open(p, "rb").read()


# ------------------------------------------------------------------------------ viewer helpers
@pytest.mark.parametrize("n", [0, 1, 63, 4095, 4096, 4097, 70001, 1_000_003])
def test_crop_aabb_equals_numpy_mask(cuda, n):
    """test/kuangxuan.py:69-79: inclusive bounds, order preserved, NaN rows dropped."""
    rng = np.random.default_rng(n + 1)
    pts = rng.random((n, 3)) * [400.0, 100.0, 60.0] + OFFSET
    pts[: n // 4] = np.round(pts[: n // 4], 0)                  # many rows exactly ON a bound
    if n > 10:
        pts[5, 1] = np.nan
    c = OFFSET + [200.0, 50.0, 20.0]
    w, h = 40.0, 15.0
    lo = np.array([np.round(c[0] - w / 1), c[1] - w / 2, c[2] - h / 1])       # the reference's own box arithmetic
    hi = np.array([np.round(c[0] + w / 0.6), c[1] + w / 1, c[2] + h * 2])
    mask = ((pts[:, 0] >= lo[0]) & (pts[:, 0] <= hi[0]) & (pts[:, 1] >= lo[1]) & (pts[:, 1] <= hi[1])
            & (pts[:, 2] >= lo[2]) & (pts[:, 2] <= hi[2]))
    got, idx = ops.crop_aabb(_dev(pts, cuda), lo, hi, want_index=True)
    np.testing.assert_array_equal(got.cpu().numpy(), pts[mask])
    np.testing.assert_array_equal(idx.cpu().numpy(), np.flatnonzero(mask))
    if n:
        everything = ops.crop_aabb(_dev(pts, cuda), [-np.inf] * 3, [np.inf] * 3)
        assert everything.shape[0] == n - (1 if n > 10 else 0)
        assert ops.crop_aabb(_dev(pts, cuda), [1e9] * 3, [2e9] * 3).shape[0] == 0


@pytest.mark.parametrize("n,k", [(1, 1), (10, 10), (1000, 7), (200001, 200000), (3_000_000, 500000), (5, 0)])
def test_decimate_is_a_sample_without_replacement(cuda, n, k):
    """np.random.choice(n, k, replace=False) is unseeded in the reference: parity is the property - exactly k rows,
    none twice, all from the input; the same seed gives the same rows, another seed other rows."""
    rng = np.random.default_rng(n)
    pts = rng.random((n, 3)) + OFFSET
    dev = _dev(pts, cuda)
    out, idx = ops.decimate(dev, k, seed=12345, want_index=True)
    idx_h = idx.cpu().numpy()
    assert out.shape == (k, 3) and len(np.unique(idx_h)) == k
    if k:
        assert idx_h.min() >= 0 and idx_h.max() < n
    np.testing.assert_array_equal(out.cpu().numpy(), pts[idx_h])
    again = ops.decimate(dev, k, seed=12345, want_index=True)[1].cpu().numpy()
    np.testing.assert_array_equal(again, idx_h)
    if 0 < k < n and n > 100:
        other = ops.decimate(dev, k, seed=999, want_index=True)[1].cpu().numpy()
        assert not np.array_equal(np.sort(other), np.sort(idx_h))
    if k >= 1000 and k < n:                                      # roughly uniform: every tenth of the rows gets its share
        share = np.bincount((idx_h * 10 // n).astype(np.int64), minlength=10) / k
        assert share.min() > 0.05 and share.max() < 0.2
    with pytest.raises(ValueError):
        ops.decimate(dev, n + 1)


@pytest.mark.parametrize("case", ["cauchy_single", "cauchy_chunked", "outlier_1e30", "two_blobs_1e9_apart", "line_of_isolated"])
def test_dbscan_grids_beyond_the_64bit_key(cuda, oracle_clib, case):
    """extent/eps far beyond what a 64-bit [chunk|cz|cy|cx] key can hold (round 1: PCH_ERR_RANGE for the whole call):
    chunks are then fitted one by one with their own boxes, and a single fit falls back to per-axis compressed
    cell coordinates - same labels as the all-pairs oracle."""
    rng = np.random.default_rng(abs(hash(case)) % 2**32)
    eps, ms, chunk = 0.5, 5, 0
    if case == "cauchy_single":
        X = rng.standard_cauchy((4000, 3)) * 50.0
    elif case == "cauchy_chunked":
        X, chunk = rng.standard_cauchy((6000, 3)) * 80.0, 700
        X[1500] = [3e12, -2e12, 1e12]
    elif case == "outlier_1e30":
        X = np.vstack([rng.normal(0, 1.0, (1500, 3)), rng.normal([40, 0, 0], 1.0, (1500, 3)), [[1e30, -1e30, 1e29]],
                       rng.uniform(-200, 200, (500, 3))])
        eps, ms = 1.0, 10
    elif case == "two_blobs_1e9_apart":
        X = np.vstack([rng.normal(0, 0.05, (2000, 3)), rng.normal([1e9, 1e9, -1e9], 40.0, (2000, 3))])
        eps, ms = 0.3, 50
    else:
        X = np.column_stack([np.arange(3000) * 1e6, np.zeros(3000), np.zeros(3000)])
        X[100:140] = X[100] + rng.normal(0, 0.1, (40, 3))
        eps, ms = 1.0, 8
    X = X[rng.permutation(len(X))].astype(np.float32)
    want = odb.dbscan_chunked(X, eps, ms, chunk, fit="c")
    lab, core, k = ops.dbscan(_dev(X, cuda), eps, ms, chunk, want_core=True)
    np.testing.assert_array_equal(lab.cpu().numpy(), want)
    assert k == (want.max() + 1 if (want >= 0).any() else 0)
    if chunk == 0:
        _, wcore = odb.dbscan_fit_c(X, eps, ms)
        np.testing.assert_array_equal(core.cpu().numpy(), wcore)


# ------------------------------------------------------------------------------ order-free properties
@pytest.mark.parametrize("seed", [1, 2, 3])
def test_voxel_set_is_invariant_under_permutation_within_a_chunk(cuda, seed):
    """Open3D's output order is unspecified, so parity is defined on the SET of voxels: shuffling the points of a chunk
    must leave (index, count) unchanged and move the means only by summation-order rounding."""
    rng = np.random.default_rng(seed)
    pts = rng.random((60000, 3)) * [80.0, 40.0, 12.0] + OFFSET
    a = ops.voxel_downsample(_dev(pts, cuda), 0.3, 0)
    b = ops.voxel_downsample(_dev(pts[rng.permutation(len(pts))], cuda), 0.3, 0)
    a = [torch.from_numpy(x) for x in ovx.canonical(a[0].cpu().numpy(), a[1].cpu().numpy(), a[2].cpu().numpy(),
                                                    a[3].cpu().numpy())]
    b = [torch.from_numpy(x) for x in ovx.canonical(b[0].cpu().numpy(), b[1].cpu().numpy(), b[2].cpu().numpy(),
                                                    b[3].cpu().numpy())]
    np.testing.assert_array_equal(a[0].cpu().numpy(), b[0].cpu().numpy())
    np.testing.assert_array_equal(a[2].cpu().numpy(), b[2].cpu().numpy())
    np.testing.assert_allclose(a[1].cpu().numpy(), b[1].cpu().numpy(), rtol=0, atol=1e-8)
    assert int(a[2].sum()) == len(pts)
    # every mean lies in its own voxel, and voxelising the means again keeps every voxel (idempotence of the index)
    lo = pts.min(0) - 0.15
    np.testing.assert_array_equal(np.floor((a[1].cpu().numpy() - lo) / 0.3).astype(np.int32), a[0].cpu().numpy())


@pytest.mark.parametrize("seed", [4, 5])
def test_dbscan_partition_is_invariant_under_permutation(cuda, seed):
    """sklearn's sweep numbers clusters by their smallest core index, so a permutation renumbers them - but the
    PARTITION of the core points, the core mask and the noise set are order-free (border points may legitimately
    switch between clusters they touch)."""
    rng = np.random.default_rng(seed)
    X = np.vstack([rng.normal([c, 50, 20], [2.5, 2.5, 8.0], (6000, 3)) for c in (40, 160, 300, 420)]
                  + [np.column_stack([rng.uniform(0, 460, 6000), rng.uniform(0, 100, 6000), rng.uniform(0, 40, 6000)])]
                  ).astype(np.float32)
    perm = rng.permutation(len(X))
    la, ca, ka = ops.dbscan(_dev(X, cuda), 8.0, 80, 0, want_core=True)
    lb, cb, kb = ops.dbscan(_dev(X[perm], cuda), 8.0, 80, 0, want_core=True)
    la, ca, lb, cb = la.cpu().numpy(), ca.cpu().numpy().astype(bool), lb.cpu().numpy(), cb.cpu().numpy().astype(bool)
    assert ka == kb
    np.testing.assert_array_equal(ca[perm], cb)                        # same core points
    np.testing.assert_array_equal(la[perm] == -1, lb == -1)            # same noise
    pairs = np.unique(np.stack([la[perm][cb], lb[cb]], axis=1), axis=0)
    assert len(pairs) == ka and len(np.unique(pairs[:, 0])) == ka and len(np.unique(pairs[:, 1])) == ka   # a bijection
    # ids are dense and ordered by first core row in either order
    for lab, core in ((la, ca), (lb, cb)):
        first = [np.flatnonzero((lab == c) & core)[0] for c in range(ka)]
        assert first == sorted(first)


def test_ground_filter_is_idempotent_on_its_own_threshold(cuda):
    """keep = z - cz > thr: filtering the kept points again with the SAME centroid and threshold keeps them all -
    checked through the returned index / points relation on a 5 M-point tile."""
    raw = synth_tile(cuda, 5_000_000)
    gf = ops.ground_filter(raw, want_index=True)
    c = torch.tensor(gf["centroid"], device=cuda)
    z = raw[:, 2] - c[2]
    assert int((z > float(gf["threshold"])).sum()) == gf["count"]
    assert torch.equal(gf["points"], raw[gf["index"].long()] - c)
    assert bool((gf["points"][:, 2] > float(gf["threshold"])).all())


def synth_tile(cuda, n):
    from pointcloudhookup_amd import synth
    return synth.corridor_torch(n, seed=synth.SEED0 + 9, kind="corridor", offset=True, device=cuda, dtype=torch.float32)


@pytest.mark.parametrize("case", ["normal", "all_equal", "ties_at_quantile", "sorted", "reverse_sorted", "with_nan",
                                  "two_values", "blocks_of_16"])
@pytest.mark.parametrize("q", [0.0, 25.0, 50.0, 99.9, 100.0])
def test_percentile_bracketed_select_is_exact(cuda, case, q):
    """From 4 Mi values on, the percentile reads the column once: a sample brackets the order statistic, one pass
    collects the bracket, the exact select runs on it - and the three full passes run instead whenever the device
    finds that the bracket missed (overflowing ties, adversarial order).  Always np.percentile bit for bit."""
    n = (1 << 22) + 12345
    rng = np.random.default_rng(abs(hash(case)) % 2**32)
    if case == "normal":
        z = rng.normal(80.0, 7.0, n)
    elif case == "all_equal":
        z = np.full(n, 81.5)
    elif case == "ties_at_quantile":
        z = rng.normal(80.0, 7.0, n)
        z[rng.random(n) < 0.6] = np.percentile(z, q)             # 60 % of the values ARE the quantile
    elif case == "sorted":
        z = np.sort(rng.normal(80.0, 7.0, n))
    elif case == "reverse_sorted":
        z = np.sort(rng.normal(80.0, 7.0, n))[::-1].copy()
    elif case == "with_nan":
        z = rng.normal(80.0, 7.0, n)
        z[rng.integers(0, n, 3)] = np.nan
    elif case == "two_values":
        z = np.where(rng.random(n) < 0.2499999, 1.0, 2.0)
    else:                                                         # the sample takes 16 of every 1024: make those special
        z = rng.normal(80.0, 7.0, n)
        z.reshape(-1)[: (n // 1024) * 1024].reshape(-1, 1024)[:, :16] = 1000.0
    z = z.astype(np.float32)
    want = np.float32(np.percentile(z, q))
    got = ops.percentile_f32(_dev(z, cuda), q).cpu().numpy()[0]
    assert (np.isnan(want) and np.isnan(got)) or got.view(np.uint32) == want.view(np.uint32), (case, q, got, want)


def test_voxel_scatter_staged_through_lds_equals_the_direct_form(cuda):
    """Large inputs take the partition whose tiles are put in digit order inside LDS before they are copied out
    (vx_scatter_lds_k; chosen by size, PCH_VX_SCATTER=lds forces it).  A child process with the switch set runs ragged
    tiles, two chunks with a dense column, and a 3 M-row corridor against the oracle (set equality per chunk)."""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    script = r'''
import sys, numpy as np, torch
sys.path.insert(0, sys.argv[1])
from oracle import voxel as ovx
from pointcloudhookup_amd import ops, synth
rng = np.random.default_rng(77)
OFF = np.asarray(synth.GLOBAL_OFFSET, dtype=np.float64)
g = np.column_stack([rng.uniform(0, 50, 150000), rng.uniform(0, 100, 150000), rng.normal(0, 0.05, 150000)])
t = rng.normal([25, 50, 22], [2.5, 2.5, 9], (150000, 3))
tower = np.vstack([g, t])
cases = [(rng.random((50001, 3)) * [50.0, 20.0, 5.0] + OFF, 0.25, 12345),
         (tower[rng.permutation(len(tower))] + OFF, 0.2, 200000),
         (synth.corridor_numpy(3_000_000, seed=synth.SEED0 + 9, kind="corridor", offset=True, towers=6), 0.1, 500000)]
for pts, voxel, chunk in cases:
    idx, mean, count, offs = ops.voxel_downsample(torch.from_numpy(np.ascontiguousarray(pts)).cuda(), voxel, chunk)
    ridx, rmean, rcount, roffs = ovx.voxel_down_sample_chunked(pts, voxel, chunk)
    assert np.array_equal(offs.cpu().numpy(), roffs)
    gi, gm, gc = ovx.canonical(idx.cpu().numpy(), mean.cpu().numpy(), count.cpu().numpy(), roffs)
    assert np.array_equal(gi, ridx) and np.array_equal(gc, rcount) and np.array_equal(gm.view(np.uint64), rmean.view(np.uint64))
print("staged scatter ok", len(cases))
'''
    env = dict(os.environ, PCH_VX_SCATTER="lds")
    r = subprocess.run([sys.executable, "-c", script, root], env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and "staged scatter ok 3" in r.stdout, r.stdout[-500:] + r.stderr[-2000:]
