#!/usr/bin/env python3
"""Randomised parity stress (not part of the test suite): random sizes, parameters and
distributions through the C ABI against the oracle.  python tools/fuzz_parity.py [seconds] [seed]"""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import dbscan as odb                      # noqa: E402
from oracle import ground_filter as ogf               # noqa: E402
from oracle import voxel as ovx                       # noqa: E402
from pointcloudhookup_amd import ops                  # noqa: E402
from pointcloudhookup_amd._lib import PchError        # noqa: E402

budget = float(sys.argv[1]) if len(sys.argv) > 1 else 120.0
seed = int(sys.argv[2]) if len(sys.argv) > 2 else 1
rng = np.random.default_rng(seed)
dev = torch.device("cuda:0")
t_end = time.time() + budget
stats = dict(filter=0, dbscan=0, voxel=0)


def cloud(n):
    kind = rng.integers(0, 5)
    off = np.array([437000.0, 3139000.0, 80.0]) * rng.integers(0, 2)
    if kind == 0:       # towers + ground
        k = int(rng.integers(1, 6))
        parts = [rng.normal([rng.uniform(0, 400), rng.uniform(0, 100), 22], [2.5, 2.5, 9], (n // (2 * k), 3))
                 for _ in range(k)]
        m = n - sum(len(p) for p in parts)
        parts.append(np.column_stack([rng.uniform(0, 400, m), rng.uniform(0, 100, m), rng.normal(0, 0.05, m)]))
        X = np.vstack(parts)
    elif kind == 1:     # uniform box
        X = rng.uniform(0, 1, (n, 3)) * np.array([rng.uniform(10, 500), rng.uniform(10, 100), rng.uniform(1, 40)])
    elif kind == 2:     # heavy tails / mixed magnitudes
        X = rng.standard_cauchy((n, 3)) * rng.uniform(0.1, 100)
    elif kind == 3:     # quantised (ties)
        X = np.round(rng.normal(0, 20, (n, 3)) * 4) / 4
    else:               # zero-mean, tiny and huge columns
        X = rng.normal(0, 1, (n, 3)) * np.array([1e-3, 1.0, 1e4])
    X = X[rng.permutation(len(X))] + off
    return X.astype(np.float32)


it = 0
t_say = time.time() + 30
while time.time() < t_end:
    it += 1
    if time.time() > t_say:                        # a line every half minute: long runs must not look hung
        print("fuzz", it, stats, flush=True)
        t_say = time.time() + 30
    n = int(rng.choice([1, 2, 63, 65, 1000, 1024, 5000, 16385, 40000, 100003, 262145, 700001]))
    raw = cloud(n)
    d = torch.from_numpy(raw).to(dev)
    # centroid + percentile + filter
    ref = ogf.ground_filter(raw)
    got = ops.ground_filter(d)
    assert np.array_equal(got["centroid"].view(np.uint32), ref["centroid"].view(np.uint32)), ("centroid", n, it)
    assert np.float32(got["base"]).view(np.uint32) == ref["base"].view(np.uint32), ("base", n, it)
    assert got["count"] == len(ref["filtered"]) and got["used_fallback"] == ref["used_fallback"], ("count", n, it)
    assert np.array_equal(got["points"].cpu().numpy().view(np.uint32), ref["filtered"].view(np.uint32)), ("points", n, it)
    stats["filter"] += 1
    # voxel stage: random voxel size / chunking on the float64 view of the same cloud (quantised like LAS in half of
    # the cases, which puts coordinates exactly on voxel faces)
    if it % 3 == 0 and n >= 2:
        p64 = raw.astype(np.float64)
        if rng.integers(0, 2):
            p64 = np.round(p64, int(rng.integers(1, 4)))
        voxel = float(rng.choice([0.05, 0.1, 0.2, 0.25, 1.0, 7.3]))
        vchunk = int(rng.choice([0, 1000, 4096, 4097, 50000, 500000]))
        ext = (p64.max(axis=0) - p64.min(axis=0)).max()
        if np.isfinite(p64).all() and ext / voxel < 2.0e6:
            ridx, rmean, rcount, roffs = ovx.voxel_down_sample_chunked(p64, voxel, vchunk if vchunk else len(p64))
            vi, vm, vc, vo = ops.voxel_downsample(torch.from_numpy(p64).to(dev), voxel, vchunk)
            assert np.array_equal(vo.cpu().numpy(), roffs), ("voxel offsets", n, voxel, vchunk, it)
            # set equality per chunk: the library's order inside a chunk is its own (Open3D's is unordered_map order)
            ci, cm, cc = ovx.canonical(vi.cpu().numpy(), vm.cpu().numpy(), vc.cpu().numpy(), roffs)
            assert np.array_equal(ci, ridx), ("voxel idx", n, voxel, vchunk, it)
            assert np.array_equal(cc, rcount), ("voxel count", n, voxel, vchunk, it)
            assert np.array_equal(cm.view(np.uint64), rmean.view(np.uint64)), ("voxel mean", n, voxel, vchunk, it)
            stats["voxel"] += 1
    # clustering on a bounded subset (the C oracle is all-pairs)
    m = min(len(ref["filtered"]), int(rng.choice([500, 3000, 12000])))
    if m >= 2:
        pts = np.ascontiguousarray(ref["filtered"][:m])
        eps = float(rng.choice([0.5, 2.0, 8.0, 30.0]))
        ms = int(rng.choice([1, 3, 10, 80]))
        chunk = int(rng.choice([0, 777, 5000, 50000]))
        want = odb.dbscan_chunked(pts, eps, ms, chunk, fit="c")
        try:
            lab, _, k = ops.dbscan(torch.from_numpy(pts).to(dev), eps, ms, chunk)
        except PchError as e:                  # documented limit: extent/eps beyond the 64-bit cell key
            assert e.code == -4, e
            stats["range"] = stats.get("range", 0) + 1
            continue
        assert np.array_equal(lab.cpu().numpy(), want), ("dbscan", n, m, eps, ms, chunk, it)
        assert k == (want.max() + 1 if (want >= 0).any() else 0)
        stats["dbscan"] += 1
        g2, l2, k2, perm, offs, st = ops.tower_clusters(torch.from_numpy(pts + ref["centroid"]).to(dev), eps, ms,
                                                        chunk if chunk else 50000)
        assert g2["count"] <= m
        # the fused call (cluster boxes gathered by the label kernels) against the separate stage calls
        if g2["count"] >= 2:
            l3, _, k3 = ops.dbscan(g2["points"], eps, ms, chunk if chunk else 50000, aabb=g2["aabb"])
            assert k3 == k2 and torch.equal(l3, l2), ("fused labels", n, m, it)
            p3, o3, s3 = ops.segment_by_label(l3, g2["points"], k3)
            assert torch.equal(p3, perm) and torch.equal(o3, offs), ("fused grouping", n, m, it)
            assert np.array_equal(s3.cpu().numpy().view(np.uint32), st.cpu().numpy().view(np.uint32)), ("fused boxes", n, m, it)
            stats["fused"] = stats.get("fused", 0) + 1
print("fuzz ok", stats, "iterations", it)
