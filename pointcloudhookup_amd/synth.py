"""Seeded synthetic clouds of SURVEY.md section 8(d) (the reference ships no data).

corridor: width W = 100 m along y, length L = N / (rho * W) with rho = 100 pts/m^2.
  ground (90 %): x~U(0,L), y~U(0,W), z~N(0, 0.05)
  towers (10 %): T = max(3, floor(L/300)) towers at x = (t+1/2) L/T, y = W/2, Gaussian
                 sigma = (2.5, 2.5, 9) m about z = 22 m, z clipped to [0.5, 45]
uniform : same box, x,y uniform, z~U(0,30)
Point order: stable sort by 50 m x-strip, then a shuffle inside every consecutive 500 000
point block (flight-line order + hash-map order of the voxel stage).
GLOBAL_OFFSET (EPSG:4547-like magnitudes) is added before any float32 cast when asked.
"""
from __future__ import annotations

import numpy as np

GLOBAL_OFFSET = np.array([437000.0, 3139000.0, 80.0])
W = 100.0
RHO = 100.0
SEED0 = 20250829


def corridor_length(n):
    return float(n) / (RHO * W)


def n_towers(n, exact=None):
    if exact is not None:
        return int(exact)
    return max(3, int(corridor_length(n) // 300))


def corridor_numpy(n, seed=SEED0, kind="corridor", offset=False, towers=None, order=True):
    """float64 (n,3) cloud on the host (parity tests, golden fixtures)."""
    rng = np.random.Generator(np.random.PCG64(seed))
    L = corridor_length(n)
    if kind == "uniform":
        pts = np.column_stack([rng.uniform(0, L, n), rng.uniform(0, W, n), rng.uniform(0, 30, n)])
    else:
        nt = n // 10
        ng = n - nt
        g = np.column_stack([rng.uniform(0, L, ng), rng.uniform(0, W, ng), rng.normal(0, 0.05, ng)])
        T = n_towers(n, towers)
        which = rng.integers(0, T, nt)
        cx = (which + 0.5) * L / T
        t = np.column_stack([rng.normal(cx, 2.5), rng.normal(W / 2, 2.5, nt),
                             np.clip(rng.normal(22.0, 9.0, nt), 0.5, 45.0)])
        pts = np.vstack([g, t])
    if order:
        strip = np.floor(pts[:, 0] / 50.0).astype(np.int64)
        pts = pts[np.argsort(strip, kind="stable")]
        for s in range(0, n, 500000):
            e = min(s + 500000, n)
            pts[s:e] = pts[s:e][rng.permutation(e - s)]
    if offset:
        pts = pts + GLOBAL_OFFSET
    return pts


def corridor_torch(n, seed=SEED0, kind="corridor", offset=False, towers=None, device="cuda",
                   dtype=None):
    """Same distribution generated on the device (bench sizes: 10^7..10^8 points).
    Returns float64 [n,3] (or ``dtype``).  The random stream differs from corridor_numpy."""
    import torch
    gen = torch.Generator(device=device)
    gen.manual_seed(int(seed))
    L = corridor_length(n)
    f64 = torch.float64

    def U(lo, hi, k):
        return torch.rand(k, generator=gen, device=device, dtype=f64) * (hi - lo) + lo

    def N(k):
        return torch.randn(k, generator=gen, device=device, dtype=f64)

    if kind == "uniform":
        x, y, z = U(0, L, n), U(0, W, n), U(0, 30, n)
    else:
        nt = n // 10
        ng = n - nt
        T = n_towers(n, towers)
        which = torch.randint(0, T, (nt,), generator=gen, device=device)
        cx = (which.to(f64) + 0.5) * (L / T)
        x = torch.cat([U(0, L, ng), cx + 2.5 * N(nt)])
        y = torch.cat([U(0, W, ng), W / 2 + 2.5 * N(nt)])
        z = torch.cat([0.05 * N(ng), torch.clamp(22.0 + 9.0 * N(nt), 0.5, 45.0)])
    # 50 m strips (stable), then shuffle inside consecutive 500k blocks: one sort on a
    # composite key reproduces both steps
    strip = torch.floor(x / 50.0).to(torch.int64)
    order = torch.sort(strip, stable=True).indices
    pos = torch.arange(n, device=device, dtype=torch.int64)
    key = (pos // 500000).to(f64) + torch.rand(n, generator=gen, device=device, dtype=f64) * 0.999
    order = order[torch.sort(key).indices]
    pts = torch.stack([x[order], y[order], z[order]], dim=1)
    if offset:
        pts = pts + torch.tensor(GLOBAL_OFFSET, device=device, dtype=f64)
    if dtype is not None:
        pts = pts.to(dtype)
    return pts.contiguous()
