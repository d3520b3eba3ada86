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


# ------------------------------------------------------------------------------------------------
# The same corridor, generated strip by strip (BASELINE config 4: one cloud spread over the ranks).
# Strip s covers x in [50 s, 50 (s+1)) and is generated from a seed of its own, so ANY rank can produce any
# strip and two ranks produce the same rows for the strip they share (a tile's halo).  File order = strips in
# ascending order, the rows of a strip shuffled - the order the module docstring describes (50 m strips of
# rho * W * 50 = 500 000 points, shuffled inside 500 000-point blocks), with the block edges on the strip edges.
STRIP = 50.0


def n_strips(n_total):
    return max(1, int(np.ceil(corridor_length(n_total) / STRIP - 1e-9)))


def corridor_strip_torch(n_total, s, seed=SEED0, kind="corridor", offset=False, device="cuda"):
    """float64 [m,3]: the rows of strip s of the n_total-point corridor, in file order."""
    import torch
    S = n_strips(n_total)
    gen = torch.Generator(device=device)
    gen.manual_seed(int(seed) * 1000003 + 7919 * int(s) + 1)
    f64 = torch.float64
    x0 = STRIP * s

    def U(lo, hi, k):
        return torch.rand(k, generator=gen, device=device, dtype=f64) * (hi - lo) + lo

    if kind == "uniform":
        m = n_total // S + (1 if s < n_total % S else 0)
        pts = torch.stack([U(x0, x0 + STRIP, m), U(0, W, m), U(0, 30, m)], dim=1)
    else:
        nt = n_total // 10
        ng = n_total - nt
        g = ng // S + (1 if s < ng % S else 0)
        parts = [torch.stack([U(x0, x0 + STRIP, g), U(0, W, g),
                              0.05 * torch.randn(g, generator=gen, device=device, dtype=f64)], dim=1)]
        T = n_towers(n_total)
        Lc = S * STRIP
        for t in range(T):                               # towers whose centre lies in this strip or next to it
            cx = (t + 0.5) * Lc / T
            if not (x0 - STRIP <= cx < x0 + 2 * STRIP):
                continue
            tg = torch.Generator(device=device)          # a tower's points are the same whichever strip asks
            tg.manual_seed(int(seed) * 1000003 + 104729 * (t + 1))
            k = nt // T + (1 if t < nt % T else 0)
            tp = torch.stack([cx + 2.5 * torch.randn(k, generator=tg, device=device, dtype=f64),
                              W / 2 + 2.5 * torch.randn(k, generator=tg, device=device, dtype=f64),
                              torch.clamp(22.0 + 9.0 * torch.randn(k, generator=tg, device=device, dtype=f64),
                                          0.5, 45.0)], dim=1)
            sid = torch.floor(tp[:, 0] / STRIP).clamp_(0, S - 1).to(torch.int64)
            parts.append(tp[sid == s])
        pts = torch.cat(parts, dim=0)
    perm = torch.randperm(pts.shape[0], generator=gen, device=device)
    pts = pts[perm]
    if offset:
        pts = pts + torch.tensor(GLOBAL_OFFSET, device=device, dtype=f64)
    return pts.contiguous()


def strips_of_rank(n_total, rank, world):
    """(first strip, one past the last strip) of rank's contiguous block of strips."""
    S = n_strips(n_total)
    per, extra = divmod(S, int(world))
    a = rank * per + min(rank, extra)
    return a, a + per + (1 if rank < extra else 0)


def corridor_tile_torch(n_total, rank, world, halo, seed=SEED0, kind="corridor", offset=False, device="cuda",
                        dtype=None):
    """This rank's part of the n_total-point strip corridor: its own strips (a consecutive file-order shard AND an
    x-tile) plus the rows of the two neighbouring strips within ``halo`` of its edges.  Nothing else of the cloud
    is generated.  Returns dict(points [n_t,3] (float64 or dtype) in file order, own bool [n_t] (= the rows
    own_range[0] <= i < own_range[1]: left halo | own | right halo), local_row int64
    [n_t] (row relative to the first OWNED row: negative in the left halo, >= n_own in the right halo), n_own,
    x_lo, x_hi (tile edges in the frame of ``points``), strips (a, b)).  Global rows = local_row + the exclusive
    prefix of n_own over the ranks (tiles.global_rows)."""
    import torch
    S = n_strips(n_total)
    a, b = strips_of_rank(n_total, rank, world)
    if b - a < 1:
        raise ValueError("more ranks than 50 m strips")
    if halo >= STRIP:
        raise ValueError("halo must be narrower than a strip (50 m)")
    ox = float(GLOBAL_OFFSET[0]) if offset else 0.0
    lo, hi = STRIP * a + ox, STRIP * b + ox
    parts, own, local = [], [], []
    if a > 0:                                            # left halo: the tail of strip a-1, its rows keep their order
        left = corridor_strip_torch(n_total, a - 1, seed, kind, offset, device)
        idx = torch.nonzero(left[:, 0] >= lo - halo).flatten()
        parts.append(left[idx])
        own.append(torch.zeros(idx.numel(), dtype=torch.bool, device=device))
        local.append(idx - left.shape[0])
        del left
    n_own = 0
    for s in range(a, b):
        st = corridor_strip_torch(n_total, s, seed, kind, offset, device)
        parts.append(st)
        own.append(torch.ones(st.shape[0], dtype=torch.bool, device=device))
        local.append(torch.arange(n_own, n_own + st.shape[0], device=device, dtype=torch.int64))
        n_own += st.shape[0]
    if b < S:                                            # right halo: the head of strip b
        right = corridor_strip_torch(n_total, b, seed, kind, offset, device)
        idx = torch.nonzero(right[:, 0] < hi + halo).flatten()
        parts.append(right[idx])
        own.append(torch.zeros(idx.numel(), dtype=torch.bool, device=device))
        local.append(idx + n_own)
        del right
    if dtype is not None:
        parts = [p.to(dtype) for p in parts]
    pts = torch.cat(parts, dim=0).contiguous()
    own = torch.cat(own)
    n_left = int(parts[0].shape[0]) if a > 0 else 0
    return dict(points=pts, own=own, own_range=(n_left, n_left + int(n_own)), local_row=torch.cat(local),
                n_own=int(n_own), x_lo=lo, x_hi=hi, strips=(a, b))
