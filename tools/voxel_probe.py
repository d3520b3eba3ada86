#!/usr/bin/env python3
"""Per-kernel times of the voxel stage (BASELINE config 2 shape).  python tools/voxel_probe.py [points] [voxel] [chunk]"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pointcloudhookup_amd import ops, synth          # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 10_000_000
voxel = float(sys.argv[2]) if len(sys.argv) > 2 else 0.2
chunk = int(sys.argv[3]) if len(sys.argv) > 3 else 500000
dev = torch.device("cuda:0")
xyz = synth.corridor_torch(n, seed=synth.SEED0 + 1, kind="corridor", offset=True, device=dev)
ops.voxel_downsample(xyz, voxel, chunk)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(5):
    out = ops.voxel_downsample(xyz, voxel, chunk)
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / 5
each = []
for _ in range(4):
    torch.cuda.synchronize()
    t1 = time.perf_counter()
    out = ops.voxel_downsample(xyz, voxel, chunk)
    torch.cuda.synchronize()
    each.append(round((time.perf_counter() - t1) * 1e3, 3))
print("single calls (ms):", each, "reserved GB:", round(torch.cuda.memory_reserved() / 1e9, 2))
print(f"{n} pts, voxel {voxel}, chunk {chunk}: {dt * 1e3:.3f} ms = {n / dt / 1e6:.0f} Mpts/s, {out[0].shape[0]} voxels")
ops.set_profiling(True)
for _ in range(3):
    ops.voxel_downsample(xyz, voxel, chunk)
torch.cuda.synchronize()
for name, ms, cnt in ops.get_profile():
    print(f"  {name:18s} {ms / cnt:8.4f} ms x{cnt}")
ops.set_profiling(False)
