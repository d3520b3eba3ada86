"""Where the time of the tower table goes (stage D1-D3 on the clusters of one 100 M-point tile).
python tools/obb_probe.py [points]"""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pointcloudhookup_amd import obb, ops, pipeline, synth  # noqa: E402

N = int(sys.argv[1]) if len(sys.argv) > 1 else 100_000_000
raw = synth.corridor_torch(N, seed=synth.SEED0 + 2, kind="corridor", offset=True, device="cuda", dtype=torch.float32)
cl = pipeline.cluster_points(raw)
K = int(cl["nclusters"])
print("clusters", K, "kept", int(cl["ground"]["count"]), flush=True)


def timed(label, fn, reps=3):
    for r in range(reps):
        torch.cuda.synchronize()
        t = time.perf_counter()
        out = fn()
        torch.cuda.synchronize()
        print(f"{label} run {r}: {(time.perf_counter() - t) * 1e3:.1f} ms", flush=True)
    return out


ws = [int(v) for v in os.environ.get("PCH_PROBE_WORKERS", "8,16,24,32,48").split(",")]
REPS = int(os.environ.get("PCH_PROBE_REPS", "3"))


def stream(depth, tiles=6):
    """tables of tile k beside the device work of tile k+1"""
    jobs, t0 = [], time.perf_counter()
    for _ in range(tiles):
        c2 = pipeline.cluster_points(raw)
        jobs.append(pipeline.tower_table_async(c2))
        del c2
        while len(jobs) > depth:
            jobs.pop(0).result()
    while jobs:
        jobs.pop(0).result()
    return (time.perf_counter() - t0) / tiles * 1e3


for w in ws:                                   # ascending: the pool only grows
    os.environ["PCH_OBB_WORKERS"] = str(w)
    obb.pool(w)
    time.sleep(2.5)                            # the new workers import scipy
    tm = {}
    timed(f"exact, {w} workers", lambda: pipeline.tower_table(cl, timings=tm), reps=REPS)
    print("   split:", {k: (round(v, 2) if isinstance(v, float) else v) for k, v in tm.items()}, flush=True)
    for depth in (1, 2):
        print(f"stream, {w} workers, {depth} tables in flight: {stream(depth):.1f} ms per tile", flush=True)
timed("fast", lambda: pipeline.tower_table(cl, obb_mode="fast"))
