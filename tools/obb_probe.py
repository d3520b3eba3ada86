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


for w in (16, 8):
    os.environ["PCH_OBB_WORKERS"] = str(w)
    timed(f"exact, native search, {w} workers", lambda: pipeline.tower_table(cl))
os.environ["PCH_OBB_SEARCH"] = "python"
timed("exact, python loop, 8 workers", lambda: pipeline.tower_table(cl), reps=2)
os.environ["PCH_OBB_SEARCH"] = "native"
timed("fast", lambda: pipeline.tower_table(cl, obb_mode="fast"))

# pieces of the exact mode
offsets = cl["offsets"].cpu().numpy()
t = time.perf_counter()
rows = cl["perm"][: int(offsets[K])].long()
host = cl["ground"]["points"].index_select(0, rows).cpu().numpy()
print(f"gather + D2H of {len(host)} points: {(time.perf_counter() - t) * 1e3:.1f} ms")
parts = [host[offsets[i]:offsets[i + 1]] for i in range(K)]
t = time.perf_counter()
first = obb._per_cluster(parts, "__hull__", 8)
print(f"qhull + candidates in 8 workers: {(time.perf_counter() - t) * 1e3:.1f} ms")
ok = [h for h, e in first if e is None]
vo = np.cumsum([0] + [len(h[0]) for h in ok]); ao = np.cumsum([0] + [len(h[1]) for h in ok])
V = np.concatenate([h[0] for h in ok]); A = np.concatenate([h[1].reshape(-1, 2) for h in ok])
t = time.perf_counter()
best, vol = ops.obb_search(V, vo, A, ao)
print(f"native search, {len(ok)} hulls, {len(A)} candidates: {(time.perf_counter() - t) * 1e3:.1f} ms; "
      f"hulls with more than one candidate within 1e-9 of the best: "
      f"{sum((vol[ao[j]:ao[j + 1]] <= vol[ao[j] + best[j]] * (1 + 1e-9)).sum() > 1 for j in range(len(ok)))}")
t = time.perf_counter()
for h, b in zip(ok, best):
    obb.bounds_from_candidates(h[0], h[1], "unsorted", np.array([int(b)]))
print(f"winner evaluation in python: {(time.perf_counter() - t) * 1e3:.1f} ms")
# pieces of the fast mode
t = time.perf_counter()
keep = ops.obb_shell(cl["ground"]["points"], cl["perm"], cl["offsets"], K)
torch.cuda.synchronize()
print(f"obb_shell: {(time.perf_counter() - t) * 1e3:.2f} ms, kept {int(keep.sum())} of {len(keep)}")
