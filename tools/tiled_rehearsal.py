#!/usr/bin/env python3
"""BASELINE config 4 rehearsal: ONE corridor cut into x-tiles (2*eps halo), one rank per tile, clustered as a
whole (tiles.cluster_tiled) and checked against a single-GPU DBSCAN of the whole filtered cloud.

  python -m torch.distributed.run --nproc-per-node R --master-addr 127.0.0.1 tools/tiled_rehearsal.py [points]

On an 8-GPU node every rank takes its own GPU and the exchange runs over RCCL (backend nccl).  On the one-GPU
box RCCL refuses two ranks on one device, so the ranks share cuda:0 and exchange over gloo
(PCH_DIST_BACKEND=gloo PCH_BENCH_SINGLE_DEVICE=1): same code path except for where the exchanged tensors live.
Every rank generates the same seeded cloud (the rehearsal has no tiled file reader), computes the one value that
cannot be sharded by x - numpy's sequential float32 centroid - and keeps only its tile + halo.  The percentile
threshold is then found ACROSS the ranks (tiles.shared_percentile), every rank filters its tile with it
(ops.filter_gt) and the survivors are clustered as one cloud (tiles.cluster_tiled)."""
import json
import os
import sys
import time

import numpy as np
import torch
import torch.distributed as dist

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pointcloudhookup_amd import ops, synth, tiles          # noqa: E402

import faulthandler
faulthandler.dump_traceback_later(150, exit=True)          # a rank that hangs says where

EPS, MS = 8.0, 80
n = int(sys.argv[1]) if len(sys.argv) > 1 else 30_000_000
rank, world, local = tiles.init_from_env()
if os.environ.get("PCH_BENCH_SINGLE_DEVICE"):
    local = 0
dev = torch.device(f"cuda:{local}")
torch.cuda.set_device(dev)
raw = synth.corridor_torch(n, seed=synth.SEED0 + 4, kind="corridor", offset=False, device=dev, dtype=torch.float32)
torch.cuda.synchronize()                                     # (a hang in the generator would show here, not in the filter)
# the one value that cannot be sharded by x: numpy's SEQUENTIAL float32 centroid follows the file order
centroid = ops.mean_seq_f32(raw).cpu().numpy()
edges = tiles.tile_edges(float(raw[:, 0].min()), float(raw[:, 0].max()), world)
take, own = tiles.tile_select(raw[:, 0], edges, rank, 2 * EPS)
tile_rows = torch.nonzero(take).flatten()                    # global rows of this rank's tile, halo included
tile = raw[tile_rows].contiguous()
own_t = own[tile_rows]
# single-GPU answer for the check at the end (whole cloud on this rank's GPU), then the whole cloud is dropped
gf_all = ops.ground_filter(raw, want_index=True)
want_lab, _, k1 = ops.dbscan(gf_all["points"], EPS, MS, 0)       # the centred float32 points, as the reference clusters them
want_full = torch.full((n,), -2, dtype=torch.int32, device=dev)
want_full[gf_all["index"].long()] = want_lab
thr_all = np.float32(gf_all["threshold"])
nf = int(gf_all["count"])
del raw, gf_all, want_lab


def tiled_step():
    # shared threshold: percentile over every rank's OWN points (three all-reduced histogram passes)
    base = tiles.shared_percentile(tile[own_t][:, 2].contiguous(), 25.0, sub=centroid[2])
    thr = np.float32(base + np.float32(3.0))
    kept = ops.filter_gt(tile, centroid, thr, want_index=True)            # this tile's survivors, halo included
    loc = kept["index"].long()
    rows = tile_rows[loc]                                                  # global rows, ascending
    cx = float(centroid[0])                                                # the points are centred: so are the edges
    labels, K = tiles.cluster_tiled(kept["points"], rows, own_t[loc], edges[rank] - cx, edges[rank + 1] - cx, EPS, MS)
    return thr, labels, K, rows, own_t[loc]


def barrier():
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()


thr, labels, K, rows, own_l = tiled_step()                    # warm-up
barrier()
t0 = time.perf_counter()
steps = 3
for _ in range(steps):
    thr, labels, K, rows, own_l = tiled_step()
barrier()
dt = (time.perf_counter() - t0) / steps
# the check: threshold and labels of the owned points against the single-GPU run over the whole cloud
ok = bool(thr.view(np.uint32) == thr_all.view(np.uint32) and K == k1 and
          torch.equal(labels[own_l], want_full[rows[own_l]]))
flag = torch.tensor([1 if ok else 0], dtype=torch.int64, device=dev if world > 1 and dist.get_backend() == "nccl" else "cpu")
tmax = torch.tensor([dt], dtype=torch.float64, device=flag.device)
if world > 1:
    dist.all_reduce(flag, op=dist.ReduceOp.MIN)
    dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
if rank == 0:
    cut = -1                                                   # (needs the whole cloud: counted in the unit tests instead)
    print(json.dumps({"config": "BASELINE config 4 rehearsal: x-tiles + 2*eps halo, global DBSCAN, label reconciliation",
                      "points": n, "filtered_points": int(nf), "ranks": world,
                      "backend": dist.get_backend() if world > 1 else "none",
                      "devices": "one GPU shared by all ranks" if os.environ.get("PCH_BENCH_SINGLE_DEVICE") else "one GPU per rank",
                      "clusters": int(K),
                      "shared_threshold_and_labels_equal_single_gpu_run_on_every_rank": bool(int(flag.item())),
                      "ms_per_tiled_step_max_over_ranks": round(float(tmax.item()) * 1e3, 3),
                      "tile_points_rank0": int(tile.shape[0])}))
if world > 1:
    dist.destroy_process_group()
