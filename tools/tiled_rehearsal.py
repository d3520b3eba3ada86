#!/usr/bin/env python3
"""BASELINE config 4 rehearsal: ONE corridor cut into x-tiles (2*eps halo), one rank per tile, clustered as a
whole (tiles.cluster_tiled) and checked against a single-GPU DBSCAN of the whole filtered cloud.

  python -m torch.distributed.run --nproc-per-node R --master-addr 127.0.0.1 tools/tiled_rehearsal.py [points]

On an 8-GPU node every rank takes its own GPU and the exchange runs over RCCL (backend nccl).  On the one-GPU
box RCCL refuses two ranks on one device, so the ranks share cuda:0 and exchange over gloo
(PCH_DIST_BACKEND=gloo PCH_BENCH_SINGLE_DEVICE=1): same code path except for where the exchanged tensors live.
Every rank generates the same seeded cloud and filters it the same way (shared centroid and threshold - the
part of config 4 that is not distributed yet), then keeps only its tile."""
import json
import os
import sys
import time

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
gf = ops.ground_filter(raw, want_index=True)                 # same result on every rank
kept = raw[gf["index"].long()].contiguous()                  # UNcentred rows: one frame for all tiles
del raw
nf = kept.shape[0]
L = synth.corridor_length(n)
edges = tiles.tile_edges(float(kept[:, 0].min()), float(kept[:, 0].max()), world)
take, own = tiles.tile_select(kept[:, 0], edges, rank, 2 * EPS)
rows = torch.nonzero(take).flatten()
pts = kept[rows].contiguous()
own_l = own[rows]


def barrier():
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()


labels, K = tiles.cluster_tiled(pts, rows, own_l, edges[rank], edges[rank + 1], EPS, MS)       # warm-up
barrier()
t0 = time.perf_counter()
steps = 3
for _ in range(steps):
    labels, K = tiles.cluster_tiled(pts, rows, own_l, edges[rank], edges[rank + 1], EPS, MS)
barrier()
dt = (time.perf_counter() - t0) / steps
# the check: one DBSCAN over the whole filtered cloud on this rank's GPU
want, _, k1 = ops.dbscan(kept, EPS, MS, 0)
ok = bool(K == k1 and torch.equal(labels[own_l], want[rows[own_l]]))
flag = torch.tensor([1 if ok else 0], dtype=torch.int64, device=dev if world > 1 and dist.get_backend() == "nccl" else "cpu")
tmax = torch.tensor([dt], dtype=torch.float64, device=flag.device)
if world > 1:
    dist.all_reduce(flag, op=dist.ReduceOp.MIN)
    dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
if rank == 0:
    cut = 0
    side = torch.bucketize(kept[:, 0].cpu(), torch.tensor(edges[1:-1], dtype=torch.float32))
    for c in range(k1):
        if len(torch.unique(side[(want == c).cpu()])) > 1:
            cut += 1
    print(json.dumps({"config": "BASELINE config 4 rehearsal: x-tiles + 2*eps halo, global DBSCAN, label reconciliation",
                      "points": n, "filtered_points": int(nf), "ranks": world,
                      "backend": dist.get_backend() if world > 1 else "none",
                      "devices": "one GPU shared by all ranks" if os.environ.get("PCH_BENCH_SINGLE_DEVICE") else "one GPU per rank",
                      "clusters": int(K), "clusters_cut_by_a_tile_edge": cut,
                      "labels_equal_single_gpu_dbscan_on_every_rank": bool(int(flag.item())),
                      "ms_per_tiled_step_max_over_ranks": round(float(tmax.item()) * 1e3, 3),
                      "tile_points_rank0": int(pts.shape[0])}))
if world > 1:
    dist.destroy_process_group()
