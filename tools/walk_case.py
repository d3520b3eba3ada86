"""One column under the centroid walk's microscope: python tools/walk_case.py  (tuning aid)
z ~ N(0, 0.05) for 1.5 M rows (the local frame's flat ground): exact blocks per column."""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pointcloudhookup_amd import _lib, ops  # noqa: E402

rng = np.random.default_rng(5)
n = 1_500_000
raw = np.column_stack([rng.uniform(0, 50, n), rng.uniform(0, 100, n), rng.normal(0, 0.05, n)]).astype(np.float32)
x = torch.from_numpy(raw).cuda()
out = ops.mean_seq_f32(x).cpu().numpy()
ref = raw.mean(axis=0)
print("equal", np.array_equal(out.view(np.uint32), ref.view(np.uint32)), out, ref)
L = _lib.lib()
ws = ops._workspace(L.pch_mean_seq_f32_ws_bytes(n), x.device)
st = ws[:128].view(torch.int32).cpu().numpy()
print("per column [batches, window misses, exact blocks, descents]:", st[:12].reshape(3, 4).tolist())
