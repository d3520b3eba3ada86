"""Statistics of the centroid walk per column (level-2 batches, exactly added blocks, descents) for the two frames.
python tools/walk_probe.py [points]"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pointcloudhookup_amd import _lib, ops, synth  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 100_000_000
L = _lib.lib()
for kind, offset in (("corridor", True), ("corridor", False), ("uniform", True), ("corridor", "centred")):
    x = synth.corridor_torch(n, seed=synth.SEED0 + 2, kind=kind, offset=offset is True, device="cuda",
                             dtype=torch.float32)
    if offset == "centred":                        # a cloud normalised to its own centre: every column is zero-mean
        x = x - x.double().mean(dim=0).float()
    ops.mean_seq_f32(x)
    torch.cuda.synchronize()
    t = time.perf_counter()
    out = ops.mean_seq_f32(x)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t) * 1e3
    ws = ops._workspace(L.pch_mean_seq_f32_ws_bytes(n), x.device)
    raw_st = ws[:128].view(torch.int32).cpu().numpy()
    st = raw_st[:16].reshape(4, 4)[:3]
    ex = raw_st[16:28].reshape(3, 4)
    stamps = raw_st[27:32]                          # tuning build (-DPCH_MS_STAMPS): y column, shader cycles / 64
    ops.set_profiling(True)
    ops.mean_seq_f32(x)
    torch.cuda.synchronize()
    prof = {k: round(ms, 3) for k, ms, c in ops.get_profile()}
    ops.set_profiling(False)
    print(f"{kind}, frame {'offset' if offset is True else offset or 'local'}: {dt:.2f} ms, mean {out.cpu().numpy()}  kernels {prof}")
    if stamps.any():
        print("  z column, ms_blocks_exact phases (shader cycles / 64): between blocks", raw_st[27], "stage", raw_st[28], "chains",
              raw_st[29], "look-ups", raw_st[30], "serial rest", raw_st[31])
    for c, name in enumerate("xyz"):
        print(f"  column {name}: batches {st[c, 0]}, exact blocks {st[c, 2]} (of them {st[c, 1]} because the candidate "
              f"window missed), descents {st[c, 3]}; exact path: {ex[c, 2]} calls, {ex[c, 0]} passes, "
              f"{ex[c, 1]} elements added one by one")
    del x
