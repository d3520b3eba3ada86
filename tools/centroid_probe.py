"""Scratch probe: timing + walk statistics of the exact float32 centroid on synthetic tiles."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from pointcloudhookup_amd import ops, synth

n = int(sys.argv[1]) if len(sys.argv) > 1 else 100_000_000
for name, kw in [("local", dict(offset=False)), ("offset", dict(offset=True)), ("uniform", dict(kind="uniform"))]:
    raw = synth.corridor_torch(n, seed=synth.SEED0 + 2, device="cuda", dtype=torch.float32, **kw)
    ops.mean_seq_f32(raw)
    torch.cuda.synchronize()
    ops.set_profiling(True)
    t0 = time.perf_counter()
    c = ops.mean_seq_f32(raw)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    prof = ops.get_profile()
    ops.set_profiling(False)
    ws = ops._workspace(0, raw.device)
    stats = ws[:64].view(torch.int32).cpu().numpy().reshape(4, 4)[:3, :4]
    print(name, "ms=%.3f" % (dt * 1e3), prof, "stats[l2batches,tight,serial,descents] per col:", stats.tolist(), c.cpu().numpy())
    if n <= 20_000_000:
        ref = np.mean(raw.cpu().numpy(), axis=0)
        print("   exact:", np.array_equal(ref.view(np.uint32), c.cpu().numpy().view(np.uint32)))
    del raw
