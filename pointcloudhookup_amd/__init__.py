"""pointcloudhookup_amd - MI355X (gfx950) implementation of the pointcloudhookup
ground-removal + tower-clustering hot path behind the reference's own call surface.

  csrc/                hand-written HIP kernels + the C ABI (include/pch_hip.h) -> libpch_hip.so; the host-only
                       libpch_obbhost.so (include/pch_obbhost.h) for the box workers of stage D1
  _lib.py, ops.py      ctypes binding and tensor-level operators (no CPU fallback)
  pipeline.py          host orchestration of stages B-D0 on device tensors, tower acceptance / de-dup (D2-D3)
  obb.py               per-cluster oriented boxes (stage D1): qhull in a pool of worker processes fed from one shared,
                       HIP-registered buffer + native candidate search; opt-in fast mode with a device hull pre-filter
  resident.py          stage A -> stage B hand-over on the device (the intermediate LAS file is written, not read back)
  las.py               LAS 1.x I/O: python header parser / host writer + the library's native reader / writer
  tiles.py             one process per GPU: tile streams, x-tiles with halo, label reconciliation (RCCL / gloo)
  synth.py             the seeded synthetic clouds of SURVEY.md section 8d
  ui/, utils/          drop-in modules with the reference's names and signatures
"""
__version__ = "0.1.0"
