"""pointcloudhookup_amd - MI355X (gfx950) implementation of the pointcloudhookup
ground-removal + tower-clustering hot path behind the reference's own call surface.

  csrc/                hand-written HIP kernels + the C ABI (include/pch_hip.h) -> libpch_hip.so
  _lib.py, ops.py      ctypes binding and tensor-level operators (no CPU fallback)
  pipeline.py          host orchestration of stages B-D0 on device tensors, tower acceptance / de-dup (D2-D3)
  obb.py               per-cluster oriented boxes (stage D1): qhull in worker processes + native candidate search;
                       opt-in fast mode with a device hull pre-filter
  las.py               LAS 1.x I/O: python header parser / host writer + the library's native reader / writer
  tiles.py             one process per GPU: tile streams, x-tiles with halo, label reconciliation (RCCL / gloo)
  synth.py             the seeded synthetic clouds of SURVEY.md section 8d
  ui/, utils/          drop-in modules with the reference's names and signatures
"""
__version__ = "0.1.0"
