"""pointcloudhookup_amd - MI355X (gfx950) implementation of the pointcloudhookup
ground-removal + tower-clustering hot path behind the reference's own call surface.

  csrc/                hand-written HIP kernels + the C ABI (include/pch_hip.h) -> libpch_hip.so
  _lib.py, ops.py      ctypes binding and tensor-level operators (no CPU fallback)
  pipeline.py          host orchestration of stages B-D on device tensors
  towers.py, obb.py    per-cluster boxes, tower acceptance, de-dup (host side of stage D)
  las.py               minimal LAS 1.x reader/writer (laspy is not a dependency)
  tiles.py             one-process-per-GPU tile sharding + RCCL label reconciliation
  ui/, utils/          drop-in modules with the reference's names and signatures
"""
__version__ = "0.1.0"
