"""Minimal LAS 1.0-1.4 reader / writer (laspy is not installed in the target image).

Only what the hot path touches is implemented: the public header block, the X/Y/Z int32
record fields, scales and offsets.  Everything else in a record is preserved on read as
opaque bytes and written as zeros for new files - the same result laspy gives for a
``LasData(header)`` whose only assigned dimensions are x, y, z
(reference: ui/import_PC.py:35-42,61-65, utils/tower_extraction.py:243-257).
The integer records are handed to the GPU untouched; the scaled float64 view
(``X*scale+offset``) is computed by ``ops.las_scale`` on the device.
"""
from __future__ import annotations

import os
import struct
from dataclasses import dataclass, field

import numpy as np

# point data record length per point format id (LAS 1.4 R15 table)
RECORD_LEN = {0: 20, 1: 28, 2: 26, 3: 34, 4: 57, 5: 63, 6: 30, 7: 36, 8: 38, 9: 59, 10: 67}
HEADER_SIZE = {(1, 0): 227, (1, 1): 227, (1, 2): 227, (1, 3): 235, (1, 4): 375}


@dataclass
class LasHeader:
    point_format: int = 3
    version: tuple = (1, 2)
    scales: np.ndarray = field(default_factory=lambda: np.array([0.01, 0.01, 0.01]))
    offsets: np.ndarray = field(default_factory=lambda: np.zeros(3))
    point_count: int = 0
    record_length: int = 0
    offset_to_points: int = 0
    header_size: int = 0
    mins: np.ndarray = field(default_factory=lambda: np.zeros(3))
    maxs: np.ndarray = field(default_factory=lambda: np.zeros(3))


@dataclass
class LasData:
    header: LasHeader
    XYZ: np.ndarray          # (n,3) int32 record integers

    def __len__(self):
        return int(self.XYZ.shape[0])

    def scaled(self, axis):
        """laspy's .x/.y/.z: float64 X*scale+offset (host; the device path uses ops.las_scale)."""
        return self.XYZ[:, axis].astype(np.float64) * self.header.scales[axis] + self.header.offsets[axis]

    @property
    def x(self):
        return self.scaled(0)

    @property
    def y(self):
        return self.scaled(1)

    @property
    def z(self):
        return self.scaled(2)


def read_header(path):
    with open(path, "rb") as f:
        h = f.read(375)
    if len(h) < 227 or h[:4] != b"LASF":
        raise ValueError(f"{path}: not a LAS file")
    vmaj, vmin = h[24], h[25]
    header_size, offset_to_points = struct.unpack_from("<HI", h, 94)
    fmt, rec_len, legacy_count = struct.unpack_from("<BHI", h, 104)
    fmt &= 0x3F                                            # bits 6/7 flag compression (LAZ)
    if h[104] & 0xC0:
        raise ValueError(f"{path}: compressed LAZ records are not supported")
    sx, sy, sz, ox, oy, oz, maxx, minx, maxy, miny, maxz, minz = struct.unpack_from("<12d", h, 131)
    count = legacy_count
    if (vmaj, vmin) >= (1, 4) and len(h) >= 255:
        count64 = struct.unpack_from("<Q", h, 247)[0]
        if count64:
            count = count64
    if fmt not in RECORD_LEN:
        raise ValueError(f"{path}: unsupported point format {fmt}")
    return LasHeader(point_format=fmt, version=(vmaj, vmin), scales=np.array([sx, sy, sz]),
                     offsets=np.array([ox, oy, oz]), point_count=int(count),
                     record_length=int(rec_len), offset_to_points=int(offset_to_points),
                     header_size=int(header_size), mins=np.array([minx, miny, minz]),
                     maxs=np.array([maxx, maxy, maxz]))


def read(path):
    """Reads the header and the X,Y,Z integers of every record (memory mapped, one copy)."""
    if not os.path.exists(path):
        raise FileNotFoundError(path)
    hdr = read_header(path)
    n, rl = hdr.point_count, hdr.record_length
    if n == 0:
        return LasData(hdr, np.zeros((0, 3), np.int32))
    rec = np.dtype({"names": ["XYZ"], "formats": [("<i4", 3)], "offsets": [0], "itemsize": rl})
    mm = np.memmap(path, dtype=rec, mode="r", offset=hdr.offset_to_points, shape=(n,))
    XYZ = np.ascontiguousarray(mm["XYZ"])
    del mm
    return LasData(hdr, XYZ)


def read_header_native(path):
    """The public header block as parsed by libpch_hip.so (pch_las_read_header); needs no GPU."""
    import ctypes as C
    from . import _lib
    if not os.path.exists(path):
        raise FileNotFoundError(path)
    h = _lib.LasHeaderC()
    _lib.check(_lib.lib().pch_las_read_header(os.fsencode(path), C.addressof(h)))
    return LasHeader(point_format=int(h.point_format), version=(int(h.version_major), int(h.version_minor)),
                     scales=np.array(h.scales), offsets=np.array(h.offsets), point_count=int(h.point_count),
                     record_length=int(h.record_length), offset_to_points=int(h.offset_to_points),
                     header_size=int(h.header_size), mins=np.array(h.mins), maxs=np.array(h.maxs))


def read_device(path, device):
    """Header + X,Y,Z ON THE DEVICE, read by the library itself (pch_las_read_xyz_i32): the file is memory
    mapped, its point records travel through a pinned double buffer and are decoded by a kernel while
    the next hop is copied.  Returns (LasHeader, int32 [n,3] device tensor)."""
    import torch
    from . import _lib, ops
    hdr = read_header_native(path)
    n = hdr.point_count
    out = torch.empty((n, 3), dtype=torch.int32, device=device)
    if n == 0:
        return hdr, out
    L = _lib.lib()
    with torch.cuda.device(device):
        ws = ops._workspace(L.pch_las_read_ws_bytes(), torch.device(device))
        _lib.check(L.pch_las_read_xyz_i32(os.fsencode(path), 0, n, out.data_ptr(), ws.data_ptr(), ws.numel(),
                                          torch.cuda.current_stream().cuda_stream))
    return hdr, out


def write_device(path, header, XYZ_dev):
    """Writes int32 [n,3] X,Y,Z that live on the device (pch_las_write_xyz_i32): header's point_format /
    version / scales / offsets, every other record field zero.  Returns the header written."""
    import ctypes as C
    import torch
    from . import _lib
    XYZ_dev = XYZ_dev.contiguous()
    if not XYZ_dev.is_cuda or XYZ_dev.dtype != torch.int32:
        raise TypeError("XYZ must be an int32 CUDA/HIP tensor")
    d = os.path.dirname(os.path.abspath(path))
    os.makedirs(d, exist_ok=True)
    h = _lib.LasHeaderC()
    h.version_major, h.version_minor = int(header.version[0]), int(header.version[1])
    h.point_format = int(header.point_format)
    for a in range(3):
        h.scales[a] = float(header.scales[a])
        h.offsets[a] = float(header.offsets[a])
    n = int(XYZ_dev.shape[0])
    with torch.cuda.device(XYZ_dev.device):
        _lib.check(_lib.lib().pch_las_write_xyz_i32(os.fsencode(path), C.addressof(h), XYZ_dev.data_ptr(), n,
                                                    torch.cuda.current_stream().cuda_stream))
    return read_header_native(path)


def write(path, header, XYZ):
    """Writes a LAS file with ``header``'s point_format / version / scales / offsets and the
    given int32 X,Y,Z; every other record field is zero."""
    XYZ = np.ascontiguousarray(XYZ, dtype=np.int32).reshape(-1, 3)
    n = XYZ.shape[0]
    fmt = int(header.point_format)
    ver = tuple(header.version)
    rl = RECORD_LEN[fmt]
    hs = HEADER_SIZE.get(ver, 227)
    sc, of = np.asarray(header.scales, float), np.asarray(header.offsets, float)
    if n:
        lo = XYZ.min(axis=0).astype(np.float64) * sc + of
        hi = XYZ.max(axis=0).astype(np.float64) * sc + of
    else:
        lo = hi = np.zeros(3)
    h = bytearray(hs)
    h[0:4] = b"LASF"
    h[24], h[25] = ver
    h[26:58] = b"pointcloudhookup_amd".ljust(32, b"\0")
    h[58:90] = b"pch-hip".ljust(32, b"\0")
    struct.pack_into("<HH", h, 90, 1, 2025)
    struct.pack_into("<HI", h, 94, hs, hs)                  # header size, offset to points
    struct.pack_into("<I", h, 100, 0)                       # no VLRs
    legacy = n if (n < 2**32 and fmt < 6) else 0
    struct.pack_into("<BHI", h, 104, fmt, rl, legacy)
    struct.pack_into("<5I", h, 111, legacy, 0, 0, 0, 0)
    struct.pack_into("<12d", h, 131, sc[0], sc[1], sc[2], of[0], of[1], of[2],
                     hi[0], lo[0], hi[1], lo[1], hi[2], lo[2])
    if ver >= (1, 4):
        struct.pack_into("<Q", h, 247, n)
        struct.pack_into("<Q", h, 255, n)
    d = os.path.dirname(os.path.abspath(path))
    os.makedirs(d, exist_ok=True)
    rec = np.dtype({"names": ["XYZ"], "formats": [("<i4", 3)], "offsets": [0], "itemsize": rl})
    with open(path, "wb") as f:
        f.write(bytes(h))
        step = 4_000_000
        for s in range(0, n, step):
            buf = np.zeros(min(step, n - s), dtype=rec)
            buf["XYZ"] = XYZ[s:s + step]
            f.write(buf.tobytes())
    out = LasHeader(point_format=fmt, version=ver, scales=sc.copy(), offsets=of.copy(), point_count=n,
                    record_length=rl, offset_to_points=hs, header_size=hs, mins=lo, maxs=hi)
    return out
