"""LAS files assembled byte by byte from the specification (ASPRS LAS 1.4 R15, public header block and
point data record layouts) - independent of pointcloudhookup_amd/las.py, so that the native reader is
checked against the format, not against its own writer."""
import struct

import numpy as np

REC_LEN = {0: 20, 1: 28, 2: 26, 3: 34, 6: 30, 7: 36}


def build(path, XYZ, point_format=3, version=(1, 2), scales=(0.001, 0.001, 0.01), offsets=(437000.0, 3139000.0, 0.0),
          vlr_payloads=(), extra_bytes=0, pad_before_points=0, legacy_count_zero=False, seed=0):
    """Writes a LAS file and returns the dict of header values that were put in."""
    XYZ = np.asarray(XYZ, dtype="<i4").reshape(-1, 3)
    n = len(XYZ)
    vmaj, vmin = version
    hs = 375 if vmin >= 4 else (235 if vmin == 3 else 227)
    rl = REC_LEN[point_format] + extra_bytes
    vlrs = b""
    for k, payload in enumerate(vlr_payloads):                       # 54-byte VLR header + payload
        vlrs += struct.pack("<H16sHH32s", 0, b"test_vlr", 100 + k, len(payload), b"payload %d" % k) + payload
    otp = hs + len(vlrs) + pad_before_points
    h = bytearray(hs)
    h[0:4] = b"LASF"
    struct.pack_into("<HH", h, 4, 7, 0)                               # file source id, global encoding
    h[24], h[25] = vmaj, vmin
    h[26:58] = b"byte-built test file".ljust(32, b"\0")
    h[58:90] = b"tests/las_bytes.py".ljust(32, b"\0")
    struct.pack_into("<HH", h, 90, 120, 2024)
    struct.pack_into("<HII", h, 94, hs, otp, len(vlr_payloads))
    legacy = 0 if (legacy_count_zero or point_format >= 6) else n
    struct.pack_into("<BHI", h, 104, point_format, rl, legacy)
    struct.pack_into("<5I", h, 111, legacy, 0, 0, 0, 0)
    sc, of = np.asarray(scales, float), np.asarray(offsets, float)
    lo = XYZ.min(0) * sc + of if n else np.zeros(3)
    hi = XYZ.max(0) * sc + of if n else np.zeros(3)
    struct.pack_into("<12d", h, 131, sc[0], sc[1], sc[2], of[0], of[1], of[2], hi[0], lo[0], hi[1], lo[1], hi[2], lo[2])
    if vmin >= 4:
        struct.pack_into("<QI", h, 235, 0, 0)                         # EVLR start, count
        struct.pack_into("<Q", h, 247, n)
        struct.pack_into("<Q", h, 255, n)
    rng = np.random.default_rng(seed)
    recs = rng.integers(1, 255, size=(n, rl), dtype=np.uint8)         # every other field: arbitrary non-zero bytes
    recs[:, 0:12] = XYZ.view(np.uint8).reshape(n, 12)
    with open(path, "wb") as f:
        f.write(bytes(h))
        f.write(vlrs)
        f.write(b"\xAB" * pad_before_points)
        f.write(recs.tobytes())
    return dict(n=n, header_size=hs, offset_to_points=otp, record_length=rl, point_format=point_format,
                version=(vmaj, vmin), scales=sc, offsets=of, mins=lo, maxs=hi, num_vlrs=len(vlr_payloads))


def parse_xyz(path):
    """Spec-level reader (numpy + struct only) used to check files the library WROTE."""
    raw = open(path, "rb").read()
    assert raw[:4] == b"LASF"
    vmaj, vmin = raw[24], raw[25]
    hs, otp, nvlr = struct.unpack_from("<HII", raw, 94)
    fmt, rl, legacy = struct.unpack_from("<BHI", raw, 104)
    vals = struct.unpack_from("<12d", raw, 131)
    n = legacy
    if vmin >= 4:
        n64 = struct.unpack_from("<Q", raw, 247)[0]
        n = n64 or n
    body = np.frombuffer(raw, dtype=np.uint8, count=n * rl, offset=otp).reshape(n, rl)
    XYZ = body[:, :12].copy().view("<i4").reshape(n, 3)
    return dict(version=(vmaj, vmin), header_size=hs, offset_to_points=otp, num_vlrs=nvlr, point_format=fmt,
                record_length=rl, n=n, scales=np.array(vals[0:3]), offsets=np.array(vals[3:6]),
                maxs=np.array([vals[6], vals[8], vals[10]]), mins=np.array([vals[7], vals[9], vals[11]]),
                XYZ=XYZ, other_bytes=body[:, 12:])
