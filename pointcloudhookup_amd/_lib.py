"""ctypes binding of libpch_hip.so (C ABI declared in include/pch_hip.h).

The product path has no CPU fallback: if the library is missing or fails to load this
module raises, loudly.  Build it with ``python -c "import __graft_entry__ as g; g.build()"``
or ``make -C pointcloudhookup_amd/csrc``.
"""
from __future__ import annotations

import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libpch_hip.so")

PCH_OK = 0
ERR_NAMES = {-1: "PCH_ERR_ARG", -2: "PCH_ERR_WORKSPACE", -3: "PCH_ERR_HIP",
             -4: "PCH_ERR_RANGE", -5: "PCH_ERR_NODEVICE", -6: "PCH_ERR_TIMEOUT"}


class PchError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__(f"{ERR_NAMES.get(code, code)}: {msg}")
        self.code = code


_vp, _i64, _i32, _f64, _f32, _sz = C.c_void_p, C.c_int64, C.c_int32, C.c_double, C.c_float, C.c_size_t

_SIGS = {
    "pch_version": (C.c_int, []),
    "pch_last_error": (C.c_char_p, []),
    "pch_device_count": (C.c_int, []),
    "pch_voxel_downsample_ws_bytes": (_sz, [_i64, _i64]),
    "pch_voxel_downsample_f64": (C.c_int, [_vp, _i64, _f64, _i64, _vp, _vp, _vp, _vp, _vp, _vp, _sz, _vp]),
    "pch_las_read_header": (C.c_int, [C.c_char_p, _vp]),
    "pch_las_read_ws_bytes": (_sz, []),
    "pch_las_read_xyz_i32": (C.c_int, [C.c_char_p, _i64, _i64, _vp, _vp, _sz, _vp]),
    "pch_las_write_xyz_i32": (C.c_int, [C.c_char_p, _vp, _vp, _i64, _vp]),
    "pch_las_records_xyz_i32": (C.c_int, [_vp, _i64, _i32, _vp, _vp]),
    "pch_las_scale_i32_f64": (C.c_int, [_vp, _i64, _vp, _vp, _vp, _vp]),
    "pch_las_unscale_f64_i32": (C.c_int, [_vp, _i64, _vp, _vp, _vp, _vp]),
    "pch_cast_f64_f32": (C.c_int, [_vp, _i64, _vp, _vp]),
    "pch_mean_seq_f32_ws_bytes": (_sz, [_i64]),
    "pch_mean_seq_f32": (C.c_int, [_vp, _i64, _vp, _vp, _sz, _vp]),
    "pch_mean_seq_partial_f32": (C.c_int, [_vp, _i64, _vp, _i64, _vp, _vp, _i32, _vp, _sz, _vp]),
    "pch_mean_seq_serial_f32": (C.c_int, [_vp, _i64, _vp, _vp]),
    "pch_percentile_f32_ws_bytes": (_sz, [_i64]),
    "pch_percentile_f32": (C.c_int, [_vp, _i64, _i64, _vp, _f64, _vp, _vp, _sz, _vp]),
    "pch_select_hist_f32": (C.c_int, [_vp, _i64, _i64, _i32, C.c_uint32, _vp, _vp, _vp, _sz, _vp]),
    "pch_select_min_above_f32": (C.c_int, [_vp, _i64, _i64, C.c_uint32, _vp, _vp, _sz, _vp]),
    "pch_filter_gt_ws_bytes": (_sz, [_i64]),
    "pch_filter_gt_f32": (C.c_int, [_vp, _i64, _vp, _f32, _vp, _vp, _vp, _vp, _vp, _sz, _vp]),
    "pch_ground_filter_ws_bytes": (_sz, [_i64]),
    "pch_ground_filter_f32": (C.c_int, [_vp, _i64, _f64, _f32, _f32, _i64, _vp, _vp, _vp, _vp, _vp,
                                        _vp, _sz, _vp]),
    "pch_dbscan_set_sort_mode": (None, [C.c_int]),
    "pch_first_nonfinite_row_f32": (C.c_int, [_vp, _i64, _vp, _vp]),
    "pch_dbscan_relabel_i32": (C.c_int, [_vp, _i32, _i64, _vp, _vp, _sz, _vp]),
    "pch_dbscan_first_core_rows_i32": (C.c_int, [_i64, _vp, _vp, _sz, _vp]),
    "pch_dbscan_set_pair_counting": (None, [C.c_int]),
    "pch_dbscan_pair_stats": (C.c_int, [_i64, _vp, _vp, _sz, _vp]),
    "pch_dbscan_strip_pairs_i32": (C.c_int, [_i64, _f32, _f32, _i32, _vp, _vp, _vp, _sz, _vp]),
    "pch_strip_lattice_reps_ws_bytes": (_sz, [_i32]),
    "pch_strip_lattice_reps_f32": (C.c_int, [_vp, _vp, _vp, _vp, _i64, _i32, _vp, _f64, _i32, _vp, _vp, _vp, _vp, _sz, _vp]),
    "pch_dbscan_ws_bytes": (_sz, [_i64]),
    "pch_dbscan_f32": (C.c_int, [_vp, _i64, _f64, _i32, _i64, _vp, _vp, _vp, _vp, _vp, _sz, _vp]),
    "pch_segment_by_label_ws_bytes": (_sz, [_i64, _i32]),
    "pch_segment_by_label": (C.c_int, [_vp, _vp, _i64, _i32, _vp, _vp, _vp, _vp, _sz, _vp]),
    "pch_crop_aabb_ws_bytes": (_sz, [_i64]),
    "pch_crop_aabb_f64": (C.c_int, [_vp, _i64, _vp, _vp, _vp, _vp, _vp, _vp, _sz, _vp]),
    "pch_decimate_f64": (C.c_int, [_vp, _i64, _i64, C.c_uint64, _vp, _vp, _vp]),
    "pch_selftest_lookback_timeout": (C.c_int, [C.c_int, _vp, _sz, _vp]),
    "pch_obb_shell_ws_bytes": (_sz, [_i32]),
    "pch_obb_shell_f32": (C.c_int, [_vp, _vp, _vp, _i32, _i64, _vp, _vp, _sz, _vp]),
    "pch_obb_search_f64": (C.c_int, [_vp, _vp, _vp, _vp, _i32, _i32, _vp, _vp]),
    "pch_obb_min_boxes_f64": (C.c_int, [_vp, _vp, _vp, _vp, _i32, _i32, _i32, _vp, _vp, _vp]),
    "pch_tower_clusters_ws_bytes": (_sz, [_i64, _i64, _i32]),
    "pch_tower_clusters_f32": (C.c_int, [_vp, _i64, _f64, _f32, _f32, _i64, _f64, _i32, _i64, _vp, _vp, _vp,
                                         _vp, _vp, _vp, _i64, _i32, _vp, _vp, _sz, _vp]),
    "pch_set_profiling": (None, [C.c_int]),
    "pch_set_profiling_filter": (None, [C.c_char_p]),
    "pch_get_profile": (C.c_int, [C.c_int, _vp, _vp, _vp]),
}



class LasHeaderC(C.Structure):
    """PchLasHeader (include/pch_hip.h)."""
    _fields_ = [("version_major", C.c_uint8), ("version_minor", C.c_uint8), ("point_format", C.c_uint8),
                ("reserved0", C.c_uint8), ("header_size", C.c_uint16), ("record_length", C.c_uint16),
                ("offset_to_points", C.c_uint32), ("num_vlrs", C.c_uint32), ("point_count", C.c_uint64),
                ("scales", C.c_double * 3), ("offsets", C.c_double * 3), ("mins", C.c_double * 3),
                ("maxs", C.c_double * 3)]


class TowerClustersInfo(C.Structure):
    """PchTowerClusters (include/pch_hip.h)."""
    _fields_ = [("centroid", C.c_float * 3), ("base", C.c_float), ("threshold", C.c_float),
                ("used_fallback", C.c_int32), ("count_at_offset", C.c_int64), ("aabb", C.c_float * 6),
                ("count", C.c_int64), ("nclusters", C.c_int32), ("reserved", C.c_int32)]


_lib = None


def exported_symbols():
    """Every symbol include/pch_hip.h declares (used by the CPU symbol test)."""
    return sorted(_SIGS)


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise ImportError(
                f"{LIB_PATH} not found - the HIP extension is not built. There is no CPU "
                "fallback; run `make -C pointcloudhookup_amd/csrc` (needs hipcc).")
        # PyTorch-ROCm ships its own HIP runtime (libamdhip64) and must be the one that loads it: had this
        # library pulled in /opt/rocm's copy first, the process would hold two runtimes and ours would not
        # know torch's allocations (hipPointerGetAttributes fails on them)
        import torch  # noqa: F401
        handle = C.CDLL(LIB_PATH)
        for name, (res, args) in _SIGS.items():
            fn = getattr(handle, name)          # AttributeError if the .so is stale
            fn.restype = res
            fn.argtypes = args
        _lib = handle
    return _lib


def check(rc):
    if rc != PCH_OK:
        raise PchError(rc, lib().pch_last_error().decode("utf-8", "replace"))


PCH_ERR_TIMEOUT = -6


def check_count(count, what):
    """A count word the device wrote reads negative when a bounded device-side wait ran out of its budget
    (include/pch_hip.h, PCH_ERR_TIMEOUT): the outputs are undefined, so this raises instead of returning them."""
    count = int(count)
    if count < 0:
        raise PchError(PCH_ERR_TIMEOUT, f"{what}: a device-side look-back wait ran out of its budget "
                                        "(is the GPU shared with other processes?); outputs are undefined")
    return count


def get_profile(cap=128):
    """[(kernel name, total ms, launches)] of the last pch_* call of this thread."""
    names = (C.c_char * 48 * cap)()
    ms = (C.c_float * cap)()
    cnt = (C.c_int * cap)()
    n = lib().pch_get_profile(cap, C.cast(names, _vp), C.cast(ms, _vp), C.cast(cnt, _vp))
    return [(names[i].value.decode(), float(ms[i]), int(cnt[i])) for i in range(n)]
