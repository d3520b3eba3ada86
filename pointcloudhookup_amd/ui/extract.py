"""Drop-in for the reference's ``ui/extract.py``: tower dicts -> wire boxes for the viewer.

Public names, argument order, defaults and return shapes follow
/root/reference/ui/extract.py:7-452.  This stage has no kernel work (SURVEY.md section 0,
fact 4); the only O(N) step is handing the cloud back as a float64 (N,3) array, which here
is decoded from the LAS integers on the GPU.
"""
from __future__ import annotations

import os

import numpy as np

_CORNER_BITS = ((0, 0, 0), (1, 0, 0), (1, 1, 0), (0, 1, 0), (0, 0, 1), (1, 0, 1), (1, 1, 1), (0, 1, 1))
_EDGE_ENDS = (0, 1, 1, 2, 2, 3, 3, 0, 4, 5, 5, 6, 6, 7, 7, 4, 0, 4, 1, 5, 2, 6, 3, 7)

_KX = ("x_left_factor", "x_right_factor", "y_down_factor", "y_up_factor", "z_down_factor", "z_up_factor")


def _kx(values):
    return dict(zip(_KX, values))


BBOX_PRESETS = {
    "kuangxuan_original": {"method": "kuangxuan", "params": _kx((1.0, 1.67, 0.5, 1.0, 1.0, 2.0))},
    "kuangxuan_conservative": {"method": "kuangxuan", "params": _kx((0.8, 1.2, 0.4, 0.8, 0.5, 1.5))},
    "kuangxuan_aggressive": {"method": "kuangxuan", "params": _kx((1.5, 2.0, 0.8, 1.5, 1.5, 3.0))},
    "symmetric_moderate": {"method": "symmetric", "params": {"x_scale": 2.0, "y_scale": 2.0, "z_scale": 1.5}},
    "symmetric_large": {"method": "symmetric", "params": {"x_scale": 3.0, "y_scale": 3.0, "z_scale": 2.0}},
}


def get_bbox_preset(preset_name: str):
    preset = BBOX_PRESETS.get(preset_name, BBOX_PRESETS["kuangxuan_original"])
    return preset["method"], preset["params"]


def create_bbox_using_kuangxuan_method(center, width, height,
                                       x_left_factor=1.0, x_right_factor=1.67,
                                       y_down_factor=0.5, y_up_factor=1.0,
                                       z_down_factor=1.0, z_up_factor=2.0):
    """Asymmetric axis-aligned box around a tower centre -> (min_coords, max_coords)."""
    cx, cy, cz = center
    lo = np.array([cx - width * x_left_factor, cy - width * y_down_factor, cz - height * z_down_factor])
    hi = np.array([cx + width * x_right_factor, cy + width * y_up_factor, cz + height * z_up_factor])
    return lo, hi


def create_bbox_lineset_from_bounds(min_coords, max_coords, color=(1.0, 0.0, 0.0)):
    """12 box edges as 24 end points, ((24,3) float64, color)."""
    span = (min_coords, max_coords)
    corners = [[span[b][axis] for axis, b in enumerate(bits)] for bits in _CORNER_BITS]
    return np.array([corners[i] for i in _EDGE_ENDS]), color


def _tower_size(tower_info):
    ext = np.array(tower_info['extent'])
    return max(ext[0], ext[1]), ext[2]           # width = larger horizontal extent, height = extent[2]


def _bounds_for(center, width, height, bbox_method, bbox_params):
    if bbox_method == "kuangxuan":
        return create_bbox_using_kuangxuan_method(center, width, height, **bbox_params)
    if bbox_method == "symmetric":
        half = np.array([width * bbox_params.get("x_scale", 2.0), width * bbox_params.get("y_scale", 2.0),
                         height * bbox_params.get("z_scale", 1.5)]) / 2
        return center - half, center + half
    raise ValueError(f"未知的包围盒方法: {bbox_method}")


def _read_cloud(las_path):
    if not os.path.exists(las_path):
        raise FileNotFoundError(f"未找到文件: {las_path}")
    import torch
    from .. import las as _las
    from .. import ops
    dev = os.environ.get("PCH_DEVICE", "cuda:0")
    hdr, XYZ = _las.read_device(las_path, dev)
    if XYZ.shape[0] == 0:
        return np.zeros((0, 3))
    return ops.las_scale(XYZ, hdr.scales, hdr.offsets).cpu().numpy()


def extract_and_visualize_towers_kuangxuan(las_path: str, tower_obbs: list,
                                           bbox_method: str = "kuangxuan",
                                           bbox_params: dict = None,
                                           line_color: tuple = (1.0, 0.0, 0.0)):
    """Returns (full cloud (N,3) float64, [(24x3 line points, color), ...])."""
    if bbox_params is None:
        bbox_params = dict(BBOX_PRESETS["kuangxuan_original"]["params"])
    full_pcd = _read_cloud(las_path)
    tower_geometries = []
    print(f"🔧 开始处理 {len(tower_obbs)} 个杆塔，使用方法: {bbox_method}")
    print(f"📊 包围盒参数: {bbox_params}")
    for i, tower_info in enumerate(tower_obbs):
        try:
            center = tower_info['center']
            width, height = _tower_size(tower_info)
            lo, hi = _bounds_for(center, width, height, bbox_method, bbox_params)
            if bbox_method == "kuangxuan":
                sx, sy, sz = hi - lo
                print(f"📏 杆塔{i}: 原始宽度{width:.1f}m, 高度{height:.1f}m")
                print(f"📐 杆塔{i}: kuangxuan方法 -> X:{sx:.1f}m, Y:{sy:.1f}m, Z:{sz:.1f}m")
            tower_geometries.append(create_bbox_lineset_from_bounds(lo, hi, line_color))
            print(f"✅ 杆塔{i}处理成功，中心：{center}")
        except Exception as e:
            print(f"⚠️ 杆塔{i}可视化失败: {str(e)}")
            continue
    print(f"✅ 成功处理 {len(tower_geometries)} 个杆塔几何体")
    return full_pcd, tower_geometries


def create_enhanced_tower_boxes_kuangxuan(tower_obbs: list,
                                          bbox_method: str = "kuangxuan",
                                          bbox_params: dict = None,
                                          add_center_marker: bool = True,
                                          add_height_indicator: bool = True):
    """Main box (red) + optional centre marker (yellow cube) + height line (green) per tower."""
    if bbox_params is None:
        bbox_params = dict(BBOX_PRESETS["kuangxuan_original"]["params"])
    out = []
    for tower_info in tower_obbs:
        try:
            center = tower_info['center']
            width, height = _tower_size(tower_info)
            lo, hi = _bounds_for(center, width, height, bbox_method, bbox_params)
            out.append((create_bbox_lineset_from_bounds(lo, hi)[0], (1.0, 0.0, 0.0)))
            if add_center_marker:
                h = np.full(3, min(width, height) * 0.1 / 2)
                out.append((create_bbox_lineset_from_bounds(center - h, center + h)[0], (1.0, 1.0, 0.0)))
            if add_height_indicator:
                out.append((np.array([[center[0], center[1], lo[2]], [center[0], center[1], hi[2]]]),
                            (0.0, 1.0, 0.0)))
        except Exception:
            continue
    return out


def visualize_towers_with_point_cloud_kuangxuan(las_path: str, tower_obbs: list,
                                                preset_name: str = "kuangxuan_original",
                                                output_path: str = None):
    try:
        bbox_method, bbox_params = get_bbox_preset(preset_name)
        full_pcd, tower_geometries = extract_and_visualize_towers_kuangxuan(
            las_path, tower_obbs, bbox_method, bbox_params)
        if output_path:
            print(f"💾 结果将保存到: {output_path}")
        return full_pcd, tower_geometries
    except Exception as e:
        print(f"❌ 可视化失败: {str(e)}")
        return None, []


_OBB_EDGES = ((0, 1), (0, 2), (0, 3), (1, 6), (1, 7), (2, 5), (2, 7), (3, 5), (3, 6), (4, 5), (4, 6), (4, 7))


def _obb_line_points(center, rotation, extent):
    """Corner numbering / edge list of Open3D's OrientedBoundingBox::GetBoxPoints and
    LineSet::CreateFromOrientedBoundingBox (from memory - open3d is not installed here)."""
    R = np.asarray(rotation, float)
    ex, ey, ez = (R[:, k] * (extent[k] * 0.5) for k in range(3))
    c = np.asarray(center, float)
    pts = [c - ex - ey - ez, c + ex - ey - ez, c - ex + ey - ez, c - ex - ey + ez,
           c + ex + ey + ez, c - ex + ey + ez, c + ex - ey + ez, c + ex + ey - ez]
    return np.array([pts[i] for e in _OBB_EDGES for i in e])


def extract_and_visualize_towers_original(las_path: str, tower_obbs: list,
                                          scale_factors: list = None,
                                          line_color: tuple = (1.0, 0.0, 0.0),
                                          adaptive_scaling: bool = True):
    """Scaled oriented boxes (the non-default branch, reference :345-420)."""
    if scale_factors is None:
        scale_factors = [2.8, 2.8, 4.5]
    full_pcd = _read_cloud(las_path)
    tower_geometries = []
    print(f"🔧 开始处理 {len(tower_obbs)} 个杆塔，使用放大因子: {scale_factors}")
    for i, tower_info in enumerate(tower_obbs):
        try:
            ext = np.array(tower_info['extent'])
            if adaptive_scaling:
                h = ext[2]
                scale = [3.2, 3.2, 5.0] if h < 20 else ([3.0, 3.0, 4.8] if h < 40 else [2.8, 2.8, 4.5])
                print(f"📏 杆塔{i}: 高度{h:.1f}m, 自适应缩放{scale}")
            else:
                scale = scale_factors
                print(f"📏 杆塔{i}: 固定缩放{scale_factors}")
            big = ext * np.array(scale)
            print(f"📐 杆塔{i}: 原始尺寸{ext} -> 增强尺寸{big}")
            tower_geometries.append((_obb_line_points(tower_info['center'], tower_info['rotation'], big),
                                     line_color))
            print(f"✅ 杆塔{i}处理成功，中心：{tower_info['center']}")
        except Exception as e:
            print(f"⚠️ 杆塔{i}可视化失败: {str(e)}")
            continue
    print(f"✅ 成功处理 {len(tower_geometries)} 个杆塔几何体")
    return full_pcd, tower_geometries


def extract_and_visualize_towers(las_path: str, tower_obbs: list,
                                 scale_factors: list = None,
                                 line_color: tuple = (1.0, 0.0, 0.0),
                                 adaptive_scaling: bool = True,
                                 use_kuangxuan_method: bool = True,
                                 kuangxuan_preset: str = "kuangxuan_original"):
    """Entry point the GUI calls (reference ui/extract.py:423-452)."""
    if use_kuangxuan_method:
        bbox_method, bbox_params = get_bbox_preset(kuangxuan_preset)
        return extract_and_visualize_towers_kuangxuan(las_path, tower_obbs, bbox_method, bbox_params,
                                                      line_color)
    return extract_and_visualize_towers_original(las_path, tower_obbs, scale_factors, line_color,
                                                 adaptive_scaling)
