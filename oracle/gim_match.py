"""Oracle for the consumer of the hot path's output: matching point-cloud towers to GIM towers
(TEST INFRASTRUCTURE).  Restates what ``/root/reference/utils/table_match_gim.py`` does with the tower
dicts ``extract_towers`` returns - which is the contract those dicts have to honour:

* ``haversine`` (:17-34), earth radius 6371.0 km;
* ``convert_pointcloud_ellipsoid_to_orthometric`` (:37-142): reads ``tower['center']`` (x, y projected,
  z ellipsoid height), ``tower.get('height', 0)``, ``tower.get('north_angle', 0)``; x, y go through the
  caller's transformer, z through ``ElevationConverter`` (utils/elevation_converter.py:41-55), which falls back
  to ``z - region_n_value`` when no geoid grid is available (always the case offline);
* ``match_towers`` (:145-196): for every GIM tower the FIRST point-cloud tower within
  ``distance_threshold`` metres (haversine) and ``height_threshold`` metres of height difference.

Pinned by tests/golden/gim_match.json, produced by running the reference's own module (PyQt5 / pyproj
replaced by empty placeholder modules, its geoid transformer therefore in fallback mode).
"""
from __future__ import annotations

import math


def haversine(lat1, lon1, lat2, lon2):
    R = 6371.0
    lat1, lon1, lat2, lon2 = map(math.radians, [lat1, lon1, lat2, lon2])
    dlat = lat2 - lat1
    dlon = lon2 - lon1
    a = math.sin(dlat / 2) ** 2 + math.cos(lat1) * math.cos(lat2) * math.sin(dlon / 2) ** 2
    c = 2 * math.atan2(math.sqrt(a), math.sqrt(1 - a))
    return R * c * 1000


def convert_towers(pointcloud_towers, transform_xy, region_n_value=25.0):
    out = []
    for i, tower in enumerate(pointcloud_towers):
        c = tower["center"]
        lon, lat = transform_xy(c[0], c[1])
        ell = c[2]
        ortho = ell - region_n_value
        out.append({"id": f"PC-{i + 1}", "converted_center": [lon, lat, ortho], "height": tower.get("height", 0),
                    "north_angle": tower.get("north_angle", 0), "original_center": c, "ellipsoid_height": ell,
                    "orthometric_height": ortho, "n_value": ell - ortho, "height_conversion_applied": True})
    return out


def match_towers(gim_list, pointcloud_towers, transform_xy, distance_threshold=50, height_threshold=100,
                 region_n_value=25.0):
    conv = convert_towers(pointcloud_towers, transform_xy, region_n_value)
    matched = []
    for i, g in enumerate(gim_list):
        glat, glon, gh = g.get("lat", 0), g.get("lng", 0), g.get("h", 0)
        for j, t in enumerate(conv):
            lon, lat, h = t["converted_center"]
            if haversine(glat, glon, lat, lon) <= distance_threshold and abs(gh - h) <= height_threshold:
                matched.append((i, j))
                break
    return matched, conv
