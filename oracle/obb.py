"""Oracle for stage D1: approximate minimum-volume oriented bounding box of one
cluster (TEST INFRASTRUCTURE).

PARITY UNPINNED: the reference calls trimesh
(``trimesh.PointCloud(cluster_points).bounding_box_oriented``,
``/root/reference/utils/tower_extraction.py:137-139``); trimesh is not installed in
this image and the reference pins no version.  This file restates trimesh's published
algorithm (``trimesh.bounds.oriented_bounds`` / ``oriented_bounds_2D``,
``trimesh.convex.convex_hull``, ``trimesh.util.vector_hemisphere``,
``vector_to_spherical``, ``grouping.unique_rows``, ``transformations.spherical_matrix``)
on top of scipy's qhull binding, the same qhull trimesh itself calls:

1. 3-D convex hull of the points, qhull options ``QbB Pp Qt``; hull vertices kept in
   ascending input order; unit face normals from the triangle cross products
   (zero-area triangles dropped).
2. Candidate directions = face normals folded onto one hemisphere, converted to
   spherical (theta, phi), de-duplicated after rounding to ``angle_digits=1`` decimal
   (first face of every rounded direction is kept).
3. For every candidate: rotate it onto +Z, take the z-extent of the hull vertices and
   the minimum-area rectangle of their xy projection (rotating calipers over the 2-D
   hull edges, ``oriented_bounds_2D``); volume = area * z-extent; keep the smallest
   (strict ``<``, first wins).
4. ``to_origin`` = planar rotation * candidate rotation, translated so the box is
   centred at the origin; ``transform = inv(to_origin)``.

``extent_order``: recent trimesh sorts the three extents ascending and permutes the
axes accordingly (``ordered=True``); the authors' recorded run
(``/root/reference/test/kuangxuan.py:30``: 17.4 m high, 20.1 m wide) can only come
from a trimesh that did *not* sort, i.e. extents = [rect_long, rect_short, normal_extent].
Both are provided; the drop-in default is ``"unsorted"``.
"""
from __future__ import annotations

import numpy as np
from scipy.spatial import ConvexHull

TOL_ZERO = np.finfo(np.float64).resolution * 100


def _hull3d(points):
    pts = np.asarray(points, dtype=np.float64)
    hull = ConvexHull(pts, qhull_options="QbB Pp Qt")
    vid = np.sort(hull.vertices)
    mask = np.zeros(len(hull.points), dtype=np.int64)
    mask[vid] = np.arange(len(vid))
    faces = mask[hull.simplices].copy()
    vertices = hull.points[vid].copy()
    tri = vertices[faces]
    crosses = np.cross(tri[:, 1] - tri[:, 0], tri[:, 2] - tri[:, 0])
    norm = np.sqrt((crosses ** 2).sum(axis=1))
    valid = norm > TOL_ZERO
    normals = crosses[valid] / norm[valid][:, None]
    return vertices, normals


def _vector_hemisphere(vectors):
    negative = vectors < -TOL_ZERO
    zero = np.logical_not(np.logical_or(negative, vectors > TOL_ZERO))
    signs = np.ones(len(vectors), dtype=np.float64)
    signs[negative[:, 2]] = -1.0
    signs[np.logical_and(zero[:, 2], negative[:, 1])] = -1.0
    signs[np.logical_and(np.logical_and(zero[:, 2], zero[:, 1]), negative[:, 0])] = -1.0
    return vectors * signs.reshape((-1, 1))


def _vector_to_spherical(cartesian):
    x, y, z = cartesian.T
    return np.column_stack((np.arctan2(y, x), np.arccos(np.clip(z, -1.0, 1.0))))


def _unique_rows_first(data, digits):
    as_int = np.round(data * 10 ** digits).astype(np.int64)
    precision = 64 // as_int.shape[1]
    hashable = np.zeros(len(as_int), dtype=np.int64)
    for offset, column in enumerate(as_int.T):
        np.bitwise_xor(hashable, column << (offset * precision), out=hashable)
    _, unique = np.unique(hashable, return_index=True)
    return unique


def _spherical_matrix(theta, phi):
    """Rz(theta) @ Ry(phi) as a 4x4 (== trimesh euler_matrix(0, phi, theta, 'sxyz'))."""
    cj, sj = np.cos(phi), np.sin(phi)
    ck, sk = np.cos(theta), np.sin(theta)
    M = np.eye(4)
    M[0, :3] = [cj * ck, -sk, sj * ck]
    M[1, :3] = [cj * sk, ck, sj * sk]
    M[2, :3] = [-sj, 0.0, cj]
    return M


def _planar_matrix(offset=(0.0, 0.0), theta=0.0):
    T = np.eye(3)
    s, c = np.sin(theta), np.cos(theta)
    T[0, :2] = [c, s]
    T[1, :2] = [-s, c]
    T[:2, 2] = offset
    return T


def oriented_bounds_2d(points):
    pts = np.asarray(points, dtype=np.float64)
    convex = ConvexHull(pts, qhull_options="QbB")
    hull_edges = convex.points[convex.simplices]
    hull_points = convex.points[convex.vertices]
    edge_vectors = hull_edges[:, 1] - hull_edges[:, 0]
    edge_norm = np.sqrt(np.sum(edge_vectors ** 2, axis=1))
    nz = edge_norm > 1e-10
    edge_vectors = edge_vectors[nz] / edge_norm[nz].reshape((-1, 1))
    perp_vectors = np.fliplr(edge_vectors) * [-1.0, 1.0]
    x = np.dot(edge_vectors, hull_points.T)
    y = np.dot(perp_vectors, hull_points.T)
    bounds = np.column_stack((x.min(axis=1), y.min(axis=1), x.max(axis=1), y.max(axis=1)))
    extents = np.diff(bounds.reshape((-1, 2, 2)), axis=1).reshape((-1, 2))
    area = np.prod(extents, axis=1)
    k = int(area.argmin())
    rectangle = extents[k]
    offset = -bounds[k][:2] - (rectangle * 0.5)
    theta = np.arctan2(*edge_vectors[k][::-1])
    transform = _planar_matrix(offset, theta)
    if rectangle[0] < rectangle[1]:
        transform = np.dot(_planar_matrix(theta=np.pi / 2), transform)
        rectangle = np.roll(rectangle, 1)
    return transform, rectangle


def oriented_bounds(points, angle_digits=1, extent_order="unsorted"):
    """Returns (to_origin 4x4, extents[3])."""
    vertices, normals = _hull3d(points)
    spherical = _vector_to_spherical(_vector_hemisphere(normals))
    uniq = _unique_rows_first(spherical, angle_digits)
    hom = np.column_stack((vertices, np.ones(len(vertices))))
    min_volume = np.inf
    min_extents = min_2D = rotation_Z = None
    for th, ph in spherical[uniq]:
        to_2D = np.linalg.inv(_spherical_matrix(th, ph))
        projected = np.dot(to_2D, hom.T).T[:, :3]
        height = np.ptp(projected[:, 2])
        rotation_2D, box = oriented_bounds_2d(projected[:, :2])
        volume = np.prod(box) * height
        if volume < min_volume:
            min_volume = volume
            min_extents = np.append(box, height)
            min_2D = to_2D.copy()
            rotation_2D[:2, 2] = 0.0
            rotation_Z = np.eye(4)
            rotation_Z[0:2, 0:2] = rotation_2D[0:2, 0:2]
    to_origin = np.dot(rotation_Z, min_2D)
    transformed = np.dot(to_origin, hom.T).T[:, :3]
    box_center = transformed.min(axis=0) + np.ptp(transformed, axis=0) * 0.5
    to_origin[:3, 3] = -box_center
    if extent_order == "trimesh_sorted":
        order = min_extents.argsort()
        flip = np.eye(4)
        flip[:3, :3] = -np.eye(3)[order]
        if np.isclose(np.trace(flip[:3, :3]), 0.0):
            flip[:3, :3] = np.dot(flip[:3, :3], -np.eye(3))
        to_origin = np.dot(flip, to_origin)
        min_extents = min_extents[order]
    elif extent_order != "unsorted":
        raise ValueError(extent_order)
    return to_origin, min_extents


def bounding_box_oriented(points, extent_order="unsorted"):
    """(extents[3], transform 4x4 box->world) like trimesh's Box primitive fields used at
    utils/tower_extraction.py:139,151,165."""
    to_origin, extents = oriented_bounds(points, extent_order=extent_order)
    return extents, np.linalg.inv(to_origin)
