/* libpch_obbhost.so - host-only C ABI of stage D1 for box WORKER processes.
 *
 * The reference boxes every cluster with trimesh (utils/tower_extraction.py:137-139:
 * trimesh.PointCloud(cluster_points).bounding_box_oriented).  The default ("exact") mode of this package keeps
 * qhull on the full cluster (pointcloudhookup_amd/obb.py) in a pool of worker processes; a worker prices the
 * candidate directions of the hull it has just built with this call - the same code as pch_obb_search_f64
 * (include/pch_hip.h) for one hull, in a library that does NOT link the HIP runtime: dozens of workers must not
 * open the GPU.  Built from pointcloudhookup_amd/csrc/pch_obb_host.cpp with -ffp-contract=off. */
#ifndef PCH_OBBHOST_H
#define PCH_OBBHOST_H
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

int pch_obbhost_version(void);

/* One convex hull: verts [nv,3] float64, angles [nc,2] float64 = (theta, phi) per candidate direction in
 * evaluation order.  out_volumes [nc]: box volume per candidate (inf: no rectangle); *out_best: index of the
 * first candidate of smallest volume, -1 if none.  Returns 0, -1 for a null pointer / negative size. */
int pch_obbhost_search_f64(const double* verts, int64_t nv, const double* angles, int64_t nc, int32_t* out_best,
                           double* out_volumes);

/* What qhull is shown of one cluster: the rows that are NOT strictly inside the tetrahedron qhull starts from (such
 * rows are inert in qhull's run; about a third of a tower cluster), as float64, in input order.  The tetrahedron is
 * predicted from qhull's own rule (libqhull_r 2019.1: qh_maxmin + qh_maxsimplex); where that choice could hinge on
 * rounding or on qhull's 'search all points' rule the call stands down and copies every row.
 * pts [n,3] float32 / float64; out [n,3] float64 capacity; *out_rows: rows written.
 * Returns 1 = reduced, 0 = all rows copied, -1 = bad argument.  See pointcloudhookup_amd/obb.py for the argument why
 * the hull, its facet order and hence the box search are unchanged, and how that is checked. */
int pch_obbhost_reduce_f32(const float* pts, int64_t n, double* out, int64_t* out_rows);
int pch_obbhost_reduce_f64(const double* pts, int64_t n, double* out, int64_t* out_rows);

#ifdef __cplusplus
}
#endif
#endif
