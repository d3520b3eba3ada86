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

#ifdef __cplusplus
}
#endif
#endif
