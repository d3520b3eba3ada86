// libpch_obbhost.so: the host-only part of stage D1 for the box WORKER processes of pointcloudhookup_amd/obb.py.
// A worker prices the candidate directions of the hull it has just built (qhull, scipy) without leaving its
// process - and without loading libpch_hip.so, i.e. without the HIP runtime: a pool of dozens of workers must
// not open the GPU.  Same code as pch_obb_search_f64 (pch_obb_host.h), same compiler flags (-ffp-contract=off).
// Declared in include/pch_obbhost.h.  Reference: utils/tower_extraction.py:137-139.
#include "pch_obb_host.h"
#include "../../include/pch_obbhost.h"

extern "C" int pch_obbhost_search_f64(const double* verts, int64_t nv, const double* angles, int64_t nc,
                                      int32_t* out_best, double* out_volumes) {
    if (nv < 0 || nc < 0 || !out_best) return -1;
    if (nc > 0 && (!verts || !angles || !out_volumes)) return -1;
    std::vector<pch::obbhost::Pt2> pts, hp;
    *out_best = pch::obbhost::search_hull(verts, nv, angles, nc, out_volumes, pts, hp);
    return 0;
}

extern "C" int pch_obbhost_reduce_f32(const float* pts, int64_t n, double* out, int64_t* out_rows) {
    if (n < 0 || !out_rows || (n > 0 && (!pts || !out))) return -1;
    return pch::obbhost::reduce_for_qhull(pts, n, out, out_rows) ? 1 : 0;
}

extern "C" int pch_obbhost_reduce_f64(const double* pts, int64_t n, double* out, int64_t* out_rows) {
    if (n < 0 || !out_rows || (n > 0 && (!pts || !out))) return -1;
    return pch::obbhost::reduce_for_qhull(pts, n, out, out_rows) ? 1 : 0;
}

extern "C" int pch_obbhost_version(void) { return 2; }
