// Stage D1, fast mode: the bandwidth part of the oriented bounding box of every cluster
// (utils/tower_extraction.py:131-139: trimesh.PointCloud(cluster_points).bounding_box_oriented, i.e.
// qhull's 3-D hull of the whole cluster, then a search over the hull's facet normals).
//
//   device : obb_support_k  - per cluster, the support point (largest dot product) in 218 fixed directions
//            obb_planes_k   - the tetrahedra (c, s_i, s_j, s_k) over a fixed triangulation of the direction
//                             sphere, c = mean of the support points: every one lies inside the hull
//            obb_shell_k    - a point strictly inside one of these tetrahedra (margin 1e-7 m) is strictly
//                             inside the hull, hence no hull vertex: dropped.  What is kept (~1 %) is a
//                             superset of the hull's vertices, in the cluster's row order.
//   host   : pch_obb_min_boxes_f64 - trimesh's candidate search (facet normals folded to a hemisphere and
//                             de-duplicated at 0.1 rad, minimum-area rectangle of the projected hull per
//                             candidate, smallest volume wins) for many hulls at once on C++ threads.
//
// The exact mode (pointcloudhookup_amd/obb.py, qhull on the full cluster) stays the default: qhull's facet
// merging depends on the points it was shown, so the fast mode's box is *a* minimum-volume box by the same
// procedure, not always the same one.
#include "pch_common.h"

#include <math.h>

#include <algorithm>
#include <atomic>
#include <thread>
#include <vector>

namespace pch {

constexpr int OB_RINGS   = 9;                               // latitude rings between the poles
constexpr int OB_SECT    = 24;                              // points per ring
constexpr int OB_DIRS    = OB_RINGS * OB_SECT + 2;          // 218 directions
constexpr int OB_DPAD    = 256;                             // padded row of the support table
constexpr int OB_TETS    = 2 * OB_SECT * OB_RINGS;          // 432 tetrahedra
constexpr int OB_THREADS = 256;
constexpr int OB_SLICE   = 4096;                            // grouped rows per workgroup
constexpr int OB_MIN     = 2048;                            // smaller clusters are kept whole
constexpr double OB_MARGIN = 1e-7;                          // metres

struct ObPlanes { double p[4][4]; };                        // inside: p[i][0..2].q - p[i][3] > margin for all i

__device__ __forceinline__ int ob_cluster_of(const int64_t* __restrict__ offsets, int ncl, int64_t pos) {
    int lo = 0, hi = ncl;                                   // largest k with offsets[k] <= pos
    while (hi - lo > 1) {
        const int mid = (lo + hi) >> 1;
        if (offsets[mid] <= pos) lo = mid; else hi = mid;
    }
    return lo;
}

__device__ __forceinline__ void ob_direction(int t, float& dx, float& dy, float& dz) {
    if (t == 0) { dx = 0.f; dy = 0.f; dz = 1.f; return; }
    if (t >= OB_DIRS - 1) { dx = 0.f; dy = 0.f; dz = -1.f; return; }
    const int i = (t - 1) / OB_SECT + 1, j = (t - 1) % OB_SECT;
    const float phi = 3.14159265358979f * (float)i / (float)(OB_RINGS + 1);
    const float th  = 6.28318530717959f * (float)j / (float)OB_SECT;
    float sp, cp, st, ct;
    sincosf(phi, &sp, &cp);
    sincosf(th, &st, &ct);
    dx = sp * ct; dy = sp * st; dz = cp;
}

// support points: thread t owns direction t and walks the tile of points held in LDS (broadcast reads),
// so there is no cross-lane reduction; one 64-bit atomicMax per direction, segment and workgroup.
__global__ __launch_bounds__(OB_THREADS) void obb_support_k(const float* __restrict__ xyz,
                                                            const int32_t* __restrict__ perm,
                                                            const int64_t* __restrict__ offsets, int ncl,
                                                            unsigned long long* __restrict__ best) {
    __shared__ float4 tile[OB_THREADS];
    const int64_t n = offsets[ncl];
    int64_t pos = (int64_t)blockIdx.x * OB_SLICE;
    const int64_t stop = pos + OB_SLICE < n ? pos + OB_SLICE : n;
    if (pos >= stop) return;
    int k = ob_cluster_of(offsets, ncl, pos);
    float dx, dy, dz;
    ob_direction(threadIdx.x < OB_DIRS ? threadIdx.x : 0, dx, dy, dz);
    while (pos < stop) {
        const int64_t seg_end = offsets[k + 1] < stop ? offsets[k + 1] : stop;
        if (offsets[k + 1] - offsets[k] >= OB_MIN) {
            const int32_t r0 = perm[offsets[k]];
            const float ox = xyz[3 * (int64_t)r0], oy = xyz[3 * (int64_t)r0 + 1], oz = xyz[3 * (int64_t)r0 + 2];
            float bv = -INFINITY;
            uint32_t bp = 0;
            for (int64_t base = pos; base < seg_end; base += OB_THREADS) {
                const int64_t i = base + threadIdx.x;
                __syncthreads();
                if (i < seg_end) {
                    const int64_t r = perm[i];
                    tile[threadIdx.x] = make_float4(xyz[3 * r] - ox, xyz[3 * r + 1] - oy, xyz[3 * r + 2] - oz, 0.f);
                }
                __syncthreads();
                const int m = (int)(seg_end - base < OB_THREADS ? seg_end - base : OB_THREADS);
                for (int q = 0; q < m; ++q) {
                    const float4 p = tile[q];
                    const float v = fmaf(dx, p.x, fmaf(dy, p.y, dz * p.z));
                    if (v > bv) { bv = v; bp = (uint32_t)(base + q); }
                }
            }
            if (threadIdx.x < OB_DIRS && bv > -INFINITY)
                atomicMax(&best[(size_t)k * OB_DPAD + threadIdx.x],
                          ((unsigned long long)f32_ordered(bv) << 32) | (0xFFFFFFFFu - bp));
        }
        pos = seg_end;
        ++k;
    }
}

__device__ __forceinline__ void ob_cross(const double* a, const double* b, double* c) {
    c[0] = a[1] * b[2] - a[2] * b[1];
    c[1] = a[2] * b[0] - a[0] * b[2];
    c[2] = a[0] * b[1] - a[1] * b[0];
}
__device__ __forceinline__ double ob_dot(const double* a, const double* b) {
    return a[0] * b[0] + a[1] * b[1] + a[2] * b[2];
}
// unit plane through `through` with normal a x b, oriented so that `inside` lies on its positive side;
// false if the triangle or the orientation is degenerate
__device__ __forceinline__ bool ob_plane(const double* a, const double* b, const double* through,
                                         const double* inside, double* out) {
    double nrm[3];
    ob_cross(a, b, nrm);
    const double len = sqrt(ob_dot(nrm, nrm));
    if (!(len > 1e-12)) return false;
    nrm[0] /= len; nrm[1] /= len; nrm[2] /= len;
    const double off = ob_dot(nrm, through);
    const double side = ob_dot(nrm, inside) - off;
    if (!(fabs(side) > 1e-6)) return false;                 // a sliver: not worth a test
    const double sg = side > 0 ? 1.0 : -1.0;
    out[0] = sg * nrm[0]; out[1] = sg * nrm[1]; out[2] = sg * nrm[2]; out[3] = sg * off;
    return true;
}

// one workgroup per cluster: centre = mean of the support points, then one thread per triangle of the
// direction mesh builds the four planes of (centre, s_i, s_j, s_k) in coordinates relative to the centre
__global__ __launch_bounds__(OB_THREADS) void obb_planes_k(const float* __restrict__ xyz,
                                                           const int32_t* __restrict__ perm,
                                                           const int64_t* __restrict__ offsets,
                                                           const unsigned long long* __restrict__ best,
                                                           double* __restrict__ centre,
                                                           ObPlanes* __restrict__ planes) {
    __shared__ double sx[OB_DPAD], sy[OB_DPAD], sz[OB_DPAD];
    __shared__ double red[3][OB_THREADS / 64];
    const int k = blockIdx.x, t = threadIdx.x;
    if (offsets[k + 1] - offsets[k] < OB_MIN) return;
    double x = 0, y = 0, z = 0;
    if (t < OB_DIRS) {
        const uint32_t posn = 0xFFFFFFFFu - (uint32_t)(best[(size_t)k * OB_DPAD + t] & 0xFFFFFFFFull);
        const int64_t r = perm[posn];
        x = (double)xyz[3 * r]; y = (double)xyz[3 * r + 1]; z = (double)xyz[3 * r + 2];
    }
    sx[t] = x; sy[t] = y; sz[t] = z;
    const double wx = wave_reduce_add(x), wy = wave_reduce_add(y), wz = wave_reduce_add(z);
    if (lane_id() == 0) { red[0][wave_id()] = wx; red[1][wave_id()] = wy; red[2][wave_id()] = wz; }
    __syncthreads();
    double c[3];
#pragma unroll
    for (int a = 0; a < 3; ++a) c[a] = (red[a][0] + red[a][1] + red[a][2] + red[a][3]) / (double)OB_DIRS;
    if (t == 0) { centre[4 * k] = c[0]; centre[4 * k + 1] = c[1]; centre[4 * k + 2] = c[2]; centre[4 * k + 3] = 0; }
    for (int tri = t; tri < OB_TETS; tri += OB_THREADS) {
        int ia, ib, id;
        if (tri < OB_SECT) {                                            // fan around the north pole
            ia = 0; ib = 1 + tri; id = 1 + (tri + 1) % OB_SECT;
        } else if (tri >= OB_TETS - OB_SECT) {                          // fan around the south pole
            const int j = tri - (OB_TETS - OB_SECT);
            ia = OB_DIRS - 1; ib = 1 + (OB_RINGS - 1) * OB_SECT + (j + 1) % OB_SECT;
            id = 1 + (OB_RINGS - 1) * OB_SECT + j;
        } else {                                                        // two triangles per ring cell
            const int u = tri - OB_SECT, cell = u >> 1, ring = cell / OB_SECT, j = cell % OB_SECT;
            const int a0 = 1 + ring * OB_SECT + j, a1 = 1 + ring * OB_SECT + (j + 1) % OB_SECT;
            const int b0 = a0 + OB_SECT, b1 = a1 + OB_SECT;
            if (u & 1) { ia = a0; ib = b1; id = a1; } else { ia = a0; ib = b0; id = b1; }
        }
        const double A[3] = {sx[ia] - c[0], sy[ia] - c[1], sz[ia] - c[2]};
        const double B[3] = {sx[ib] - c[0], sy[ib] - c[1], sz[ib] - c[2]};
        const double D[3] = {sx[id] - c[0], sy[id] - c[1], sz[id] - c[2]};
        const double AB[3] = {B[0] - A[0], B[1] - A[1], B[2] - A[2]};
        const double AD[3] = {D[0] - A[0], D[1] - A[1], D[2] - A[2]};
        const double zero[3] = {0, 0, 0};
        ObPlanes pl;
        const bool ok = ob_plane(A, B, zero, D, pl.p[0]) && ob_plane(B, D, zero, A, pl.p[1]) &&
                        ob_plane(D, A, zero, B, pl.p[2]) && ob_plane(AB, AD, A, zero, pl.p[3]);
        if (!ok) {                                                      // nobody is inside this one
            for (int a = 0; a < 4; ++a) { pl.p[a][0] = pl.p[a][1] = pl.p[a][2] = 0; pl.p[a][3] = 1; }
        }
        planes[(size_t)k * OB_TETS + tri] = pl;
    }
}

__global__ __launch_bounds__(OB_THREADS) void obb_shell_k(const float* __restrict__ xyz,
                                                          const int32_t* __restrict__ perm,
                                                          const int64_t* __restrict__ offsets, int ncl,
                                                          const double* __restrict__ centre,
                                                          const ObPlanes* __restrict__ planes,
                                                          uint8_t* __restrict__ keep) {
    extern __shared__ double lds[];                                     // OB_TETS * 16 doubles
    const int64_t n = offsets[ncl];
    int64_t pos = (int64_t)blockIdx.x * OB_SLICE;
    const int64_t stop = pos + OB_SLICE < n ? pos + OB_SLICE : n;
    if (pos >= stop) return;
    int k = ob_cluster_of(offsets, ncl, pos);
    while (pos < stop) {
        const int64_t seg_end = offsets[k + 1] < stop ? offsets[k + 1] : stop;
        if (offsets[k + 1] - offsets[k] < OB_MIN) {
            for (int64_t i = pos + threadIdx.x; i < seg_end; i += OB_THREADS) keep[i] = 1;
        } else {
            __syncthreads();
            const double* src = reinterpret_cast<const double*>(planes + (size_t)k * OB_TETS);
            for (int i = threadIdx.x; i < OB_TETS * 16; i += OB_THREADS) lds[i] = src[i];
            __syncthreads();
            const double cx = centre[4 * k], cy = centre[4 * k + 1], cz = centre[4 * k + 2];
            for (int64_t i = pos + threadIdx.x; i < seg_end; i += OB_THREADS) {
                const int64_t r = perm[i];
                const double qx = (double)xyz[3 * r] - cx, qy = (double)xyz[3 * r + 1] - cy,
                             qz = (double)xyz[3 * r + 2] - cz;
                bool inside = false;
                for (int t = 0; t < OB_TETS && !inside; ++t) {
                    const double* p = lds + t * 16;
                    if (p[0] * qx + p[1] * qy + p[2] * qz - p[3] > OB_MARGIN &&
                        p[4] * qx + p[5] * qy + p[6] * qz - p[7] > OB_MARGIN &&
                        p[8] * qx + p[9] * qy + p[10] * qz - p[11] > OB_MARGIN &&
                        p[12] * qx + p[13] * qy + p[14] * qz - p[15] > OB_MARGIN)
                        inside = true;
                }
                keep[i] = inside ? 0 : 1;
            }
        }
        pos = seg_end;
        ++k;
    }
}

struct ObWs { unsigned long long* best; double* centre; ObPlanes* planes; };
static void ob_plan(Arena& a, ObWs& w, int32_t ncl) {
    const size_t k = ncl > 0 ? (size_t)ncl : 1;
    w.best   = a.take<unsigned long long>(k * OB_DPAD);
    w.centre = a.take<double>(k * 4);
    w.planes = a.take<ObPlanes>(k * OB_TETS);
}

}  // namespace pch

#include "pch_obb_host.h"

using namespace pch;
using namespace pch::obbhost;

extern "C" size_t pch_obb_shell_ws_bytes(int32_t nclusters) {
    Arena a;
    ObWs w;
    ob_plan(a, w, nclusters);
    return a.off + 256;
}

extern "C" int pch_obb_shell_f32(const float* xyz, const int32_t* perm, const int64_t* offsets, int32_t nclusters,
                                 int64_t n_grouped, uint8_t* out_keep, void* ws, size_t ws_bytes, void* stream) {
    PCH_REQUIRE(nclusters >= 0 && n_grouped >= 0, "negative size");
    PCH_REQUIRE(n_grouped < (int64_t)0xFFFFFFFFll, "more than 2^32-1 grouped rows");
    if (nclusters == 0 || n_grouped == 0) return PCH_OK;
    PCH_REQUIRE(xyz && perm && offsets && out_keep && ws, "null pointer");
    PCH_DEVICE_GUARD(xyz);
    hipStream_t s = static_cast<hipStream_t>(stream);
    Arena a(ws, ws_bytes);
    ObWs w;
    ob_plan(a, w, nclusters);
    if (a.overflow) {
        set_error("pch_obb_shell_f32: workspace too small (%zu bytes, need %zu)", ws_bytes,
                  pch_obb_shell_ws_bytes(nclusters));
        return PCH_ERR_WORKSPACE;
    }
    PCH_HIP_TRY(hipMemsetAsync(w.best, 0, sizeof(unsigned long long) * (size_t)nclusters * OB_DPAD, s));
    const unsigned grid = (unsigned)ceil_div(n_grouped, OB_SLICE);
    PCH_LAUNCH("obb_support", obb_support_k, dim3(grid), dim3(OB_THREADS), 0, s, xyz, perm, offsets, (int)nclusters,
               w.best);
    PCH_LAUNCH("obb_planes", obb_planes_k, dim3((unsigned)nclusters), dim3(OB_THREADS), 0, s, xyz, perm, offsets,
               w.best, w.centre, w.planes);
    PCH_LAUNCH("obb_shell", obb_shell_k, dim3(grid), dim3(OB_THREADS), sizeof(double) * OB_TETS * 16, s, xyz, perm,
               offsets, (int)nclusters, w.centre, w.planes, out_keep);
    return PCH_OK;
}

extern "C" int pch_obb_min_boxes_f64(const double* verts, const int64_t* vert_offsets, const int32_t* tris,
                                     const int64_t* tri_offsets, int32_t nhulls, int32_t sorted_extents,
                                     int32_t nthreads, double* out_to_origin, double* out_extents,
                                     int32_t* out_status) {
    PCH_REQUIRE(nhulls >= 0, "negative count");
    if (nhulls == 0) return PCH_OK;
    PCH_REQUIRE(verts && vert_offsets && tris && tri_offsets && out_to_origin && out_extents && out_status,
                "null pointer");
    for (int32_t k = 0; k < nhulls; ++k) {
        PCH_REQUIRE(vert_offsets[k + 1] >= vert_offsets[k] && tri_offsets[k + 1] >= tri_offsets[k],
                    "offsets must not decrease");
        const int64_t nv = vert_offsets[k + 1] - vert_offsets[k];
        for (int64_t f = 3 * tri_offsets[k]; f < 3 * tri_offsets[k + 1]; ++f)
            PCH_REQUIRE(tris[f] >= 0 && tris[f] < nv, "triangle index outside its hull's vertices");
    }
    std::atomic<int32_t> next(0);
    auto work = [&]() {
        for (;;) {
            const int32_t k = next.fetch_add(1);
            if (k >= nhulls) return;
            out_status[k] = min_box(verts + 3 * vert_offsets[k], vert_offsets[k + 1] - vert_offsets[k],
                                    tris + 3 * tri_offsets[k], tri_offsets[k + 1] - tri_offsets[k],
                                    sorted_extents != 0, out_to_origin + 16 * (size_t)k, out_extents + 3 * (size_t)k);
        }
    };
    int nt = nthreads > 0 ? nthreads : (int)std::thread::hardware_concurrency();
    nt = nt < 1 ? 1 : (nt > 32 ? 32 : nt);
    nt = nt > nhulls ? nhulls : nt;
    if (nt <= 1) {
        work();
    } else {
        std::vector<std::thread> pool;
        for (int i = 0; i < nt; ++i) pool.emplace_back(work);
        for (auto& t : pool) t.join();
    }
    return PCH_OK;
}

extern "C" int pch_obb_search_f64(const double* verts, const int64_t* vert_offsets, const double* angles,
                                  const int64_t* angle_offsets, int32_t nhulls, int32_t nthreads,
                                  int32_t* out_best, double* out_volumes) {
    PCH_REQUIRE(nhulls >= 0, "negative count");
    if (nhulls == 0) return PCH_OK;
    PCH_REQUIRE(verts && vert_offsets && angles && angle_offsets && out_best && out_volumes, "null pointer");
    for (int32_t k = 0; k < nhulls; ++k)
        PCH_REQUIRE(vert_offsets[k + 1] >= vert_offsets[k] && angle_offsets[k + 1] >= angle_offsets[k],
                    "offsets must not decrease");
    std::atomic<int32_t> next(0);
    auto work = [&]() {
        std::vector<Pt2> pts, hp;
        for (;;) {
            const int32_t k = next.fetch_add(1);
            if (k >= nhulls) return;
            const double* v = verts + 3 * vert_offsets[k];
            const int64_t nv = vert_offsets[k + 1] - vert_offsets[k];
            out_best[k] = search_hull(v, nv, angles + 2 * angle_offsets[k], angle_offsets[k + 1] - angle_offsets[k],
                                      out_volumes + angle_offsets[k], pts, hp);
        }
    };
    int nt = nthreads > 0 ? nthreads : (int)std::thread::hardware_concurrency();
    nt = nt < 1 ? 1 : (nt > 32 ? 32 : nt);
    nt = nt > nhulls ? nhulls : nt;
    if (nt <= 1) {
        work();
    } else {
        std::vector<std::thread> pool;
        for (int i = 0; i < nt; ++i) pool.emplace_back(work);
        for (auto& t : pool) t.join();
    }
    return PCH_OK;
}
