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

// ------------------------------------------------------------------------------------------------
// host: minimum-volume box of one hull (trimesh bounds.oriented_bounds, restated in
// pointcloudhookup_amd/obb.py:oriented_bounds; this is its native form for many hulls)
// ------------------------------------------------------------------------------------------------
namespace {

const double OB_TOL = 1e-13;                                // np.finfo(float64).resolution * 100

struct Pt2 { double x, y; };

inline double cross2(const Pt2& o, const Pt2& a, const Pt2& b) {
    return (a.x - o.x) * (b.y - o.y) - (a.y - o.y) * (b.x - o.x);
}

// Andrew's monotone chain, counter-clockwise, collinear points dropped
void hull2d(std::vector<Pt2>& pts, std::vector<Pt2>& out) {
    std::sort(pts.begin(), pts.end(), [](const Pt2& a, const Pt2& b) { return a.x < b.x || (a.x == b.x && a.y < b.y); });
    pts.erase(std::unique(pts.begin(), pts.end(), [](const Pt2& a, const Pt2& b) { return a.x == b.x && a.y == b.y; }),
              pts.end());
    const size_t n = pts.size();
    out.clear();
    if (n < 3) { out = pts; return; }
    out.resize(2 * n);
    size_t m = 0;
    for (size_t i = 0; i < n; ++i) {
        while (m >= 2 && cross2(out[m - 2], out[m - 1], pts[i]) <= 0) --m;
        out[m++] = pts[i];
    }
    for (size_t i = n - 1, t = m + 1; i > 0; --i) {
        while (m >= t && cross2(out[m - 2], out[m - 1], pts[i - 1]) <= 0) --m;
        out[m++] = pts[i - 1];
    }
    out.resize(m - 1);
}

struct Rect { double t[3][3]; double ext[2]; bool ok; };

void planar(double theta, double ox, double oy, double t[3][3]) {
    const double c = cos(theta), s = sin(theta);
    t[0][0] = c;  t[0][1] = s; t[0][2] = ox;
    t[1][0] = -s; t[1][1] = c; t[1][2] = oy;
    t[2][0] = 0;  t[2][1] = 0; t[2][2] = 1;
}

void mul3(const double a[3][3], const double b[3][3], double c[3][3]) {
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j) c[i][j] = a[i][0] * b[0][j] + a[i][1] * b[1][j] + a[i][2] * b[2][j];
}

// trimesh oriented_bounds_2D
void min_area_rectangle(std::vector<Pt2>& pts, std::vector<Pt2>& hp, Rect& r) {
    hull2d(pts, hp);
    r.ok = false;
    const size_t m = hp.size();
    if (m < 3) return;
    double best_area = 0, best_ev[2] = {1, 0}, best_lo[2] = {0, 0}, best_ext[2] = {0, 0};
    for (size_t e = 0; e < m; ++e) {
        const Pt2 &a = hp[e], &b = hp[(e + 1) % m];
        double ex = b.x - a.x, ey = b.y - a.y;
        const double ln = sqrt(ex * ex + ey * ey);
        if (!(ln > 1e-10)) continue;
        ex /= ln; ey /= ln;
        double lox = INFINITY, hix = -INFINITY, loy = INFINITY, hiy = -INFINITY;
        for (size_t i = 0; i < m; ++i) {
            const double px = ex * hp[i].x + ey * hp[i].y, py = -ey * hp[i].x + ex * hp[i].y;
            lox = px < lox ? px : lox; hix = px > hix ? px : hix;
            loy = py < loy ? py : loy; hiy = py > hiy ? py : hiy;
        }
        const double area = (hix - lox) * (hiy - loy);
        if (!r.ok || area < best_area) {
            r.ok = true;
            best_area = area;
            best_ev[0] = ex; best_ev[1] = ey;
            best_lo[0] = lox; best_lo[1] = loy;
            best_ext[0] = hix - lox; best_ext[1] = hiy - loy;
        }
    }
    if (!r.ok) return;
    planar(atan2(best_ev[1], best_ev[0]), -best_lo[0] - best_ext[0] * 0.5, -best_lo[1] - best_ext[1] * 0.5, r.t);
    r.ext[0] = best_ext[0]; r.ext[1] = best_ext[1];
    if (r.ext[0] < r.ext[1]) {
        double q[3][3], tmp[3][3];
        planar(M_PI / 2, 0, 0, q);
        mul3(q, r.t, tmp);
        memcpy(r.t, tmp, sizeof(tmp));
        std::swap(r.ext[0], r.ext[1]);
    }
}

// one candidate direction (theta, phi): rotation that turns it onto +Z, height along it and the
// minimum-area rectangle of the projected vertices
void eval_candidate(const double* v, int64_t nv, double theta, double phi, std::vector<Pt2>& pts,
                    std::vector<Pt2>& hp, double rot[3][3], Rect& r, double& h) {
    const double ct = cos(theta), st = sin(theta), cp = cos(phi), sp = sin(phi);
    const double m[3][3] = {{cp * ct, cp * st, -sp}, {-st, ct, 0.0}, {sp * ct, sp * st, cp}};
    memcpy(rot, m, sizeof(m));
    double zlo = INFINITY, zhi = -INFINITY;
    pts.resize((size_t)nv);
    for (int64_t i = 0; i < nv; ++i) {
        const double* p = v + 3 * i;
        pts[(size_t)i].x = m[0][0] * p[0] + m[0][1] * p[1] + m[0][2] * p[2];
        pts[(size_t)i].y = m[1][0] * p[0] + m[1][1] * p[1] + m[1][2] * p[2];
        const double z = m[2][0] * p[0] + m[2][1] * p[1] + m[2][2] * p[2];
        zlo = z < zlo ? z : zlo; zhi = z > zhi ? z : zhi;
    }
    h = zhi - zlo;
    min_area_rectangle(pts, hp, r);
}

struct Cand { long long code; double theta, phi; int first; };

// returns 0, or 1 if the hull gives no candidate / no rectangle
int min_box(const double* v, int64_t nv, const int32_t* tri, int64_t nt, bool sorted_extents, double* to_origin,
            double* extents) {
    std::vector<Cand> cand;
    cand.reserve((size_t)nt);
    for (int64_t f = 0; f < nt; ++f) {
        const double* a = v + 3 * (int64_t)tri[3 * f];
        const double* b = v + 3 * (int64_t)tri[3 * f + 1];
        const double* c = v + 3 * (int64_t)tri[3 * f + 2];
        const double u[3] = {b[0] - a[0], b[1] - a[1], b[2] - a[2]}, w[3] = {c[0] - a[0], c[1] - a[1], c[2] - a[2]};
        double n[3] = {u[1] * w[2] - u[2] * w[1], u[2] * w[0] - u[0] * w[2], u[0] * w[1] - u[1] * w[0]};
        const double ln = sqrt(n[0] * n[0] + n[1] * n[1] + n[2] * n[2]);
        if (!(ln > OB_TOL)) continue;
        n[0] /= ln; n[1] /= ln; n[2] /= ln;
        bool neg[3], zero[3];
        for (int i = 0; i < 3; ++i) { neg[i] = n[i] < -OB_TOL; zero[i] = !(neg[i] || n[i] > OB_TOL); }
        if (neg[2] || (zero[2] && neg[1]) || (zero[2] && zero[1] && neg[0])) { n[0] = -n[0]; n[1] = -n[1]; n[2] = -n[2]; }
        Cand cd;
        cd.theta = atan2(n[1], n[0]);
        cd.phi = acos(n[2] < -1.0 ? -1.0 : (n[2] > 1.0 ? 1.0 : n[2]));
        const long long q0 = (long long)nearbyint(cd.theta * 10.0), q1 = (long long)nearbyint(cd.phi * 10.0);
        cd.code = q0 ^ (long long)((unsigned long long)q1 << 32);
        cd.first = (int)cand.size();
        cand.push_back(cd);
    }
    if (cand.empty()) return 1;
    // np.unique(code, return_index=True): ascending codes, first occurrence of each
    std::sort(cand.begin(), cand.end(),
              [](const Cand& a, const Cand& b) { return a.code < b.code || (a.code == b.code && a.first < b.first); });
    std::vector<Pt2> pts, hp;
    bool have = false;
    double best_vol = 0, best_ext[3] = {0, 0, 0}, best_rot[3][3] = {{0}}, best_t2[3][3] = {{0}};
    for (size_t ci = 0; ci < cand.size(); ++ci) {
        if (ci && cand[ci].code == cand[ci - 1].code) continue;
        double rot[3][3], h;
        Rect r;
        eval_candidate(v, nv, cand[ci].theta, cand[ci].phi, pts, hp, rot, r, h);
        if (!r.ok) continue;
        const double vol = r.ext[0] * r.ext[1] * h;
        if (!have || vol < best_vol) {
            have = true;
            best_vol = vol;
            best_ext[0] = r.ext[0]; best_ext[1] = r.ext[1]; best_ext[2] = h;
            memcpy(best_rot, rot, sizeof(rot));
            memcpy(best_t2, r.t, sizeof(r.t));
        }
    }
    if (!have) return 1;
    double rz[3][3] = {{best_t2[0][0], best_t2[0][1], 0}, {best_t2[1][0], best_t2[1][1], 0}, {0, 0, 1}};
    double R[3][3];
    mul3(rz, best_rot, R);
    double lo[3] = {INFINITY, INFINITY, INFINITY}, hi[3] = {-INFINITY, -INFINITY, -INFINITY};
    for (int64_t i = 0; i < nv; ++i) {
        const double* p = v + 3 * i;
        for (int a = 0; a < 3; ++a) {
            const double m = R[a][0] * p[0] + R[a][1] * p[1] + R[a][2] * p[2];
            lo[a] = m < lo[a] ? m : lo[a]; hi[a] = m > hi[a] ? m : hi[a];
        }
    }
    double T[4][4] = {{0}};
    for (int a = 0; a < 3; ++a) {
        for (int b = 0; b < 3; ++b) T[a][b] = R[a][b];
        T[a][3] = -(lo[a] + (hi[a] - lo[a]) * 0.5);
    }
    T[3][3] = 1;
    double ext[3] = {best_ext[0], best_ext[1], best_ext[2]};
    if (sorted_extents) {                                   // current trimesh: ascending extents, axes permuted
        int order[3] = {0, 1, 2};
        std::stable_sort(order, order + 3, [&](int a, int b) { return ext[a] < ext[b]; });
        double F[3][3] = {{0}};
        for (int a = 0; a < 3; ++a) F[a][order[a]] = -1.0;
        const double tr = F[0][0] + F[1][1] + F[2][2];
        if (fabs(tr) <= 1e-8)
            for (int a = 0; a < 3; ++a) for (int b = 0; b < 3; ++b) F[a][b] = -F[a][b];
        double T2[4][4] = {{0}};
        for (int a = 0; a < 3; ++a)
            for (int b = 0; b < 4; ++b) T2[a][b] = F[a][0] * T[0][b] + F[a][1] * T[1][b] + F[a][2] * T[2][b];
        T2[3][3] = 1;
        memcpy(T, T2, sizeof(T));
        const double e2[3] = {ext[order[0]], ext[order[1]], ext[order[2]]};
        memcpy(ext, e2, sizeof(ext));
    }
    memcpy(to_origin, T, sizeof(T));
    memcpy(extents, ext, sizeof(ext));
    return 0;
}

}  // namespace
}  // namespace pch

using namespace pch;

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
            int32_t best = -1;
            double v1 = INFINITY;
            for (int64_t c = angle_offsets[k]; c < angle_offsets[k + 1]; ++c) {
                double rot[3][3], h;
                Rect r;
                eval_candidate(v, nv, angles[2 * c], angles[2 * c + 1], pts, hp, rot, r, h);
                const double vol = r.ok ? r.ext[0] * r.ext[1] * h : INFINITY;
                out_volumes[c] = vol;
                if (vol < v1) { v1 = vol; best = (int32_t)(c - angle_offsets[k]); }
            }
            out_best[k] = best;
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
