// Host side of stage D1 (no device code, no HIP headers): trimesh bounds.oriented_bounds restated natively -
// the 2-D hull, the minimum-area rectangle and the pricing of one candidate direction, shared by libpch_hip.so
// (pch_obb_search_f64 / pch_obb_min_boxes_f64, pch_obb.hip) and by libpch_obbhost.so (pch_obb_host.cpp), the
// host-only library the box workers of pointcloudhookup_amd/obb.py load: a worker process must never open the GPU.
// Reference: utils/tower_extraction.py:137-139 (trimesh.PointCloud(...).bounding_box_oriented).
#pragma once
#include <math.h>
#include <stdint.h>
#include <string.h>

#include <algorithm>
#include <vector>

namespace pch {
namespace obbhost {


constexpr double OB_TOL = 1e-13;                            // np.finfo(float64).resolution * 100

struct Pt2 { double x, y; };

inline double cross2(const Pt2& o, const Pt2& a, const Pt2& b) {
    return (a.x - o.x) * (b.y - o.y) - (a.y - o.y) * (b.x - o.x);
}

// Andrew's monotone chain, counter-clockwise, collinear points dropped
inline void hull2d(std::vector<Pt2>& pts, std::vector<Pt2>& out) {
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

inline void planar(double theta, double ox, double oy, double t[3][3]) {
    const double c = cos(theta), s = sin(theta);
    t[0][0] = c;  t[0][1] = s; t[0][2] = ox;
    t[1][0] = -s; t[1][1] = c; t[1][2] = oy;
    t[2][0] = 0;  t[2][1] = 0; t[2][2] = 1;
}

inline void mul3(const double a[3][3], const double b[3][3], double c[3][3]) {
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j) c[i][j] = a[i][0] * b[0][j] + a[i][1] * b[1][j] + a[i][2] * b[2][j];
}

// trimesh oriented_bounds_2D
inline void min_area_rectangle(std::vector<Pt2>& pts, std::vector<Pt2>& hp, Rect& r) {
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
inline void eval_candidate(const double* v, int64_t nv, double theta, double phi, std::vector<Pt2>& pts,
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
inline int min_box(const double* v, int64_t nv, const int32_t* tri, int64_t nt, bool sorted_extents, double* to_origin,
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


// every candidate direction of ONE hull priced: volumes[c] (INFINITY where no rectangle exists) and the index of the
// smallest (first of equals), -1 if none
inline int32_t search_hull(const double* v, int64_t nv, const double* angles, int64_t nc, double* volumes,
                           std::vector<Pt2>& pts, std::vector<Pt2>& hp) {
    int32_t best = -1;
    double v1 = INFINITY;
    for (int64_t c = 0; c < nc; ++c) {
        double rot[3][3], h;
        Rect r;
        eval_candidate(v, nv, angles[2 * c], angles[2 * c + 1], pts, hp, rot, r, h);
        const double vol = r.ok ? r.ext[0] * r.ext[1] * h : INFINITY;
        volumes[c] = vol;
        if (vol < v1) { v1 = vol; best = (int32_t)c; }
    }
    return best;
}

// ---- what qhull is shown (pointcloudhookup_amd/obb.py, "what qhull is shown"): the points of a cluster that are NOT
// strictly inside the tetrahedron qhull starts from.  Prediction of that tetrahedron after libqhull_r 2019.1
// (qh_maxmin + qh_maxsimplex): candidates = first minimum and maximum point of every coordinate; the x-extremes, then
// the candidate with the largest |2x2 determinant| of the (unit-box scaled) x,y differences, then the largest |3x3
// determinant|.  Stands down (returns false: show qhull everything) when the choice could hinge on rounding or on
// qhull's 'search all points' rule.  pts: n rows of three floats or doubles; out: room for n rows of three doubles.
template <typename T>
inline bool reduce_for_qhull(const T* pts, int64_t n, double* out, int64_t* out_rows) {
    auto all = [&]() {
        for (int64_t i = 0; i < 3 * n; ++i) out[i] = (double)pts[i];
        *out_rows = n;
        return false;
    };
    if (n < 64) return all();
    int64_t mp[6];
    double lo[3], hi[3];
    {
        // the extreme VALUES in one sweep the compiler can vectorise (48 independent lanes over the flat array: lane
        // j holds coordinate j % 3), then the FIRST row holding each of them - qhull's strict comparisons keep the
        // first occurrence
        constexpr int LANES = 48;
        T mn[LANES], mx[LANES];
        const int64_t flat = 3 * n;
        for (int j = 0; j < LANES; ++j) { mn[j] = pts[j % 3]; mx[j] = pts[j % 3]; }
        int64_t i = 0;
        for (; i + LANES <= flat; i += LANES)
            for (int j = 0; j < LANES; ++j) {
                const T v = pts[i + j];
                mn[j] = v < mn[j] ? v : mn[j];
                mx[j] = v > mx[j] ? v : mx[j];
            }
        for (; i < flat; ++i) {
            const T v = pts[i];
            const int j = (int)(i % 3);
            mn[j] = v < mn[j] ? v : mn[j];
            mx[j] = v > mx[j] ? v : mx[j];
        }
        for (int k = 0; k < 3; ++k) {
            T a = mn[k], b = mx[k];
            for (int j = k; j < LANES; j += 3) { a = mn[j] < a ? mn[j] : a; b = mx[j] > b ? mx[j] : b; }
            lo[k] = (double)a; hi[k] = (double)b;
            if (!(hi[k] > lo[k]) || !(fabs(hi[k]) < INFINITY) || !(fabs(lo[k]) < INFINITY)) return all();
            int64_t imin = -1, imax = -1;
            for (int64_t r = 0; r < n && (imin < 0 || imax < 0); ++r) {
                const T v = pts[3 * r + k];
                if (imin < 0 && v == a) imin = r;
                if (imax < 0 && v == b) imax = r;
            }
            if (imin < 0 || imax < 0) return all();        // (cannot happen: the values were read from these rows)
            mp[2 * k] = imin; mp[2 * k + 1] = imax;
        }
    }
    double q[6][3];
    for (int i = 0; i < 6; ++i)
        for (int k = 0; k < 3; ++k) q[i][k] = ((double)pts[3 * mp[i] + k] - lo[k]) / (hi[k] - lo[k]);
    int sel[4] = {0, 1, -1, -1};                           // positions in mp[]: the x-extremes first
    if (mp[0] == mp[1]) return all();
    double prev = q[1][0] - q[0][0];
    for (int k = 2; k <= 3; ++k) {
        double best = -1.0, second = -1.0;
        int ibest = -1;
        for (int i = 0; i < 6; ++i) {
            bool used = false;
            for (int j = 0; j < k; ++j) used |= mp[sel[j]] == mp[i];
            if (used) continue;
            double d;
            if (k == 2) {
                const double a0 = q[sel[0]][0] - q[i][0], a1 = q[sel[0]][1] - q[i][1];
                const double b0 = q[sel[1]][0] - q[i][0], b1 = q[sel[1]][1] - q[i][1];
                d = fabs(a0 * b1 - a1 * b0);
            } else {
                double r[3][3];
                for (int j = 0; j < 3; ++j) for (int c = 0; c < 3; ++c) r[j][c] = q[sel[j]][c] - q[i][c];
                d = fabs(r[0][0] * (r[1][1] * r[2][2] - r[1][2] * r[2][1]) - r[0][1] * (r[1][0] * r[2][2] - r[1][2] * r[2][0]) +
                         r[0][2] * (r[1][0] * r[2][1] - r[1][1] * r[2][0]));
            }
            if (ibest >= 0 && mp[i] == mp[ibest]) continue;               // the same point twice among the candidates
            if (d > best) { second = best; best = d; ibest = i; }
            else if (d > second) second = d;
        }
        if (ibest < 0) return all();
        if (second > best * (1.0 - 1e-6)) return all();    // qhull's own rounding would decide
        if (!(best > 0.05 * prev)) return all();           // too close to qhull's 'search all points' regime
        prev = best;
        sel[k] = ibest;
    }
    // barycentric coordinates with respect to the tetrahedron (in the points' own frame)
    double t0[3], m[3][3];
    for (int c = 0; c < 3; ++c) t0[c] = (double)pts[3 * mp[sel[0]] + c];
    for (int j = 0; j < 3; ++j)
        for (int c = 0; c < 3; ++c) m[c][j] = (double)pts[3 * mp[sel[j + 1]] + c] - t0[c];   // columns = edges
    const double det = m[0][0] * (m[1][1] * m[2][2] - m[1][2] * m[2][1]) - m[0][1] * (m[1][0] * m[2][2] - m[1][2] * m[2][0]) +
                       m[0][2] * (m[1][0] * m[2][1] - m[1][1] * m[2][0]);
    if (!(fabs(det) > 0.0) || !(fabs(det) < INFINITY)) return all();
    double inv[3][3];
    inv[0][0] = (m[1][1] * m[2][2] - m[1][2] * m[2][1]) / det; inv[0][1] = (m[0][2] * m[2][1] - m[0][1] * m[2][2]) / det;
    inv[0][2] = (m[0][1] * m[1][2] - m[0][2] * m[1][1]) / det; inv[1][0] = (m[1][2] * m[2][0] - m[1][0] * m[2][2]) / det;
    inv[1][1] = (m[0][0] * m[2][2] - m[0][2] * m[2][0]) / det; inv[1][2] = (m[0][2] * m[1][0] - m[0][0] * m[1][2]) / det;
    inv[2][0] = (m[1][0] * m[2][1] - m[1][1] * m[2][0]) / det; inv[2][1] = (m[0][1] * m[2][0] - m[0][0] * m[2][1]) / det;
    inv[2][2] = (m[0][0] * m[1][1] - m[0][1] * m[1][0]) / det;
    // blocks of rows: the inside test as a branch-free sweep into flags, then the rows that stay are copied
    int64_t kept = 0;
    constexpr int BLOCK = 512;
    unsigned char in[BLOCK];
    for (int64_t r0 = 0; r0 < n; r0 += BLOCK) {
        const int m = (int)(n - r0 < BLOCK ? n - r0 : BLOCK);
        const T* p = pts + 3 * r0;
        for (int i = 0; i < m; ++i) {
            const double dx = (double)p[3 * i] - t0[0], dy = (double)p[3 * i + 1] - t0[1], dz = (double)p[3 * i + 2] - t0[2];
            const double b0 = inv[0][0] * dx + inv[0][1] * dy + inv[0][2] * dz;
            const double b1 = inv[1][0] * dx + inv[1][1] * dy + inv[1][2] * dz;
            const double b2 = inv[2][0] * dx + inv[2][1] * dy + inv[2][2] * dz;
            in[i] = (unsigned char)((b0 > 1e-6) & (b1 > 1e-6) & (b2 > 1e-6) & ((b0 + b1 + b2) < 1.0 - 1e-6));
        }
        for (int i = 0; i < m; ++i) {
            out[3 * kept] = (double)p[3 * i]; out[3 * kept + 1] = (double)p[3 * i + 1]; out[3 * kept + 2] = (double)p[3 * i + 2];
            kept += 1 - in[i];                              // an inside row is overwritten by the next one
        }
    }
    *out_rows = kept;
    return kept < n;
}

}  // namespace obbhost
}  // namespace pch
