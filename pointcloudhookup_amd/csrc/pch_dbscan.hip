// Stage C: exact DBSCAN for every file-order chunk at once (reference:
// utils/tower_extraction.py:96-117 -> sklearn.cluster.DBSCAN(...).fit(chunk)).
//
// Method (grid DBSCAN, exact): points are binned into cubic cells of side
//   s = eps/sqrt(3) * (1 - 2^-16)
// so any two points of one cell are closer than eps under sklearn's own float64 predicate
// (DESIGN.md "cell-side margin").  Consequences used below:
//   * a cell holding >= min_samples points consists of core points only (no distance tests);
//   * all core points of one cell belong to one cluster, so clusters are the connected
//     components of a graph over CELLS (edge: some core pair of the two cells within eps);
//   * every neighbour of a point lies in the 5x5x5 block of cells around its own cell.
// Cell key = [chunk | cz | cy | cx] (x in the low bits), so for a fixed (dy,dz) the five
// x-neighbour cells are one contiguous run of the sorted point array: 25 runs per cell.
// Cluster numbering reproduces sklearn's sweep: id = rank of the component's smallest core
// index; a border point takes the smallest id among its core neighbours.
#include "pch_prims.h"

namespace pch {

constexpr int DB_THREADS = 256;
constexpr int DB_WAVES   = DB_THREADS / 64;
constexpr int DB_ROWS    = 25;
constexpr int INT_BIG    = 0x7fffffff;
constexpr int DB_FEW_QUERIES = 24;
constexpr int DB_AHEAD = 3;              // db_core_k, long sweeps: 64-candidate groups whose loads are in flight together
constexpr long long DB_LONG_TOT = 8192;  // ... a cell counts as long from this many candidates in its neighbourhood
constexpr int DB_SEGS = 43;          // 9 inner + 9 + 9 end pieces of the near runs + 16 outer runs

struct DbGrid {
    float   ox, oy, oz;          // grid origin (lower corner of the bounding box)
    double  cell;                // cell side s
    double  inv_cell;            // 1/s: cell index = floor((x - origin) * inv_cell); the 2^-16 slack in s
                                 // covers the rounding of this product (< 2^-21 cells for indices < 2^31)
    double  eps2;                // eps*eps (sklearn _dist_to_rdist)
    float   eps2_lo, eps2_hi;    // float32 pre-filter: d32 <= lo is surely inside, d32 >= hi surely outside
    int     bx, by, bz;          // key bits per axis
    int     mx, my, mz;          // largest valid cell coordinate per axis
    int64_t chunk_size;
    const uint32_t* chunk_bad;   // != 0: the chunk holds NaN/inf and stays noise as a whole
    const uint32_t* chunk_cells; // [chunks + 1] first cell of every chunk (cells are sorted by chunk first)
    int     min_samples;
};

// (dy,dz) rows ordered by distance so that early exits trigger as soon as possible
__constant__ int8_t DB_ROW_DY[DB_ROWS] = {0, 1, -1, 0, 0, 1, 1, -1, -1, 2, -2, 0, 0,
                                          2, 2, -2, -2, 1, 1, -1, -1, 2, 2, -2, -2};
__constant__ int8_t DB_ROW_DZ[DB_ROWS] = {0, 0, 0, 1, -1, 1, -1, 1, -1, 0, 0, 2, -2,
                                          1, -1, 1, -1, 2, -2, 2, -2, 2, -2, 2, -2};

__device__ __forceinline__ uint64_t db_pack(const DbGrid& g, uint64_t chunk, uint64_t cz,
                                            uint64_t cy, uint64_t cx) {
    return ((((chunk << g.bz) | cz) << g.by | cy) << g.bx) | cx;
}

__device__ __forceinline__ bool db_within(const float4& q, const float4& p, double eps2) {
    // euclidean_rdist: tmp = x1[j] - x2[j]; d += tmp * tmp   (float64, j = 0,1,2)
    const double dx = (double)q.x - (double)p.x;
    const double dy = (double)q.y - (double)p.y;
    const double dz = (double)q.z - (double)p.z;
    double d = dx * dx;
    d += dy * dy;
    d += dz * dz;
    return d <= eps2;
}

// Same predicate, cheaper: the float32 value d32 of the squared distance carries a relative error
// below 5*2^-24 (one rounding per difference, product and accumulation, all terms non-negative),
// so with a 2^-20 guard band d32 <= eps2*(1-2^-20) implies d64 <= eps2 and d32 >= eps2*(1+2^-20)
// implies d64 > eps2; only pairs inside the band are evaluated exactly.
__device__ __forceinline__ bool db_within2(const float4& q, const float4& p, const DbGrid& g) {
    const float dx = q.x - p.x, dy = q.y - p.y, dz = q.z - p.z;
    float d = dx * dx;
    d = __builtin_fmaf(dy, dy, d);
    d = __builtin_fmaf(dz, dz, d);
    if (d <= g.eps2_lo) return true;
    if (d >= g.eps2_hi) return false;
    return db_within(q, p, g.eps2);
}

// ---- one query per lane against a staged tile of candidates --------------------------------
// The tile holds candidates in pairs (x0 x1 | y0 y1 | z0 z1), so the float32 pre-filter of db_within2 runs on
// packed two-wide arithmetic (v_pk_add/mul/fma_f32: same operations, same roundings) and without a branch per
// candidate: the lane counts d <= eps2_lo and d < eps2_hi separately; only if the two counts differ - some
// candidate fell into the 2^-20 guard band - is the tile counted again with the exact predicate.  The LDS reads
// of four pairs are issued ahead of their arithmetic.  7 VALU instructions per test (before: 13 and a wait for
// LDS per candidate).
typedef float f32x2 __attribute__((ext_vector_type(2)));
struct __attribute__((aligned(16))) DbPair { f32x2 x, y, z, pad; };

__device__ __forceinline__ void db_tile_put(DbPair* __restrict__ tile, int l, const float4& p) {
    float* f = reinterpret_cast<float*>(&tile[l >> 1]);
    f[l & 1] = p.x; f[2 + (l & 1)] = p.y; f[4 + (l & 1)] = p.z;
}
__device__ __forceinline__ float4 db_tile_get(const DbPair* __restrict__ tile, int k) {
    const float* f = reinterpret_cast<const float*>(&tile[k >> 1]);
    float4 p;
    p.x = f[k & 1]; p.y = f[2 + (k & 1)]; p.z = f[4 + (k & 1)]; p.w = 0.0f;
    return p;
}
// nj candidates staged (padding beyond nj up to a multiple of 8 must be far away: +3e38); returns the number
// of them within eps of q
__device__ __forceinline__ int db_tile_count(const float4& q, const DbPair* __restrict__ tile, int nj,
                                             const DbGrid& g) {
    const f32x2 qx = {q.x, q.x}, qy = {q.y, q.y}, qz = {q.z, q.z};
    const float lo = g.eps2_lo, hi = g.eps2_hi;
    int c_lo = 0, c_hi = 0;
    const int np = ((nj + 7) & ~7) >> 1;
    for (int k = 0; k < np; k += 4) {
        f32x2 px[4], py[4], pz[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) { px[u] = tile[k + u].x; py[u] = tile[k + u].y; pz[u] = tile[k + u].z; }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const f32x2 dx = qx - px[u], dy = qy - py[u], dz = qz - pz[u];
            f32x2 d = dx * dx;
            d = __builtin_elementwise_fma(dy, dy, d);
            d = __builtin_elementwise_fma(dz, dz, d);
            c_lo += (d.x <= lo ? 1 : 0) + (d.y <= lo ? 1 : 0);
            c_hi += (d.x < hi ? 1 : 0) + (d.y < hi ? 1 : 0);
        }
    }
    if (c_lo != c_hi) {                                    // a candidate inside the guard band: exact recount
        c_lo = 0;
        for (int k = 0; k < nj; ++k) c_lo += db_within2(q, db_tile_get(tile, k), g) ? 1 : 0;
    }
    return c_lo;
}

// squared distance from a point to an axis-aligned box, same operation order as db_within;
// never larger than the computed distance to any point inside the box (monotone rounding)
__device__ __forceinline__ double db_box_d2(const float4& q, const float* __restrict__ box) {
    const double gx = fmax(fmax((double)box[0] - (double)q.x, (double)q.x - (double)box[3]), 0.0);
    const double gy = fmax(fmax((double)box[1] - (double)q.y, (double)q.y - (double)box[4]), 0.0);
    const double gz = fmax(fmax((double)box[2] - (double)q.z, (double)q.z - (double)box[5]), 0.0);
    double d = gx * gx;
    d += gy * gy;
    d += gz * gz;
    return d;
}
__device__ __forceinline__ double db_boxbox_d2(const float* __restrict__ a, const float* __restrict__ b) {
    const double gx = fmax(fmax((double)b[0] - (double)a[3], (double)a[0] - (double)b[3]), 0.0);
    const double gy = fmax(fmax((double)b[1] - (double)a[4], (double)a[1] - (double)b[4]), 0.0);
    const double gz = fmax(fmax((double)b[2] - (double)a[5], (double)a[2] - (double)b[5]), 0.0);
    double d = gx * gx;
    d += gy * gy;
    d += gz * gz;
    return d;
}

// ---- chunks that contain NaN/inf: sklearn raises for such a chunk (utils/tower_extraction.py:118-119);
// its rows keep their own chunk key and sit in ONE cell (0,0,0) of that chunk that is never core, so
// every chunk owns at least one cell and chunk_cells[] is complete (db_cells_k)
__global__ __launch_bounds__(DB_THREADS) void db_chunkbad_k(const float* __restrict__ xyz, int64_t n,
                                                            int64_t chunk_size, uint32_t* __restrict__ bad) {
    for (int64_t i = (int64_t)blockIdx.x * DB_THREADS + threadIdx.x; i < n;
         i += (int64_t)gridDim.x * DB_THREADS) {
        const uint32_t a = __float_as_uint(xyz[3 * i + 0]) & 0x7FFFFFFFu;
        const uint32_t b = __float_as_uint(xyz[3 * i + 1]) & 0x7FFFFFFFu;
        const uint32_t c = __float_as_uint(xyz[3 * i + 2]) & 0x7FFFFFFFu;
        if (a >= 0x7F800000u || b >= 0x7F800000u || c >= 0x7F800000u) atomicOr(&bad[i / chunk_size], 1u);
    }
}

// ---- first row holding NaN/inf (what sklearn's check_array rejects before DBSCAN.fit starts) ----
__global__ __launch_bounds__(DB_THREADS) void db_first_bad_k(const float* __restrict__ xyz, int64_t n,
                                                             unsigned long long* __restrict__ first) {
    unsigned long long best = ~0ull;
    for (int64_t i = (int64_t)blockIdx.x * DB_THREADS + threadIdx.x; i < n;
         i += (int64_t)gridDim.x * DB_THREADS) {
        const uint32_t a = __float_as_uint(xyz[3 * i + 0]) & 0x7FFFFFFFu;
        const uint32_t b = __float_as_uint(xyz[3 * i + 1]) & 0x7FFFFFFFu;
        const uint32_t c = __float_as_uint(xyz[3 * i + 2]) & 0x7FFFFFFFu;
        if ((a >= 0x7F800000u || b >= 0x7F800000u || c >= 0x7F800000u) && (unsigned long long)i < best)
            best = (unsigned long long)i;
    }
    best = wave_reduce_min(best);
    if (lane_id() == 0 && best != ~0ull) atomicMin(first, best);
}

// ---- bounding box of the finite input points (only when the caller did not provide one) ------
__global__ __launch_bounds__(DB_THREADS) void db_aabb_in_k(const float* __restrict__ xyz, int64_t n,
                                                           uint32_t* __restrict__ mm) {
    uint32_t lo[3] = {0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu}, hi[3] = {0u, 0u, 0u};
    for (int64_t i = (int64_t)blockIdx.x * DB_THREADS + threadIdx.x; i < n;
         i += (int64_t)gridDim.x * DB_THREADS) {
        const float x = xyz[3 * i + 0], y = xyz[3 * i + 1], z = xyz[3 * i + 2];
        if (!(fabsf(x) < INFINITY && fabsf(y) < INFINITY && fabsf(z) < INFINITY)) continue;
        const float v[3] = {x, y, z};
#pragma unroll
        for (int a = 0; a < 3; ++a) {
            const uint32_t k = f32_ordered(v[a]);
            lo[a] = k < lo[a] ? k : lo[a];
            hi[a] = k > hi[a] ? k : hi[a];
        }
    }
#pragma unroll
    for (int a = 0; a < 3; ++a) {
        lo[a] = wave_reduce_min(lo[a]);
        hi[a] = wave_reduce_max(hi[a]);
        if (lane_id() == 0) {
            if (lo[a] != 0xFFFFFFFFu) atomicMin(&mm[a], lo[a]);
            if (hi[a] != 0u) atomicMax(&mm[3 + a], hi[a]);
        }
    }
}

// Cell coordinates of a point: floor((x - origin) / s) per axis, as a product with 1/s in float64
// (error < 2^-21 cells for indices < 2^31, covered by the 2^-16 slack in s).  false: outside the
// grid (or NaN), coordinates 0.
__device__ __forceinline__ bool db_cell_coords(const DbGrid& g, float x, float y, float z,
                                               uint32_t& cx, uint32_t& cy, uint32_t& cz) {
    const double fx = floor(((double)x - (double)g.ox) * g.inv_cell);
    const double fy = floor(((double)y - (double)g.oy) * g.inv_cell);
    const double fz = floor(((double)z - (double)g.oz) * g.inv_cell);
    const bool ok = fx >= 0.0 && fx <= (double)g.mx && fy >= 0.0 && fy <= (double)g.my &&
                    fz >= 0.0 && fz <= (double)g.mz;
    cx = ok ? (uint32_t)fx : 0u; cy = ok ? (uint32_t)fy : 0u; cz = ok ? (uint32_t)fz : 0u;
    return ok;
}

// ---- cell keys -----------------------------------------------------------------------
__global__ __launch_bounds__(DB_THREADS) void db_keys_k(const float* __restrict__ xyz, int64_t n,
                                                        DbGrid g, const uint32_t* __restrict__ bad,
                                                        uint64_t* __restrict__ keys,
                                                        uint32_t* __restrict__ vals,
                                                        uint32_t* __restrict__ status) {
    const int64_t i = (int64_t)blockIdx.x * DB_THREADS + threadIdx.x;
    if (i >= n) return;
    vals[i] = (uint32_t)i;
    if (bad[i / g.chunk_size]) {                           // the whole chunk stays noise: one cell, never core
        keys[i] = db_pack(g, (uint64_t)(i / g.chunk_size), 0, 0, 0);
        return;
    }
    const float x = xyz[3 * i + 0], y = xyz[3 * i + 1], z = xyz[3 * i + 2];
    uint32_t cx, cy, cz;
    if (!db_cell_coords(g, x, y, z, cx, cy, cz)) atomicOr(status, 1u);
    keys[i] = db_pack(g, (uint64_t)(i / g.chunk_size), cz, cy, cx);
}

// ---- fallback for grids that do not fit the 64-bit key (extent/eps astronomically large: outliers, heavy
// tails): per-axis COMPRESSED cell coordinates.  The points are sorted along the axis; a gap of >= 3 cells
// between consecutive points starts a new segment (nothing on one side of such a gap is within eps of
// anything on the other side), cell indices are taken relative to the segment's first point (small, so the
// float64 product is as exact as on the main path) and segments are laid out 3 units apart.  Equal
// compressed index <=> same segment and same local cell; indices that differ by <= 2 are exactly as far
// apart as the true ones; everything else stays >= 3 apart: the 5x5x5 neighbourhood logic is unchanged,
// with at most 3n index values per axis.
__global__ __launch_bounds__(DB_THREADS) void dbc_axis_keys_k(const float* __restrict__ xyz, int64_t n, int axis,
                                                              uint64_t* __restrict__ keys, uint32_t* __restrict__ vals) {
    const int64_t i = (int64_t)blockIdx.x * DB_THREADS + threadIdx.x;
    if (i >= n) return;
    keys[i] = f32_ordered(xyz[3 * i + axis]);
    vals[i] = (uint32_t)i;
}
__global__ __launch_bounds__(DB_THREADS) void dbc_heads_k(const uint64_t* __restrict__ keys, int64_t n, double inv_cell,
                                                          uint32_t* __restrict__ head) {
    const int64_t i = (int64_t)blockIdx.x * DB_THREADS + threadIdx.x;
    if (i >= n) return;
    bool h = i == 0;
    if (i > 0) {
        const double a = (double)f32_unordered((uint32_t)keys[i - 1]), b = (double)f32_unordered((uint32_t)keys[i]);
        h = (b - a) * inv_cell >= 3.0;
    }
    head[i] = h ? 1u : 0u;
}
__global__ __launch_bounds__(DB_THREADS) void dbc_local_k(const uint64_t* __restrict__ keys, const uint32_t* __restrict__ flag,
                                                          const uint32_t* __restrict__ excl, int64_t n, double inv_cell,
                                                          float* __restrict__ headx, int phase,
                                                          uint32_t* __restrict__ li, uint32_t* __restrict__ seglen,
                                                          uint32_t* __restrict__ status) {
    const int64_t i = (int64_t)blockIdx.x * DB_THREADS + threadIdx.x;
    if (i >= n) return;
    const uint32_t seg = excl[i] + flag[i] - 1u;                     // flags are kept beside their scan
    const float x = f32_unordered((uint32_t)keys[i]);
    if (phase == 0) {
        if (flag[i]) headx[seg] = x;
        return;
    }
    const double v = floor(((double)x - (double)headx[seg]) * inv_cell);
    uint32_t u = 0;
    if (v >= 0.0 && v < 2147483000.0) u = (uint32_t)v; else atomicOr(status, 2u);
    li[i] = u;
    const bool last = i + 1 == n || flag[i + 1] != 0u;
    if (last) seglen[seg] = u + 3u;                                  // next segment starts 3 units behind this one's last cell
}
__global__ __launch_bounds__(DB_THREADS) void dbc_comp_k(const uint32_t* __restrict__ vals, const uint32_t* __restrict__ flag,
                                                         const uint32_t* __restrict__ excl, const uint32_t* __restrict__ li,
                                                         const uint32_t* __restrict__ segoff, int64_t n, int axis,
                                                         uint32_t* __restrict__ comp, uint32_t* __restrict__ cmax) {
    const int64_t i = (int64_t)blockIdx.x * DB_THREADS + threadIdx.x;
    if (i >= n) return;
    const uint32_t seg = excl[i] + flag[i] - 1u;
    const uint32_t c = segoff[seg] + li[i];
    comp[3 * (int64_t)vals[i] + axis] = c;
    if (i == n - 1) cmax[axis] = c;                                  // sorted: the last one is the largest
}
__global__ __launch_bounds__(DB_THREADS) void db_keys_comp_k(const uint32_t* __restrict__ comp, int64_t n, DbGrid g,
                                                             uint64_t* __restrict__ keys, uint32_t* __restrict__ vals) {
    const int64_t i = (int64_t)blockIdx.x * DB_THREADS + threadIdx.x;
    if (i >= n) return;
    vals[i] = (uint32_t)i;
    keys[i] = db_pack(g, 0, comp[3 * i + 2], comp[3 * i + 1], comp[3 * i + 0]);
}
__global__ __launch_bounds__(DB_THREADS) void db_add_offset_k(int32_t* __restrict__ labels, int64_t n, int32_t off) {
    const int64_t i = (int64_t)blockIdx.x * DB_THREADS + threadIdx.x;
    if (i < n && labels[i] >= 0) labels[i] += off;
}

// sorted order: gather coordinates (+ original row in .w)
__global__ __launch_bounds__(DB_THREADS) void db_gather_k(const float* __restrict__ xyz,
                                                          const uint64_t* __restrict__ keys,
                                                          const uint32_t* __restrict__ vals, int64_t n,
                                                          float4* __restrict__ pts) {
    const int64_t i = (int64_t)blockIdx.x * DB_THREADS + threadIdx.x;
    if (i >= n) return;
    const uint32_t o = vals[i];
    float4 p;
    p.x = xyz[3 * (int64_t)o + 0];
    p.y = xyz[3 * (int64_t)o + 1];
    p.z = xyz[3 * (int64_t)o + 2];
    p.w = __uint_as_float(o);
    pts[i] = p;
}

// ---- chunk-local sort: keys + sort + gather for one chunk per workgroup --------------------
// Chunks are runs of consecutive input rows, so the chunk field of the key is already sorted;
// only the cell bits (<= 32) need sorting, and only inside a chunk.  One 1024-thread workgroup
// per chunk.  What moves through the LSD passes is the row itself, (x, y, z, original row) as one
// float4: the cell key is a cheap function of (x, y, z) and is recomputed wherever a digit is
// needed, so there is neither a key/index stream nor a random gather at the end (a 12-byte
// gather costs a whole cache line).  Sweep B finds the chunk's own box in cell coordinates: what is
// sorted is the key RELATIVE to it - same order, but as many bits as the chunk's extent needs (15-17 for a
// 50 000-row chunk) instead of the tile's (21+), i.e. two 8/9-bit passes instead of three.  Sweep H builds
// the digit histograms of every pass; pass 0 reads the input rows, the last pass writes the sorted rows and
// their full keys (the cell heads are read off those keys by db_heads_k / db_cells_k).  Replaces db_chunkbad, db_keys, every radix pass (histogram + 3 scan
// kernels + scatter) and db_gather of the global path.
struct Row3 { float x, y, z; };               // 4-byte aligned: loads as one global_load_dwordx3
constexpr int CS_THREADS = 1024;
constexpr int CS_WAVES   = CS_THREADS / 64;
constexpr int CS_ROUNDS  = 4;
constexpr int CS_TILE    = CS_THREADS * CS_ROUNDS;
constexpr int CS_PASSES  = 4;
constexpr int CS_HREP    = 4;         // copies of every digit histogram (power of two)
constexpr int CS_BINS    = 512;       // digits of up to 9 bits
constexpr int64_t CS_MAX_CHUNK = 1 << 17;     // larger chunks use the global sort
constexpr int64_t CS_MIN_CHUNKS = 96;         // fewer chunks: the global sort keeps more of the GPU busy

// cell key of a row inside its chunk (0 when the row is outside the grid); ok = inside
__device__ __forceinline__ uint32_t cs_cell_key(const DbGrid& g, float x, float y, float z, bool& ok) {
    uint32_t cx, cy, cz;
    ok = db_cell_coords(g, x, y, z, cx, cy, cz);
    return (((cz << g.by) | cy) << g.bx) | cx;
}

__global__ __launch_bounds__(CS_THREADS) void db_chunksort_k(
    const float* __restrict__ xyz, int64_t n, DbGrid g, uint32_t* __restrict__ bad,
    float4* __restrict__ xbuf, float4* __restrict__ pts, uint64_t* __restrict__ keys_out,
    uint32_t* __restrict__ status, unsigned long long* __restrict__ stamps) {
#ifdef PCH_CS_STAMPS                                    // phase timing of one workgroup (tuning builds only)
    int stamp_i = 0;
#define CS_STAMP() if (stamps && blockIdx.x == gridDim.x / 2 && threadIdx.x == 0) stamps[stamp_i++] = wall_clock64();
    unsigned long long tacc[3] = {0, 0, 0}, tlast = 0;       // inside the tile loop: rank | offsets | scatter
#define CS_TILE_STAMP(k) do { const unsigned long long _t = wall_clock64(); tacc[k] += _t - tlast; tlast = _t; } while (0)
#else
#define CS_STAMP()
#define CS_TILE_STAMP(k)
#endif
    CS_STAMP();
    __shared__ uint32_t hist[CS_PASSES][CS_HREP][CS_BINS];   // replicated: lanes of a wave that share a bin (most do: a
                                                             // chunk covers few cells) spread over CS_HREP addresses
    __shared__ uint32_t cnt[CS_WAVES][CS_BINS];
    __shared__ uint32_t off[CS_WAVES][CS_BINS];
    __shared__ uint32_t base[CS_BINS];
    __shared__ uint32_t wsum[CS_BINS / 64];
    __shared__ uint32_t flags[2];                       // [0] chunk holds NaN/inf, [1] point outside the box
    __shared__ uint32_t cbox[6];                        // the chunk's own box in cell coordinates: min xyz, max xyz
    const int tid = threadIdx.x, w = wave_id(), l = lane_id();
    const int64_t c = blockIdx.x;
    const int64_t lo = c * g.chunk_size;
    const int cn = (int)((n - lo) < g.chunk_size ? (n - lo) : g.chunk_size);
    const Row3* __restrict__ rows = reinterpret_cast<const Row3*>(xyz) + lo;
    for (int j = tid; j < CS_PASSES * CS_HREP * CS_BINS; j += CS_THREADS) (&hist[0][0][0])[j] = 0;
    for (int j = tid; j < CS_WAVES * CS_BINS; j += CS_THREADS) (&cnt[0][0])[j] = 0;
    if (tid < 2) flags[tid] = 0;
    if (tid < 6) cbox[tid] = tid < 3 ? 0xFFFFFFFFu : 0u;
    __syncthreads();
    constexpr int HU = 8;                               // rows per thread in flight
    // ---- sweep B: NaN/inf and range checks, the chunk's box in cell coordinates
    {
        uint32_t mn[3] = {0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu}, mx[3] = {0u, 0u, 0u};
        for (int i0 = 0; i0 < cn; i0 += HU * CS_THREADS) {
            Row3 q[HU];
#pragma unroll
            for (int u = 0; u < HU; ++u) {
                const int i = i0 + u * CS_THREADS + tid;
                q[u] = rows[i < cn ? i : 0];
            }
#pragma unroll
            for (int u = 0; u < HU; ++u) {
                const bool in = i0 + u * CS_THREADS + tid < cn;
                const bool fin = fabsf(q[u].x) < INFINITY && fabsf(q[u].y) < INFINITY && fabsf(q[u].z) < INFINITY;
                uint32_t cc[3];
                const bool ok = db_cell_coords(g, q[u].x, q[u].y, q[u].z, cc[0], cc[1], cc[2]);
                if (in && !fin) flags[0] = 1u;
                else if (in && !ok) flags[1] = 1u;
                if (in && fin && ok) {
#pragma unroll
                    for (int a = 0; a < 3; ++a) { mn[a] = cc[a] < mn[a] ? cc[a] : mn[a]; mx[a] = cc[a] > mx[a] ? cc[a] : mx[a]; }
                }
            }
        }
#pragma unroll
        for (int a = 0; a < 3; ++a) {
            const uint32_t lo_w = wave_reduce_min(mn[a]), hi_w = wave_reduce_max(mx[a]);
            if (l == 0) { atomicMin(&cbox[a], lo_w); atomicMax(&cbox[3 + a], hi_w); }
        }
    }
    __syncthreads();
    CS_STAMP();
    // widths of the relative key; rows outside the grid (the call fails for them) land on arbitrary digits
    const uint32_t ox = cbox[0], oy = cbox[1], oz = cbox[2];
    const bool any = cbox[3] >= ox && cbox[4] >= oy && cbox[5] >= oz;
    const int wx = any && cbox[3] > ox ? 32 - __clz(cbox[3] - ox) : 0;
    const int wy = any && cbox[4] > oy ? 32 - __clz(cbox[4] - oy) : 0;
    const int wz = any && cbox[5] > oz ? 32 - __clz(cbox[5] - oz) : 0;
    const int tb = wx + wy + wz;                         // <= bx + by + bz <= 32
    const int passes = (tb + 8) / 9;
    const int dbits = passes ? (tb + passes - 1) / passes : 1;
    const uint32_t mask = (1u << dbits) - 1u;
    const uint32_t gxm = (1u << g.bx) - 1u, gym = (1u << g.by) - 1u;
    auto relkey = [&](uint32_t k) -> uint32_t {
        const uint32_t cx = k & gxm, cy = (k >> g.bx) & gym, cz = k >> (g.bx + g.by);
        return (uint32_t)((((((uint64_t)(cz - oz)) << wy) | (uint64_t)(cy - oy)) << wx) | (uint64_t)(cx - ox));   // tb <= 32
    };
    // ---- sweep H: digit histograms of every pass
    for (int i0 = 0; i0 < cn; i0 += HU * CS_THREADS) {  // workgroup-uniform trip count
        Row3 q[HU];
#pragma unroll
        for (int u = 0; u < HU; ++u) {
            const int i = i0 + u * CS_THREADS + tid;
            q[u] = rows[i < cn ? i : 0];
        }
#pragma unroll
        for (int u = 0; u < HU; ++u) {
            const bool in = i0 + u * CS_THREADS + tid < cn;
            bool ok;
            const uint32_t k = relkey(cs_cell_key(g, q[u].x, q[u].y, q[u].z, ok));
            if (in)
                for (int p = 0; p < passes; ++p) atomicAdd(&hist[p][l & (CS_HREP - 1)][(k >> (p * dbits)) & mask], 1u);
        }
    }
    __syncthreads();
    const bool isbad = flags[0] != 0;                   // the whole chunk stays noise: one cell, never core
    CS_STAMP();
    if (tid == 0) {
        bad[c] = isbad ? 1u : 0u;
        if (!isbad && flags[1]) atomicOr(status, 1u);
    }
    const int sh = g.bx + g.by + g.bz;
    const uint64_t hi = sh < 64 ? ((uint64_t)c << sh) : 0ull;
    if (isbad || passes == 0) {                         // rows stay where they are: one cell (cell 0 if bad)
        for (int i = tid; i < cn; i += CS_THREADS) {
            const Row3 q = rows[i];
            float4 o4;
            o4.x = q.x; o4.y = q.y; o4.z = q.z; o4.w = __uint_as_float((uint32_t)(lo + i));
            pts[lo + i] = o4;
            bool ok;
            keys_out[lo + i] = isbad ? hi : (hi | cs_cell_key(g, q.x, q.y, q.z, ok));
        }
        return;
    }
    // One pass, unswitched on what the compiler has to know statically: FIRST (the source rows are 12-byte input
    // rows), LAST (the full keys are written too) and, per tile, FULL (every lane holds a row).  In a full tile the
    // scattered stores are unconditional, so their number is a constant - and with it the wait for the NEXT tile's
    // rows, which are requested before them: the memory counter is one in-order queue of loads and stores, and a wait
    // that cannot count the stores behind a load drains them all (that drain, once per tile, was 40 % of the kernel).
    auto run_pass = [&](int p, auto FIRST_T, auto LAST_T) {
        constexpr bool FIRST = decltype(FIRST_T)::value, LAST = decltype(LAST_T)::value;
        const int shift = p * dbits;
        // pass p writes the sorted-rows array when an even number of passes follows, else the spare one
        const float4* __restrict__ src = ((passes - p) & 1) ? xbuf + lo : pts + lo;     // what pass p-1 wrote
        float4* __restrict__ dst = ((passes - 1 - p) & 1) ? xbuf + lo : pts + lo;
        // exclusive scan of this pass' histogram
        uint32_t hv = 0, incl = 0;
        if (tid < CS_BINS) {
            hv = 0;
#pragma unroll
            for (int r = 0; r < CS_HREP; ++r) hv += hist[p][r][tid];
            incl = wave_scan_incl(hv);
            if (l == 63) wsum[w] = incl;
        }
        __syncthreads();
        if (tid < CS_BINS) {
            uint32_t b = incl - hv;
            for (int w2 = 0; w2 < w; ++w2) b += wsum[w2];
            base[tid] = b;
        }
        __syncthreads();
        auto fetch = [&](int t0, float4 (&out)[CS_ROUNDS]) {
            const int seg = t0 + w * (64 * CS_ROUNDS);
#pragma unroll
            for (int r = 0; r < CS_ROUNDS; ++r) {
                const int i = seg + r * 64 + l;
                const int j = i < cn ? i : 0;
                if constexpr (FIRST) {
                    const Row3 q = rows[j];
                    out[r].x = q.x; out[r].y = q.y; out[r].z = q.z;
                    out[r].w = __uint_as_float((uint32_t)(lo + j));
                } else {
                    out[r] = src[j];
                }
            }
        };
        float4 nxt[CS_ROUNDS];
        auto tile = [&](int t0, auto FULL_T) {
            constexpr bool FULL = decltype(FULL_T)::value;
            float4 row[CS_ROUNDS];
            uint32_t key[CS_ROUNDS], rank[CS_ROUNDS];
            const int seg = t0 + w * (64 * CS_ROUNDS);
#pragma unroll
            for (int r = 0; r < CS_ROUNDS; ++r) row[r] = nxt[r];
            if (t0 + CS_TILE < cn) fetch(t0 + CS_TILE, nxt);        // in flight while this tile is ranked
#ifdef PCH_CS_STAMPS
            tlast = wall_clock64();
#endif
#pragma unroll
            for (int r = 0; r < CS_ROUNDS; ++r) {
                const bool valid = FULL || seg + r * 64 + l < cn;
                bool ok;
                key[r] = cs_cell_key(g, row[r].x, row[r].y, row[r].z, ok);
                const uint32_t d = (relkey(key[r]) >> shift) & mask;
                uint32_t np;
                const uint32_t rk = dbits <= 8 ? wave_match<8>(d, valid, np) : wave_match<9>(d, valid, np);
                const uint32_t prior = cnt[w][d];
                __builtin_amdgcn_wave_barrier();
                if (valid && rk == 0) cnt[w][d] = prior + np;
                __builtin_amdgcn_wave_barrier();
                rank[r] = prior + rk;
            }
            __syncthreads();
            CS_TILE_STAMP(0);
            if (tid < CS_BINS) {                        // digit tid: waves in order, counters cleared for the next tile
                uint32_t run = base[tid];
#pragma unroll
                for (int w2 = 0; w2 < CS_WAVES; ++w2) {
                    const uint32_t cc = cnt[w2][tid];
                    off[w2][tid] = run;
                    cnt[w2][tid] = 0;
                    run += cc;
                }
                base[tid] = run;
            }
            __syncthreads();
            CS_TILE_STAMP(1);
#pragma unroll
            for (int r = 0; r < CS_ROUNDS; ++r) {
                if (FULL || seg + r * 64 + l < cn) {
                    const uint32_t d = (relkey(key[r]) >> shift) & mask;
                    const uint32_t pos = off[w][d] + rank[r];
                    dst[pos] = row[r];
                    if constexpr (LAST) keys_out[lo + pos] = hi | key[r];
                }
            }
            // off[] is rewritten only behind the next tile's first barrier, which every wave reaches
            // after these reads
            CS_TILE_STAMP(2);
        };
        fetch(0, nxt);
        int t0 = 0;
        for (; t0 + CS_TILE <= cn; t0 += CS_TILE) tile(t0, std::true_type{});
        if (t0 < cn) tile(t0, std::false_type{});
        __syncthreads();                                // this pass' stores are visible to the whole workgroup
        CS_STAMP();
    };
    for (int p = 0; p < passes; ++p) {
        const bool first = p == 0, last = p == passes - 1;
        if (first && last) run_pass(p, std::true_type{}, std::true_type{});
        else if (first)    run_pass(p, std::true_type{}, std::false_type{});
        else if (last)     run_pass(p, std::false_type{}, std::true_type{});
        else               run_pass(p, std::false_type{}, std::false_type{});
    }
#ifdef PCH_CS_STAMPS
    __syncthreads();
    CS_STAMP();
    if (stamps && blockIdx.x == gridDim.x / 2 && threadIdx.x == 0) for (int k = 0; k < 3; ++k) stamps[8 + k] = tacc[k];
#endif
}

// ---- cells of the sorted rows.  A row is a cell head when its key differs from the key in front of it; nobody
// materialises those flags: db_heads_k counts them per SCAN_TILE rows, scan_tile_sums_u32 turns the counts into tile
// offsets (and the number of cells), and db_cells_k scans its tile again while it writes what depends on the cell
// index.  (Until round 3 the sort kernels wrote a flag per row, a three-launch scan rewrote it and db_cells_k read
// it back - and the chunk sort's own flag sweep ran on 196 workgroups.)
constexpr int DC_ITEMS = SCAN_TILE / DB_THREADS;          // 8 consecutive rows per thread
static_assert(DC_ITEMS == 8, "two 64-byte key loads per thread");

__device__ __forceinline__ void dc_load_keys(const uint64_t* __restrict__ keys, int64_t base, int64_t n,
                                             uint64_t (&k)[DC_ITEMS], uint64_t& prev) {
    if (base + DC_ITEMS <= n) {
        const ulonglong2* q = reinterpret_cast<const ulonglong2*>(keys + base);     // base is a multiple of 8
#pragma unroll
        for (int j = 0; j < DC_ITEMS / 2; ++j) { const ulonglong2 v = q[j]; k[2 * j] = v.x; k[2 * j + 1] = v.y; }
    } else {
#pragma unroll
        for (int j = 0; j < DC_ITEMS; ++j) k[j] = base + j < n ? keys[base + j] : 0ull;
    }
    prev = (base > 0 && base < n) ? keys[base - 1] : 0ull;
}

__global__ __launch_bounds__(DB_THREADS) void db_heads_k(const uint64_t* __restrict__ keys, int64_t n,
                                                         uint32_t* __restrict__ tile_sums) {
    __shared__ uint32_t wsum[DB_WAVES];
    const int64_t base = (int64_t)blockIdx.x * SCAN_TILE + (int64_t)threadIdx.x * DC_ITEMS;
    uint64_t k[DC_ITEMS], prev;
    dc_load_keys(keys, base, n, k, prev);
    uint32_t heads = 0;
#pragma unroll
    for (int j = 0; j < DC_ITEMS; ++j) {
        const int64_t i = base + j;
        heads += (i < n && (i == 0 || k[j] != (j ? k[j - 1] : prev))) ? 1u : 0u;
    }
    heads = wave_reduce_add(heads);
    if (lane_id() == 0) wsum[wave_id()] = heads;
    __syncthreads();
    if (threadIdx.x == 0) {
        uint32_t t = 0;
#pragma unroll
        for (int w = 0; w < DB_WAVES; ++w) t += wsum[w];
        tile_sums[blockIdx.x] = t;
    }
}

__global__ __launch_bounds__(DB_THREADS) void db_cells_k(const uint64_t* __restrict__ keys,
                                                         const uint32_t* __restrict__ tile_excl, int64_t n,
                                                         int sh, int64_t nchunks,
                                                         uint32_t* __restrict__ cid,
                                                         uint32_t* __restrict__ cell_start,
                                                         uint64_t* __restrict__ cell_key,
                                                         uint32_t* __restrict__ chunk_cells,
                                                         uint32_t* __restrict__ cell_acc) {
    __shared__ uint32_t wsum[DB_WAVES];
    const int64_t base = (int64_t)blockIdx.x * SCAN_TILE + (int64_t)threadIdx.x * DC_ITEMS;
    uint64_t k[DC_ITEMS], prev;
    dc_load_keys(keys, base, n, k, prev);
    bool hd[DC_ITEMS];
    uint32_t heads = 0;
#pragma unroll
    for (int j = 0; j < DC_ITEMS; ++j) {
        const int64_t i = base + j;
        hd[j] = i < n && (i == 0 || k[j] != (j ? k[j - 1] : prev));
        heads += hd[j] ? 1u : 0u;
    }
    const uint32_t incl = wave_scan_incl(heads);
    if (lane_id() == 63) wsum[wave_id()] = incl;
    __syncthreads();
    uint32_t c = tile_excl[blockIdx.x] + incl - heads;           // heads in front of this thread's first row
    for (int w = 0; w < wave_id(); ++w) c += wsum[w];
    uint32_t cc[DC_ITEMS];
#pragma unroll
    for (int j = 0; j < DC_ITEMS; ++j) {
        const int64_t i = base + j;
        c += hd[j] ? 1u : 0u;
        cc[j] = c - 1u;                                          // cell of row i (row 0 is a head, so c >= 1)
        if (hd[j]) {
            const uint64_t key = k[j], pk = j ? k[j - 1] : prev;
            cell_start[cc[j]] = (uint32_t)i;
            cell_key[cc[j]] = key;
            uint4* a4 = reinterpret_cast<uint4*>(cell_acc + 8 * (int64_t)cc[j]);   // neutral start of db_cellstats_k
            a4[0] = make_uint4(0u, 0u, 0u, 0u);
            a4[1] = make_uint4(0u, 0u, 0u, 0u);
            const uint64_t ch = sh < 64 ? key >> sh : 0, pch = sh < 64 ? pk >> sh : 0;
            if (i == 0 || ch != pch) chunk_cells[ch] = cc[j];    // every chunk holds rows, hence cells
        }
        if (i == n - 1) { cell_start[cc[j] + 1] = (uint32_t)n; chunk_cells[nchunks] = cc[j] + 1; }
    }
    if (base + DC_ITEMS <= n) {
        uint4* o = reinterpret_cast<uint4*>(cid + base);
        o[0] = make_uint4(cc[0], cc[1], cc[2], cc[3]);
        o[1] = make_uint4(cc[4], cc[5], cc[6], cc[7]);
    } else {
#pragma unroll
        for (int j = 0; j < DC_ITEMS; ++j)
            if (base + j < n) cid[base + j] = cc[j];
    }
}
// ---- neighbour rows of one cell: lanes 0..24 each binary-search one (dy,dz) row ---------
struct RowSet {
    int ca[DB_ROWS], cb[DB_ROWS];      // cell index range of every row
};

__device__ __forceinline__ int db_lower(const uint64_t* __restrict__ a, int m, uint64_t k) {
    int lo = 0, hi = m;
    while (lo < hi) { const int mid = (lo + hi) >> 1; if (a[mid] < k) lo = mid + 1; else hi = mid; }
    return lo;
}
__device__ __forceinline__ int db_upper(const uint64_t* __restrict__ a, int m, uint64_t k) {
    int lo = 0, hi = m;
    while (lo < hi) { const int mid = (lo + hi) >> 1; if (a[mid] <= k) lo = mid + 1; else hi = mid; }
    return lo;
}

__device__ __forceinline__ void db_rows(const DbGrid& g, const uint64_t* __restrict__ cell_key, int m,
                                        uint64_t key, RowSet* __restrict__ rs,
                                        const int2* __restrict__ rowtab = nullptr, int cell = 0) {
    const int l = lane_id();
    if (rowtab) {                                          // computed once by db_rowtab_k
        if (l < DB_ROWS) {
            const int2 v = rowtab[(int64_t)cell * DB_ROWS + l];
            rs->ca[l] = v.x;
            rs->cb[l] = v.y;
        }
        __builtin_amdgcn_wave_barrier();
        return;
    }
    if (l < DB_ROWS) {
        const uint64_t cx = key & ((1ull << g.bx) - 1);
        const uint64_t cy = (key >> g.bx) & ((1ull << g.by) - 1);
        const uint64_t cz = (key >> (g.bx + g.by)) & ((1ull << g.bz) - 1);
        const int sh = g.bx + g.by + g.bz;
        const uint64_t chunk = sh < 64 ? (key >> sh) : 0;
        const int ny = (int)cy + DB_ROW_DY[l], nz = (int)cz + DB_ROW_DZ[l];
        int a = 0, b = 0;
        if (ny >= 0 && ny <= g.my && nz >= 0 && nz <= g.mz) {
            const int xlo = (int)cx - 2 < 0 ? 0 : (int)cx - 2;
            const int xhi = (int)cx + 2 > g.mx ? g.mx : (int)cx + 2;
            const int c0 = (int)g.chunk_cells[chunk], c1 = (int)g.chunk_cells[chunk + 1];   // neighbours share the chunk
            a = c0 + db_lower(cell_key + c0, c1 - c0, db_pack(g, chunk, (uint64_t)nz, (uint64_t)ny, (uint64_t)xlo));
            b = c0 + db_upper(cell_key + c0, c1 - c0, db_pack(g, chunk, (uint64_t)nz, (uint64_t)ny, (uint64_t)xhi));
        }
        rs->ca[l] = a;
        rs->cb[l] = b;
    }
    __builtin_amdgcn_wave_barrier();
}

// neighbour-row table: [cell][25] cell-index ranges, shared by the core / union / border kernels
__global__ __launch_bounds__(DB_THREADS) void db_rowtab_k(DbGrid g, const uint64_t* __restrict__ cell_key,
                                                          int m, int2* __restrict__ rowtab) {
    const int c = (blockIdx.x * DB_WAVES + wave_id()) * 2 + (lane_id() >> 5);   // two cells per wave
    const int l = lane_id() & 31;
    if (c >= m || l >= DB_ROWS) return;
    const uint64_t key = cell_key[c];
    const uint64_t cx = key & ((1ull << g.bx) - 1);
    const uint64_t cy = (key >> g.bx) & ((1ull << g.by) - 1);
    const uint64_t cz = (key >> (g.bx + g.by)) & ((1ull << g.bz) - 1);
    const int sh = g.bx + g.by + g.bz;
    const uint64_t chunk = sh < 64 ? (key >> sh) : 0;
    const int ny = (int)cy + DB_ROW_DY[l], nz = (int)cz + DB_ROW_DZ[l];
    int2 v;
    v.x = 0; v.y = 0;
    if (ny >= 0 && ny <= g.my && nz >= 0 && nz <= g.mz) {
        const int xlo = (int)cx - 2 < 0 ? 0 : (int)cx - 2;
        const int xhi = (int)cx + 2 > g.mx ? g.mx : (int)cx + 2;
        const int c0 = (int)g.chunk_cells[chunk], c1 = (int)g.chunk_cells[chunk + 1];   // neighbours share the chunk
        v.x = c0 + db_lower(cell_key + c0, c1 - c0, db_pack(g, chunk, (uint64_t)nz, (uint64_t)ny, (uint64_t)xlo));
        // the end of the run: at most five cells (x - 2 .. x + 2) follow v.x, their keys ascending - five loads side by
        // side instead of a second binary search (fourteen dependent loads in a chunk of 11 000 cells)
        const uint64_t khi = db_pack(g, chunk, (uint64_t)nz, (uint64_t)ny, (uint64_t)xhi);
        uint64_t kk[5];
#pragma unroll
        for (int i = 0; i < 5; ++i) kk[i] = v.x + i < c1 ? cell_key[v.x + i] : ~0ull;
        int cntx = 0;
#pragma unroll
        for (int i = 0; i < 5; ++i) cntx += kk[i] <= khi ? 1 : 0;
        v.y = v.x + cntx;
    }
    rowtab[(int64_t)c * DB_ROWS + l] = v;
}

// ---- core points ---------------------------------------------------------------------
// one wave per cell.  Dense cell: all core.  Sparse cell: n-body tile loop - every lane owns one
// query point of the cell, candidate tiles (64 points of the sorted neighbour runs) are staged
// once in LDS and broadcast to all queries; the wave leaves as soon as every query has reached
// min_samples (checked per tile).  Each candidate is loaded once per cell, not once per query.
// COUNT (measurement builds of the SAME control flow, pch_dbscan_set_pair_counting): the wave tallies the distance
// tests it makes - useful ones (a real query against a real candidate) and issued lane slots (64 per
// wave-instruction group, padding and idle lanes included) - and adds them to stats[0..3] once, at its end:
// [0] useful pair tests, [1] issued lane slots, [2] cells that reached the test path, [3] LDS tiles staged.
template <bool COUNT>
__global__ __launch_bounds__(DB_THREADS) void db_core_k(DbGrid g, const float4* __restrict__ pts,
                                                        const uint32_t* __restrict__ cell_start,
                                                        const uint64_t* __restrict__ cell_key, int m,
                                                        const int2* __restrict__ rowtab,
                                                        uint8_t* __restrict__ core_s,
                                                        uint32_t* __restrict__ cell_ncore,
                                                        unsigned long long* __restrict__ stats) {
    unsigned long long n_useful = 0, n_slots = 0, n_tiles = 0;
    auto tally = [&]() {
        if (COUNT && lane_id() == 0) {
            atomicAdd(&stats[0], n_useful); atomicAdd(&stats[1], n_slots);
            atomicAdd(&stats[2], 1ull); atomicAdd(&stats[3], n_tiles);
        }
    };
    __shared__ RowSet rows[DB_WAVES];
    __shared__ DbPair tiles[DB_WAVES][32];
    __shared__ uint32_t seg_a[DB_WAVES][DB_SEGS], seg_b[DB_WAVES][DB_SEGS];
    const int c = blockIdx.x * DB_WAVES + wave_id();
    if (c >= m) return;
    const int l = lane_id();
    const uint32_t s = cell_start[c], e = cell_start[c + 1];
    const int cnt = (int)(e - s);
    {
        const int sh = g.bx + g.by + g.bz;
        if (g.chunk_bad[sh < 64 ? (cell_key[c] >> sh) : 0]) {            // points of NaN/inf chunks
            for (uint32_t i = s + l; i < e; i += 64) core_s[i] = 0;
            if (l == 0) cell_ncore[c] = 0;
            return;
        }
    }
    if (cnt >= g.min_samples) {
        for (uint32_t i = s + l; i < e; i += 64) core_s[i] = 1;
        if (l == 0) cell_ncore[c] = (uint32_t)cnt;
        return;
    }
    RowSet* rs = &rows[wave_id()];
    db_rows(g, cell_key, m, cell_key[c], rs, rowtab, c);
    // candidate total: if even all candidates together are too few, nobody is core
    long long tot = 0;
    if (l < DB_ROWS) tot = (long long)cell_start[rs->cb[l]] - (long long)cell_start[rs->ca[l]];
    tot = wave_reduce_add(tot);
    if (tot < (long long)g.min_samples) {
        for (uint32_t i = s + l; i < e; i += 64) core_s[i] = 0;
        if (l == 0) cell_ncore[c] = 0;
        return;
    }
    // candidate segments in nearest-first order: the 27-cell neighbourhood first (x-1..x+1 of the
    // nine nearest runs), then the x-2 / x+2 ends of those runs, then the sixteen outer runs
    uint32_t* sa = seg_a[wave_id()];
    uint32_t* sb = seg_b[wave_id()];
    {
        const uint64_t xmask = (1ull << g.bx) - 1;
        const int cx = (int)(cell_key[c] & xmask);
        if (l < DB_ROWS) {
            const int ca = rs->ca[l], cb = rs->cb[l];
            if (l < 9) {
                int ia = ca, ib = cb;
                if (ca < cb) {
                    if ((int)(cell_key[ca] & xmask) == cx - 2) ia = ca + 1;
                    if (ib > ia && (int)(cell_key[cb - 1] & xmask) == cx + 2) ib = cb - 1;
                }
                sa[l] = cell_start[ia];      sb[l] = cell_start[ib];          // inner part
                sa[9 + l] = cell_start[ca];  sb[9 + l] = cell_start[ia];      // x-2 end
                sa[18 + l] = cell_start[ib]; sb[18 + l] = cell_start[cb];     // x+2 end
            } else {
                sa[18 + l] = cell_start[ca]; sb[18 + l] = cell_start[cb];     // outer runs: slots 27..42
            }
        }
        __builtin_amdgcn_wave_barrier();
    }
    uint32_t ncore = 0;
    const bool long_sweep = tot >= DB_LONG_TOT;            // wave-uniform
    if (cnt < DB_FEW_QUERIES) {
        // a handful of queries (cluster fringe): lanes sweep the candidates of one query at a
        // time and leave at min_samples - usually within the first tile of a dense neighbour
        for (uint32_t q = s; q < e; ++q) {
            const float4 qp = pts[q];
            int count = 0;
            if (long_sweep) {
                // A lone point beside a tower core that is NOT core passes every candidate of its neighbourhood (13 585
                // in the bench tile) - one dependent load per 64 candidates made such a wave the tail of the kernel.
                // Where the neighbourhood is that large, DB_AHEAD groups per trip, their loads in flight together (the
                // same candidates in the same order; a sweep leaves at min_samples either way).
                for (int r = 0; r < DB_SEGS && count < g.min_samples; ++r) {
                    const uint32_t pa = sa[r], pb = sb[r];
                    for (uint32_t j0 = pa; j0 < pb && count < g.min_samples; j0 += 64u * DB_AHEAD) {
                        float4 P[DB_AHEAD];
#pragma unroll
                        for (int u = 0; u < DB_AHEAD; ++u) {
                            const uint32_t j = j0 + 64u * u + l;
                            P[u] = pts[j < pb ? j : pb - 1];
                        }
#pragma unroll
                        for (int u = 0; u < DB_AHEAD; ++u) {
                            const uint32_t jb = j0 + 64u * u;
                            const bool hit = jb + l < pb && db_within2(qp, P[u], g);
                            count += (int)__popcll(__ballot(hit));
                            if (COUNT && jb < pb) { n_useful += (pb - jb) < 64u ? (pb - jb) : 64u; n_slots += 64; }
                        }
                    }
                }
            } else {
                for (int r = 0; r < DB_SEGS && count < g.min_samples; ++r) {
                    const uint32_t pa = sa[r], pb = sb[r];
                    for (uint32_t j0 = pa; j0 < pb && count < g.min_samples; j0 += 64) {
                        const uint32_t j = j0 + l;
                        bool hit = false;
                        if (j < pb) hit = db_within2(qp, pts[j], g);
                        count += (int)__popcll(__ballot(hit));
                        if (COUNT) { n_useful += (pb - j0) < 64u ? (pb - j0) : 64u; n_slots += 64; }
                    }
                }
            }
            const bool is_core = count >= g.min_samples;
            if (l == 0) core_s[q] = is_core ? 1 : 0;
            ncore += is_core;
        }
        if (l == 0) cell_ncore[c] = ncore;
        tally();
        return;
    }
    DbPair* tile = tiles[wave_id()];
    for (uint32_t q0 = s; q0 < e; q0 += 64) {              // 64 query points per round
        const bool valid = q0 + l < e;
        float4 Q;
        Q.x = Q.y = Q.z = 3.0e37f; Q.w = 0.0f;             // idle lanes sit far away (finite)
        if (valid) Q = pts[q0 + l];
        int count = 0;
        unsigned long long active = __ballot(valid);        // queries still below min_samples
        int r = 0;
        uint32_t j0 = sa[0];
        // phase 1: all queries against one staged candidate tile at a time, while enough of
        // them are still counting to keep the lanes busy
        while (r < DB_SEGS && __popcll(active) >= DB_FEW_QUERIES / 2) {
            const uint32_t pb = sb[r];
            if (j0 >= pb) { ++r; if (r < DB_SEGS) j0 = sa[r]; continue; }
            const int nj = (int)((pb - j0) < 64u ? (pb - j0) : 64u);
            float4 P;
            P.x = P.y = P.z = 3.0e38f; P.w = 0.0f;         // padding: squared distance overflows to +inf
            if (l < nj) P = pts[j0 + l];
            __builtin_amdgcn_wave_barrier();
            db_tile_put(tile, l, P);
            __builtin_amdgcn_wave_barrier();
            count += db_tile_count(Q, tile, nj, g);
            if (COUNT) {
                n_useful += (unsigned long long)__popcll(__ballot(valid)) * (unsigned)nj;
                n_slots += 64ull * (unsigned)((nj + 7) & ~7);
                ++n_tiles;
            }
            j0 += 64;
            active = __ballot(valid && count < g.min_samples);
        }
        // phase 2: the few stragglers one by one, lanes over the remaining candidates
        while (active) {
            const int ql = (int)__builtin_ctzll(active);
            active &= active - 1;
            float4 qp;
            qp.x = __shfl(Q.x, ql, 64); qp.y = __shfl(Q.y, ql, 64); qp.z = __shfl(Q.z, ql, 64); qp.w = 0.0f;
            int cq = __shfl(count, ql, 64);
            int rr = r;
            uint32_t jj = j0;
            if (long_sweep) {                              // (as above)
                while (rr < DB_SEGS && cq < g.min_samples) {
                    const uint32_t pb = sb[rr];
                    if (jj >= pb) { ++rr; if (rr < DB_SEGS) jj = sa[rr]; continue; }
                    float4 P[DB_AHEAD];
#pragma unroll
                    for (int u = 0; u < DB_AHEAD; ++u) {
                        const uint32_t j = jj + 64u * u + l;
                        P[u] = pts[j < pb ? j : pb - 1];
                    }
#pragma unroll
                    for (int u = 0; u < DB_AHEAD; ++u) {
                        const uint32_t jb = jj + 64u * u;
                        const bool hit = jb + l < pb && db_within2(qp, P[u], g);
                        cq += (int)__popcll(__ballot(hit));
                        if (COUNT && jb < pb) { n_useful += (pb - jb) < 64u ? (pb - jb) : 64u; n_slots += 64; }
                    }
                    jj += 64u * DB_AHEAD;
                }
            } else {
                while (rr < DB_SEGS && cq < g.min_samples) {
                    const uint32_t pb = sb[rr];
                    if (jj >= pb) { ++rr; if (rr < DB_SEGS) jj = sa[rr]; continue; }
                    const uint32_t j = jj + l;
                    bool hit = false;
                    if (j < pb) hit = db_within2(qp, pts[j], g);
                    cq += (int)__popcll(__ballot(hit));
                    if (COUNT) { n_useful += (pb - jj) < 64u ? (pb - jj) : 64u; n_slots += 64; }
                    jj += 64;
                }
            }
            if (l == ql) count = cq;
        }
        const bool is_core = valid && count >= g.min_samples;
        if (valid) core_s[q0 + l] = is_core ? 1 : 0;
        ncore += (uint32_t)__popcll(__ballot(is_core));
    }
    if (l == 0) cell_ncore[c] = ncore;
    tally();
}

// ---- per-cell box of the core points, union-find init ----------------------------------
// ---- per-cell statistics of the core points: bounding box and smallest original row ----------
// Points-parallel (a dense tower cell holds thousands of points - one wave per cell would leave a
// long tail): a wave walks 1024 consecutive sorted points, 64 per round.  Inside a round the
// values are combined by a segmented scan over the lanes (the points are sorted by cell, so a
// cell is a run of lanes); a run that ends inside the round is folded into the cell's accumulator
// with atomics, the run that reaches lane 63 is carried into the next round.  Accumulators hold
// ordered uint32 keys folded with atomicMax (minima as the complement), so 0 = nothing yet.
constexpr int DB_CS_ROUNDS = 16;

__device__ __forceinline__ void db_cellstats_flush(uint32_t* __restrict__ acc, uint32_t c, const uint32_t (&v)[7]) {
#pragma unroll
    for (int k = 0; k < 7; ++k)
        if (v[k]) atomicMax(&acc[8 * (int64_t)c + k], v[k]);
}

__global__ __launch_bounds__(DB_THREADS) void db_cellstats_k(const float4* __restrict__ pts,
                                                             const uint32_t* __restrict__ cid,
                                                             const uint8_t* __restrict__ core_s, int64_t n,
                                                             uint32_t* __restrict__ acc,
                                                             uint32_t* __restrict__ bits, int64_t nw) {
    {   // the row bitmap db_mark_k sets bits in starts empty: n/32 words, and this grid has n/16 threads
        const int64_t t = (int64_t)blockIdx.x * DB_THREADS + threadIdx.x;
        if (t < nw) bits[t] = 0u;
    }
    const int64_t base = ((int64_t)blockIdx.x * DB_WAVES + wave_id()) * (64 * DB_CS_ROUNDS);
    if (base >= n) return;
    const int l = lane_id();
    constexpr uint32_t NONE = 0xFFFFFFFFu;
    uint32_t carry_c = NONE;                                   // cell of the run that reached lane 63 ...
    uint32_t carry[7] = {0, 0, 0, 0, 0, 0, 0};                 // ... its values so far (wave-uniform) ...
    uint32_t mine[7] = {0, 0, 0, 0, 0, 0, 0};                  // ... plus whole rounds of it, still per lane
    auto fold_mine = [&]() {
#pragma unroll
        for (int k = 0; k < 7; ++k) {
            const uint32_t t = wave_reduce_max(mine[k]);
            carry[k] = t > carry[k] ? t : carry[k];
            mine[k] = 0;
        }
    };
    for (int r0 = 0; r0 < DB_CS_ROUNDS; r0 += 4) {
    if (base + r0 * 64 >= n) break;
    uint32_t c4[4], core4[4];
    float4 p4[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) {                              // four rounds of loads in flight
        const int64_t i = base + (r0 + u) * 64 + l;
        const bool in = i < n;
        c4[u] = in ? cid[i] : NONE - 1u;                       // past the end: a run of its own
        core4[u] = in ? core_s[i] : 0u;
        p4[u] = pts[in ? i : base];
    }
#pragma unroll
    for (int u = 0; u < 4; ++u) {
        const int64_t i = base + (r0 + u) * 64 + l;
        const bool in = i < n;
        const uint32_t c = c4[u];
        uint32_t v[7] = {0, 0, 0, 0, 0, 0, 0};
        if (core4[u]) {
            const float4 p = p4[u];
            const uint32_t kx = f32_ordered(p.x), ky = f32_ordered(p.y), kz = f32_ordered(p.z);
            v[0] = ~kx; v[1] = ~ky; v[2] = ~kz; v[3] = kx; v[4] = ky; v[5] = kz;
            v[6] = ~__float_as_uint(p.w);                      // original row (< 2^31)
        }
        if (__ballot(c != carry_c) == 0) {                     // the whole round belongs to the carried cell
#pragma unroll
            for (int k = 0; k < 7; ++k) mine[k] = v[k] > mine[k] ? v[k] : mine[k];
            continue;
        }
        if (carry_c != NONE) fold_mine();
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) {                     // segmented inclusive max-scan
            const uint32_t cu = __shfl_up(c, o, 64);
            const bool same = l >= o && cu == c;
#pragma unroll
            for (int k = 0; k < 7; ++k) {
                const uint32_t t = __shfl_up(v[k], o, 64);
                if (same) v[k] = t > v[k] ? t : v[k];
            }
        }
        const uint32_t c0 = (uint32_t)__builtin_amdgcn_readlane((int)c, 0);
        if (carry_c != NONE) {
            if (carry_c == c0) {                               // the carried run continues: its end lane needs the carry
                if (c == c0) {
#pragma unroll
                    for (int k = 0; k < 7; ++k) v[k] = carry[k] > v[k] ? carry[k] : v[k];
                }
            } else if (l == 0) {
                db_cellstats_flush(acc, carry_c, carry);
            }
        }
        const uint32_t cn = __shfl_down(c, 1, 64);
        const bool run_end = l == 63 || cn != c;
        if (run_end && l != 63 && in) db_cellstats_flush(acc, c, v);
        carry_c = (uint32_t)__builtin_amdgcn_readlane((int)c, 63);
        if (carry_c == NONE - 1u) carry_c = NONE;              // lane 63 is past the end: nothing to carry
#pragma unroll
        for (int k = 0; k < 7; ++k) carry[k] = (uint32_t)__builtin_amdgcn_readlane((int)v[k], 63);
    }
    }
    if (carry_c != NONE) {
        fold_mine();
        if (l == 0) db_cellstats_flush(acc, carry_c, carry);
    }
}

// accumulators -> bounding box floats (no core point: +inf / -inf) and smallest core row; also
// resets the union-find forest
__global__ __launch_bounds__(DB_THREADS) void db_cellfin_k(const uint32_t* __restrict__ acc, int m,
                                                           float* __restrict__ cell_box,
                                                           int* __restrict__ cell_min,
                                                           int* __restrict__ parent,
                                                           int* __restrict__ comp_min) {
    const int c = blockIdx.x * DB_THREADS + threadIdx.x;
    if (c >= m) return;
    uint32_t a[7];
#pragma unroll
    for (int k = 0; k < 7; ++k) a[k] = acc[8 * (int64_t)c + k];
    float box[6];
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        box[k] = INFINITY;
        box[3 + k] = -INFINITY;
        if (a[k] != 0u) box[k] = f32_unordered(~a[k]);
        if (a[3 + k] != 0u) box[3 + k] = f32_unordered(a[3 + k]);
    }
#pragma unroll
    for (int k = 0; k < 6; ++k) cell_box[6 * (int64_t)c + k] = box[k];
    int mn = INT_BIG;
    if (a[6] != 0u) mn = (int)(~a[6]);
    cell_min[c] = mn;
    parent[c] = c;
    comp_min[c] = INT_BIG;
}

// ---- wave-wide test: do cells A and B hold a pair of core points within eps? -------------------
// A plain double loop finds a hit at once when most pairs are hits, but between two large cells
// that touch only at a corner it can scan all of B for thousands of A points before it reaches one
// that has a partner.  So first a few steps of alternating nearest-point descent (the point of A
// nearest to B's box, its nearest point in B, that one's nearest point in A, ...) - every
// candidate pair is checked with the exact predicate, so a hit is a proof; only if the descent
// stalls above eps does the exhaustive search run, the A side filtered 64 points at a time.
__device__ __forceinline__ float db_d2f(const float4& a, const float4& b) {
    const float dx = a.x - b.x, dy = a.y - b.y, dz = a.z - b.z;
    return __builtin_fmaf(dz, dz, __builtin_fmaf(dy, dy, dx * dx));
}
__device__ __forceinline__ float db_box_d2f(const float4& q, const float* __restrict__ box) {
    const float gx = fmaxf(fmaxf(box[0] - q.x, q.x - box[3]), 0.0f);
    const float gy = fmaxf(fmaxf(box[1] - q.y, q.y - box[4]), 0.0f);
    const float gz = fmaxf(fmaxf(box[2] - q.z, q.z - box[5]), 0.0f);
    return __builtin_fmaf(gz, gz, __builtin_fmaf(gy, gy, gx * gx));
}
// index of the core point of [s, e) that minimises f (lane-parallel, four loads in flight); -1: none
template <typename F>
__device__ __forceinline__ int db_argmin(const float4* __restrict__ pts, const uint8_t* __restrict__ core_s,
                                         uint32_t s, uint32_t e, bool dense, F f) {
    const int l = lane_id();
    unsigned long long best = ~0ull;
    for (uint32_t i0 = s; i0 < e; i0 += 256) {
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const uint32_t i = i0 + 64 * u + l;
            if (i < e && (dense || core_s[i])) {
                const float d = f(pts[i]);
                const unsigned long long k = ((unsigned long long)__float_as_uint(d) << 32) | i;   // d >= 0: bits order
                best = k < best ? k : best;
            }
        }
    }
    best = wave_reduce_min(best);
    return best == ~0ull ? -1 : (int)(uint32_t)best;
}

__device__ __forceinline__ bool db_cells_connected(const DbGrid& g, const float4* __restrict__ pts,
                                                   const uint8_t* __restrict__ core_s, uint32_t as, uint32_t ae,
                                                   bool a_dense, uint32_t bs, uint32_t be, bool b_dense,
                                                   const float* __restrict__ boxB, uint32_t a_from) {
    const int l = lane_id();
    if (ae - as > 64 || be - bs > 64) {                    // descent only pays between larger cells
        int ia = db_argmin(pts, core_s, as, ae, a_dense, [&](const float4& p) { return db_box_d2f(p, boxB); });
        if (ia < 0) return false;
        float4 q = pts[ia];
        for (int it = 0; it < 3; ++it) {
            const int jb = db_argmin(pts, core_s, bs, be, b_dense, [&](const float4& p) { return db_d2f(q, p); });
            if (jb < 0) return false;
            const float4 pb = pts[jb];
            if (db_within2(q, pb, g)) return true;
            const int ia2 = db_argmin(pts, core_s, as, ae, a_dense, [&](const float4& p) { return db_d2f(p, pb); });
            const float4 pa = pts[ia2];
            if (db_within2(pa, pb, g)) return true;
            if (ia2 == ia) break;                          // a local minimum above eps: decide exhaustively
            ia = ia2;
            q = pa;
        }
    }
    for (uint32_t a0 = a_from; a0 < ae; a0 += 64) {
        const uint32_t ia = a0 + l;
        float4 pa;
        pa.x = pa.y = pa.z = pa.w = 0.0f;
        bool near = false;
        if (ia < ae && (a_dense || core_s[ia])) {
            pa = pts[ia];
            near = !(db_box_d2(pa, boxB) > g.eps2);
        }
        unsigned long long todo = __ballot(near);
        while (todo) {
            const int src = (int)__builtin_ctzll(todo);
            todo &= todo - 1;
            float4 q;
            q.x = __shfl(pa.x, src, 64); q.y = __shfl(pa.y, src, 64); q.z = __shfl(pa.z, src, 64); q.w = 0.0f;
            for (uint32_t j0 = bs; j0 < be; j0 += 64) {
                const uint32_t j = j0 + l;
                bool hit = false;
                if (j < be && (b_dense || core_s[j])) hit = db_within2(q, pts[j], g);
                if (__ballot(hit)) return true;
            }
        }
    }
    return false;
}

// ---- union-find over cells (hook larger root under smaller; lock free) ------------------
// Invariant: parent[x] <= x, only roots (parent[x] == x) are ever hooked, and only by a CAS, so
// every value ever stored in parent[x] is an ancestor of x.  Loads bypass the (incoherent) L1;
// path compression is a PLAIN store of an ancestor - racing writers all write ancestors, and a
// stale read merely costs an extra hop or a failed CAS, whose return value is the truth.
__device__ __forceinline__ int uf_load(int* p) {
    return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ int uf_find(int* __restrict__ parent, int x) {
    int p = uf_load(&parent[x]);
    while (p != x) {
        const int gp = uf_load(&parent[p]);
        if (gp != p) __hip_atomic_store(&parent[x], gp, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        x = p;
        p = gp;
    }
    return x;
}
// a, b: any members (ideally already roots) of the two sets
__device__ __forceinline__ void uf_union(int* __restrict__ parent, int a, int b) {
    for (;;) {
        a = uf_find(parent, a);
        b = uf_find(parent, b);
        if (a == b) return;
        if (a < b) { const int t = a; a = b; b = t; }   // a > b: hook a under b
        const int old = atomicCAS(&parent[a], a, b);
        if (old == a) return;
        a = old;                                          // somebody hooked a first: follow it
    }
}

// one wave per core cell A.  Phase 1 (64 candidate cells at a time, one per lane): neighbour core
// cells B > A whose core boxes are within eps and whose root differs from A's survive.
// Phase 2 (wave-wide per survivor): look for one core pair within eps (lanes over B's points,
// scalar loop over A's points, leave at the first hit), then unite.
// db_union_face_k has looked at the (up to 3) face-adjacent cells with a larger index before: for
// dense data they connect at the first tile and leave almost nothing but root comparisons here.
__global__ __launch_bounds__(DB_THREADS) void db_union_k(DbGrid g, const float4* __restrict__ pts,
                                                         const uint32_t* __restrict__ cell_start,
                                                         const uint64_t* __restrict__ cell_key, int m,
                                                         const int2* __restrict__ rowtab,
                                                         const uint8_t* __restrict__ core_s,
                                                         const uint32_t* __restrict__ cell_ncore,
                                                         const float* __restrict__ cell_box,
                                                         int* __restrict__ parent) {
    __shared__ RowSet rows[DB_WAVES];
    __shared__ int cand[DB_WAVES][128];
    const int A = blockIdx.x * DB_WAVES + wave_id();
    if (A >= m) return;
    if (cell_ncore[A] == 0) return;
    const int l = lane_id();
    RowSet* rs = &rows[wave_id()];
    int* cd = cand[wave_id()];
    const uint64_t keyA = cell_key[A];
    db_rows(g, cell_key, m, keyA, rs, rowtab, A);
    int total;
    {
        // flatten the <= 25 runs of <= 5 cells into one candidate list (prefix over the run lengths)
        int len = 0;
        if (l < DB_ROWS) { len = rs->cb[l] - rs->ca[l]; len = len < 0 ? 0 : len; }
        const int incl = wave_scan_incl(len);
        total = __shfl(incl, 63, 64);
        if (l < DB_ROWS)
            for (int k = 0; k < len; ++k) cd[incl - len + k] = rs->ca[l] + k;
    }
    __builtin_amdgcn_wave_barrier();
    const uint32_t as = cell_start[A], ae = cell_start[A + 1];
    const bool a_dense = cell_ncore[A] == (ae - as);
    const float* boxA = cell_box + 6 * (int64_t)A;
    for (int base = 0; base < total; base += 64) {
        int B = -1;
        if (base + l < total) B = cd[base + l];
        bool live = false;
        if (B > A) {                                       // every unordered pair once
            if (cell_ncore[B] != 0 && !(db_boxbox_d2(boxA, cell_box + 6 * (int64_t)B) > g.eps2)) {
                // plain (possibly stale) loads first: equal parents were in one set at some time,
                // and sets only ever merge
                live = parent[A] != parent[B];
                if (live) live = uf_find(parent, A) != uf_find(parent, B);
            }
        }
        unsigned long long todo = __ballot(live);
        while (todo) {
            const int src = (int)__builtin_ctzll(todo);
            todo &= todo - 1;
            const int Bs = __builtin_amdgcn_readlane(B, src);
            {                                              // united meanwhile through another cell?
                int same = 0;
                if (l == 0) same = uf_find(parent, A) == uf_find(parent, Bs);
                if (__builtin_amdgcn_readfirstlane(same)) continue;
            }
            const uint32_t bs = cell_start[Bs], be = cell_start[Bs + 1];
            const bool b_dense = cell_ncore[Bs] == (be - bs);
            const float* boxB = cell_box + 6 * (int64_t)Bs;
            const bool connected = db_cells_connected(g, pts, core_s, as, ae, a_dense, bs, be, b_dense, boxB, as);
            if (connected && l == 0) uf_union(parent, A, Bs);
        }
    }
}

// ROUND 0, first half: one LANE per (cell, face direction).  A wave per cell walks one chain of dependent loads
// (neighbour lookup -> counts and boxes -> roots -> first points) per cell; on sparse data (millions of cells of
// a few dozen points) that chain, not arithmetic, is the whole cost.  Here 64 such chains are in flight per wave:
// the lane finds its neighbour, compares boxes and parents and tries the first DB_PAIR_TRIES core points of either
// cell against each other - adjacent cells almost always connect there - and unites on a hit.  Pairs it cannot
// decide are flagged in face_todo[cell]; only those cells are looked at by the wave-wide db_union_face_k.
constexpr int DB_PAIR_TRIES = 3;

__global__ __launch_bounds__(DB_THREADS) void db_union_pairs_k(DbGrid g, const float4* __restrict__ pts,
                                                               const uint32_t* __restrict__ cell_start,
                                                               const uint64_t* __restrict__ cell_key, int m,
                                                               const int2* __restrict__ rowtab,
                                                               const uint8_t* __restrict__ core_s,
                                                               const uint32_t* __restrict__ cell_ncore,
                                                               const float* __restrict__ cell_box,
                                                               int* __restrict__ parent,
                                                               uint8_t* __restrict__ face_todo) {
    const int64_t t = (int64_t)blockIdx.x * DB_THREADS + threadIdx.x;
    const int A = (int)(t >> 2), dir = (int)(t & 3);       // four lanes per cell: +x, +y, +z, idle
    const bool lane_on = A < m && dir < 3;
    bool undecided = false;
    if (lane_on) {
        const uint32_t ncoreA = cell_ncore[A];
        if (ncoreA != 0) {
            const uint64_t keyA = cell_key[A];
            int B = -1;
            if (dir == 0) {
                if (A + 1 < m && cell_key[A + 1] == keyA + 1ull) B = A + 1;
            } else {
                const int2 run = rowtab[(int64_t)A * DB_ROWS + (dir == 1 ? 1 : 3)];
                const uint64_t want = keyA + (dir == 1 ? (1ull << g.bx) : (1ull << (g.bx + g.by)));
                for (int k = run.x; k < run.y; ++k)
                    if (cell_key[k] == want) { B = k; break; }
            }
            if (B > A) {
                const uint32_t nB = cell_ncore[B];
                if (nB != 0) {
                    float boxB[6];
#pragma unroll
                    for (int a = 0; a < 6; ++a) boxB[a] = cell_box[6 * (int64_t)B + a];
                    bool pend = !(db_boxbox_d2(cell_box + 6 * (int64_t)A, boxB) > g.eps2);
                    if (pend) pend = parent[A] != parent[B];
                    int rootA = A, rootB = B;
                    if (pend) {
                        rootA = uf_find(parent, A);
                        rootB = uf_find(parent, B);
                        pend = rootA != rootB;
                    }
                    if (pend) {
                        const uint32_t as = cell_start[A], ae = cell_start[A + 1];
                        const uint32_t bs = cell_start[B], be = cell_start[B + 1];
                        const bool a_dense = ncoreA == (ae - as), b_dense = nB == (be - bs);
                        float4 pb[DB_PAIR_TRIES];
                        int nb = 0;
                        for (uint32_t j = bs; j < be && nb < DB_PAIR_TRIES; ++j)
                            if (b_dense || core_s[j]) pb[nb++] = pts[j];
                        bool connected = false;
                        int na = 0;
                        for (uint32_t i = as; i < ae && na < DB_PAIR_TRIES && !connected; ++i) {
                            if (!a_dense && !core_s[i]) continue;
                            ++na;
                            const float4 pa = pts[i];
                            if (db_box_d2(pa, boxB) > g.eps2) continue;
#pragma unroll
                            for (int j = 0; j < DB_PAIR_TRIES; ++j)
                                if (j < nb && db_within2(pa, pb[j], g)) connected = true;
                        }
                        if (connected) uf_union(parent, rootA, rootB);
                        else undecided = true;
                    }
                }
            }
        }
    }
    const unsigned long long um = __ballot(undecided);
    if (A < m && dir == 0) face_todo[A] = (uint8_t)((um >> (lane_id() & ~3)) & 7ull);
}

// ROUND 0 as its own kernel: the (up to) three face neighbours with a larger key are examined side
// by side, 21 lanes each, so that the chain of dependent loads (neighbour lookup, roots, first
// points) is walked once per cell instead of once per neighbour.  Adjacent dense cells connect at
// the first point pair; a pair that is still undecided after DB_FACE_TRIES points of A is handed
// to the full-wave search of the general round.
constexpr int DB_FACE_TRIES = 8;

__global__ __launch_bounds__(DB_THREADS) void db_union_face_k(DbGrid g, const float4* __restrict__ pts,
                                                              const uint32_t* __restrict__ cell_start,
                                                              const uint64_t* __restrict__ cell_key, int m,
                                                              const int2* __restrict__ rowtab,
                                                              const uint8_t* __restrict__ core_s,
                                                              const uint32_t* __restrict__ cell_ncore,
                                                              const float* __restrict__ cell_box,
                                                              int* __restrict__ parent,
                                                              const uint8_t* __restrict__ face_todo) {
    __shared__ RowSet rows[DB_WAVES];
    const int A = blockIdx.x * DB_WAVES + wave_id();
    if (A >= m) return;
    if (face_todo && face_todo[A] == 0) return;            // db_union_pairs_k has settled this cell's faces
    const uint32_t ncoreA = cell_ncore[A];
    if (ncoreA == 0) return;
    const int l = lane_id();
    const int grp = l / 21, gl = l - 21 * grp;             // lane 63: group 3, idle
    const unsigned long long gmask = grp < 3 ? (0x1FFFFFull << (21 * grp)) : 0ull;
    RowSet* rs = &rows[wave_id()];
    const uint64_t keyA = cell_key[A];
    db_rows(g, cell_key, m, keyA, rs, rowtab, A);
    // +x is the next cell in sorted order, +y / +z sit in the runs (dy,dz) = (1,0) / (0,1)
    // (rows 1 and 3 of DB_ROW_DY/DZ)
    int B = -1;
    {
        bool found = false;
        int k = -1;
        if (grp == 0) {
            k = A + 1;
            found = gl == 0 && k < m && cell_key[k] == keyA + 1ull;
        } else if (grp < 3) {
            const int r = grp == 1 ? 1 : 3;
            const uint64_t want = keyA + (grp == 1 ? (1ull << g.bx) : (1ull << (g.bx + g.by)));
            k = rs->ca[r] + gl;
            found = k < rs->cb[r] && cell_key[k] == want;
        }
        const unsigned long long fm = __ballot(found) & gmask;
        if (fm) B = __shfl(k, (int)__builtin_ctzll(fm), 64);
    }
    const uint32_t as = cell_start[A], ae = cell_start[A + 1];
    const bool a_dense = ncoreA == (ae - as);
    const float4 pa0 = pts[as];                            // in flight with the neighbour look-ups below
    uint32_t bs = 0, be = 0;
    bool b_dense = false, pend = false;
    float boxB[6] = {0, 0, 0, 0, 0, 0};
    if (B > A) {
        const uint32_t nB = cell_ncore[B];
        bs = cell_start[B];
        be = cell_start[B + 1];
        b_dense = nB == (be - bs);
#pragma unroll
        for (int a = 0; a < 6; ++a) boxB[a] = cell_box[6 * (int64_t)B + a];
        pend = nB != 0 && !(db_boxbox_d2(cell_box + 6 * (int64_t)A, boxB) > g.eps2);
        // plain (possibly stale) loads first: equal parents were in one set at some time, and sets only merge
        if (pend) pend = parent[A] != parent[B];
    }
    if (gl != 0) pend = false;                             // one lane per group looks the roots up
    int rootA = A, rootB = B;
    if (pend) {
        rootA = uf_find(parent, A);
        rootB = uf_find(parent, B);
        pend = rootA != rootB;
    }
    pend = __shfl((int)pend, grp < 3 ? 21 * grp : 0, 64) != 0 && grp < 3;
    bool connected = false;
    uint32_t ia = as;
    for (int tries = 0; ia < ae && tries < DB_FACE_TRIES; ++ia) {
        if (__ballot(pend) == 0) break;
        if (!a_dense && !core_s[ia]) continue;
        ++tries;
        const float4 pa = ia == as ? pa0 : pts[ia];
        bool near = pend && !(db_box_d2(pa, boxB) > g.eps2);
        const uint32_t maxlen = wave_reduce_max(near ? be - bs : 0u);
        for (uint32_t j0 = 0; j0 < maxlen; j0 += 21) {
            const uint32_t j = bs + j0 + gl;
            bool hit = false;
            if (near && j < be && (b_dense || core_s[j])) hit = db_within2(pa, pts[j], g);
            if (__ballot(hit) & gmask) { connected = true; pend = false; near = false; }
            if (__ballot(near) == 0) break;
        }
    }
    if (connected && gl == 0) uf_union(parent, rootA, rootB);      // starts from the roots found above
    // undecided pairs (rare): the full-wave search, one pair after the other
    unsigned long long todo = __ballot(pend && gl == 0);
    while (todo) {
        const int src = (int)__builtin_ctzll(todo);
        todo &= todo - 1;
        const int Bs = __shfl(B, src, 64);
        const uint32_t bs2 = cell_start[Bs], be2 = cell_start[Bs + 1];
        const bool b_dense2 = cell_ncore[Bs] == (be2 - bs2);
        const float* boxB2 = cell_box + 6 * (int64_t)Bs;
        const bool conn = db_cells_connected(g, pts, core_s, as, ae, a_dense, bs2, be2, b_dense2, boxB2, ia);
        if (conn && l == 0) uf_union(parent, A, Bs);
    }
}

// path compression between the union rounds: afterwards parent[c] is the root of c
__global__ __launch_bounds__(DB_THREADS) void db_flatten_k(int* __restrict__ parent, int m) {
    const int c = blockIdx.x * DB_THREADS + threadIdx.x;
    if (c >= m) return;
    const int r = uf_find(parent, c);
    if (r != c) parent[c] = r;
}

// root of every core cell + smallest original row among the component's core points
__global__ __launch_bounds__(DB_THREADS) void db_compmin_k(const int* __restrict__ cell_min,
                                                           const uint32_t* __restrict__ cell_ncore, int m,
                                                           int* __restrict__ parent,
                                                           int* __restrict__ root,
                                                           int* __restrict__ comp_min) {
    const int c = blockIdx.x * DB_THREADS + threadIdx.x;
    if (c >= m) return;
    if (cell_ncore[c] == 0) { root[c] = -1; return; }
    const int r = uf_find(parent, c);
    root[c] = r;
    atomicMin(&comp_min[r], cell_min[c]);
}

__global__ __launch_bounds__(DB_THREADS) void db_mark_k(const int* __restrict__ root,
                                                        const int* __restrict__ comp_min, int m,
                                                        uint32_t* __restrict__ flag) {
    const int c = blockIdx.x * DB_THREADS + threadIdx.x;
    if (c >= m) return;
    if (root[c] == c) {                                    // one bit per original row: the smallest core row of a cluster
        const uint32_t r = (uint32_t)comp_min[c];
        atomicOr(&flag[r >> 5], 1u << (r & 31u));
    }
}

// Per-cluster bounding boxes, accumulated where the labels are made (the grouping stage then needs no pass of its
// own over the points): acc[8 k + a], a = 0..2: max of ~ordered(x|y|z) (the minimum), a = 3..5: max of ordered(.)
// - folded with atomicMax so that a zeroed table is the neutral start.
constexpr int DB_LAB_ROUNDS = 8;                       // 64-point rounds per wave in db_label_k
constexpr int DB_LAB_TILE   = DB_THREADS * DB_LAB_ROUNDS;

__device__ __forceinline__ void db_box_flush(uint32_t* __restrict__ acc, int cur, uint32_t (&m)[6]) {
    const int l = lane_id();
    uint32_t v = 0;
#pragma unroll
    for (int a = 0; a < 6; ++a) {
        const uint32_t t = wave_reduce_max(m[a]);
        if (l == a) v = t;
        m[a] = 0u;
    }
    if (cur >= 0 && l < 6 && v) atomicMax(&acc[8 * (int64_t)cur + l], v);
}

// labels of core points (original order), label of every cell, optional core mask.
// A wave walks DB_LAB_ROUNDS consecutive groups of 64 sorted points.  The points are sorted by cell, so a wave
// mostly sees ONE cluster: it keeps a per-lane running box while the label stays the same and pays the wave
// reduction + six atomics only when the label changes (one flush per 64 points cost more than the separate pass
// over the points it replaced: ~10^6 atomics on a few hundred lines).
// Block -> sorted range: with many chunks (bpc > 0) the blocks of one chunk share blockIdx % 8, i.e. one XCD and
// one L2 (MI355X deals blocks round-robin over its 8 XCDs; a speed assumption only): the 4-byte label stores of a
// chunk scatter over that chunk's own 200 KB window of `labels`, and lines written from several XCDs would leave
// every L2 as partial lines (measured: 94.6 -> 68.8 us).
__global__ __launch_bounds__(DB_THREADS) void db_label_k(const float4* __restrict__ pts,
                                                         const uint32_t* __restrict__ cid,
                                                         const uint8_t* __restrict__ core_s,
                                                         const int* __restrict__ root,
                                                         const int* __restrict__ comp_min,
                                                         const uint32_t* __restrict__ bits,
                                                         const uint32_t* __restrict__ rank, int64_t n,
                                                         const uint32_t* __restrict__ cell_start,
                                                         int* __restrict__ cell_label,
                                                         int32_t* __restrict__ labels,
                                                         uint8_t* __restrict__ core_out, int64_t chunk_size, int bpc,
                                                         int64_t nchunks, uint32_t* __restrict__ box_acc,
                                                         int32_t box_cap) {
    int64_t lo, hi;                                        // this wave's sorted range [lo, hi)
    {
        const int64_t woff = (int64_t)wave_id() * (64 * DB_LAB_ROUNDS);
        if (bpc > 0) {
            const int64_t slot = blockIdx.x >> 3;
            const int64_t c = (slot / bpc) * 8 + (blockIdx.x & 7);
            const int64_t within = (slot % bpc) * DB_LAB_TILE + woff;
            lo = c * chunk_size + within;
            hi = (c < nchunks && within < chunk_size) ? (c + 1) * chunk_size : lo;
        } else {
            lo = (int64_t)blockIdx.x * DB_LAB_TILE + woff;
            hi = n;
        }
        if (hi > n) hi = n;
        if (hi > lo + 64 * DB_LAB_ROUNDS) hi = lo + 64 * DB_LAB_ROUNDS;
    }
    if (lo >= hi) return;                                  // wave-uniform
    const int l = lane_id();
    // the gather chain cid -> root -> comp_min -> (rank, bits) is four dependent loads deep: all rounds of the wave
    // go through it level by level, eight loads in flight per lane and level
    constexpr int R = DB_LAB_ROUNDS;
    bool valid[R], is_core[R];
    uint32_t c[R], cs[R], row[R], rk[R], bw[R];
    int rt[R], lab[R];
    float4 p[R];
#pragma unroll
    for (int u = 0; u < R; ++u) {
        const int64_t i = lo + u * 64 + l;
        valid[u] = i < hi;
        c[u] = valid[u] ? cid[i] : 0u;
        p[u] = pts[valid[u] ? i : lo];
        is_core[u] = valid[u] && core_s[i] != 0;
    }
#pragma unroll
    for (int u = 0; u < R; ++u) {
        rt[u] = valid[u] ? root[c[u]] : -1;
        cs[u] = valid[u] ? cell_start[c[u]] : 0xFFFFFFFFu;
    }
#pragma unroll
    for (int u = 0; u < R; ++u) row[u] = rt[u] >= 0 ? (uint32_t)comp_min[rt[u]] : 0u;
#pragma unroll
    for (int u = 0; u < R; ++u) {
        rk[u] = rt[u] >= 0 ? rank[row[u] >> 5] : 0u;
        bw[u] = rt[u] >= 0 ? bits[row[u] >> 5] : 0u;
    }
#pragma unroll
    for (int u = 0; u < R; ++u) {
        const int64_t i = lo + u * 64 + l;
        // clusters are numbered by their smallest core row
        lab[u] = rt[u] >= 0 ? (int)(rk[u] + (uint32_t)__popc(bw[u] & ((1u << (row[u] & 31u)) - 1u))) : INT_BIG;
        if (valid[u]) {
            const uint32_t o = __float_as_uint(p[u].w);
            labels[o] = is_core[u] ? lab[u] : -1;
            if (core_out) core_out[o] = is_core[u] ? 1 : 0;
            if (cs[u] == (uint32_t)i) cell_label[c[u]] = lab[u];
        }
    }
    if (!box_acc) return;
    int cur = -1;                                          // label of the box carried in m[] (wave-uniform)
    uint32_t m[6] = {0u, 0u, 0u, 0u, 0u, 0u};              // per-lane running box of `cur`
#pragma unroll
    for (int u = 0; u < R; ++u) {
        const int mine = (is_core[u] && lab[u] != INT_BIG && lab[u] >= 0 && lab[u] < box_cap) ? lab[u] : -1;
        const uint32_t k[3] = {f32_ordered(p[u].x), f32_ordered(p[u].y), f32_ordered(p[u].z)};
        unsigned long long todo = __ballot(mine >= 0);
        while (todo) {
            const int L = __builtin_amdgcn_readlane(mine, (int)__builtin_ctzll(todo));
            const bool in = mine == L;
            todo &= ~__ballot(in);
            if (L != cur) {                                // another cluster: hand the carried box over
                db_box_flush(box_acc, cur, m);
                cur = L;
            }
            if (in) {
#pragma unroll
                for (int a = 0; a < 3; ++a) {
                    m[a] = ~k[a] > m[a] ? ~k[a] : m[a];
                    m[3 + a] = k[a] > m[3 + a] ? k[a] : m[3 + a];
                }
            }
        }
    }
    db_box_flush(box_acc, cur, m);
}

// border points: smallest cluster id among the core points within eps
__global__ __launch_bounds__(DB_THREADS) void db_border_k(DbGrid g, const float4* __restrict__ pts,
                                                          const uint32_t* __restrict__ cell_start,
                                                          const uint64_t* __restrict__ cell_key, int m,
                                                          const int2* __restrict__ rowtab,
                                                          const uint8_t* __restrict__ core_s,
                                                          const uint32_t* __restrict__ cell_ncore,
                                                          const float* __restrict__ cell_box,
                                                          const int* __restrict__ cell_label,
                                                          int32_t* __restrict__ labels,
                                                          uint32_t* __restrict__ box_acc, int32_t box_cap) {
    __shared__ RowSet rows[DB_WAVES];
    const int A = blockIdx.x * DB_WAVES + wave_id();
    if (A >= m) return;
    const uint32_t as = cell_start[A], ae = cell_start[A + 1];
    if (cell_ncore[A] == (ae - as)) return;                // no border candidates here
    const int l = lane_id();
    RowSet* rs = &rows[wave_id()];
    db_rows(g, cell_key, m, cell_key[A], rs, rowtab, A);
    // quick reject: no core cell anywhere around
    int any = 0;
    if (l < DB_ROWS)
        for (int B = rs->ca[l]; B < rs->cb[l]; ++B) any |= (cell_ncore[B] != 0);
    if (!__ballot(any != 0)) return;
    // The neighbour cells with core points, ONE PER LANE (three rounds of twelve rows x five cells - a row holds the
    // cells x-2 .. x+2 of one (y, z)): count, label and core box are read once per cell A, side by side.  One cell after
    // the other - count, label, six box words, each a dependent load, for up to 125 cells and every query again - was
    // this kernel: a lone noise point beside a tower tested ~100 boxes at ~0.4 us each, 51 us for 2.6 MB of traffic.
    constexpr int BR = 3, BROWS = 12;
    int Bc[BR], Bl[BR];
    bool Bd[BR];
    float Bx[BR][6];
    bool wide = false;                                       // a row of more than five cells (never by construction)
#pragma unroll
    for (int rd = 0; rd < BR; ++rd) {
        const int row = rd * BROWS + l / 5, k = l % 5;
        Bc[rd] = -1; Bl[rd] = INT_BIG; Bd[rd] = false;
#pragma unroll
        for (int a = 0; a < 6; ++a) Bx[rd][a] = 0.0f;
        if (l < 5 * BROWS && row < DB_ROWS) {
            const int ca = rs->ca[row], cb = rs->cb[row];
            wide |= cb - ca > 5;
            const int B = ca + k;
            if (B < cb) {
                const uint32_t nb = cell_ncore[B];
                if (nb != 0) {
                    Bc[rd] = B;
                    Bl[rd] = cell_label[B];
                    Bd[rd] = nb == cell_start[B + 1] - cell_start[B];
#pragma unroll
                    for (int a = 0; a < 6; ++a) Bx[rd][a] = cell_box[6 * (int64_t)B + a];
                }
            }
        }
    }
    if (__ballot(wide)) {                                    // the plain loop, cell after cell
        for (uint32_t q = as; q < ae; ++q) {
            if (core_s[q]) continue;
            const float4 qp = pts[q];
            int best = INT_BIG;
            for (int r = 0; r < DB_ROWS; ++r) {
                const int cb = rs->cb[r];
                for (int B = rs->ca[r]; B < cb; ++B) {
                    const uint32_t nb = cell_ncore[B];
                    if (nb == 0) continue;
                    const int lab = cell_label[B];
                    if (lab >= best) continue;
                    if (db_box_d2(qp, cell_box + 6 * (int64_t)B) > g.eps2) continue;
                    const uint32_t bs = cell_start[B], be = cell_start[B + 1];
                    const bool b_dense = nb == (be - bs);
                    for (uint32_t j0 = bs; j0 < be; j0 += 64) {
                        const uint32_t j = j0 + l;
                        bool hit = false;
                        if (j < be && (b_dense || core_s[j])) hit = db_within2(qp, pts[j], g);
                        if (__ballot(hit)) { best = lab; break; }
                    }
                }
            }
            if (l == 0 && best != INT_BIG) labels[__float_as_uint(qp.w)] = best;
            if (box_acc && best != INT_BIG && best >= 0 && best < box_cap && l < 6) {
                const float v = l % 3 == 0 ? qp.x : (l % 3 == 1 ? qp.y : qp.z);
                const uint32_t k = f32_ordered(v);
                atomicMax(&box_acc[8 * (int64_t)best + l], l < 3 ? ~k : k);
            }
        }
        return;
    }
    for (uint32_t q = as; q < ae; ++q) {
        if (core_s[q]) continue;
        const float4 qp = pts[q];
        // cells whose core box reaches the query, then: smallest label first - the first cell that really holds a core
        // point within eps decides (what is asked for is the smallest label among such cells; the order of the cells
        // with larger labels does not matter)
        bool cand[BR];
#pragma unroll
        for (int rd = 0; rd < BR; ++rd) cand[rd] = Bc[rd] >= 0 && !(db_box_d2(qp, Bx[rd]) > g.eps2);
        int best = INT_BIG;
        for (;;) {
            int mine = INT_BIG, sel = -1;
#pragma unroll
            for (int rd = 0; rd < BR; ++rd)
                if (cand[rd] && Bl[rd] < mine) { mine = Bl[rd]; sel = rd; }
            const int lo = wave_reduce_min(mine);
            if (lo == INT_BIG) break;
            const int owner = (int)__builtin_ctzll(__ballot(mine == lo));
            int Bsel = -1;
            bool dsel = false;
#pragma unroll
            for (int rd = 0; rd < BR; ++rd)
                if (sel == rd) { Bsel = Bc[rd]; dsel = Bd[rd]; }
            const int B = __builtin_amdgcn_readlane(Bsel, owner);
            const bool b_dense = __builtin_amdgcn_readlane((int)dsel, owner) != 0;
            const uint32_t bs = cell_start[B], be = cell_start[B + 1];
            bool found = false;
            for (uint32_t j0 = bs; j0 < be; j0 += 64) {
                const uint32_t j = j0 + l;
                bool hit = false;
                if (j < be && (b_dense || core_s[j])) hit = db_within2(qp, pts[j], g);
                if (__ballot(hit)) { found = true; break; }
            }
            if (found) { best = lo; break; }
            if (l == owner) {
#pragma unroll
                for (int rd = 0; rd < BR; ++rd)
                    if (sel == rd) cand[rd] = false;
            }
        }
        if (l == 0 && best != INT_BIG) labels[__float_as_uint(qp.w)] = best;
        if (box_acc && best != INT_BIG && best >= 0 && best < box_cap && l < 6) {     // a border point joins its box
            const float v = l % 3 == 0 ? qp.x : (l % 3 == 1 ? qp.y : qp.z);
            const uint32_t k = f32_ordered(v);
            atomicMax(&box_acc[8 * (int64_t)best + l], l < 3 ? ~k : k);
        }
    }
}

// ---- relabelling with an external cluster map (cross-tile reconciliation, tiles.py): core points and
// cells take map[old id], every other point goes back to -1 and is assigned again by db_border_k, so
// that a border point takes the smallest NEW id among its core neighbours
__global__ __launch_bounds__(DB_THREADS) void db_remap_points_k(const float4* __restrict__ pts,
                                                                const uint8_t* __restrict__ core_s, int64_t n,
                                                                const int32_t* __restrict__ map, int32_t nmap,
                                                                int32_t* __restrict__ labels) {
    const int64_t i = (int64_t)blockIdx.x * DB_THREADS + threadIdx.x;
    if (i >= n) return;
    const uint32_t o = __float_as_uint(pts[i].w);
    int32_t lab = -1;
    if (core_s[i]) {
        const int32_t old = labels[o];
        lab = (old >= 0 && old < nmap) ? map[old] : -1;
    }
    labels[o] = lab;
}
__global__ __launch_bounds__(DB_THREADS) void db_remap_cells_k(int* __restrict__ cell_label,
                                                               const uint32_t* __restrict__ cell_ncore, int m,
                                                               const int32_t* __restrict__ map, int32_t nmap) {
    const int c = blockIdx.x * DB_THREADS + threadIdx.x;
    if (c >= m) return;
    if (cell_ncore[c] == 0) return;
    const int old = cell_label[c];
    const int neu = (old >= 0 && old < nmap) ? map[old] : -1;
    cell_label[c] = neu >= 0 ? neu : INT_BIG;              // a dropped cluster attracts no border points
}

// cluster id -> its smallest core row (the set bits of the row bitmap, in order)
__global__ __launch_bounds__(DB_THREADS) void db_first_rows_k(const uint32_t* __restrict__ bits,
                                                              const uint32_t* __restrict__ rank, int64_t nw,
                                                              int32_t* __restrict__ out) {
    const int64_t i = (int64_t)blockIdx.x * DB_THREADS + threadIdx.x;
    if (i >= nw) return;
    uint32_t b = bits[i];
    uint32_t c = rank[i];
    while (b) {
        const int t = __builtin_ctz(b);
        b &= b - 1;
        out[c++] = (int32_t)(i * 32 + t);
    }
}

// Cross-tile links (config 4): one representative per cell of the core points with x in [x_lo, x_hi) - the strip
// around a tile edge.  All core points of a cell are mutually within eps (cell side eps/sqrt(3)), so they are one
// cluster in ANY tile that holds them with exact core flags: one (row, cluster) pair per cell tells the
// neighbouring tile everything the pairs of all its strip points would.  One wave per cell; cells whose core box
// misses the strip are rejected from the cell table alone.
__global__ __launch_bounds__(DB_THREADS) void db_strip_pairs_k(const float4* __restrict__ pts,
                                                               const uint32_t* __restrict__ cell_start,
                                                               const uint8_t* __restrict__ core_s,
                                                               const uint32_t* __restrict__ cell_ncore,
                                                               const float* __restrict__ cell_box,
                                                               const int* __restrict__ cell_label, int m, float x_lo,
                                                               float x_hi, int32_t cap, int32_t* __restrict__ out_pairs,
                                                               int32_t* __restrict__ out_count) {
    const int c = blockIdx.x * DB_WAVES + wave_id();
    if (c >= m) return;                                   // wave-uniform
    if (cell_ncore[c] == 0) return;
    if (!(cell_box[6 * (int64_t)c + 3] >= x_lo && cell_box[6 * (int64_t)c] < x_hi)) return;
    const int l = lane_id();
    const uint32_t a = cell_start[c], b = cell_start[c + 1];
    uint32_t best = 0xFFFFFFFFu;
    for (uint32_t i = a + l; i < b; i += 64) {
        if (!core_s[i]) continue;
        const float4 p = pts[i];
        if (p.x >= x_lo && p.x < x_hi) {
            const uint32_t row = __float_as_uint(p.w);
            best = row < best ? row : best;
        }
    }
    best = wave_reduce_min(best);
    if (l == 0 && best != 0xFFFFFFFFu) {
        const int32_t slot = atomicAdd(out_count, 1);     // the count keeps growing beyond cap: the caller sees it
        if (slot < cap) {
            out_pairs[2 * slot] = (int32_t)best;
            out_pairs[2 * slot + 1] = cell_label[c];
        }
    }
}

// publishes the cluster count and zeroes the box accumulators (neutral start of db_box_fold)
__global__ void db_prelabel_k(const uint32_t* __restrict__ total, int32_t* __restrict__ out_nclusters,
                              uint32_t* __restrict__ box_acc, int64_t box_words) {
    if (threadIdx.x == 0 && blockIdx.x == 0) *out_nclusters = (int32_t)*total;
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (box_acc && i < box_words) box_acc[i] = 0u;
}

struct DbWs {
    uint32_t* meta;         // [0..5] aabb ordered keys, [6] status, [7] m, [8] nclusters
    uint64_t *k0, *k1, *cell_key;
    uint32_t *v0, *v1, *head, *cid, *cell_start, *cell_ncore, *flag, *radix_ws, *scan_ws;
    float4   *pts, *xbuf;
    uint8_t*  core_s;
    float*    cell_box;
    uint32_t* cell_acc;     // [cells][8] ordered-key accumulators of db_cellstats_k
    int*      cell_min;     // smallest original row among a cell's core points
    int      *parent, *root, *comp_min, *cell_label;
    int2*     rowtab;
    int64_t   rowtab_cells;
    uint32_t* chunk_bad;
    uint32_t* chunk_cells;
    uint32_t* comp;          // [n][3] compressed cell coordinates (fallback for grids beyond the 64-bit key)
    uint32_t* flag2;         // [n + 8] head flags kept beside their scan
    unsigned long long* core_stats;   // [4] tallies of db_core_k<true> (pch_dbscan_set_pair_counting)
    uint32_t* scan1_b;                // zeroed words of the single-pass scan of the cluster ranks
};

static void db_plan(Arena& a, int64_t n, DbWs& w) {
    const int64_t nn = n > 0 ? n : 1;
    w.meta = a.take<uint32_t>(64);                       // exactly one 256-byte arena block ...
    w.scan1_b = a.take<uint32_t>(scan1_ws_u32(nn / 32 + 1));   // ... directly followed by the zero-initialised words
                                                         // of the single-pass scan and by the per-chunk table:
    w.chunk_cells = a.take<uint32_t>(nn + 8);            // ONE fill clears them all (db_plan keeps them adjacent)
    w.core_stats = a.take<unsigned long long>(4);
    w.chunk_bad = a.take<uint32_t>(nn + 8);              // one word per chunk (chunk_size >= 1)
    w.k0 = a.take<uint64_t>(nn);
    w.k1 = a.take<uint64_t>(nn);
    w.v0 = a.take<uint32_t>(nn);
    w.v1 = a.take<uint32_t>(nn);
    w.head = a.take<uint32_t>(nn + 8);
    w.cid = a.take<uint32_t>(nn);
    w.pts = a.take<float4>(nn);
    w.xbuf = a.take<float4>(nn);                        // spare rows array of the chunk-local sort
    w.core_s = a.take<uint8_t>(nn);
    w.cell_key = a.take<uint64_t>(nn);
    w.cell_start = a.take<uint32_t>(nn + 8);
    w.cell_ncore = a.take<uint32_t>(nn);
    w.cell_box = a.take<float>(6 * nn);
    w.cell_acc = a.take<uint32_t>(8 * nn);
    w.cell_min = a.take<int>(nn);
    w.parent = a.take<int>(nn);
    w.root = a.take<int>(nn);
    w.comp_min = a.take<int>(nn);
    w.cell_label = a.take<int>(nn);
    w.flag = a.take<uint32_t>(nn / 16 + 256);             // row bitmap (n/32 words) + its scanned word counts
    w.radix_ws = a.take<uint32_t>(radix_ws_u32(nn));
    w.scan_ws = a.take<uint32_t>(scan_ws_u32(nn));
    // neighbour-row table (200 B per cell) for up to max(n/4, 64Ki) cells; beyond that the rows are
    // searched on the fly
    w.rowtab_cells = nn / 4 > 65536 ? nn / 4 : (nn < 65536 ? nn : 65536);
    w.rowtab = a.take<int2>((size_t)w.rowtab_cells * DB_ROWS);
    w.comp = a.take<uint32_t>(3 * nn);
    w.flag2 = a.take<uint32_t>(nn + 8);
}

// PCH_DBSCAN_SORT (chunk | global), read once per process unless pch_dbscan_set_sort_mode() overrides it:
// 0 automatic, 1 chunk-local, 2 global
static int g_sort_mode = -1;
static int db_sort_mode() {
    int m = __atomic_load_n(&g_sort_mode, __ATOMIC_RELAXED);
    if (m < 0) {
        const char* e = getenv("PCH_DBSCAN_SORT");
        m = (e && strcmp(e, "chunk") == 0) ? 1 : (e && strcmp(e, "global") == 0) ? 2 : 0;
        __atomic_store_n(&g_sort_mode, m, __ATOMIC_RELAXED);
    }
    return m;
}

static thread_local bool g_count_pairs = false;

// what pch_dbscan_relabel_i32 needs to know about the run whose workspace it continues
struct DbLastRun { void* ws; size_t ws_bytes; int64_t n; int m; DbGrid g; bool has_rowtab; };
static thread_local DbLastRun g_last = {nullptr, 0, 0, 0, {}, false};

void ws_touched(const void* base, size_t bytes) {
    if (!g_last.ws) return;
    const char* a0 = static_cast<const char*>(base);
    const char* b0 = static_cast<const char*>(g_last.ws);
    if (a0 < b0 + g_last.ws_bytes && b0 < a0 + bytes) g_last.ws = nullptr;
}

// host mirror of f32_unordered
static float host_unordered(uint32_t k) {
    uint32_t u = (k & 0x80000000u) ? (k & 0x7fffffffu) : ~k;
    float f;
    memcpy(&f, &u, 4);
    return f;
}

}  // namespace pch

using namespace pch;

extern "C" int pch_first_nonfinite_row_f32(const float* xyz, int64_t n, int64_t* out_row, void* stream) {
    PCH_DEVICE_GUARD(out_row);
    hipStream_t s = (hipStream_t)stream;
    PCH_REQUIRE(n >= 0 && out_row && (n == 0 || xyz), "bad argument");
    PCH_HIP_TRY(hipMemsetAsync(out_row, 0xFF, sizeof(int64_t), s));             // -1: every row finite
    if (n == 0) return PCH_OK;
    int64_t gb = ceil_div(n, DB_THREADS * 8);
    if (gb > 2048) gb = 2048;
    PCH_LAUNCH("db_first_bad", db_first_bad_k, dim3((unsigned)gb), dim3(DB_THREADS), 0, s, xyz, n,
               reinterpret_cast<unsigned long long*>(out_row));
    return PCH_OK;
}

extern "C" void pch_dbscan_set_sort_mode(int mode) {
    __atomic_store_n(&g_sort_mode, (mode == 1 || mode == 2) ? mode : 0, __ATOMIC_RELAXED);
}

extern "C" size_t pch_dbscan_ws_bytes(int64_t n) {
    if (n < 0) return 0;
    Arena a;
    DbWs w;
    db_plan(a, n, w);
    return a.off;
}

int pch::dbscan_run(const float* xyz, int64_t n, double eps, int32_t min_samples, int64_t chunk_size,
                    const float* aabb_host, int32_t* labels, uint8_t* core, int32_t* out_nclusters, void* ws,
                    size_t ws_bytes, hipStream_t s, int32_t* k_host, DbBoxOut* boxes) {
    if (k_host) *k_host = 0;
    if (boxes) boxes->done = false;
    PCH_REQUIRE(n >= 0 && n < (int64_t(1) << 31), "n out of range [0, 2^31)");
    PCH_REQUIRE(eps > 0.0, "eps must be > 0 (sklearn: InvalidParameterError)");
    PCH_REQUIRE(min_samples >= 1, "min_samples must be >= 1 (sklearn: InvalidParameterError)");
    PCH_REQUIRE(out_nclusters != nullptr, "out_nclusters is null");
    if (n == 0) {
        PCH_HIP_TRY(hipMemsetAsync(out_nclusters, 0, sizeof(int32_t), s));
        return PCH_OK;
    }
    PCH_REQUIRE(xyz && labels && ws, "null buffer");
    Arena a(ws, ws_bytes);
    DbWs w;
    db_plan(a, n, w);
    if (a.overflow) { set_error("workspace too small: need %zu bytes", a.off); return PCH_ERR_WORKSPACE; }
    if (chunk_size <= 0 || chunk_size > n) chunk_size = n;
    const int64_t nchunks = ceil_div(n, chunk_size);

    int64_t gstride = ceil_div(n, DB_THREADS * 8);
    if (gstride > 2048) gstride = 2048;
    float box[6];
    if (aabb_host) {
        memcpy(box, aabb_host, sizeof(box));
    } else {
        PCH_HIP_TRY(hipMemsetAsync(w.meta, 0xFF, 3 * sizeof(uint32_t), s));
        PCH_HIP_TRY(hipMemsetAsync(w.meta + 3, 0, 3 * sizeof(uint32_t), s));
        PCH_LAUNCH("db_aabb_in", db_aabb_in_k, dim3((unsigned)gstride), dim3(DB_THREADS), 0, s, xyz, n, w.meta);
        uint32_t mm[6];
        PCH_HIP_TRY(hipMemcpyAsync(mm, w.meta, sizeof(mm), hipMemcpyDeviceToHost, s));
        PCH_HIP_TRY(hipStreamSynchronize(s));
        if (mm[0] == 0xFFFFFFFFu) {                        // not a single finite point: every chunk fails
            PCH_HIP_TRY(hipMemsetAsync(labels, 0xFF, sizeof(int32_t) * (size_t)n, s));
            if (core) PCH_HIP_TRY(hipMemsetAsync(core, 0, (size_t)n, s));
            PCH_HIP_TRY(hipMemsetAsync(out_nclusters, 0, sizeof(int32_t), s));
            return PCH_OK;
        }
        for (int k = 0; k < 6; ++k) box[k] = host_unordered(mm[k]);
    }
    for (int k = 0; k < 6; ++k) {
        if (!(box[k] == box[k]) || box[k] > 3.0e38f || box[k] < -3.0e38f) {
            set_error("bounding box is not finite");
            return PCH_ERR_ARG;
        }
    }
    DbGrid g;
    g.ox = box[0]; g.oy = box[1]; g.oz = box[2];
    g.cell = eps / 1.7320508075688772 * (1.0 - 1.0 / 65536.0);
    g.inv_cell = 1.0 / g.cell;
    g.eps2 = eps * eps;
    g.eps2_lo = nextafterf((float)(g.eps2 * (1.0 - 1.0 / 1048576.0)), -INFINITY);
    g.eps2_hi = nextafterf((float)(g.eps2 * (1.0 + 1.0 / 1048576.0)), INFINITY);
    if (!(g.eps2_hi < 3.0e38f)) { g.eps2_lo = -1.0f; g.eps2_hi = NAN; }    // absurd eps: exact path only
    g.chunk_size = chunk_size;
    g.chunk_bad = w.chunk_bad;
    g.chunk_cells = w.chunk_cells;
    g.min_samples = min_samples;
    double ext[3];
    int mc[3] = {0, 0, 0};
    bool overflow = false;
    for (int k = 0; k < 3; ++k) {
        ext[k] = ((double)box[3 + k] - (double)box[k]) * g.inv_cell;
        if (!(ext[k] < 2.0e9)) overflow = true;
        else mc[k] = (int)ext[k] + 1;            // +1: slack for the rounding of the division
    }
    g.mx = mc[0]; g.my = mc[1]; g.mz = mc[2];
    g.bx = bits_for((uint64_t)g.mx + 1);
    g.by = bits_for((uint64_t)g.my + 1);
    g.bz = bits_for((uint64_t)g.mz + 1);
    int cellbits = g.bx + g.by + g.bz;
    int nbits = cellbits + bits_for((uint64_t)nchunks);
    if (nbits > 64) overflow = true;
    const unsigned gn = (unsigned)ceil_div(n, DB_THREADS);
    bool compressed = false;
    if (overflow) {
        // extent/eps beyond the 64-bit cell key (outliers, heavy tails).  Several chunks: every chunk on its own,
        // with its own bounding box (the reference fits them one by one anyway).  One chunk: compressed coordinates.
        if (nchunks > 1) {
            int32_t offset = 0;
            for (int64_t c = 0; c < nchunks; ++c) {
                const int64_t lo = c * chunk_size, cn = (n - lo) < chunk_size ? (n - lo) : chunk_size;
                int32_t kc = 0;
                PCH_TRY(dbscan_run(xyz + 3 * lo, cn, eps, min_samples, 0, nullptr, labels + lo, core ? core + lo : nullptr,
                                   out_nclusters, ws, ws_bytes, s, &kc));
                if (kc > 0 && offset > 0)
                    PCH_LAUNCH("db_add_offset", db_add_offset_k, dim3((unsigned)ceil_div(cn, DB_THREADS)), dim3(DB_THREADS), 0, s,
                               labels + lo, cn, offset);
                offset += kc;                    // utils/tower_extraction.py:114-116
            }
            PCH_HIP_TRY(hipMemcpyAsync(out_nclusters, &offset, sizeof(int32_t), hipMemcpyHostToDevice, s));
            PCH_HIP_TRY(hipStreamSynchronize(s));
            if (k_host) *k_host = offset;
            g_last.ws = nullptr;                 // no single grid is left to relabel on
            return PCH_OK;
        }
        compressed = true;
        int64_t* first_bad = reinterpret_cast<int64_t*>(w.meta + 12);
        PCH_HIP_TRY(hipMemsetAsync(first_bad, 0xFF, sizeof(int64_t), s));
        PCH_LAUNCH("db_first_bad", db_first_bad_k, dim3((unsigned)gstride), dim3(DB_THREADS), 0, s, xyz, n,
                   reinterpret_cast<unsigned long long*>(first_bad));
        PCH_HIP_TRY(hipMemsetAsync(w.meta + 6, 0, 6 * sizeof(uint32_t), s));
        for (int axis = 0; axis < 3; ++axis) {
            PCH_LAUNCH("dbc_axis_keys", dbc_axis_keys_k, dim3(gn), dim3(DB_THREADS), 0, s, xyz, n, axis, w.k0, w.v0);
            PCH_TRY(radix_sort_pairs(w.k0, w.v0, w.k1, w.v1, n, 32, w.radix_ws, s));
            const bool in1 = radix_sort_result_buffer(32) == 1;
            const uint64_t* ksa = in1 ? w.k1 : w.k0;
            const uint32_t* vsa = in1 ? w.v1 : w.v0;
            PCH_LAUNCH("dbc_heads", dbc_heads_k, dim3(gn), dim3(DB_THREADS), 0, s, ksa, n, g.inv_cell, w.flag2);
            PCH_TRY(scan_exclusive_u32(w.flag2, w.head, n, w.scan_ws, nullptr, s));
            PCH_HIP_TRY(hipMemsetAsync(w.cell_start, 0, sizeof(uint32_t) * (size_t)(n + 8), s));
            for (int phase = 0; phase < 2; ++phase)
                PCH_LAUNCH("dbc_local", dbc_local_k, dim3(gn), dim3(DB_THREADS), 0, s, ksa, (const uint32_t*)w.flag2,
                           (const uint32_t*)w.head, n, g.inv_cell, w.cell_box, phase, w.cid, w.cell_start, w.meta + 6);
            PCH_TRY(scan_exclusive_u32(w.cell_start, w.cell_start, n, w.scan_ws, nullptr, s));
            PCH_LAUNCH("dbc_comp", dbc_comp_k, dim3(gn), dim3(DB_THREADS), 0, s, vsa, (const uint32_t*)w.flag2,
                       (const uint32_t*)w.head, (const uint32_t*)w.cid, (const uint32_t*)w.cell_start, n, axis, w.comp,
                       w.meta + 9);
        }
        uint32_t back[8];                        // [0] status, [3..5] largest compressed index per axis, [6..7] first bad row
        PCH_HIP_TRY(hipMemcpyAsync(back, w.meta + 6, sizeof(back), hipMemcpyDeviceToHost, s));
        PCH_HIP_TRY(hipStreamSynchronize(s));
        int64_t bad_row;
        memcpy(&bad_row, &back[6], sizeof(bad_row));
        if (bad_row >= 0) {                      // NaN / inf in a single fit: sklearn rejects it, everything stays noise
            PCH_HIP_TRY(hipMemsetAsync(labels, 0xFF, sizeof(int32_t) * (size_t)n, s));
            if (core) PCH_HIP_TRY(hipMemsetAsync(core, 0, (size_t)n, s));
            PCH_HIP_TRY(hipMemsetAsync(out_nclusters, 0, sizeof(int32_t), s));
            g_last.ws = nullptr;
            return PCH_OK;
        }
        if (back[0] != 0) { set_error("compressed cell coordinates out of range"); return PCH_ERR_RANGE; }
        g.mx = (int)back[3]; g.my = (int)back[4]; g.mz = (int)back[5];
        g.bx = bits_for((uint64_t)g.mx + 1);
        g.by = bits_for((uint64_t)g.my + 1);
        g.bz = bits_for((uint64_t)g.mz + 1);
        cellbits = g.bx + g.by + g.bz;
        nbits = cellbits;
        if (nbits > 64) {
            set_error("cell key needs %d bits even with compressed coordinates (%lld isolated points?)", nbits, (long long)n);
            return PCH_ERR_RANGE;
        }
    }

    {   // status / cell count / cluster count words and the per-chunk first-cell table, in one 16-byte aligned fill
        // (meta[4..5] are bounding-box keys nobody reads any more; chunk_cells is written at the first cell of every
        // chunk by db_cells_k, the zeros make a chunk without a cell an empty range)
        const size_t bytes = (size_t)(reinterpret_cast<char*>(w.chunk_cells + nchunks + 1) - reinterpret_cast<char*>(w.meta + 4));
        PCH_HIP_TRY(hipMemsetAsync(w.meta + 4, 0, (bytes + 15) & ~size_t(15), s));
    }
    const uint64_t* ks;
    // One workgroup per chunk only pays with enough chunks to fill the GPU (measured break-even near
    // 100 chunks of 50 000 rows); PCH_DBSCAN_SORT=chunk / global forces a path (tests compare them)
    const int sort_mode = db_sort_mode();
    const bool force_global = sort_mode == 2;
    const bool force_chunk = sort_mode == 1;
    if (compressed) {
        PCH_HIP_TRY(hipMemsetAsync(w.chunk_bad, 0, sizeof(uint32_t) * (size_t)(nchunks + 1), s));
        PCH_LAUNCH("db_keys_comp", db_keys_comp_k, dim3(gn), dim3(DB_THREADS), 0, s, (const uint32_t*)w.comp, n, g, w.k0,
                   w.v0);
        PCH_TRY(radix_sort_pairs(w.k0, w.v0, w.k1, w.v1, n, nbits, w.radix_ws, s));
        const bool in1 = radix_sort_result_buffer(nbits) == 1;
        ks = in1 ? w.k1 : w.k0;
        const uint32_t* vs = in1 ? w.v1 : w.v0;
        PCH_LAUNCH("db_gather", db_gather_k, dim3(gn), dim3(DB_THREADS), 0, s, xyz, ks, vs, n, w.pts);
    } else if (cellbits <= 32 && chunk_size <= CS_MAX_CHUNK && !force_global && (force_chunk || nchunks >= CS_MIN_CHUNKS)) {
        // chunk-local path: one workgroup per chunk builds keys, sorts and gathers
        PCH_LAUNCH("db_chunksort", db_chunksort_k, dim3((unsigned)nchunks), dim3(CS_THREADS), 0, s, xyz, n, g,
                   w.chunk_bad, w.xbuf, w.pts, w.k1, w.meta + 6, (unsigned long long*)w.cell_box);
#ifdef PCH_CS_STAMPS
        {
            unsigned long long t[11];                    // start | sweep B | sweep H | one per pass | heads
            PCH_HIP_TRY(hipStreamSynchronize(s));
            PCH_HIP_TRY(hipMemcpy(t, w.cell_box, sizeof(t), hipMemcpyDeviceToHost));
            for (int q = 1; q <= 6; ++q)
                fprintf(stderr, "chunksort phase %d: %.2f us\n", q, (double)(long long)(t[q] - t[q - 1]) / 100.0);
            fprintf(stderr, "chunksort tiles: rank %.2f us, offsets %.2f us, scatter %.2f us\n", (double)t[8] / 100.0,
                    (double)t[9] / 100.0, (double)t[10] / 100.0);
        }
#endif
        ks = w.k1;
    } else {
        PCH_HIP_TRY(hipMemsetAsync(w.chunk_bad, 0, sizeof(uint32_t) * (size_t)(nchunks + 1), s));
        PCH_LAUNCH("db_chunkbad", db_chunkbad_k, dim3((unsigned)gstride), dim3(DB_THREADS), 0, s, xyz, n,
                   chunk_size, w.chunk_bad);
        PCH_LAUNCH("db_keys", db_keys_k, dim3(gn), dim3(DB_THREADS), 0, s, xyz, n, g,
                   (const uint32_t*)w.chunk_bad, w.k0, w.v0, w.meta + 6);
        PCH_TRY(radix_sort_pairs(w.k0, w.v0, w.k1, w.v1, n, nbits, w.radix_ws, s));
        const bool in1 = radix_sort_result_buffer(nbits) == 1;
        ks = in1 ? w.k1 : w.k0;
        const uint32_t* vs = in1 ? w.v1 : w.v0;
        PCH_LAUNCH("db_gather", db_gather_k, dim3(gn), dim3(DB_THREADS), 0, s, xyz, ks, vs, n, w.pts);
    }
    const int64_t ntile = ceil_div(n, (int64_t)SCAN_TILE);
    PCH_LAUNCH("db_heads", db_heads_k, dim3((unsigned)ntile), dim3(DB_THREADS), 0, s, ks, n, w.scan_ws);
    PCH_TRY(scan_tile_sums_u32(w.scan_ws, ntile, w.meta + 7, s));
    // the cell count sizes the next grids: fetch it while db_cells_k (sized by n) runs
    uint32_t st_m[2];
    PCH_TRY(peek_enqueue(w.meta + 6, sizeof(st_m), s));
    PCH_LAUNCH("db_cells", db_cells_k, dim3((unsigned)ntile), dim3(DB_THREADS), 0, s, ks, (const uint32_t*)w.scan_ws, n,
               cellbits, nchunks, w.cid, w.cell_start, w.cell_key, w.chunk_cells, w.cell_acc);
    PCH_TRY(peek_wait(st_m, sizeof(st_m)));
    if (st_m[0] != 0) {
        set_error("finite coordinates outside the supplied bounding box");
        return PCH_ERR_ARG;
    }
    const int m = (int)st_m[1];
    const unsigned gc = (unsigned)ceil_div(m, DB_WAVES);
    const int2* rowtab = nullptr;
    if (m <= w.rowtab_cells) {
        PCH_LAUNCH("db_rowtab", db_rowtab_k, dim3((unsigned)ceil_div(m, 2 * DB_WAVES)), dim3(DB_THREADS), 0, s,
                   g, (const uint64_t*)w.cell_key, m, w.rowtab);
        rowtab = w.rowtab;
    }
    if (g_count_pairs) {
        PCH_HIP_TRY(hipMemsetAsync(w.core_stats, 0, 4 * sizeof(unsigned long long), s));
        PCH_LAUNCH("db_core_counting", db_core_k<true>, dim3(gc), dim3(DB_THREADS), 0, s, g, (const float4*)w.pts,
                   (const uint32_t*)w.cell_start, (const uint64_t*)w.cell_key, m, rowtab, w.core_s, w.cell_ncore,
                   w.core_stats);
    } else {
        PCH_LAUNCH("db_core", db_core_k<false>, dim3(gc), dim3(DB_THREADS), 0, s, g, (const float4*)w.pts,
                   (const uint32_t*)w.cell_start, (const uint64_t*)w.cell_key, m, rowtab, w.core_s, w.cell_ncore,
                   (unsigned long long*)nullptr);
    }
    PCH_LAUNCH("db_cellstats", db_cellstats_k, dim3((unsigned)ceil_div(n, (int64_t)DB_WAVES * 64 * DB_CS_ROUNDS)),
               dim3(DB_THREADS), 0, s, (const float4*)w.pts, (const uint32_t*)w.cid, (const uint8_t*)w.core_s, n,
               w.cell_acc, w.flag, ceil_div(n, 32));
    PCH_LAUNCH("db_cellfin", db_cellfin_k, dim3((unsigned)ceil_div(m, DB_THREADS)), dim3(DB_THREADS), 0, s,
               (const uint32_t*)w.cell_acc, m, w.cell_box, w.cell_min, w.parent, w.comp_min);
    // face neighbours: lane-per-pair first (needs the row table), the wave-wide search for what that left open;
    // the flags live in w.root, which nobody needs before db_compmin_k
    const uint8_t* face_todo = nullptr;
    if (rowtab) {
        PCH_LAUNCH("db_union_pairs", db_union_pairs_k, dim3((unsigned)ceil_div(4 * (int64_t)m, DB_THREADS)),
                   dim3(DB_THREADS), 0, s, g, (const float4*)w.pts, (const uint32_t*)w.cell_start,
                   (const uint64_t*)w.cell_key, m, rowtab, (const uint8_t*)w.core_s, (const uint32_t*)w.cell_ncore,
                   (const float*)w.cell_box, w.parent, reinterpret_cast<uint8_t*>(w.root));
        face_todo = reinterpret_cast<const uint8_t*>(w.root);
    }
    PCH_LAUNCH("db_union0", db_union_face_k, dim3(gc), dim3(DB_THREADS), 0, s, g, (const float4*)w.pts,
               (const uint32_t*)w.cell_start, (const uint64_t*)w.cell_key, m, rowtab, (const uint8_t*)w.core_s,
               (const uint32_t*)w.cell_ncore, (const float*)w.cell_box, w.parent, face_todo);
    PCH_LAUNCH("db_flatten", db_flatten_k, dim3((unsigned)ceil_div(m, DB_THREADS)), dim3(DB_THREADS), 0, s,
               w.parent, m);
    PCH_LAUNCH("db_union1", db_union_k, dim3(gc), dim3(DB_THREADS), 0, s, g, (const float4*)w.pts,
               (const uint32_t*)w.cell_start, (const uint64_t*)w.cell_key, m, rowtab, (const uint8_t*)w.core_s,
               (const uint32_t*)w.cell_ncore, (const float*)w.cell_box, w.parent);
    PCH_LAUNCH("db_compmin", db_compmin_k, dim3((unsigned)ceil_div(m, DB_THREADS)), dim3(DB_THREADS), 0, s,
               (const int*)w.cell_min, (const uint32_t*)w.cell_ncore, m, w.parent, w.root, w.comp_min);
    // cluster id = rank of the cluster's smallest core row: a bitmap over the rows + a scan of its
    // word counts (n/32 elements instead of n)
    const int64_t nw = ceil_div(n, 32);
    uint32_t* bits = w.flag;
    uint32_t* wrank = w.flag + ((nw + 63) & ~int64_t(63));
    static_assert(DB_CS_ROUNDS * 64 * DB_WAVES / DB_THREADS <= 32, "db_cellstats_k's grid has a thread per bitmap word");
    PCH_LAUNCH("db_mark", db_mark_k, dim3((unsigned)ceil_div(m, DB_THREADS)), dim3(DB_THREADS), 0, s,
               (const int*)w.root, (const int*)w.comp_min, m, bits);
    // word ranks: popcount on load
    if (scan1_pays(nw)) PCH_TRY(scan1_exclusive_popc_u32(bits, wrank, nw, w.scan1_b, w.meta + 8, s));
    else PCH_TRY(scan_exclusive_popc_u32(bits, wrank, nw, w.scan_ws, w.meta + 8, s));
    static const bool no_fold = getenv("PCH_DB_NO_BOXFOLD") != nullptr;     // tuning toggle
    uint32_t* box_acc = (boxes && boxes->acc && boxes->cap > 0 && !no_fold) ? boxes->acc : nullptr;
    const int32_t box_cap = box_acc ? boxes->cap : 0;
    PCH_LAUNCH("db_prelabel", db_prelabel_k, dim3((unsigned)(box_acc ? ceil_div(8 * (int64_t)box_cap, 256) : 1)),
               dim3(256), 0, s, (const uint32_t*)(w.meta + 8), out_nclusters, box_acc, 8 * (int64_t)box_cap);
    if (k_host) PCH_TRY(peek_enqueue(out_nclusters, sizeof(int32_t), s));    // read while the labels are written
    {
        // many chunks: the blocks of one chunk share an XCD (see db_label_k); otherwise blocks in sorted order
        static const bool no_xcd = getenv("PCH_DB_NO_XCD") != nullptr;      // tuning toggle
        const int bpc = (nchunks >= 16 && !no_xcd) ? (int)ceil_div(chunk_size, DB_LAB_TILE) : 0;
        const unsigned gl = bpc > 0 ? (unsigned)(8 * ceil_div(nchunks, 8) * bpc) : (unsigned)ceil_div(n, DB_LAB_TILE);
        PCH_LAUNCH("db_label", db_label_k, dim3(gl), dim3(DB_THREADS), 0, s, (const float4*)w.pts,
                   (const uint32_t*)w.cid, (const uint8_t*)w.core_s, (const int*)w.root,
                   (const int*)w.comp_min, (const uint32_t*)bits, (const uint32_t*)wrank, n,
                   (const uint32_t*)w.cell_start, w.cell_label, labels, core, chunk_size, bpc, nchunks, box_acc, box_cap);
    }
    PCH_LAUNCH("db_border", db_border_k, dim3(gc), dim3(DB_THREADS), 0, s, g, (const float4*)w.pts,
               (const uint32_t*)w.cell_start, (const uint64_t*)w.cell_key, m, rowtab, (const uint8_t*)w.core_s,
               (const uint32_t*)w.cell_ncore, (const float*)w.cell_box, (const int*)w.cell_label, labels, box_acc, box_cap);
    if (box_acc) boxes->done = true;
    if (k_host) {
        PCH_TRY(peek_wait(k_host, sizeof(int32_t)));
        if (*k_host < 0) {                              // the rank scan's bounded wait gave up: the count reads -1
            set_error("stage C: a device-side look-back wait ran out of its budget; outputs are undefined");
            return PCH_ERR_TIMEOUT;
        }
    }
    g_last.ws = ws; g_last.ws_bytes = ws_bytes; g_last.n = n; g_last.m = m; g_last.g = g;
    g_last.has_rowtab = rowtab != nullptr;
    return PCH_OK;
}

extern "C" int pch_dbscan_first_core_rows_i32(int64_t n, int32_t* out_rows, void* ws, size_t ws_bytes, void* stream) {
    PCH_DEVICE_GUARD(out_rows ? (const void*)out_rows : (const void*)ws);
    PCH_REQUIRE(n >= 0, "bad argument");
    if (n == 0) return PCH_OK;
    if (g_last.ws != ws || g_last.ws_bytes != ws_bytes || g_last.n != n || ws == nullptr) {
        set_error("pch_dbscan_first_core_rows_i32 must follow pch_dbscan_f32 of this thread on the same, untouched workspace");
        return PCH_ERR_ARG;
    }
    PCH_REQUIRE(out_rows != nullptr, "null output");
    Arena a(ws, ws_bytes, true);
    DbWs w;
    db_plan(a, n, w);
    const int64_t nw = ceil_div(n, 32);
    const uint32_t* bits = w.flag;
    const uint32_t* wrank = w.flag + ((nw + 63) & ~int64_t(63));
    PCH_LAUNCH("db_first_rows", db_first_rows_k, dim3((unsigned)ceil_div(nw, DB_THREADS)), dim3(DB_THREADS), 0,
               (hipStream_t)stream, bits, wrank, nw, out_rows);
    return PCH_OK;
}

extern "C" void pch_dbscan_set_pair_counting(int enable) { g_count_pairs = enable != 0; }

extern "C" int pch_dbscan_pair_stats(int64_t n, uint64_t* out4_host, void* ws, size_t ws_bytes, void* stream) {
    PCH_DEVICE_GUARD(ws);
    PCH_REQUIRE(n >= 0 && out4_host, "bad argument");
    memset(out4_host, 0, 4 * sizeof(uint64_t));
    if (n == 0) return PCH_OK;
    if (g_last.ws != ws || g_last.ws_bytes != ws_bytes || g_last.n != n || ws == nullptr) {
        set_error("pch_dbscan_pair_stats must follow pch_dbscan_f32 of this thread on the same, untouched workspace");
        return PCH_ERR_ARG;
    }
    Arena a(ws, ws_bytes, true);
    DbWs w;
    db_plan(a, n, w);
    PCH_HIP_TRY(hipMemcpyAsync(out4_host, w.core_stats, 4 * sizeof(uint64_t), hipMemcpyDeviceToHost, (hipStream_t)stream));
    PCH_HIP_TRY(hipStreamSynchronize((hipStream_t)stream));
    return PCH_OK;
}

extern "C" int pch_dbscan_strip_pairs_i32(int64_t n, float x_lo, float x_hi, int32_t cap, int32_t* out_pairs,
                                          int32_t* out_count, void* ws, size_t ws_bytes, void* stream) {
    PCH_DEVICE_GUARD(out_count);
    hipStream_t s = (hipStream_t)stream;
    PCH_REQUIRE(n >= 0 && cap >= 0 && out_count && (cap == 0 || out_pairs), "bad argument");
    PCH_HIP_TRY(hipMemsetAsync(out_count, 0, sizeof(int32_t), s));
    if (n == 0 || !(x_lo < x_hi)) return PCH_OK;
    if (g_last.ws != ws || g_last.ws_bytes != ws_bytes || g_last.n != n || ws == nullptr) {
        set_error("pch_dbscan_strip_pairs_i32 must follow pch_dbscan_f32 of this thread on the same, untouched workspace");
        return PCH_ERR_ARG;
    }
    Arena a(ws, ws_bytes, true);
    DbWs w;
    db_plan(a, n, w);
    const int m = g_last.m;
    PCH_LAUNCH("db_strip_pairs", db_strip_pairs_k, dim3((unsigned)ceil_div(m, DB_WAVES)), dim3(DB_THREADS), 0, s,
               (const float4*)w.pts, (const uint32_t*)w.cell_start, (const uint8_t*)w.core_s,
               (const uint32_t*)w.cell_ncore, (const float*)w.cell_box, (const int*)w.cell_label, m, x_lo, x_hi, cap,
               out_pairs, out_count);
    return PCH_OK;
}

extern "C" int pch_dbscan_relabel_i32(const int32_t* map, int32_t nmap, int64_t n, int32_t* labels, void* ws,
                                      size_t ws_bytes, void* stream) {
    PCH_DEVICE_GUARD(labels);
    hipStream_t s = (hipStream_t)stream;
    PCH_REQUIRE(n >= 0 && nmap >= 0 && (nmap == 0 || map) && (n == 0 || labels), "bad argument");
    if (n == 0) return PCH_OK;
    if (g_last.ws != ws || g_last.ws_bytes != ws_bytes || g_last.n != n || ws == nullptr) {
        set_error("pch_dbscan_relabel_i32 must follow pch_dbscan_f32 of this thread on the same, untouched workspace");
        return PCH_ERR_ARG;
    }
    Arena a(ws, ws_bytes, true);
    DbWs w;
    db_plan(a, n, w);
    const DbGrid g = g_last.g;
    const int m = g_last.m;
    const unsigned gn = (unsigned)ceil_div(n, DB_THREADS);
    const unsigned gc = (unsigned)ceil_div(m, DB_WAVES);
    PCH_LAUNCH("db_remap_points", db_remap_points_k, dim3(gn), dim3(DB_THREADS), 0, s, (const float4*)w.pts,
               (const uint8_t*)w.core_s, n, map, nmap, labels);
    PCH_LAUNCH("db_remap_cells", db_remap_cells_k, dim3((unsigned)ceil_div(m, DB_THREADS)), dim3(DB_THREADS), 0, s,
               w.cell_label, (const uint32_t*)w.cell_ncore, m, map, nmap);
    PCH_LAUNCH("db_border", db_border_k, dim3(gc), dim3(DB_THREADS), 0, s, g, (const float4*)w.pts,
               (const uint32_t*)w.cell_start, (const uint64_t*)w.cell_key, m,
               g_last.has_rowtab ? (const int2*)w.rowtab : (const int2*)nullptr, (const uint8_t*)w.core_s,
               (const uint32_t*)w.cell_ncore, (const float*)w.cell_box, (const int*)w.cell_label, labels,
               (uint32_t*)nullptr, 0);
    return PCH_OK;
}

extern "C" int pch_dbscan_f32(const float* xyz, int64_t n, double eps, int32_t min_samples,
                              int64_t chunk_size, const float* aabb_host, int32_t* labels,
                              uint8_t* core, int32_t* out_nclusters, void* ws, size_t ws_bytes,
                              void* stream) {
    PCH_DEVICE_GUARD(xyz ? (const void*)xyz : (const void*)out_nclusters);
    return dbscan_run(xyz, n, eps, min_samples, chunk_size, aabb_host, labels, core, out_nclusters, ws, ws_bytes,
                      (hipStream_t)stream, nullptr);
}
