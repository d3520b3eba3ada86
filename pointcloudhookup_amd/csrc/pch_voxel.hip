// Stage A: per-chunk voxel-grid downsample (Open3D VoxelDownSample semantics): the rows of every
// chunk are sorted by voxel index (one MSD partition in HBM, the rest inside LDS) and reduced in
// file order, one float64 running sum per voxel.
// Reference call site: ui/import_PC.py:8-13 inside the chunk loop ui/import_PC.py:45-58.
#include "pch_prims.h"
#include "pch_lookback.h"

namespace pch {

constexpr int VX_THREADS = 256;
constexpr int VX_MM_ROUNDS = 4;

// ---- per-chunk float64 min / max (ordered-uint64 atomics) ---------------------------
__global__ __launch_bounds__(VX_THREADS) void vx_minmax_k(const double* __restrict__ xyz,
                                                          int64_t n, int64_t chunk_size,
                                                          int64_t blocks_per_chunk,
                                                          unsigned long long* __restrict__ mm) {
    __shared__ unsigned long long sm[VX_THREADS / 64][6];
    const int64_t chunk = blockIdx.x / blocks_per_chunk;
    const int64_t tile  = blockIdx.x % blocks_per_chunk;
    const int64_t cbeg = chunk * chunk_size;
    const int64_t cend = (cbeg + chunk_size < n) ? cbeg + chunk_size : n;
    unsigned long long lo[3] = {~0ull, ~0ull, ~0ull}, hi[3] = {0ull, 0ull, 0ull};
    for (int r = 0; r < 4; ++r) {
        const int64_t i = cbeg + (tile * 4 + r) * VX_THREADS + threadIdx.x;
        if (i < cend) {
#pragma unroll
            for (int a = 0; a < 3; ++a) {
                const unsigned long long k = f64_ordered(xyz[3 * i + a]);
                lo[a] = k < lo[a] ? k : lo[a];
                hi[a] = k > hi[a] ? k : hi[a];
            }
        }
    }
#pragma unroll
    for (int a = 0; a < 3; ++a) {
        lo[a] = wave_reduce_min(lo[a]);
        hi[a] = wave_reduce_max(hi[a]);
    }
    if (lane_id() == 0) {
#pragma unroll
        for (int a = 0; a < 3; ++a) { sm[wave_id()][a] = lo[a]; sm[wave_id()][3 + a] = hi[a]; }
    }
    __syncthreads();
    if (threadIdx.x < 6) {
        const int a = threadIdx.x;
        unsigned long long v = sm[0][a];
        for (int w = 1; w < VX_THREADS / 64; ++w) {
            const unsigned long long o = sm[w][a];
            v = (a < 3) ? (o < v ? o : v) : (o > v ? o : v);
        }
        if (a < 3) atomicMin(&mm[chunk * 6 + a], v);
        else       atomicMax(&mm[chunk * 6 + a], v);
    }
}

// one thread per chunk: grid origin, largest voxel index per axis, Open3D's range check
__global__ void vx_bounds_k(const unsigned long long* __restrict__ mm, int64_t nchunks,
                            double voxel, double* __restrict__ minb, int* __restrict__ gmeta) {
    const int64_t c = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= nchunks) return;
    double ext = 0.0;
#pragma unroll
    for (int a = 0; a < 3; ++a) {
        const double lo = f64_unordered(mm[c * 6 + a]);
        const double hi = f64_unordered(mm[c * 6 + 3 + a]);
        const double mb = lo - voxel * 0.5;           // voxel_min_bound
        const double xb = hi + voxel * 0.5;           // voxel_max_bound
        minb[c * 3 + a] = mb;
        const double e = xb - mb;
        ext = e > ext ? e : ext;
        const double q = floor((hi - mb) / voxel);
        int qi = (q >= 0.0 && q < 2147483647.0) ? (int)q : 2147483647;
        if (!(q == q)) qi = 2147483647;               // NaN coordinates
        atomicMax(&gmeta[a], qi);
    }
    if (voxel * 2147483647.0 < ext || !(ext == ext)) atomicMax(&gmeta[3], 1);
}

// =====================================================================================
// Sorting the rows of every chunk by voxel index WITHOUT a key/index stream and without a final
// gather (a 24-byte gather through a sorted index costs a whole cache line per point):
//
//   level 1  one stable MSD partition of each chunk by the top D1 <= 9 bits of the packed voxel
//            index [ix|iy|iz]: what moves is the row itself (three doubles).  Three launches:
//            per-tile digit histograms, a per-chunk scan, the scatter (tile-local stable ranks from
//            wave ballots, as db_chunksort_k).  Runs of a tile are ~8 rows per bin: coalesced enough.
//   finish   the (chunk, digit) units are taken in order by persistent workgroups (tickets).  A unit
//            of <= VF_CAP rows is sorted inside LDS - only (key, row) words move there, the rows stay
//            in L2 where the unit was just read - and reduced straight from there: one thread per
//            voxel adds its rows in file order (the level-1 partition and the LDS sort are stable),
//            which is Open3D's AccumulatedPoint::AddPoint order, so the means are bit-exact.
//            A larger unit (tower cores, degenerate inputs) is sorted by its workgroup with LSD passes
//            in global memory first.  The first output slot of a unit comes from a decoupled
//            look-back over the units in front of it.
// HBM traffic per point: 24 (min/max) + 24 (histogram) + 48 (scatter) + 24 (finish) + 40 per voxel.
// =====================================================================================
constexpr int VP_THREADS = 512;
constexpr int VP_WAVES   = VP_THREADS / 64;
constexpr int VP_ROUNDS  = 8;                        // rows per thread and tile
constexpr int VP_TILE    = VP_THREADS * VP_ROUNDS;   // 4096 rows
constexpr int VP_MAXBITS = 9;
constexpr int VP_MAXBINS = 1 << VP_MAXBITS;
constexpr int VG_ROUNDS  = 4;                        // general (global-memory) path: rows per thread and tile
constexpr int VG_TILE    = VP_THREADS * VG_ROUNDS;
constexpr int VF_CAP     = 4096;                     // rows of a unit that is sorted inside LDS
constexpr int VF_ROWBITS = 12;                       // log2(VF_CAP): local row field of a packed LDS item

struct Row { double x, y, z; };

struct VoxelPlan {
    int64_t n, chunk_size, nchunks, tiles_per_chunk;
    double  voxel;
    int     bx, by, bz;      // bits per axis
    int     T;               // bx + by + bz
    int     d1;              // level-1 digit bits
    int     nb;              // 1 << d1
    int     rem;             // T - d1: bits the finisher sorts on
};

__device__ __forceinline__ uint64_t vx_key(const VoxelPlan& g, const double* __restrict__ mb, const Row& q) {
    // ref_coord = (p - voxel_min_bound) / voxel_size ; index = floor(ref_coord)   (float64, IEEE division)
    const uint64_t ix = (uint64_t)(int64_t)floor((q.x - mb[0]) / g.voxel);
    const uint64_t iy = (uint64_t)(int64_t)floor((q.y - mb[1]) / g.voxel);
    const uint64_t iz = (uint64_t)(int64_t)floor((q.z - mb[2]) / g.voxel);
    return (((ix << g.by) | iy) << g.bz) | iz;
}

// rank of this lane among the lanes of its wave that hold the same `bits`-bit digit (valid lanes only),
// and the number of such lanes
template <int BITS>
__device__ __forceinline__ uint32_t vx_match(uint32_t d, bool valid, uint32_t& peers_out) {
    uint64_t peers = __ballot(valid);
#pragma unroll
    for (int b = 0; b < BITS; ++b) {
        const bool bit = (d >> b) & 1u;
        const uint64_t m = __ballot(bit);
        peers &= bit ? m : ~m;
    }
    peers_out = (uint32_t)__popcll(peers);
    return (uint32_t)__popcll(peers & lanemask_lt());
}

// ---- level 1a: digit histogram of every tile ------------------------------------------------
__global__ __launch_bounds__(VP_THREADS) void vx_tilehist_k(const double* __restrict__ xyz, VoxelPlan g,
                                                            const double* __restrict__ minb,
                                                            uint32_t* __restrict__ tile_hist) {
    __shared__ uint32_t hist[VP_MAXBINS];
    const int64_t c = blockIdx.x / g.tiles_per_chunk, t = blockIdx.x % g.tiles_per_chunk;
    const int64_t cbeg = c * g.chunk_size, cend = (cbeg + g.chunk_size < g.n) ? cbeg + g.chunk_size : g.n;
    const int64_t t0 = cbeg + t * VP_TILE;
    for (int j = threadIdx.x; j < g.nb; j += VP_THREADS) hist[j] = 0;
    __syncthreads();
    const double mb[3] = {minb[3 * c + 0], minb[3 * c + 1], minb[3 * c + 2]};
    const Row* __restrict__ rows = reinterpret_cast<const Row*>(xyz);
    Row q[VP_ROUNDS];
#pragma unroll
    for (int r = 0; r < VP_ROUNDS; ++r) {
        const int64_t i = t0 + r * VP_THREADS + threadIdx.x;
        q[r] = rows[i < cend ? i : cbeg];
    }
#pragma unroll
    for (int r = 0; r < VP_ROUNDS; ++r) {
        const int64_t i = t0 + r * VP_THREADS + threadIdx.x;
        if (i < cend) atomicAdd(&hist[(uint32_t)(vx_key(g, mb, q[r]) >> g.rem)], 1u);
    }
    __syncthreads();
    for (int j = threadIdx.x; j < g.nb; j += VP_THREADS) tile_hist[(int64_t)blockIdx.x * g.nb + j] = hist[j];
}

// ---- level 1b: one workgroup per chunk: tile_hist -> exclusive offsets inside every bin, bin starts ----
__global__ __launch_bounds__(VP_MAXBINS) void vx_binscan_k(VoxelPlan g, uint32_t* __restrict__ tile_hist,
                                                           uint32_t* __restrict__ unit_start,
                                                           uint32_t* __restrict__ unit_count) {
    __shared__ uint32_t wsum[VP_MAXBINS / 64];
    const int64_t c = blockIdx.x;
    const int b = threadIdx.x;
    uint32_t run = 0;
    if (b < g.nb) {
        for (int64_t t = 0; t < g.tiles_per_chunk; ++t) {
            const int64_t at = (c * g.tiles_per_chunk + t) * g.nb + b;
            const uint32_t h = tile_hist[at];
            tile_hist[at] = run;
            run += h;
        }
    }
    const uint32_t incl = wave_scan_incl(run);
    if (lane_id() == 63) wsum[wave_id()] = incl;
    __syncthreads();
    uint32_t base = incl - run;
    for (int w = 0; w < wave_id(); ++w) base += wsum[w];
    if (b < g.nb) {
        unit_start[c * g.nb + b] = (uint32_t)(c * g.chunk_size) + base;
        unit_count[c * g.nb + b] = run;
    }
}

// ---- level 1c: stable scatter of the rows into their (chunk, digit) units -----------------------
__global__ __launch_bounds__(VP_THREADS) void vx_scatter_k(const double* __restrict__ xyz, VoxelPlan g,
                                                           const double* __restrict__ minb,
                                                           const uint32_t* __restrict__ tile_hist,
                                                           const uint32_t* __restrict__ unit_start,
                                                           Row* __restrict__ out) {
    __shared__ uint32_t cnt[VP_WAVES][VP_MAXBINS];
    __shared__ uint32_t base[VP_MAXBINS];
    const int64_t c = blockIdx.x / g.tiles_per_chunk, t = blockIdx.x % g.tiles_per_chunk;
    const int64_t cbeg = c * g.chunk_size, cend = (cbeg + g.chunk_size < g.n) ? cbeg + g.chunk_size : g.n;
    const int64_t t0 = cbeg + t * VP_TILE;
    if (t0 >= cend) return;
    const int w = wave_id(), l = lane_id();
    for (int j = threadIdx.x; j < VP_WAVES * g.nb; j += VP_THREADS) cnt[j / g.nb][j % g.nb] = 0;
    for (int j = threadIdx.x; j < g.nb; j += VP_THREADS)
        base[j] = unit_start[c * g.nb + j] + tile_hist[(int64_t)blockIdx.x * g.nb + j];
    __syncthreads();
    const double mb[3] = {minb[3 * c + 0], minb[3 * c + 1], minb[3 * c + 2]};
    const Row* __restrict__ rows = reinterpret_cast<const Row*>(xyz);
    // wave w owns rows [w*512, w*512+512) of the tile, 64 per round: (wave, round, lane) order = file order
    const int64_t seg = t0 + (int64_t)w * (64 * VP_ROUNDS);
    Row q[VP_ROUNDS];
    uint32_t dig[VP_ROUNDS], rank[VP_ROUNDS];
#pragma unroll
    for (int r = 0; r < VP_ROUNDS; ++r) {
        const int64_t i = seg + r * 64 + l;
        q[r] = rows[i < cend ? i : cbeg];
    }
#pragma unroll
    for (int r = 0; r < VP_ROUNDS; ++r) {
        const bool valid = seg + r * 64 + l < cend;
        dig[r] = (uint32_t)(vx_key(g, mb, q[r]) >> g.rem);
        uint32_t np;
        const uint32_t rk = vx_match<VP_MAXBITS>(dig[r], valid, np);
        const uint32_t prior = cnt[w][dig[r]];
        __builtin_amdgcn_wave_barrier();
        if (valid && rk == 0) cnt[w][dig[r]] = prior + np;
        __builtin_amdgcn_wave_barrier();
        rank[r] = prior + rk;
    }
    __syncthreads();
    for (int d = threadIdx.x; d < g.nb; d += VP_THREADS) {       // per digit: waves in order
        uint32_t run = base[d];
#pragma unroll
        for (int w2 = 0; w2 < VP_WAVES; ++w2) {
            const uint32_t cc = cnt[w2][d];
            cnt[w2][d] = run;
            run += cc;
        }
    }
    __syncthreads();
#pragma unroll
    for (int r = 0; r < VP_ROUNDS; ++r)
        if (seg + r * 64 + l < cend) out[cnt[w][dig[r]] + rank[r]] = q[r];
}

// ---- finish ------------------------------------------------------------------------------------
struct VfShared {
    union {
        // LDS sort of one unit: packed (key << VF_ROWBITS | local row) items, two buffers
        unsigned long long item[2][VF_CAP];      // 64 KB (the 32-bit variant uses the first half of each)
        uint32_t hist[8][256];                   // general path: digit histograms of every LSD pass
    };
    uint32_t cnt[VP_WAVES][256];                 // per-wave digit counters / offsets
    uint32_t base[256];
    uint32_t wsum[VP_WAVES];
    uint32_t unit, nvox, vbase;
};

// block-wide exclusive scan of one value per thread (VP_THREADS threads); total returned to all
__device__ __forceinline__ uint32_t vf_block_scan(uint32_t v, uint32_t* wsum, uint32_t& total) {
    const uint32_t incl = wave_scan_incl(v);
    __syncthreads();
    if (lane_id() == 63) wsum[wave_id()] = incl;
    __syncthreads();
    uint32_t before = 0, tot = 0;
#pragma unroll
    for (int w = 0; w < VP_WAVES; ++w) {
        const uint32_t s = wsum[w];
        if (w < wave_id()) before += s;
        tot += s;
    }
    total = tot;
    return before + incl - v;
}

__device__ __forceinline__ void vf_emit(const VoxelPlan& g, uint64_t key, double ax, double ay, double az,
                                        uint32_t count, int64_t slot, int32_t* __restrict__ out_idx,
                                        double* __restrict__ out_mean, int32_t* __restrict__ out_count) {
    const double cnt = (double)count;
    out_mean[3 * slot + 0] = ax / cnt;                  // GetAveragePoint
    out_mean[3 * slot + 1] = ay / cnt;
    out_mean[3 * slot + 2] = az / cnt;
    out_count[slot] = (int32_t)count;
    out_idx[3 * slot + 2] = (int32_t)(key & ((1ull << g.bz) - 1));
    out_idx[3 * slot + 1] = (int32_t)((key >> g.bz) & ((1ull << g.by) - 1));
    out_idx[3 * slot + 0] = (int32_t)((key >> (g.bz + g.by)) & ((1ull << g.bx) - 1));
}

// P = uint32_t when rem + VF_ROWBITS <= 32, else unsigned long long
template <typename P>
__global__ __launch_bounds__(VP_THREADS, 4) void vx_finish_k(
    VoxelPlan g, const double* __restrict__ minb, const uint32_t* __restrict__ unit_start,
    const uint32_t* __restrict__ unit_count, Row* __restrict__ bufA, Row* __restrict__ bufB,
    uint32_t* __restrict__ vstart_g, uint64_t* __restrict__ status, uint32_t* __restrict__ ticket,
    int32_t* __restrict__ out_idx, double* __restrict__ out_mean, int32_t* __restrict__ out_count,
    int64_t* __restrict__ out_chunk_offsets, int64_t* __restrict__ out_m) {
    extern __shared__ __attribute__((aligned(16))) unsigned char vf_raw[];
    VfShared& sh = *reinterpret_cast<VfShared*>(vf_raw);
    const int tid = threadIdx.x, w = wave_id(), l = lane_id();
    const int64_t nunits = g.nchunks * g.nb;
    const uint64_t remmask = g.rem >= 64 ? ~0ull : ((1ull << g.rem) - 1);
    P* it0 = reinterpret_cast<P*>(sh.item[0]);
    P* it1 = reinterpret_cast<P*>(sh.item[1]);
    for (;;) {
        __syncthreads();                                   // previous unit's LDS reads are done
        if (tid == 0) sh.unit = atomicAdd(ticket, 1u);     // units are taken in order of arrival (look-back below)
        __syncthreads();
        const int64_t u = sh.unit;
        if (u >= nunits) return;
        const int64_t c = u / g.nb;
        const uint64_t dtop = (uint64_t)(u % g.nb);
        const uint32_t s = unit_start[u], R = unit_count[u];
        const double mb[3] = {minb[3 * c + 0], minb[3 * c + 1], minb[3 * c + 2]};
        const int passes = (g.rem + 7) / 8;
        uint32_t nvox = 0;
        const Row* fin = bufA + s;                         // where the unit's rows are when they are reduced
        bool in_lds = false;
        if (R == 0) {
            // empty unit: publishes zero voxels below
        } else if (R <= (uint32_t)VF_CAP && g.rem + VF_ROWBITS <= (int)(8 * sizeof(P))) {
            // ---------------- LDS path: load keys, stable LSD passes over packed items ----------------
            in_lds = true;
            // item index of (wave, round, lane): wave w owns a block of `per` consecutive items
            const uint32_t per = ((R + VP_WAVES * 64 - 1) / (VP_WAVES * 64)) * 64;   // multiple of 64, <= 512
            const int rounds = (int)(per / 64);
            for (uint32_t i = tid; i < R; i += VP_THREADS) {
                const uint64_t k = vx_key(g, mb, bufA[s + i]) & remmask;
                it0[i] = (P)((k << VF_ROWBITS) | i);
            }
            __syncthreads();
            P* src = it0;
            P* dst = it1;
            for (int p = 0; p < passes; ++p) {
                const int shift = VF_ROWBITS + 8 * p;
                for (int j = tid; j < VP_WAVES * 256; j += VP_THREADS) (&sh.cnt[0][0])[j] = 0;
                __syncthreads();
                P item[VP_ROUNDS];
                uint32_t rank[VP_ROUNDS];
#pragma unroll
                for (int r = 0; r < VP_ROUNDS; ++r) {
                    if (r < rounds) {                      // wave-uniform
                        const uint32_t i = w * per + r * 64 + l;
                        const bool valid = i < R;
                        item[r] = valid ? src[i] : (P)0;
                        const uint32_t d = (uint32_t)(item[r] >> shift) & 255u;
                        uint32_t np;
                        const uint32_t rk = vx_match<8>(d, valid, np);
                        const uint32_t prior = sh.cnt[w][d];
                        __builtin_amdgcn_wave_barrier();
                        if (valid && rk == 0) sh.cnt[w][d] = prior + np;
                        __builtin_amdgcn_wave_barrier();
                        rank[r] = prior + rk;
                    }
                }
                __syncthreads();
                uint32_t tot = 0;
                if (tid < 256) {
#pragma unroll
                    for (int w2 = 0; w2 < VP_WAVES; ++w2) tot += sh.cnt[w2][tid];
                }
                uint32_t all;
                const uint32_t ex = vf_block_scan(tid < 256 ? tot : 0u, sh.wsum, all);
                if (tid < 256) {
                    uint32_t run = ex;
#pragma unroll
                    for (int w2 = 0; w2 < VP_WAVES; ++w2) {
                        const uint32_t cc = sh.cnt[w2][tid];
                        sh.cnt[w2][tid] = run;
                        run += cc;
                    }
                }
                __syncthreads();
#pragma unroll
                for (int r = 0; r < VP_ROUNDS; ++r) {
                    if (r < rounds) {
                        const uint32_t i = w * per + r * 64 + l;
                        if (i < R) {
                            const uint32_t d = (uint32_t)(item[r] >> shift) & 255u;
                            dst[sh.cnt[w][d] + rank[r]] = item[r];
                        }
                    }
                }
                __syncthreads();
                P* tswap = src; src = dst; dst = tswap;
            }
            // ---- voxel heads of the sorted items; voxel v starts at item vstart[v] (kept in `dst`, as P)
            uint32_t heads = 0;
            const uint32_t i0 = tid * VP_ROUNDS;            // 8 consecutive items per thread
            P mine[VP_ROUNDS];
            bool hd[VP_ROUNDS];
#pragma unroll
            for (int j = 0; j < VP_ROUNDS; ++j) {
                const uint32_t i = i0 + j;
                mine[j] = i < R ? src[i] : (P)0;
                const P prev = (i > 0 && i < R) ? src[i - 1] : (P)0;
                hd[j] = i < R && (i == 0 || (mine[j] >> VF_ROWBITS) != (prev >> VF_ROWBITS));
                heads += hd[j];
            }
            uint32_t vi = vf_block_scan(heads, sh.wsum, nvox);
#pragma unroll
            for (int j = 0; j < VP_ROUNDS; ++j)
                if (hd[j]) dst[vi++] = (P)(i0 + j);
            __syncthreads();
            if (tid == 0) sh.nvox = nvox;
            // (src = sorted items, dst = voxel starts) are read below
            it0 = src; it1 = dst;                            // remembered for the reduce; restored per unit below
        } else {
            // ---------------- general path: LSD passes over the rows in global memory ----------------
            for (int j = tid; j < 8 * 256; j += VP_THREADS) (&sh.hist[0][0])[j] = 0;
            __syncthreads();
            const int np = passes > 8 ? 8 : passes;          // rem <= 63
            for (uint32_t i = tid; i < R; i += VP_THREADS) {
                const uint64_t k = vx_key(g, mb, bufA[s + i]) & remmask;
                for (int p = 0; p < np; ++p) atomicAdd(&sh.hist[p][(k >> (8 * p)) & 255u], 1u);
            }
            __syncthreads();
            Row* a = bufA + s;
            Row* b = bufB + s;
            for (int p = 0; p < np; ++p) {
                uint32_t all;
                const uint32_t ex = vf_block_scan(tid < 256 ? sh.hist[p][tid] : 0u, sh.wsum, all);
                if (tid < 256) sh.base[tid] = ex;
                __syncthreads();
                for (uint32_t t0 = 0; t0 < R; t0 += VG_TILE) {
                    for (int j = tid; j < VP_WAVES * 256; j += VP_THREADS) (&sh.cnt[0][0])[j] = 0;
                    __syncthreads();
                    const uint32_t segb = t0 + w * (64 * VG_ROUNDS);
                    Row q[VG_ROUNDS];
                    uint32_t dig[VG_ROUNDS], rank[VG_ROUNDS];
#pragma unroll
                    for (int r = 0; r < VG_ROUNDS; ++r) {
                        const uint32_t i = segb + r * 64 + l;
                        q[r] = a[i < R ? i : 0];
                    }
#pragma unroll
                    for (int r = 0; r < VG_ROUNDS; ++r) {
                        const bool valid = segb + r * 64 + l < R;
                        dig[r] = (uint32_t)((vx_key(g, mb, q[r]) & remmask) >> (8 * p)) & 255u;
                        uint32_t npeer;
                        const uint32_t rk = vx_match<8>(dig[r], valid, npeer);
                        const uint32_t prior = sh.cnt[w][dig[r]];
                        __builtin_amdgcn_wave_barrier();
                        if (valid && rk == 0) sh.cnt[w][dig[r]] = prior + npeer;
                        __builtin_amdgcn_wave_barrier();
                        rank[r] = prior + rk;
                    }
                    __syncthreads();
                    if (tid < 256) {
                        uint32_t run = sh.base[tid];
#pragma unroll
                        for (int w2 = 0; w2 < VP_WAVES; ++w2) {
                            const uint32_t cc = sh.cnt[w2][tid];
                            sh.cnt[w2][tid] = run;
                            run += cc;
                        }
                        sh.base[tid] = run;
                    }
                    __syncthreads();
#pragma unroll
                    for (int r = 0; r < VG_ROUNDS; ++r)
                        if (segb + r * 64 + l < R) b[sh.cnt[w][dig[r]] + rank[r]] = q[r];
                    __syncthreads();
                }
                __threadfence_block();
                __syncthreads();                             // this pass' rows are visible to the whole workgroup
                Row* tswap = a; a = b; b = tswap;
            }
            fin = a;
            // ---- voxel heads: vstart_g[s + v] = first row of voxel v (rows are physically sorted now)
            uint32_t carry = 0;
            for (uint32_t t0 = 0; t0 < R; t0 += VP_TILE) {
                const uint32_t i0 = t0 + tid * VP_ROUNDS;
                uint64_t prevk = 0;
                if (i0 > 0 && i0 < R) prevk = vx_key(g, mb, fin[i0 - 1]);
                bool hd[VP_ROUNDS];
                uint32_t heads = 0;
#pragma unroll
                for (int j = 0; j < VP_ROUNDS; ++j) {
                    const uint32_t i = i0 + j;
                    uint64_t k = prevk;
                    if (i < R) k = vx_key(g, mb, fin[i]);
                    hd[j] = i < R && (i == 0 || k != prevk);
                    heads += hd[j];
                    prevk = k;
                }
                uint32_t tot;
                uint32_t vi = carry + vf_block_scan(heads, sh.wsum, tot);
#pragma unroll
                for (int j = 0; j < VP_ROUNDS; ++j)
                    if (hd[j]) vstart_g[s + vi++] = i0 + j;
                carry += tot;
            }
            nvox = carry;
            __threadfence_block();
            __syncthreads();
        }
        // ---- first output slot of this unit: look-back over the units in front (ticket order)
        if (w == 0) {
            const uint32_t e = gf_lookback(status, u, nvox);
            if (l == 0) sh.vbase = e;
        }
        __syncthreads();
        const int64_t vbase = sh.vbase;
        if (tid == 0) {
            if (dtop == 0 && out_chunk_offsets) out_chunk_offsets[c] = vbase;
            if (u == nunits - 1) {
                *out_m = vbase + nvox;
                if (out_chunk_offsets) out_chunk_offsets[g.nchunks] = vbase + nvox;
            }
        }
        // ---- reduce: one thread per voxel, rows added in file order (AccumulatedPoint::AddPoint)
        if (in_lds) {
            const P* srt = it0;
            const P* vst = it1;
            for (uint32_t v = tid; v < nvox; v += VP_THREADS) {
                const uint32_t a0 = (uint32_t)vst[v], a1 = v + 1 < nvox ? (uint32_t)vst[v + 1] : R;
                double ax = 0.0, ay = 0.0, az = 0.0;
                for (uint32_t i = a0; i < a1; ++i) {
                    const Row q = bufA[s + ((uint32_t)srt[i] & (uint32_t)(VF_CAP - 1))];
                    ax += q.x; ay += q.y; az += q.z;
                }
                const uint64_t key = (dtop << g.rem) | ((uint64_t)(srt[a0] >> VF_ROWBITS));
                vf_emit(g, key, ax, ay, az, a1 - a0, vbase + v, out_idx, out_mean, out_count);
            }
            it0 = reinterpret_cast<P*>(sh.item[0]);
            it1 = reinterpret_cast<P*>(sh.item[1]);
        } else {
            for (uint32_t v = tid; v < nvox; v += VP_THREADS) {
                const uint32_t a0 = vstart_g[s + v], a1 = v + 1 < nvox ? vstart_g[s + v + 1] : R;
                double ax = 0.0, ay = 0.0, az = 0.0;
                for (uint32_t i = a0; i < a1; ++i) {
                    const Row q = fin[i];
                    ax += q.x; ay += q.y; az += q.z;
                }
                const uint64_t key = vx_key(g, mb, fin[a0]);
                vf_emit(g, key, ax, ay, az, a1 - a0, vbase + v, out_idx, out_mean, out_count);
            }
        }
    }
}

struct VoxelWs {
    unsigned long long* mm;
    double*   minb;
    int*      gmeta;
    Row      *bufA, *bufB;
    uint32_t *tile_hist, *unit_start, *unit_count, *vstart, *ticket;
    uint64_t* status;
    size_t    clear_bytes;       // ticket .. end of status: zeroed before the finisher
};

static int64_t voxel_nchunks(int64_t n, int64_t& chunk_size) {
    if (chunk_size <= 0 || chunk_size > n) chunk_size = n > 0 ? n : 1;
    return n > 0 ? ceil_div(n, chunk_size) : 1;
}

static void voxel_plan(Arena& a, int64_t n, int64_t nchunks, int64_t chunk_size, VoxelWs& w) {
    const int64_t nn = n > 0 ? n : 1;
    const int64_t tiles = nchunks * ceil_div(chunk_size, VP_TILE);
    w.mm = a.take<unsigned long long>(nchunks * 6);
    w.minb = a.take<double>(nchunks * 3);
    w.gmeta = a.take<int>(4);
    w.bufA = a.take<Row>(nn);
    w.bufB = a.take<Row>(nn);
    w.vstart = a.take<uint32_t>(nn);
    w.tile_hist = a.take<uint32_t>(tiles * VP_MAXBINS);
    w.unit_start = a.take<uint32_t>(nchunks * VP_MAXBINS);
    w.unit_count = a.take<uint32_t>(nchunks * VP_MAXBINS);
    const size_t off0 = a.off;
    w.ticket = a.take<uint32_t>(4);
    w.status = a.take<uint64_t>(nchunks * VP_MAXBINS);
    w.clear_bytes = a.off - off0;
}

}  // namespace pch

using namespace pch;

extern "C" size_t pch_voxel_downsample_ws_bytes(int64_t n, int64_t chunk_size) {
    if (n < 0) return 0;
    const int64_t nchunks = voxel_nchunks(n, chunk_size);
    Arena a;
    VoxelWs w;
    voxel_plan(a, n, nchunks, chunk_size, w);
    return a.off;
}

extern "C" int pch_voxel_downsample_f64(const double* xyz, int64_t n, double voxel_size,
                                        int64_t chunk_size, int32_t* out_idx, double* out_mean,
                                        int32_t* out_count, int64_t* out_chunk_offsets,
                                        int64_t* out_m, void* ws, size_t ws_bytes, void* stream) {
    PCH_DEVICE_GUARD(xyz ? (const void*)xyz : (const void*)out_m);
    hipStream_t s = (hipStream_t)stream;
    PCH_REQUIRE(n >= 0 && n < (int64_t(1) << 31), "n out of range [0, 2^31)");
    PCH_REQUIRE(voxel_size > 0.0, "voxel_size must be > 0");
    PCH_REQUIRE(out_m != nullptr, "out_m is null");
    const int64_t nchunks = voxel_nchunks(n, chunk_size);
    if (n == 0) {
        PCH_HIP_TRY(hipMemsetAsync(out_m, 0, sizeof(int64_t), s));
        if (out_chunk_offsets) PCH_HIP_TRY(hipMemsetAsync(out_chunk_offsets, 0, 2 * sizeof(int64_t), s));
        return PCH_OK;
    }
    PCH_REQUIRE(xyz && out_idx && out_mean && out_count && ws, "null buffer");
    Arena a(ws, ws_bytes);
    VoxelWs w;
    voxel_plan(a, n, nchunks, chunk_size, w);
    if (a.overflow) { set_error("workspace too small: need %zu bytes", a.off); return PCH_ERR_WORKSPACE; }

    // per-chunk bounds: [chunk][min3,max3]; min slots start at all-ones, max slots at zero
    PCH_HIP_TRY(hipMemsetAsync(w.mm, 0, sizeof(unsigned long long) * nchunks * 6, s));
    PCH_HIP_TRY(hipMemset2DAsync(w.mm, 48, 0xFF, 24, nchunks, s));
    PCH_HIP_TRY(hipMemsetAsync(w.gmeta, 0, sizeof(int) * 4, s));
    const int64_t bpc = ceil_div(chunk_size, VX_THREADS * VX_MM_ROUNDS);
    PCH_LAUNCH("voxel_minmax", vx_minmax_k, dim3((unsigned)(bpc * nchunks)), dim3(VX_THREADS), 0, s,
               xyz, n, chunk_size, bpc, w.mm);
    PCH_LAUNCH("voxel_bounds", vx_bounds_k, dim3((unsigned)ceil_div(nchunks, 256)), dim3(256), 0, s,
               (const unsigned long long*)w.mm, nchunks, voxel_size, w.minb, w.gmeta);
    int gmeta[4];
    PCH_TRY(peek_enqueue(w.gmeta, sizeof(gmeta), s));
    PCH_HIP_TRY(hipMemsetAsync(w.ticket, 0, w.clear_bytes, s));      // overlaps the host's wait
    PCH_TRY(peek_wait(gmeta, sizeof(gmeta)));
    if (gmeta[3] != 0) {   // Open3D: "[VoxelDownSample] voxel_size is too small."
        set_error("voxel_size is too small (or non-finite coordinates)");
        return PCH_ERR_RANGE;
    }
    VoxelPlan g;
    g.n = n;
    g.chunk_size = chunk_size;
    g.nchunks = nchunks;
    g.tiles_per_chunk = ceil_div(chunk_size, VP_TILE);
    g.voxel = voxel_size;
    g.bx = bits_for((uint64_t)gmeta[0] + 1);
    g.by = bits_for((uint64_t)gmeta[1] + 1);
    g.bz = bits_for((uint64_t)gmeta[2] + 1);
    g.T = g.bx + g.by + g.bz;
    if (g.T > 63) {
        set_error("voxel grid needs %d key bits (> 63): reduce chunk extent or enlarge voxel", g.T);
        return PCH_ERR_RANGE;
    }
    g.d1 = g.T < VP_MAXBITS ? g.T : VP_MAXBITS;
    g.nb = 1 << g.d1;
    g.rem = g.T - g.d1;
    const unsigned gt = (unsigned)(nchunks * g.tiles_per_chunk);
    PCH_LAUNCH("voxel_tilehist", vx_tilehist_k, dim3(gt), dim3(VP_THREADS), 0, s, xyz, g, (const double*)w.minb,
               w.tile_hist);
    PCH_LAUNCH("voxel_binscan", vx_binscan_k, dim3((unsigned)nchunks), dim3(VP_MAXBINS), 0, s, g, w.tile_hist,
               w.unit_start, w.unit_count);
    PCH_LAUNCH("voxel_scatter", vx_scatter_k, dim3(gt), dim3(VP_THREADS), 0, s, xyz, g, (const double*)w.minb,
               (const uint32_t*)w.tile_hist, (const uint32_t*)w.unit_start, w.bufA);
    // persistent finisher: two workgroups per CU draw the units in order
    int dev = 0, cus = 256;
    if (hipGetDevice(&dev) == hipSuccess) (void)hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev);
    int64_t fg = (int64_t)cus * 2;
    if (fg > nchunks * g.nb) fg = nchunks * g.nb;
    const size_t shm = sizeof(VfShared);
    if (g.rem + VF_ROWBITS <= 32) {
        PCH_HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(vx_finish_k<uint32_t>),
                                        hipFuncAttributeMaxDynamicSharedMemorySize, (int)shm));
        PCH_LAUNCH("voxel_finish", vx_finish_k<uint32_t>, dim3((unsigned)fg), dim3(VP_THREADS), shm, s, g,
                   (const double*)w.minb, (const uint32_t*)w.unit_start, (const uint32_t*)w.unit_count, w.bufA, w.bufB,
                   w.vstart, w.status, w.ticket, out_idx, out_mean, out_count, out_chunk_offsets, out_m);
    } else {
        PCH_HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(vx_finish_k<unsigned long long>),
                                        hipFuncAttributeMaxDynamicSharedMemorySize, (int)shm));
        PCH_LAUNCH("voxel_finish", vx_finish_k<unsigned long long>, dim3((unsigned)fg), dim3(VP_THREADS), shm, s, g,
                   (const double*)w.minb, (const uint32_t*)w.unit_start, (const uint32_t*)w.unit_count, w.bufA, w.bufB,
                   w.vstart, w.status, w.ticket, out_idx, out_mean, out_count, out_chunk_offsets, out_m);
    }
    return PCH_OK;
}

// ---- LAS integer <-> scaled float64 (laspy ScaledArrayView semantics) ----------------
namespace pch {
struct D3 { double v[3]; };

__global__ void las_scale_k(const int32_t* __restrict__ X, int64_t count, D3 sc, D3 of,
                            double* __restrict__ out) {
    const int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= count) return;
    const int a = (int)(e % 3);
    out[e] = (double)X[e] * sc.v[a] + of.v[a];          // separate mul, add (-ffp-contract=off)
}
// one thread per coordinate: three 4-byte loads at the head of each record (records need not be
// 4-byte aligned: formats 2, 7, 8 ... have odd lengths), assembled from bytes when misaligned
__global__ void las_records_k(const uint8_t* __restrict__ rec, int64_t count, int record_len,
                              int32_t* __restrict__ out) {
    const int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= count) return;
    const int64_t i = e / 3;
    const int a = (int)(e - 3 * i);
    const uint8_t* p = rec + i * record_len + 4 * a;
    uint32_t v;
    if ((reinterpret_cast<uintptr_t>(p) & 3) == 0) v = *reinterpret_cast<const uint32_t*>(p);
    else v = (uint32_t)p[0] | ((uint32_t)p[1] << 8) | ((uint32_t)p[2] << 16) | ((uint32_t)p[3] << 24);
    out[e] = (int32_t)v;
}
__global__ void las_unscale_k(const double* __restrict__ v, int64_t count, D3 sc, D3 of,
                              int32_t* __restrict__ out) {
    const int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= count) return;
    const int a = (int)(e % 3);
    out[e] = (int32_t)rint((v[e] - of.v[a]) / sc.v[a]); // np.round = half-to-even = rint
}
__global__ void cast_f64_f32_k(const double* __restrict__ in, int64_t count, float* __restrict__ out) {
    const int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (e < count) out[e] = (float)in[e];
}
}  // namespace pch

extern "C" int pch_las_records_xyz_i32(const uint8_t* records, int64_t n, int32_t record_len,
                                       int32_t* out_XYZ, void* stream) {
    PCH_DEVICE_GUARD(records);
    PCH_REQUIRE(n >= 0 && record_len >= 12, "bad argument");
    if (n == 0) return PCH_OK;
    PCH_REQUIRE(records && out_XYZ, "null buffer");
    const int64_t count = 3 * n;
    PCH_LAUNCH("las_records", las_records_k, dim3((unsigned)ceil_div(count, 256)), dim3(256), 0,
               (hipStream_t)stream, records, count, (int)record_len, out_XYZ);
    return PCH_OK;
}

extern "C" int pch_las_scale_i32_f64(const int32_t* XYZ, int64_t n, const double* scale3_host,
                                     const double* offset3_host, double* out_xyz, void* stream) {
    PCH_DEVICE_GUARD(XYZ);
    PCH_REQUIRE(n >= 0 && scale3_host && offset3_host, "bad argument");
    if (n == 0) return PCH_OK;
    PCH_REQUIRE(XYZ && out_xyz, "null buffer");
    D3 sc, of;
    for (int a = 0; a < 3; ++a) { sc.v[a] = scale3_host[a]; of.v[a] = offset3_host[a]; }
    const int64_t count = 3 * n;
    PCH_LAUNCH("las_scale", las_scale_k, dim3((unsigned)ceil_div(count, 256)), dim3(256), 0,
               (hipStream_t)stream, XYZ, count, sc, of, out_xyz);
    return PCH_OK;
}

extern "C" int pch_las_unscale_f64_i32(const double* xyz, int64_t n, const double* scale3_host,
                                       const double* offset3_host, int32_t* out_XYZ, void* stream) {
    PCH_DEVICE_GUARD(xyz);
    PCH_REQUIRE(n >= 0 && scale3_host && offset3_host, "bad argument");
    if (n == 0) return PCH_OK;
    PCH_REQUIRE(xyz && out_XYZ, "null buffer");
    D3 sc, of;
    for (int a = 0; a < 3; ++a) { sc.v[a] = scale3_host[a]; of.v[a] = offset3_host[a]; }
    const int64_t count = 3 * n;
    PCH_LAUNCH("las_unscale", las_unscale_k, dim3((unsigned)ceil_div(count, 256)), dim3(256), 0,
               (hipStream_t)stream, xyz, count, sc, of, out_XYZ);
    return PCH_OK;
}

extern "C" int pch_cast_f64_f32(const double* in, int64_t count, float* out, void* stream) {
    PCH_DEVICE_GUARD(in);
    PCH_REQUIRE(count >= 0, "bad count");
    if (count == 0) return PCH_OK;
    PCH_REQUIRE(in && out, "null buffer");
    PCH_LAUNCH("cast_f64_f32", cast_f64_f32_k, dim3((unsigned)ceil_div(count, 256)), dim3(256), 0,
               (hipStream_t)stream, in, count, out);
    return PCH_OK;
}
