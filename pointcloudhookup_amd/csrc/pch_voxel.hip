// Stage A: per-chunk voxel-grid downsample (Open3D VoxelDownSample semantics): the rows of every
// chunk are sorted by voxel index (one MSD partition in HBM, the rest inside LDS) and reduced in
// file order, one float64 running sum per voxel.
// Reference call site: ui/import_PC.py:8-13 inside the chunk loop ui/import_PC.py:45-58.
#include "pch_prims.h"
#ifdef PCH_VX_STAMPS
#define PCH_LB_COUNT
namespace pch { __device__ unsigned long long g_lb_polls = 0, g_lb_windows = 0; }
#endif
#include "pch_lookback.h"

namespace pch {

constexpr int VX_THREADS = 256;
constexpr int VX_MM_ROUNDS = 8;

// ---- per-chunk float64 min / max (ordered-uint64 atomics) ---------------------------
// A thread takes two rows per round as three 16-byte loads (a chunk starts on a 16-byte boundary
// when chunk_size is even; odd chunk sizes and the last row use the scalar path).
__global__ __launch_bounds__(VX_THREADS) void vx_minmax_k(const double* __restrict__ xyz,
                                                          int64_t n, int64_t chunk_size,
                                                          int64_t blocks_per_chunk,
                                                          unsigned long long* __restrict__ mm) {
    __shared__ unsigned long long sm[VX_THREADS / 64][6];
    const int64_t chunk = blockIdx.x / blocks_per_chunk;
    const int64_t tile  = blockIdx.x % blocks_per_chunk;
    const int64_t cbeg = chunk * chunk_size;
    const int64_t cend = (cbeg + chunk_size < n) ? cbeg + chunk_size : n;
    double lo[3] = {INFINITY, INFINITY, INFINITY}, hi[3] = {-INFINITY, -INFINITY, -INFINITY};
    bool nan = false;
    auto take = [&](int a, double v) {
        lo[a] = v < lo[a] ? v : lo[a];
        hi[a] = v > hi[a] ? v : hi[a];
        nan |= v != v;
    };
    const bool vec = ((cbeg & 1) == 0) && ((reinterpret_cast<uintptr_t>(xyz) & 15u) == 0);
    const int64_t t0 = cbeg + tile * (VX_THREADS * VX_MM_ROUNDS);
    if (vec) {
        double2 q[VX_MM_ROUNDS / 2][3];
#pragma unroll
        for (int r = 0; r < VX_MM_ROUNDS / 2; ++r) {
            const int64_t i = t0 + 2 * (r * VX_THREADS + threadIdx.x);        // rows i, i+1
            const bool both = i + 1 < cend;
            const double2* p = reinterpret_cast<const double2*>(xyz + 3 * (both ? i : cbeg));
            q[r][0] = p[0]; q[r][1] = p[1]; q[r][2] = p[2];
        }
#pragma unroll
        for (int r = 0; r < VX_MM_ROUNDS / 2; ++r) {
            const int64_t i = t0 + 2 * (r * VX_THREADS + threadIdx.x);
            if (i + 1 < cend) {
                take(0, q[r][0].x); take(1, q[r][0].y); take(2, q[r][1].x);
                take(0, q[r][1].y); take(1, q[r][2].x); take(2, q[r][2].y);
            } else if (i < cend) {
                take(0, xyz[3 * i + 0]); take(1, xyz[3 * i + 1]); take(2, xyz[3 * i + 2]);
            }
        }
    } else {
        for (int r = 0; r < VX_MM_ROUNDS; ++r) {
            const int64_t i = t0 + r * VX_THREADS + threadIdx.x;
            if (i < cend) { take(0, xyz[3 * i + 0]); take(1, xyz[3 * i + 1]); take(2, xyz[3 * i + 2]); }
        }
    }
    unsigned long long klo[3], khi[3];
#pragma unroll
    for (int a = 0; a < 3; ++a) {
        // a NaN anywhere must reach the bounds kernel (it rejects the call): it is carried in the max slot
        klo[a] = wave_reduce_min(f64_ordered(lo[a]));
        khi[a] = wave_reduce_max(nan ? f64_ordered(__longlong_as_double(0x7FF8000000000000ll)) : f64_ordered(hi[a]));
    }
    if (lane_id() == 0) {
#pragma unroll
        for (int a = 0; a < 3; ++a) { sm[wave_id()][a] = klo[a]; sm[wave_id()][3 + a] = khi[a]; }
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
        if (!(fabs(lo) < INFINITY) || !(fabs(hi) < INFINITY)) atomicMax(&gmeta[3], 1);   // NaN / inf coordinates
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
constexpr int VF_THREADS = 512;                      // finisher workgroup (two per CU)
constexpr int VF_WAVES   = VF_THREADS / 64;
constexpr int VF_CAP     = VF_THREADS * VP_ROUNDS;   // 8192 rows: what one LDS sort takes
constexpr int VF_DIGBITS = 8;                        // a batch spans at most 256 level-1 digits
constexpr int VG_ROUNDS  = 4;                        // general (global-memory) path: rows per thread and tile
constexpr int VG_TILE    = VF_THREADS * VG_ROUNDS;

struct Row { double x, y, z; };

struct VoxelPlan {
    int64_t n, chunk_size, nchunks, tiles_per_chunk;
    double  voxel, rvoxel;   // voxel size and fl(1 / voxel)
    int     bx, by, bz;      // bits per axis
    int     T;               // bx + by + bz
    int     d1;              // level-1 digit bits
    int     nb;              // 1 << d1
    int     rem;             // T - d1: bits the finisher sorts on
};

// floor((p - voxel_min_bound) / voxel_size) exactly as IEEE float64 subtraction + DIVISION + floor give it
// (Open3D: ref_coord = (p - min_bound) / voxel_size; floor), without paying for a division per coordinate:
// q = d * fl(1/v) differs from fl(d / v) by less than q * 2^-51 (three roundings of 2^-53 each), so whenever q
// is further than q * 2^-49 from the integers on both sides, both have the same floor; otherwise (a coordinate
// on a voxel face to within rounding - common for quantised LAS data) the real division decides.
__device__ __forceinline__ uint64_t vx_index(double p, double mb, double v, double rv) {
    const double d = p - mb;                            // >= v/2: min_bound is half a voxel below the smallest point
    const double q = d * rv;
    const double fq = floor(q);
    const double fr = q - fq, tol = q * 0x1p-49;
    // indices are < 2^31 (vx_bounds_k rejects larger grids): one v_cvt_i32_f64 instead of the 64-bit conversion
    if (fr > tol && (1.0 - fr) > tol) return (uint64_t)(uint32_t)(int32_t)fq;
    return (uint64_t)(uint32_t)(int32_t)floor(d / v);
}
__device__ __forceinline__ uint64_t vx_key(const VoxelPlan& g, const double* __restrict__ mb, const Row& q) {
    const uint64_t ix = vx_index(q.x, mb[0], g.voxel, g.rvoxel);
    const uint64_t iy = vx_index(q.y, mb[1], g.voxel, g.rvoxel);
    const uint64_t iz = vx_index(q.z, mb[2], g.voxel, g.rvoxel);
    return (((ix << g.by) | iy) << g.bz) | iz;
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

// ---- level 1b: one workgroup per chunk: tile_hist -> exclusive offsets inside every bin, bin starts;
// consecutive units are then grouped into BATCHES of at most VF_CAP rows (one LDS sort each); a unit
// that is larger on its own is a batch of one (sorted in global memory by its workgroup)
struct VoxelBatch {
    uint32_t row0;        // first row (absolute, in the partitioned buffer) of the units this batch reads
    uint32_t rows;        // rows this batch sorts (0: nothing to do)
    uint32_t span;        // == rows; for the reserved slots of an oversize unit (before vx_split_k): the unit's rows
    uint16_t digit0;      // first level-1 digit of the batch
    uint16_t ndigits;     // level-1 digits covered (>= 1)
    uint32_t pad;
    uint32_t from_b;      // 1: the batch's rows lie in the second row buffer (a part of an oversize unit, vx_split_k)
};
struct VoxelOversize { uint32_t slot, nsub; };             // a unit above VF_CAP rows and its reserved batch slots

__global__ __launch_bounds__(VP_MAXBINS) void vx_binscan_k(VoxelPlan g, uint32_t* __restrict__ tile_hist,
                                                           uint32_t* __restrict__ unit_start,
                                                           VoxelBatch* __restrict__ batches,
                                                           uint32_t* __restrict__ nbatch,
                                                           VoxelOversize* __restrict__ over,
                                                           uint32_t* __restrict__ nover) {
    __shared__ uint32_t wsum[VP_MAXBINS / 64];
    __shared__ __attribute__((aligned(16))) uint32_t ucount[VP_MAXBINS];
    __shared__ uint32_t ustart[VP_MAXBINS];
    const int64_t c = blockIdx.x;
    const int b = threadIdx.x;
    uint32_t run = 0;
    if (b < g.nb) {
        constexpr int U = 32;                              // loads of 32 tiles in flight (a chunk of 500 000 rows has 123:
                                                           // four round trips instead of sixteen - the kernel is one
                                                           // workgroup per chunk and nothing but these round trips)
        for (int64_t t0 = 0; t0 < g.tiles_per_chunk; t0 += U) {
            uint32_t h[U];
#pragma unroll
            for (int k = 0; k < U; ++k)
                h[k] = t0 + k < g.tiles_per_chunk ? tile_hist[(c * g.tiles_per_chunk + t0 + k) * g.nb + b] : 0u;
#pragma unroll
            for (int k = 0; k < U; ++k) {
                if (t0 + k < g.tiles_per_chunk) tile_hist[(c * g.tiles_per_chunk + t0 + k) * g.nb + b] = run;
                run += h[k];
            }
        }
    }
    const uint32_t incl = wave_scan_incl(run);
    if (lane_id() == 63) wsum[wave_id()] = incl;
    __syncthreads();
    uint32_t base = incl - run;
    for (int w = 0; w < wave_id(); ++w) base += wsum[w];
    if (b < g.nb) {
        const uint32_t st = (uint32_t)(c * g.chunk_size) + base;
        unit_start[c * g.nb + b] = st;
        ucount[b] = run;
        ustart[b] = st;
    } else {
        ucount[b] = 0;                                     // (the grouping below reads the counts four at a time)
    }
    const uint32_t nonempty = (uint32_t)__syncthreads_count(b < g.nb && run != 0);
    if (b == 0) {                                          // greedy grouping, in digit order
        VoxelBatch* out = batches + c * VP_MAXBINS;
        uint32_t nbt = 0, rows = 0, first = 0;
        const uint32_t maxdig = (g.rem >= 32) ? 1u : (g.rem + VF_DIGBITS > 32 ? (1u << (32 - g.rem)) : (1u << VF_DIGBITS));
        auto emit = [&](uint32_t r0, uint32_t nrows, uint32_t d0, uint32_t nd) {
            VoxelBatch v;
            v.row0 = r0; v.rows = nrows; v.span = nrows; v.digit0 = (uint16_t)d0; v.ndigits = (uint16_t)nd;
            v.pad = 0; v.from_b = 0;
            out[nbt++] = v;
        };
        // a batch covers the digits [first, last] of its first and last NON-EMPTY unit (empty units in between
        // cost nothing but would widen the sort key)
        uint32_t last = 0, seen = 0;
        // (one thread, in digit order: the counts come four per LDS read - one dependent read per digit was most of
        // this kernel's 70 us, which every call waits for whatever its size)
        for (uint32_t d4 = 0; d4 < (uint32_t)g.nb; d4 += 4) {
          const uint4 c4 = *reinterpret_cast<const uint4*>(&ucount[d4]);
          const uint32_t cq[4] = {c4.x, c4.y, c4.z, c4.w};
          if ((c4.x | c4.y | c4.z | c4.w) == 0u) continue;
#pragma unroll
          for (uint32_t dd = 0; dd < 4; ++dd) {
            const uint32_t d = d4 + dd;
            const uint32_t cnt = cq[dd];
            if (cnt == 0) continue;
            ++seen;
            if (rows && (rows + cnt > (uint32_t)VF_CAP || d - first >= maxdig)) {   // close the open batch
                emit(ustart[first], rows, first, last - first + 1);
                rows = 0;
            }
            if (rows == 0) first = d;
            rows += cnt;
            last = d;
            if (rows > (uint32_t)VF_CAP) {                 // a single unit above VF_CAP (rows == cnt here)
                // split by its next digit into <= 2*ceil(rows/CAP)+1 parts (filled in by vx_split_k), unless
                // batch slots or the 16-bit position field run out: then one batch, sorted in global memory
                const uint32_t nsub = 2 * ((rows + VF_CAP - 1) / VF_CAP) + 1;
                const uint32_t left = nonempty - seen;                    // units still to come need <= 1 slot each
                if (g.rem > 0 && g.rem <= 32 && nbt + nsub + left <= (uint32_t)VP_MAXBINS) {
                    const uint32_t o = atomicAdd(nover, 1u);
                    over[o].slot = (uint32_t)(c * VP_MAXBINS) + nbt;
                    over[o].nsub = nsub;
                    for (uint32_t k = 0; k < nsub; ++k) {
                        emit(ustart[d], 0, d, 1);
                        out[nbt - 1].span = rows;
                    }
                } else {
                    emit(ustart[d], rows, d, 1);
                }
                rows = 0;
            }
          }
        }
        if (rows) emit(ustart[first], rows, first, last - first + 1);
        nbatch[c] = nbt;
    }
}

// one workgroup per oversize unit (a tower core: thousands of rows in one level-1 cell): a SECOND partition, by the
// unit's next 8-bit digit, into the second row buffer (stable: histogram, then ranked scatter tile by tile); the parts of
// <= VF_CAP rows that come out of it are contiguous there and are finished like any other batch.  (Until round 4 every
// part swept the whole unit twice and picked its rows: 12 sweeps of a 20 000-row unit that was cut into six.)
constexpr int VS_THREADS = 1024, VS_WAVES = VS_THREADS / 64, VS_ROUNDS = 4, VS_TILE = VS_THREADS * VS_ROUNDS;
__global__ __launch_bounds__(VS_THREADS, 8) void vx_split_k(VoxelPlan g, const double* __restrict__ minb,
                                                       const Row* __restrict__ rows, Row* __restrict__ rows_b,
                                                       VoxelBatch* __restrict__ batches,
                                                       const VoxelOversize* __restrict__ over,
                                                       const uint32_t* __restrict__ nover) {
    __shared__ uint32_t hist[256], base[256];
    __shared__ uint32_t cnt[VS_WAVES][256];
    __shared__ uint32_t okflag;
    const uint32_t total = *nover;
    const int tid = threadIdx.x, w = wave_id(), l = lane_id();
    for (uint32_t o = blockIdx.x; o < total; o += gridDim.x) {
        const VoxelOversize ov = over[o];
        VoxelBatch* bt = batches + ov.slot;
        const uint32_t s = bt[0].row0, R = bt[0].span;
        const int64_t c = ov.slot / VP_MAXBINS;
        const double mb[3] = {minb[3 * c + 0], minb[3 * c + 1], minb[3 * c + 2]};
        const int sh2 = g.rem > 8 ? g.rem - 8 : 0;
        const uint64_t remmask = (1ull << g.rem) - 1;
        if (tid < 256) hist[tid] = 0;
        __syncthreads();
        for (uint32_t i0 = 0; i0 < R; i0 += VS_TILE) {
            Row q[VS_ROUNDS];
#pragma unroll
            for (int k = 0; k < VS_ROUNDS; ++k) {
                const uint32_t i = i0 + k * VS_THREADS + tid;
                q[k] = rows[s + (i < R ? i : 0)];
            }
#pragma unroll
            for (int k = 0; k < VS_ROUNDS; ++k)
                if (i0 + k * VS_THREADS + tid < R)
                    atomicAdd(&hist[(uint32_t)((vx_key(g, mb, q[k]) & remmask) >> sh2) & 255u], 1u);
        }
        __syncthreads();
        if (tid == 0) {
            uint32_t k = 0, acc = 0, first = 0, run = 0;
            bool ok = true;
            for (uint32_t d = 0; d < 256 && ok; ++d) {
                const uint32_t h = hist[d];
                if (h > (uint32_t)VF_CAP) { ok = false; break; }
                if (acc + h > (uint32_t)VF_CAP) {
                    bt[k].row0 = s + first; bt[k].rows = acc; bt[k].span = acc; bt[k].from_b = 1u; ++k;
                    first = run;
                    acc = 0;
                }
                acc += h;
                run += h;
            }
            if (ok) {
                if (acc) { bt[k].row0 = s + first; bt[k].rows = acc; bt[k].span = acc; bt[k].from_b = 1u; ++k; }
                for (uint32_t j = k; j < ov.nsub; ++j) { bt[j].rows = 0; bt[j].span = 0; }
            } else {                                       // one voxel column alone exceeds the LDS capacity:
                for (uint32_t j = 0; j < ov.nsub; ++j) { bt[j].rows = 0; bt[j].from_b = 0; }
                bt[0].rows = R;                            // the whole unit as one batch, sorted in global memory
            }
            okflag = ok ? 1u : 0u;
        }
        __syncthreads();
        if (okflag) {                                      // workgroup-uniform
            // exclusive scan of the digit histogram -> where every digit's rows start in the second buffer
            uint32_t hv = tid < 256 ? hist[tid] : 0u;
            const uint32_t incl = wave_scan_incl(hv);
            if (tid < 256 && l == 63) cnt[0][w] = incl;    // (cnt[0][0..3]: the four waves' totals, consumed below)
            __syncthreads();
            if (tid < 256) {
                uint32_t b = incl - hv;
                for (int w2 = 0; w2 < w; ++w2) b += cnt[0][w2];
                base[tid] = s + b;
            }
            __syncthreads();
            for (uint32_t t0 = 0; t0 < R; t0 += VS_TILE) {
                for (int j = tid; j < VS_WAVES * 256; j += VS_THREADS) (&cnt[0][0])[j] = 0;
                __syncthreads();
                const uint32_t segb = t0 + w * (64 * VS_ROUNDS);     // wave w owns 256 consecutive rows of the tile
                Row q[VS_ROUNDS];
                uint32_t dig[VS_ROUNDS], rank[VS_ROUNDS];
#pragma unroll
                for (int r = 0; r < VS_ROUNDS; ++r) {
                    const uint32_t i = segb + r * 64 + l;
                    q[r] = rows[s + (i < R ? i : 0)];
                }
#pragma unroll
                for (int r = 0; r < VS_ROUNDS; ++r) {
                    const bool valid = segb + r * 64 + l < R;
                    dig[r] = (uint32_t)((vx_key(g, mb, q[r]) & remmask) >> sh2) & 255u;
                    uint32_t npeer;
                    const uint32_t rk = wave_match<8>(dig[r], valid, npeer);
                    const uint32_t prior = cnt[w][dig[r]];
                    __builtin_amdgcn_wave_barrier();
                    if (valid && rk == 0) cnt[w][dig[r]] = prior + npeer;
                    __builtin_amdgcn_wave_barrier();
                    rank[r] = prior + rk;
                }
                __syncthreads();
                if (tid < 256) {                           // per digit: waves in order
                    uint32_t run = base[tid];
#pragma unroll
                    for (int w2 = 0; w2 < VS_WAVES; ++w2) {
                        const uint32_t cc = cnt[w2][tid];
                        cnt[w2][tid] = run;
                        run += cc;
                    }
                    base[tid] = run;
                }
                __syncthreads();
#pragma unroll
                for (int r = 0; r < VS_ROUNDS; ++r)
                    if (segb + r * 64 + l < R) rows_b[cnt[w][dig[r]] + rank[r]] = q[r];
                __syncthreads();
            }
        }
        __syncthreads();
    }
}

// exclusive scan of the per-chunk batch counts (one workgroup); prefix[nchunks] = total
__global__ __launch_bounds__(1024) void vx_batchscan_k(const uint32_t* __restrict__ nbatch, int64_t nchunks,
                                                       uint32_t* __restrict__ prefix) {
    __shared__ uint32_t wsum[16];
    __shared__ uint32_t carry_s;
    if (threadIdx.x == 0) carry_s = 0;
    __syncthreads();
    for (int64_t base = 0; base < nchunks; base += 1024) {
        const int64_t i = base + threadIdx.x;
        const uint32_t v = i < nchunks ? nbatch[i] : 0u;
        const uint32_t incl = wave_scan_incl(v);
        if (lane_id() == 63) wsum[wave_id()] = incl;
        __syncthreads();
        uint32_t before = carry_s, tot = 0;
        for (int w = 0; w < 16; ++w) { if (w < wave_id()) before += wsum[w]; tot += wsum[w]; }
        if (i < nchunks) prefix[i] = before + incl - v;
        __syncthreads();
        if (threadIdx.x == 0) carry_s += tot;
        __syncthreads();
    }
    if (threadIdx.x == 0) prefix[nchunks] = carry_s;
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
        const uint32_t rk = wave_match<VP_MAXBITS>(dig[r], valid, np);
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

// The same scatter with the tile staged through LDS: the rows of a tile are first put in digit order INSIDE the tile
// (LDS), then copied out - a (tile, digit) run of ~8 rows (192 bytes) leaves in one store instruction of adjacent lanes
// instead of reaching the L2 as eight 24-byte stores of eight waves at eight different times.
struct VsShared {
    Row      rows[VP_TILE];                       // 96 KB
    uint16_t dig[VP_TILE];                        // 8 KB: digit of the row at that LDS position
    uint32_t cnt[VP_WAVES][VP_MAXBINS];           // 16 KB
    uint32_t gbase[VP_MAXBINS];                   // where the tile's run of a digit starts in HBM
    uint32_t lstart[VP_MAXBINS];                  // ... and inside the tile
    uint32_t wsum[VP_WAVES];
};
__global__ __launch_bounds__(VP_THREADS) void vx_scatter_lds_k(const double* __restrict__ xyz, VoxelPlan g,
                                                               const double* __restrict__ minb,
                                                               const uint32_t* __restrict__ tile_hist,
                                                               const uint32_t* __restrict__ unit_start,
                                                               Row* __restrict__ out) {
    extern __shared__ __attribute__((aligned(16))) unsigned char vs_raw[];
    VsShared& sh = *reinterpret_cast<VsShared*>(vs_raw);
    const int64_t c = blockIdx.x / g.tiles_per_chunk, t = blockIdx.x % g.tiles_per_chunk;
    const int64_t cbeg = c * g.chunk_size, cend = (cbeg + g.chunk_size < g.n) ? cbeg + g.chunk_size : g.n;
    const int64_t t0 = cbeg + t * VP_TILE;
    if (t0 >= cend) return;
    const uint32_t tn = (uint32_t)((cend - t0) < VP_TILE ? (cend - t0) : VP_TILE);
    const int tid = threadIdx.x, w = wave_id(), l = lane_id();
    for (int j = tid; j < VP_WAVES * g.nb; j += VP_THREADS) sh.cnt[j / g.nb][j % g.nb] = 0;
    for (int j = tid; j < g.nb; j += VP_THREADS)
        sh.gbase[j] = unit_start[c * g.nb + j] + tile_hist[(int64_t)blockIdx.x * g.nb + j];
    __syncthreads();
    const double mb[3] = {minb[3 * c + 0], minb[3 * c + 1], minb[3 * c + 2]};
    const Row* __restrict__ rows = reinterpret_cast<const Row*>(xyz);
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
        const uint32_t rk = wave_match<VP_MAXBITS>(dig[r], valid, np);
        const uint32_t prior = sh.cnt[w][dig[r]];
        __builtin_amdgcn_wave_barrier();
        if (valid && rk == 0) sh.cnt[w][dig[r]] = prior + np;
        __builtin_amdgcn_wave_barrier();
        rank[r] = prior + rk;
    }
    __syncthreads();
    // per digit (one thread each, g.nb <= VP_THREADS): the waves in order -> offsets inside the digit's run; the runs
    // in digit order -> where each starts inside the tile
    uint32_t tot = 0;
    if (tid < g.nb) {
#pragma unroll
        for (int w2 = 0; w2 < VP_WAVES; ++w2) {
            const uint32_t cc = sh.cnt[w2][tid];
            sh.cnt[w2][tid] = tot;
            tot += cc;
        }
    }
    const uint32_t incl = wave_scan_incl(tot);
    if (l == 63) sh.wsum[w] = incl;
    __syncthreads();
    if (tid < g.nb) {
        uint32_t b = incl - tot;
        for (int w2 = 0; w2 < w; ++w2) b += sh.wsum[w2];
        sh.lstart[tid] = b;
    }
    __syncthreads();
#pragma unroll
    for (int r = 0; r < VP_ROUNDS; ++r)
        if (seg + r * 64 + l < cend) {
            const uint32_t at = sh.lstart[dig[r]] + sh.cnt[w][dig[r]] + rank[r];
            sh.rows[at] = q[r];
            sh.dig[at] = (uint16_t)dig[r];
        }
    __syncthreads();
    for (uint32_t j = tid; j < tn; j += VP_THREADS) {
        const uint32_t d = sh.dig[j];
        out[sh.gbase[d] + (j - sh.lstart[d])] = sh.rows[j];
    }
}

// ---- finish ------------------------------------------------------------------------------------
// One batch per workgroup iteration.  LDS sort: what is sorted is a 32-bit key
//   ((level-1 digit - first digit of the batch) << rem) | (low `rem` bits of the voxel key)
// together with the row's position in the batch (uint16); the rows themselves stay in L2.
constexpr uint32_t VQ_SLOTS = 2 * VF_CAP;            // hash-set slots of a batch (load factor <= 1/2)
constexpr uint32_t VQ_MAXRUN = 32;                   // rows per voxel the grouping path takes (else: the stable sort)
struct VfShared {
    union {
        struct {
            uint32_t key[2][VF_CAP];             // 32 KB (key[0..1] together: the hash set of both LDS paths)
            uint16_t perm[2][VF_CAP];            // 16 KB
        };
        struct {                                 // grouping path (vf_group): hash set, per-slot row counts (one BYTE
            uint32_t hset[VQ_SLOTS];             // per slot, four to a word), first position of every slot's rows,
            uint32_t hcnt[VQ_SLOTS / 4];         // the rows' positions grouped by slot, and one bit per position that
            uint16_t sstart[VQ_SLOTS];           // is the first row of its voxel
            uint16_t pos[VF_CAP];
            unsigned long long headbits[VF_CAP / 64];
            uint32_t headpre[VF_CAP / 64];
        } q;
        uint32_t hist[8][256];                   // general path: digit histograms of every LSD pass
    };
    uint32_t cnt[VF_WAVES][256];                 // per-wave digit counters / offsets (16 KB)
    uint32_t base[256];
    uint32_t wsum[VF_WAVES];
    uint32_t ticket, vbase, lb_failed, overflow;
};

// block-wide exclusive scan of one value per thread (VF_THREADS threads); total returned to all
__device__ __forceinline__ uint32_t vf_block_scan(uint32_t v, uint32_t* wsum, uint32_t& total) {
    const uint32_t incl = wave_scan_incl(v);
    __syncthreads();
    if (lane_id() == 63) wsum[wave_id()] = incl;
    __syncthreads();
    uint32_t before = 0, tot = 0;
#pragma unroll
    for (int w = 0; w < VF_WAVES; ++w) {
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

#ifdef PCH_VX_STAMPS        // tuning builds only: where one workgroup's time goes, phase by phase
#define VX_STAMP(k) do { __syncthreads(); if (tid == 0) { const unsigned long long _t = wall_clock64(); acc[k] += _t - t_last; t_last = _t; } } while (0)
#else
#define VX_STAMP(k)
#endif

__global__ __launch_bounds__(VF_THREADS, (2 * VF_THREADS) / 256) void vx_finish_k(
    VoxelPlan g, const double* __restrict__ minb, const VoxelBatch* __restrict__ batches,
    const uint32_t* __restrict__ batch_prefix, Row* __restrict__ bufA, Row* __restrict__ bufB,
    uint32_t* __restrict__ vstart_g, uint64_t* __restrict__ status, uint32_t* __restrict__ ticket,
    int32_t* __restrict__ out_idx, double* __restrict__ out_mean, int32_t* __restrict__ out_count,
    int64_t* __restrict__ out_chunk_offsets, int64_t* __restrict__ out_m, unsigned long long* __restrict__ stamps) {
    extern __shared__ __attribute__((aligned(16))) unsigned char vf_raw[];
    VfShared& sh = *reinterpret_cast<VfShared*>(vf_raw);
    const int tid = threadIdx.x, w = wave_id(), l = lane_id();
    const uint32_t nbatches = batch_prefix[g.nchunks];
#ifdef PCH_VX_STAMPS
    unsigned long long acc[12] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0}, t_last = wall_clock64();
#endif
    const uint64_t remmask = (1ull << g.rem) - 1;           // rem <= 54
    for (;;) {
        __syncthreads();                                   // previous batch's LDS reads are done
        if (tid == 0) sh.ticket = atomicAdd(ticket, 1u);   // batches are taken in order of arrival (look-back below)
        __syncthreads();
        const uint32_t t = sh.ticket;
        if (t >= nbatches) {
#ifdef PCH_VX_STAMPS
            if (tid == 0) for (int k = 0; k < 12; ++k) atomicAdd(&stamps[k], acc[k]);
#endif
            return;
        }
        VX_STAMP(0);
        // chunk of batch t: last c with batch_prefix[c] <= t
        int64_t lo = 0, hi = g.nchunks - 1;
        while (lo < hi) { const int64_t mid = (lo + hi + 1) >> 1; if (batch_prefix[mid] <= t) lo = mid; else hi = mid - 1; }
        const int64_t c = lo;
        const VoxelBatch bt = batches[c * VP_MAXBINS + (t - batch_prefix[c])];
        const bool first_of_chunk = t == batch_prefix[c];
        const uint32_t s = bt.row0, R = bt.rows;
        const Row* __restrict__ src = bt.from_b ? bufB : bufA;      // where this batch's rows are
        const uint64_t d0 = bt.digit0;
        const double mb[3] = {minb[3 * c + 0], minb[3 * c + 1], minb[3 * c + 2]};
        int dbits = 0;
        while ((1u << dbits) < (uint32_t)bt.ndigits) ++dbits;
        const int sortbits = g.rem + dbits;
        uint32_t nvox = 0;
        const Row* fin = bufA + s;                         // where the batch's rows are when they are reduced
        bool in_lds = false, announced = false;
        const uint32_t* srt = nullptr;                     // LDS path: sorted keys / their rows / voxel starts
        const uint16_t* sperm = nullptr;
        const uint32_t* vst = nullptr;
        // ---------------- grouping path: rows grouped by HASH SLOT, not sorted by key ----------------
        // The exact hash set that counts the batch's voxels already says which rows belong together; what Open3D's
        // accumulation needs on top is the FILE ORDER inside a voxel, not an order between voxels (its own output
        // order is unordered_map iteration order).  So: every row takes a ticket on its slot's counter (arrival
        // order, arbitrary), the slots' counts are scanned, the rows' positions are scattered to their slot's run, and
        // the thread of the row that arrived FIRST sorts that run (1-3 positions as a rule) and adds the rows in
        // ascending position = file order.  Voxels leave the batch in the order of their first rows (a bit per
        // position + prefix popcounts): deterministic, whatever the arrival order was.  One pass over the rows for
        // the keys, no radix pass at all (the stable LDS sort below costs 3-4 passes of a ballot per key bit).
        // A batch with a voxel of more than VQ_MAXRUN rows takes the stable sort below instead.
        bool grouped = false, skip_early = false;
        uint32_t gkey[VP_ROUNDS], ghs[VP_ROUNDS];           // grouping path: relative key; slot | arrival ticket << 16
        if (R != 0 && R <= (uint32_t)VF_CAP && sortbits <= 31) {
            for (uint32_t j = tid; j < VQ_SLOTS; j += VF_THREADS) sh.q.hset[j] = 0xFFFFFFFFu;
            for (uint32_t j = tid; j < VQ_SLOTS / 4; j += VF_THREADS) sh.q.hcnt[j] = 0u;
            if (tid == 0) sh.overflow = 0u;
            __syncthreads();
            uint32_t* const kreg = gkey;
            uint32_t* const hs = ghs;
            uint32_t fresh = 0;
#pragma unroll
            for (int h = 0; h < 2; ++h) {                   // four rows per thread in flight
                Row q[VP_ROUNDS / 2];
#pragma unroll
                for (int r = 0; r < VP_ROUNDS / 2; ++r) {
                    const uint32_t i = (h * (VP_ROUNDS / 2) + r) * VF_THREADS + tid;
                    q[r] = src[s + (i < R ? i : 0)];
                }
#pragma unroll
                for (int r = 0; r < VP_ROUNDS / 2; ++r) {
                    const uint64_t k = vx_key(g, mb, q[r]);
                    kreg[h * (VP_ROUNDS / 2) + r] = (uint32_t)((((k >> g.rem) - d0) << g.rem) | (k & remmask));
                }
            }
            {
#pragma unroll
                for (int rr = 0; rr < VP_ROUNDS; ++rr) {
                    const uint32_t i = rr * VF_THREADS + tid;
                    hs[rr] = 0xFFFFFFFFu;
                    if (i < R) {
                        uint32_t slot = (kreg[rr] * 2654435761u) >> (32 - 13);
                        for (;;) {
                            const uint32_t old = atomicCAS(&sh.q.hset[slot], 0xFFFFFFFFu, kreg[rr]);
                            if (old == 0xFFFFFFFFu) { ++fresh; break; }
                            if (old == kreg[rr]) break;
                            slot = (slot + 1) & (VQ_SLOTS - 1);
                        }
                        const uint32_t sft = (slot & 3u) * 8u;
                        const uint32_t tk = (atomicAdd(&sh.q.hcnt[slot >> 2], 1u << sft) >> sft) & 255u;
                        if (tk >= VQ_MAXRUN) sh.overflow = 1u;
                        hs[rr] = slot | (tk << 16);
                    }
                }
            }
            VX_STAMP(8);
            {
                uint32_t tot;
                vf_block_scan(fresh, sh.wsum, tot);        // (its barriers also publish the counters and the flag)
                if (tid == 0) gf_announce(status, (int64_t)t, tot);
                nvox = tot;
            }
            announced = true;
            // The grouping path pays where most voxels hold one or two rows (0.1 m voxels on 100 points / m^2: 0.86 voxels
            // per row, 2.55 against 2.93 ms per 100 M rows); where rows share voxels (0.2 m: 0.45 voxels per row) the
            // stable sort below is the faster of the two (0.30 against 0.34 ms per 10 M rows) - measured, both exact.
            if (sh.overflow == 0u && 10u * nvox >= 7u * R) {
                grouped = true;
                // ---- exclusive scan of the slots' counts: 16 slots (16 count bytes) per thread
                {
                    const uint4 c4 = reinterpret_cast<const uint4*>(sh.q.hcnt)[tid];
                    const uint32_t cw[4] = {c4.x, c4.y, c4.z, c4.w};
                    uint32_t pre[16], run = 0;
#pragma unroll
                    for (int j = 0; j < 16; ++j) { pre[j] = run; run += (cw[j >> 2] >> (8 * (j & 3))) & 255u; }
                    uint32_t tot;
                    const uint32_t ex = vf_block_scan(run, sh.wsum, tot);
                    uint4 o0, o1;
                    o0.x = (ex + pre[0]) | ((ex + pre[1]) << 16);   o0.y = (ex + pre[2]) | ((ex + pre[3]) << 16);
                    o0.z = (ex + pre[4]) | ((ex + pre[5]) << 16);   o0.w = (ex + pre[6]) | ((ex + pre[7]) << 16);
                    o1.x = (ex + pre[8]) | ((ex + pre[9]) << 16);   o1.y = (ex + pre[10]) | ((ex + pre[11]) << 16);
                    o1.z = (ex + pre[12]) | ((ex + pre[13]) << 16); o1.w = (ex + pre[14]) | ((ex + pre[15]) << 16);
                    reinterpret_cast<uint4*>(sh.q.sstart)[2 * tid] = o0;
                    reinterpret_cast<uint4*>(sh.q.sstart)[2 * tid + 1] = o1;
                }
                __syncthreads();
                VX_STAMP(9);
                // ---- positions to their slot's run (arrival order)
#pragma unroll
                for (int r = 0; r < VP_ROUNDS; ++r)
                    if (hs[r] != 0xFFFFFFFFu)
                        sh.q.pos[sh.q.sstart[hs[r] & 0xFFFFu] + (hs[r] >> 16)] = (uint16_t)(r * VF_THREADS + tid);
                __syncthreads();
                VX_STAMP(10);
#pragma unroll
                for (int r = 0; r < VP_ROUNDS; ++r) {
                    if (hs[r] != 0xFFFFFFFFu && (hs[r] >> 16) == 0u) {
                        const uint32_t slot = hs[r] & 0xFFFFu;
                        const uint32_t a0 = sh.q.sstart[slot];
                        const uint32_t cn = (sh.q.hcnt[slot >> 2] >> (8 * (slot & 3u))) & 255u;
                        for (uint32_t a = 1; a < cn; ++a) {
                            const uint16_t v = sh.q.pos[a0 + a];
                            uint32_t b = a;
                            while (b > 0 && sh.q.pos[a0 + b - 1] > v) { sh.q.pos[a0 + b] = sh.q.pos[a0 + b - 1]; --b; }
                            sh.q.pos[a0 + b] = v;
                        }
                    }
                }
                __syncthreads();
                // ---- one bit per position: is this row the first of its voxel?  Position r * 512 + tid sits in word
                // r * 8 + wave: a ballot per round, no atomics (64 lanes of a wave would hit one word)
#pragma unroll
                for (int r = 0; r < VP_ROUNDS; ++r) {
                    const bool head = hs[r] != 0xFFFFFFFFu &&
                                      sh.q.pos[sh.q.sstart[hs[r] & 0xFFFFu]] == (uint16_t)(r * VF_THREADS + tid);
                    const unsigned long long m = __ballot(head);
                    if (l == 0) sh.q.headbits[r * VF_WAVES + w] = m;
                }
                static_assert(VF_THREADS == 64 * VF_WAVES && VF_CAP == VP_ROUNDS * VF_THREADS, "position -> word");
                __syncthreads();
                if (tid < 64) {                                         // voxels in front of every 64-position word
                    const uint32_t c = tid < (int)(VF_CAP / 64) ? (uint32_t)__popcll(sh.q.headbits[tid]) : 0u;
                    const uint32_t incl = wave_scan_incl(c);
                    if (tid < (int)(VF_CAP / 64)) sh.q.headpre[tid] = incl - c;
                }
                static_assert(VF_CAP / 64 <= 64, "one wave scans the head words");
            } else {
                skip_early = true;                                      // the count is out already; sort stably below
            }
            __syncthreads();
            VX_STAMP(11);
        }
        if (R == 0 || grouped) {
            // a reserved slot that was not needed (publishes zero voxels below), or grouped above
        } else if (R <= (uint32_t)VF_CAP && sortbits <= 32) {
            // ---------------- LDS path: keys once, stable LSD passes over (key, position) ----------------
            in_lds = true;
            const int passes = (sortbits + 7) / 8;
            const uint32_t per = ((R + VF_WAVES * 64 - 1) / (VF_WAVES * 64)) * 64;   // items per wave: multiple of 64, <= 512
            const int rounds = (int)(per / 64);
            // The number of distinct keys (= voxels) is announced to the batches behind BEFORE the sort: the keys
            // are inserted into an exact hash set (open addressing, 2 * VF_CAP slots in the still unused key
            // buffers) while they are computed, so nobody waits for this batch's sort (the batch's own prefix is
            // only fetched at the end).
            const bool early = sortbits <= 31 && !skip_early;   // 0xFFFFFFFF marks an empty slot
            uint32_t* hset = &sh.key[0][0];
            constexpr uint32_t HSLOTS = 2 * VF_CAP;
            uint32_t fresh = 0;                            // keys this thread was the first to insert
            auto insert = [&](uint32_t k) {
                static_assert((HSLOTS & (HSLOTS - 1)) == 0, "power of two");
                uint32_t h = ((k * 2654435761u) >> 8) & (HSLOTS - 1);
                for (;;) {
                    const uint32_t old = atomicCAS(&hset[h], 0xFFFFFFFFu, k);
                    if (old == 0xFFFFFFFFu) { ++fresh; break; }
                    if (old == k) break;
                    h = (h + 1) & (HSLOTS - 1);
                }
            };
            auto count_and_announce = [&]() {
                uint32_t tot;
                vf_block_scan(fresh, sh.wsum, tot);
                if (tid == 0) gf_announce(status, (int64_t)t, tot);
                __syncthreads();
            };
            if (early) {
                for (uint32_t j = tid; j < HSLOTS; j += VF_THREADS) hset[j] = 0xFFFFFFFFu;
                __syncthreads();
            }
            announced = early || skip_early;
            if (skip_early) {
                // the grouping path above has computed the keys and announced the count, then stood down (rows share
                // voxels, or a voxel holds more rows than it takes): its keys are still in registers
#pragma unroll
                for (int r = 0; r < VP_ROUNDS; ++r) {
                    const uint32_t i = r * VF_THREADS + tid;
                    if (i < R) { sh.key[0][i] = gkey[r]; sh.perm[0][i] = (uint16_t)i; }
                }
            } else {
                uint32_t kreg[VP_ROUNDS];
#pragma unroll
                for (int h = 0; h < 2; ++h) {               // four rows per thread in flight
                    Row q[VP_ROUNDS / 2];
#pragma unroll
                    for (int r = 0; r < VP_ROUNDS / 2; ++r) {
                        const uint32_t i = (h * (VP_ROUNDS / 2) + r) * VF_THREADS + tid;
                        q[r] = src[s + (i < R ? i : 0)];
                    }
#pragma unroll
                    for (int r = 0; r < VP_ROUNDS / 2; ++r) {
                        const int rr = h * (VP_ROUNDS / 2) + r;
                        const uint32_t i = rr * VF_THREADS + tid;
                        const uint64_t k = vx_key(g, mb, q[r]);
                        kreg[rr] = (uint32_t)((((k >> g.rem) - d0) << g.rem) | (k & remmask));
                        if (early && i < R) insert(kreg[rr]);
                    }
                }
                if (early) count_and_announce();
#pragma unroll
                for (int r = 0; r < VP_ROUNDS; ++r) {
                    const uint32_t i = r * VF_THREADS + tid;
                    if (i < R) { sh.key[0][i] = kreg[r]; sh.perm[0][i] = (uint16_t)i; }
                }
            }
            __syncthreads();
            VX_STAMP(1);
            int cur = 0;
            for (int p = 0; p < passes; ++p) {
                const int shift = 8 * p;
                for (int j = tid; j < VF_WAVES * 256; j += VF_THREADS) (&sh.cnt[0][0])[j] = 0;
                __syncthreads();
                uint32_t kk[VP_ROUNDS], rank[VP_ROUNDS];
                uint16_t pp[VP_ROUNDS];
#pragma unroll
                for (int r = 0; r < VP_ROUNDS; ++r) {
                    if (r < rounds) {                      // wave-uniform
                        const uint32_t i = w * per + r * 64 + l;
                        const bool valid = i < R;
                        kk[r] = valid ? sh.key[cur][i] : 0u;
                        pp[r] = valid ? sh.perm[cur][i] : (uint16_t)0;
                        const uint32_t d = (kk[r] >> shift) & 255u;
                        uint32_t np;
                        const uint32_t rk = wave_match<8>(d, valid, np);
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
                    for (int w2 = 0; w2 < VF_WAVES; ++w2) tot += sh.cnt[w2][tid];
                }
                uint32_t all;
                const uint32_t ex = vf_block_scan(tid < 256 ? tot : 0u, sh.wsum, all);
                if (tid < 256) {
                    uint32_t run = ex;
#pragma unroll
                    for (int w2 = 0; w2 < VF_WAVES; ++w2) {
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
                            const uint32_t at = sh.cnt[w][(kk[r] >> shift) & 255u] + rank[r];
                            sh.key[cur ^ 1][at] = kk[r];
                            sh.perm[cur ^ 1][at] = pp[r];
                        }
                    }
                }
                __syncthreads();
                cur ^= 1;
            }
            VX_STAMP(2);
            // ---- voxel heads of the sorted keys; voxel v starts at item key[cur^1][v]
            uint32_t heads = 0;
            const uint32_t i0 = tid * VP_ROUNDS;            // 8 consecutive items per thread
            bool hd[VP_ROUNDS];
#pragma unroll
            for (int j = 0; j < VP_ROUNDS; ++j) {
                const uint32_t i = i0 + j;
                hd[j] = i < R && (i == 0 || sh.key[cur][i] != sh.key[cur][i - 1]);
                heads += hd[j];
            }
            uint32_t vi = vf_block_scan(heads, sh.wsum, nvox);
#pragma unroll
            for (int j = 0; j < VP_ROUNDS; ++j)
                if (hd[j]) sh.key[cur ^ 1][vi++] = i0 + j;
            __syncthreads();
            srt = sh.key[cur];
            sperm = sh.perm[cur];
            vst = sh.key[cur ^ 1];
        } else {
            // ---------------- general path (one oversize unit): LSD passes over the rows in global memory ----------------
            const int passes = (g.rem + 7) / 8;
            for (int j = tid; j < 8 * 256; j += VF_THREADS) (&sh.hist[0][0])[j] = 0;
            __syncthreads();
            const int np = passes > 8 ? 8 : passes;
            for (uint32_t i = tid; i < R; i += VF_THREADS) {
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
                    for (int j = tid; j < VF_WAVES * 256; j += VF_THREADS) (&sh.cnt[0][0])[j] = 0;
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
                        const uint32_t rk = wave_match<8>(dig[r], valid, npeer);
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
                        for (int w2 = 0; w2 < VF_WAVES; ++w2) {
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
            for (uint32_t t0 = 0; t0 < R; t0 += VF_THREADS * VP_ROUNDS) {
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
#ifdef PCH_VX_STAMPS
        if (tid == 0) { acc[6] += (R != 0 && !in_lds) ? 1 : 0; acc[7] += (R == 0) ? 1 : 0; }
#endif
        VX_STAMP(3);
        // ---- first output slot of this batch: look-back over the batches in front (ticket order)
        if (w == 0) {
            const uint32_t e0 = gf_lookback(status, (int64_t)t, nvox, announced);
            if (l == 0) {
                sh.vbase = e0 == GF_LB_FAILED ? 0u : e0;     // slot 0 keeps the writes below inside the output
                sh.lb_failed = e0 == GF_LB_FAILED ? 1u : 0u;
            }
        }
        __syncthreads();
        const int64_t vbase = sh.vbase;
        VX_STAMP(4);
        if (tid == 0) {
            // *out_m starts at 0: the last batch adds the total (< 2^62), a batch whose wait ran out of its budget
            // sets the sign bit (idempotent: every batch behind a poisoned one fails too) - the word reads negative
            // iff some batch failed, in whatever order the two happen
            unsigned long long* om = reinterpret_cast<unsigned long long*>(out_m);
            if (sh.lb_failed) atomicOr(om, 1ull << 63);
            if (first_of_chunk && out_chunk_offsets) out_chunk_offsets[c] = vbase;
            if (t == nbatches - 1 && !sh.lb_failed) {
                atomicAdd(om, (unsigned long long)(vbase + nvox));
                if (out_chunk_offsets) out_chunk_offsets[g.nchunks] = vbase + nvox;
            }
        }
        // ---- reduce: one thread per voxel, rows added in file order (AccumulatedPoint::AddPoint)
        if (grouped) {
            // the thread whose row arrived first on a slot reduces that voxel (slot and key are still in its registers).
            // The first rows of all (up to eight) voxels of a thread are requested together; further rows of a voxel
            // follow one after the other (file order), which few voxels need.
#pragma unroll
            for (int h = 0; h < 2; ++h) {                   // four voxels' first rows in flight per thread
                constexpr int HV = VP_ROUNDS / 2;
                uint32_t a0[HV], cn[HV], p0[HV];
                Row q0[HV];
#pragma unroll
                for (int j = 0; j < HV; ++j) {
                    const int r = h * HV + j;
                    const bool owner = ghs[r] != 0xFFFFFFFFu && (ghs[r] >> 16) == 0u;
                    const uint32_t slot = owner ? (ghs[r] & 0xFFFFu) : 0u;
                    a0[j] = sh.q.sstart[slot];
                    cn[j] = owner ? (sh.q.hcnt[slot >> 2] >> (8 * (slot & 3u))) & 255u : 0u;
                    p0[j] = owner ? sh.q.pos[a0[j]] : 0u;
                }
#pragma unroll
                for (int j = 0; j < HV; ++j) q0[j] = src[s + p0[j]];
#pragma unroll
                for (int j = 0; j < HV; ++j) {
                    if (cn[j] != 0u) {
                        double ax = 0.0 + q0[j].x, ay = 0.0 + q0[j].y, az = 0.0 + q0[j].z;   // AddPoint starts from +0.0
                        for (uint32_t a = 1; a < cn[j]; ++a) {
                            const Row q = src[s + sh.q.pos[a0[j] + a]];
                            ax += q.x; ay += q.y; az += q.z;
                        }
                        const uint32_t v = sh.q.headpre[p0[j] >> 6] +
                                           (uint32_t)__popcll(sh.q.headbits[p0[j] >> 6] & ((1ull << (p0[j] & 63u)) - 1ull));
                        const uint64_t sk = gkey[h * HV + j];
                        const uint64_t key = ((d0 + (sk >> g.rem)) << g.rem) | (sk & remmask);
                        vf_emit(g, key, ax, ay, az, cn[j], vbase + v, out_idx, out_mean, out_count);
                    }
                }
            }
        } else if (in_lds) {
            for (uint32_t v = tid; v < nvox; v += VF_THREADS) {
                const uint32_t a0 = vst[v], a1 = v + 1 < nvox ? vst[v + 1] : R;
                double ax = 0.0, ay = 0.0, az = 0.0;
                for (uint32_t i = a0; i < a1; ++i) {
                    const Row q = src[s + sperm[i]];
                    ax += q.x; ay += q.y; az += q.z;
                }
                const uint64_t sk = srt[a0];
                const uint64_t key = ((d0 + (sk >> g.rem)) << g.rem) | (sk & remmask);
                vf_emit(g, key, ax, ay, az, a1 - a0, vbase + v, out_idx, out_mean, out_count);
            }
        } else {
            for (uint32_t v = tid; v < nvox; v += VF_THREADS) {
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
        VX_STAMP(5);
    }
}

struct VoxelWs {
    unsigned long long* mm;
    double*   minb;
    int*      gmeta;
    Row      *bufA, *bufB;
    uint32_t *tile_hist, *unit_start, *nbatch, *batch_prefix, *vstart, *ticket;
    VoxelBatch* batches;
    VoxelOversize* over;
    uint32_t* nover;
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
    w.batches = a.take<VoxelBatch>(nchunks * VP_MAXBINS);
    w.nbatch = a.take<uint32_t>(nchunks + 1);
    w.batch_prefix = a.take<uint32_t>(nchunks + 1);
    w.over = a.take<VoxelOversize>(nchunks * VP_MAXBINS);
    const size_t off0 = a.off;
    w.ticket = a.take<uint32_t>(64);                       // [0] ticket, [1] oversize units, [8..24) stamp sums (tuning builds)
    w.nover = w.ticket + 1;
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
    PCH_HIP_TRY(hipMemsetAsync(out_m, 0, sizeof(int64_t), s));           // the finisher ADDS to it (see vx_finish_k)
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
    g.rvoxel = 1.0 / voxel_size;
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
               w.unit_start, w.batches, w.nbatch, w.over, w.nover);
    PCH_LAUNCH("voxel_batchscan", vx_batchscan_k, dim3(1), dim3(1024), 0, s, (const uint32_t*)w.nbatch, nchunks,
               w.batch_prefix);
    // staged through LDS where the partitioned rows do not stay in the Infinity Cache (256 MB): 100 M rows 1.39 -> 1.20 ms;
    // below that the direct form is the faster one (10 M rows: 0.125 against 0.134 ms).  PCH_VX_SCATTER=direct|lds: tuning
    static const int scatter_mode = [] {
        const char* e = getenv("PCH_VX_SCATTER");
        return e == nullptr ? 0 : (strcmp(e, "lds") == 0 ? 1 : (strcmp(e, "direct") == 0 ? 2 : 0));
    }();
    const bool scatter_lds = scatter_mode == 1 || (scatter_mode == 0 && n * (int64_t)sizeof(Row) > (int64_t(384) << 20));
    if (scatter_lds) {
        static bool attr_s[PCH_MAX_DEVICES] = {};
        const int slot_s = current_device_slot();
        if (!attr_s[slot_s]) {
            PCH_HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(vx_scatter_lds_k),
                                            hipFuncAttributeMaxDynamicSharedMemorySize, (int)sizeof(VsShared)));
            attr_s[slot_s] = true;
        }
        PCH_LAUNCH("voxel_scatter", vx_scatter_lds_k, dim3(gt), dim3(VP_THREADS), sizeof(VsShared), s, xyz, g,
                   (const double*)w.minb, (const uint32_t*)w.tile_hist, (const uint32_t*)w.unit_start, w.bufA);
    } else
    PCH_LAUNCH("voxel_scatter", vx_scatter_k, dim3(gt), dim3(VP_THREADS), 0, s, xyz, g, (const double*)w.minb,
               (const uint32_t*)w.tile_hist, (const uint32_t*)w.unit_start, w.bufA);
    {   // units above the LDS capacity (dense columns): split by their next digit, one workgroup each
        int64_t sg = nchunks * g.nb;
        if (sg > 2048) sg = 2048;                          // one unit per workgroup as a rule: 512 of them are resident at a time
        PCH_LAUNCH("voxel_split", vx_split_k, dim3((unsigned)sg), dim3(VS_THREADS), 0, s, g, (const double*)w.minb,
                   (const Row*)w.bufA, w.bufB, w.batches, (const VoxelOversize*)w.over, (const uint32_t*)w.nover);
    }
    // persistent finisher: two 512-thread workgroups per CU draw the batches in order
    int dev = 0, cus = 256;
    if (hipGetDevice(&dev) == hipSuccess) (void)hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev);
    int64_t fg = (int64_t)cus * 2;
    if (fg > nchunks * g.nb) fg = nchunks * g.nb;
    const size_t shm = sizeof(VfShared);
    static bool attr_set[PCH_MAX_DEVICES] = {};
    const int slot = current_device_slot();
    if (!attr_set[slot]) {
        PCH_HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(vx_finish_k),
                                        hipFuncAttributeMaxDynamicSharedMemorySize, (int)shm));
        attr_set[slot] = true;
    }
    PCH_LAUNCH("voxel_finish", vx_finish_k, dim3((unsigned)fg), dim3(VF_THREADS), shm, s, g,
               (const double*)w.minb, (const VoxelBatch*)w.batches, (const uint32_t*)w.batch_prefix, w.bufA, w.bufB,
               w.vstart, w.status, w.ticket, out_idx, out_mean, out_count, out_chunk_offsets, out_m,
               reinterpret_cast<unsigned long long*>(w.ticket + 8));
#ifdef PCH_VX_STAMPS
    {
        unsigned long long t[12];
        uint32_t nb_tot = 0, nov = 0;
        PCH_HIP_TRY(hipStreamSynchronize(s));
        PCH_HIP_TRY(hipMemcpy(t, w.ticket + 8, sizeof(t), hipMemcpyDeviceToHost));
        PCH_HIP_TRY(hipMemcpy(&nb_tot, w.batch_prefix + nchunks, 4, hipMemcpyDeviceToHost));
        PCH_HIP_TRY(hipMemcpy(&nov, w.nover, 4, hipMemcpyDeviceToHost));
        const char* nm[6] = {"ticket+lookup", "load+keys", "sort passes", "heads", "look-back", "reduce"};
        fprintf(stderr, "voxel finisher: %u batches, %u oversize units; per-workgroup average (us):", nb_tot, nov);
        for (int k = 0; k < 6; ++k) fprintf(stderr, "  %s %.1f", nm[k], (double)t[k] / 100.0 / (double)fg);
        fprintf(stderr, "  | global-path batches %llu, empty slots %llu", t[6], t[7]);
        fprintf(stderr, "  | grouping path: init+keys+insert %.1f, announce+slot scan %.1f, scatter %.1f, sort+heads %.1f",
                (double)t[8] / 100.0 / (double)fg, (double)t[9] / 100.0 / (double)fg, (double)t[10] / 100.0 / (double)fg,
                (double)t[11] / 100.0 / (double)fg);
        unsigned long long polls = 0, wins = 0;
        (void)hipMemcpyFromSymbol(&polls, HIP_SYMBOL(g_lb_polls), 8);
        (void)hipMemcpyFromSymbol(&wins, HIP_SYMBOL(g_lb_windows), 8);
        fprintf(stderr, "  | look-back totals so far: %llu windows, %llu failed polls\n", wins, polls);
    }
#endif
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
