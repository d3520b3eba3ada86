// Stage B: float32 centroid (numpy sequential-sum semantics), centring, numpy 'linear'
// percentile by radix select, and the order-preserving height-filter compaction.
// Reference: utils/tower_extraction.py:62-64 (centroid, centring), :82-89 (percentile filter).
#include "pch_prims.h"
#include "pch_mean.h"

namespace pch {

// =====================================================================================
// B2: k-th order statistics of v[i] = base[i*stride] - sub by 12/12/8-bit radix select on the
// order-preserving uint32 image of the float.  NaN sorts last (as numpy's partition).
// x -> fl(x - sub) is monotone non-decreasing, so the k-th smallest of v is fl(k-th smallest of
// base - sub): the select runs on the raw values and the subtraction is applied to the two
// selected order statistics only.
// =====================================================================================
constexpr int SEL_BINS = 4096;
constexpr int SEL_TILE = 4096;          // values per workgroup trip in the histogram passes

struct SelState {
    unsigned long long rank;       // remaining rank inside the current prefix
    unsigned long long less;       // elements strictly below the current prefix
    uint32_t prefix;               // resolved high bits
    uint32_t need_next;            // 1: k1 is not inside the final bin of k0
    uint32_t v0key, v1key;
    unsigned long long nan_count;
    uint32_t next_min;             // min key > v0key (pass 4)
    uint32_t pad;
};

__device__ __forceinline__ uint32_t sel_key(float v) {
    return (v != v) ? 0xFFFFFFFFu : f32_ordered(v);
}

// adds 1 to h[bin] for every active lane; lanes of a wave that share a bin are merged into one
// LDS atomic (z values of a flat corridor fall into a handful of bins)
__device__ __forceinline__ void sel_hist_add(uint32_t* h, bool active, uint32_t bin) {
    const unsigned long long todo = __ballot(active);
    if (!todo) return;
    // one merged atomic for the most likely bin (the first active lane's), plain atomics for the rest
    const int leader = (int)__builtin_ctzll(todo);
    const uint32_t b0 = (uint32_t)__builtin_amdgcn_readlane((int)bin, leader);
    const unsigned long long same = __ballot(active && bin == b0);
    if (lane_id() == leader) atomicAdd(&h[b0], (uint32_t)__popcll(same));
    if (active && bin != b0) atomicAdd(&h[bin], 1u);
}

template <int PASS>
__global__ __launch_bounds__(256) void sel_hist_k(const float* __restrict__ base, int64_t n,
                                                  int64_t stride, SelState* __restrict__ st,
                                                  uint32_t* __restrict__ hist) {
    __shared__ uint32_t h[SEL_BINS];
    for (int j = threadIdx.x; j < SEL_BINS; j += 256) h[j] = 0;
    __syncthreads();
    const uint32_t prefix = st->prefix;
    unsigned long long nans = 0;
    auto take = [&](bool in, float v) {
        const uint32_t k = sel_key(v);
        if (PASS == 0) {
            sel_hist_add(h, in, k >> 20);
            nans += (in && v != v);
        } else if (PASS == 1) {
            sel_hist_add(h, in && (k >> 20) == prefix, (k >> 8) & 0xFFFu);
        } else {
            sel_hist_add(h, in && (k >> 8) == prefix, k & 0xFFu);
        }
    };
    // a workgroup takes tiles of 4096 values; on a contiguous, 16-byte aligned column every thread
    // keeps four float4 loads in flight (the histogram update behind a load is a dependent chain)
    const int64_t span = (int64_t)gridDim.x * SEL_TILE;
    const bool vec = stride == 1 && (reinterpret_cast<uintptr_t>(base) & 15u) == 0;
    for (int64_t t0 = (int64_t)blockIdx.x * SEL_TILE; t0 < n; t0 += span) {    // wave-uniform trip count
        if (vec && t0 + SEL_TILE <= n) {
            const float4* b4 = reinterpret_cast<const float4*>(base + t0);
            float4 q[4];
#pragma unroll
            for (int r = 0; r < 4; ++r) q[r] = b4[r * 256 + threadIdx.x];
#pragma unroll
            for (int r = 0; r < 4; ++r) { take(true, q[r].x); take(true, q[r].y); take(true, q[r].z); take(true, q[r].w); }
        } else {
#pragma unroll 4
            for (int r = 0; r < SEL_TILE / 256; ++r) {
                const int64_t i = t0 + r * 256 + threadIdx.x;
                const bool in = i < n;
                take(in, in ? base[i * stride] : 0.0f);
            }
        }
    }
    __syncthreads();
    const int nb = (PASS == 2) ? 256 : SEL_BINS;
    for (int j = threadIdx.x; j < nb; j += 256)
        if (h[j]) atomicAdd(&hist[j], h[j]);
    if (PASS == 0) {
        nans = wave_reduce_add(nans);
        if (lane_id() == 0 && nans) atomicAdd(&st->nan_count, nans);
    }
}

// single workgroup: locate the bin holding the wanted rank, extend the prefix, clear hist
template <int PASS>
__global__ __launch_bounds__(256) void sel_pick_k(SelState* __restrict__ st, uint32_t* __restrict__ hist) {
    __shared__ unsigned long long wsum[4];
    __shared__ int found_bin;
    __shared__ unsigned long long found_below;
    const int nb = (PASS == 2) ? 256 : SEL_BINS;
    const int per = nb / 256;                       // bins per thread (16 or 1)
    const unsigned long long rank = st->rank;
    unsigned long long loc[16];
    unsigned long long tsum = 0;
    for (int j = 0; j < per; ++j) { loc[j] = hist[threadIdx.x * per + j]; tsum += loc[j]; }
    const unsigned long long incl = wave_scan_incl(tsum);
    if (lane_id() == 63) wsum[wave_id()] = incl;
    if (threadIdx.x == 0) found_bin = -1;
    __syncthreads();
    unsigned long long before = incl - tsum;
    for (int w = 0; w < wave_id(); ++w) before += wsum[w];
    // the thread whose bin range [before, before+tsum) contains rank owns the answer
    if (rank >= before && rank < before + tsum) {
        unsigned long long b = before;
        for (int j = 0; j < per; ++j) {
            if (rank < b + loc[j]) { found_bin = threadIdx.x * per + j; found_below = b; break; }
            b += loc[j];
        }
    }
    __syncthreads();
    for (int j = threadIdx.x; j < nb; j += 256) {
        if (PASS == 2 && j == found_bin) {
            // does rank+1 (the 'next' order statistic) still fall into this final bin?
            const unsigned long long cnt = hist[j];
            st->need_next = (rank + 1 < found_below + cnt) ? 0u : 1u;
        }
    }
    __syncthreads();
    for (int j = threadIdx.x; j < nb; j += 256) hist[j] = 0;
    if (threadIdx.x == 0) {
        const int bin = found_bin < 0 ? 0 : found_bin;   // n == 0 never reaches here
        st->less += found_below;
        st->rank = rank - found_below;
        if (PASS == 0) st->prefix = (uint32_t)bin;
        else if (PASS == 1) st->prefix = (st->prefix << 12) | (uint32_t)bin;
        else { st->v0key = (st->prefix << 8) | (uint32_t)bin; st->next_min = 0xFFFFFFFFu; }
    }
}

// pass 4 (only when needed): smallest key strictly above v0key
__global__ __launch_bounds__(256) void sel_next_k(const float* __restrict__ base, int64_t n,
                                                  int64_t stride, SelState* __restrict__ st) {
    if (st->need_next == 0) return;
    const uint32_t v0 = st->v0key;
    uint32_t best = 0xFFFFFFFFu;
    auto take = [&](float v) {
        const uint32_t k = sel_key(v);
        if (k > v0 && k < best) best = k;
    };
    const int64_t span = (int64_t)gridDim.x * SEL_TILE;
    const bool vec = stride == 1 && (reinterpret_cast<uintptr_t>(base) & 15u) == 0;
    for (int64_t t0 = (int64_t)blockIdx.x * SEL_TILE; t0 < n; t0 += span) {
        if (vec && t0 + SEL_TILE <= n) {
            const float4* b4 = reinterpret_cast<const float4*>(base + t0);
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const float4 q = b4[r * 256 + threadIdx.x];
                take(q.x); take(q.y); take(q.z); take(q.w);
            }
        } else {
            for (int r = 0; r < SEL_TILE / 256; ++r) {
                const int64_t i = t0 + r * 256 + threadIdx.x;
                if (i < n) take(base[i * stride]);
            }
        }
    }
    best = wave_reduce_min(best);
    if (lane_id() == 0 && best != 0xFFFFFFFFu) atomicMin(&st->next_min, best);
}

__device__ __forceinline__ float sel_key_to_float(uint32_t k) {
    return (k == 0xFFFFFFFFu) ? __uint_as_float(0x7FC00000u) : f32_unordered(k);
}

// numpy _lerp in float32: a + (b-a)*t, and b - (b-a)*(1-t) where t >= 0.5
// (numpy/lib/_function_base_impl.py:4639-4660); NaN anywhere -> NaN.
// scal: [0] = percentile, [1] = percentile + add1, [2] = percentile + add2
__global__ void sel_lerp_k(const SelState* __restrict__ st, int same_index, float gamma,
                           const float* __restrict__ sub, float add1, float add2,
                           float* __restrict__ scal) {
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    const float c = sub ? *sub : 0.0f;
    const float a = sel_key_to_float(st->v0key) - c;
    float b = a;
    if (!same_index && st->need_next) b = sel_key_to_float(st->next_min) - c;
    const float diff = b - a;
    float r = a + diff * gamma;
    if (gamma >= 0.5f) r = b - diff * (1.0f - gamma);
    if (st->nan_count) r = __uint_as_float(0x7FC00000u);
    scal[0] = r;
    scal[1] = r + add1;
    scal[2] = r + add2;
}

__global__ void sel_init_k(SelState* st, unsigned long long rank) { st->rank = rank; }

struct SelWs {
    SelState* st;
    uint32_t* hist;
    float*    scal;     // 4 floats
};
static void sel_plan(Arena& a, SelWs& w) {
    w.st = a.take<SelState>(1);
    w.hist = a.take<uint32_t>(SEL_BINS);
    w.scal = a.take<float>(4);
}

// host side of np.percentile's index arithmetic (float32 under NEP 50)
struct PctIndex { int64_t k0; int same; float gamma; };
static PctIndex pct_index(int64_t n, double q_percent) {
    PctIndex r;
    const float q = (float)q_percent / 100.0f;          // np.true_divide(q, float32(100))
    const float vi = (float)(n - 1) * q;                // (n - 1) * quantiles -> float32
    float prev = floorf(vi);
    r.gamma = vi - prev;
    r.same = 0;
    if (vi >= (float)(n - 1)) { prev = (float)(n - 1); r.same = 1; }   // indexes -> -1 (last)
    if (vi < 0.0f) { prev = 0.0f; r.same = 1; }
    int64_t k0 = (int64_t)prev;
    if (k0 > n - 1) k0 = n - 1;                          // float32(n-1) may round up
    if (k0 < 0) k0 = 0;
    if (k0 == n - 1) r.same = 1;
    r.k0 = k0;
    return r;
}

// the histogram / pick / next passes (need only the raw values) ...
static int select_passes(const float* base, int64_t n, int64_t stride, double q_percent, SelWs& w,
                         hipStream_t s) {
    const PctIndex pi = pct_index(n, q_percent);
    PCH_HIP_TRY(hipMemsetAsync(w.hist, 0, sizeof(uint32_t) * SEL_BINS, s));
    PCH_HIP_TRY(hipMemsetAsync(w.st, 0, sizeof(SelState), s));
    PCH_LAUNCH("sel_init", sel_init_k, dim3(1), dim3(1), 0, s, w.st, (unsigned long long)pi.k0);
    int64_t gb = ceil_div(n, SEL_TILE);
    if (gb > 2048) gb = 2048;
    if (gb < 1) gb = 1;
    const dim3 grid((unsigned)gb), blk(256);
    PCH_LAUNCH("sel_hist0", sel_hist_k<0>, grid, blk, 0, s, base, n, stride, w.st, w.hist);
    PCH_LAUNCH("sel_pick0", sel_pick_k<0>, dim3(1), blk, 0, s, w.st, w.hist);
    PCH_LAUNCH("sel_hist1", sel_hist_k<1>, grid, blk, 0, s, base, n, stride, w.st, w.hist);
    PCH_LAUNCH("sel_pick1", sel_pick_k<1>, dim3(1), blk, 0, s, w.st, w.hist);
    PCH_LAUNCH("sel_hist2", sel_hist_k<2>, grid, blk, 0, s, base, n, stride, w.st, w.hist);
    PCH_LAUNCH("sel_pick2", sel_pick_k<2>, dim3(1), blk, 0, s, w.st, w.hist);
    if (!pi.same)
        PCH_LAUNCH("sel_next", sel_next_k, grid, blk, 0, s, base, n, stride, w.st);
    return PCH_OK;
}
// ... and the final interpolation, which is where `sub` (the centroid) enters
static int select_lerp(int64_t n, const float* sub, double q_percent, float add1, float add2, SelWs& w,
                       hipStream_t s) {
    const PctIndex pi = pct_index(n, q_percent);
    PCH_LAUNCH("sel_lerp", sel_lerp_k, dim3(1), dim3(64), 0, s, (const SelState*)w.st, pi.same, pi.gamma,
               sub, add1, add2, w.scal);
    return PCH_OK;
}
static int select_percentile(const float* base, int64_t n, int64_t stride, const float* sub,
                             double q_percent, float add1, float add2, SelWs& w, hipStream_t s) {
    PCH_TRY(select_passes(base, n, stride, q_percent, w, s));
    return select_lerp(n, sub, q_percent, add1, add2, w, s);
}

// per-thread side stream + events: the select passes only need the raw z column, so they run
// beside the (latency-bound, 3-wave) centroid walk
struct SideStream { hipStream_t s; hipEvent_t ev_fork, ev_join; bool ok; };
static SideStream& side_stream() {
    static thread_local SideStream ss = {nullptr, nullptr, nullptr, false};
    if (!ss.ok) {
        if (hipStreamCreateWithFlags(&ss.s, hipStreamNonBlocking) == hipSuccess &&
            hipEventCreateWithFlags(&ss.ev_fork, hipEventDisableTiming) == hipSuccess &&
            hipEventCreateWithFlags(&ss.ev_join, hipEventDisableTiming) == hipSuccess)
            ss.ok = true;
    }
    return ss;
}

// =====================================================================================
// B2/B3: keep = (z - cz) > thr, order preserving.  Two launches over the raw points:
// count (both thresholds at once, so the <min_keep fallback needs no host round trip),
// then scatter of the centred coordinates.
// =====================================================================================
constexpr int GF_THREADS = 256;
constexpr int GF_ROUNDS  = 8;
constexpr int GF_TILE    = GF_THREADS * GF_ROUNDS;   // 2048 points per workgroup

struct GfState {
    uint32_t total_a, total_b;             // kept with threshold A (offset) / B (fallback): scan totals
    uint32_t use_b;
    uint32_t aabb[6];                      // ordered-uint32 min xyz / max xyz
    uint32_t pad;
};

__global__ __launch_bounds__(GF_THREADS) void gf_count_k(const float* __restrict__ zcol, int64_t n,
                                                         const float* __restrict__ centroid,
                                                         const float* __restrict__ scal,
                                                         uint32_t* __restrict__ cnt_a,
                                                         uint32_t* __restrict__ cnt_b) {
    __shared__ uint32_t sa[GF_THREADS / 64], sb[GF_THREADS / 64];
    const float cz = centroid[2];
    const float thr_a = scal[1], thr_b = scal[2];
    const int64_t base = (int64_t)blockIdx.x * GF_TILE;
    uint32_t a = 0, b = 0;
#pragma unroll
    for (int r = 0; r < GF_ROUNDS; ++r) {
        const int64_t i = base + r * GF_THREADS + threadIdx.x;
        if (i < n) {
            const float z = zcol[i] - cz;
            a += (z > thr_a);
            b += (z > thr_b);
        }
    }
    a = wave_reduce_add(a);
    b = wave_reduce_add(b);
    if (lane_id() == 0) { sa[wave_id()] = a; sb[wave_id()] = b; }
    __syncthreads();
    if (threadIdx.x == 0) {
        const uint32_t ta = sa[0] + sa[1] + sa[2] + sa[3];
        const uint32_t tb = sb[0] + sb[1] + sb[2] + sb[3];
        cnt_a[blockIdx.x] = ta;
        cnt_b[blockIdx.x] = tb;
    }
}

// picks the threshold (utils/tower_extraction.py:87-89) and publishes the scalars
__global__ void gf_decide_k(GfState* __restrict__ st, long long min_keep,
                            const float* __restrict__ centroid, float* __restrict__ scal,
                            float* __restrict__ out_scalars, int64_t* __restrict__ out_count) {
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    const bool use_b = (long long)st->total_a < min_keep;
    st->use_b = use_b ? 1u : 0u;
    const float thr = use_b ? scal[2] : scal[1];
    scal[3] = thr;
    out_scalars[0] = centroid[0];
    out_scalars[1] = centroid[1];
    out_scalars[2] = centroid[2];
    out_scalars[3] = scal[0];
    out_scalars[4] = thr;
    out_scalars[5] = use_b ? 1.0f : 0.0f;
    out_scalars[6] = (float)st->total_a;      // kept at the first threshold (exact below 2^24)
    out_scalars[7] = 0.0f;
    *out_count = (int64_t)(use_b ? st->total_b : st->total_a);
    for (int a = 0; a < 3; ++a) { st->aabb[a] = 0xFFFFFFFFu; st->aabb[3 + a] = 0u; }
}

__global__ __launch_bounds__(GF_THREADS) void gf_scatter_k(
    const float* __restrict__ raw, const float* __restrict__ zcol, int64_t n,
    const float* __restrict__ centroid,
    const float* __restrict__ scal, const uint32_t* __restrict__ off_a,
    const uint32_t* __restrict__ off_b, GfState* __restrict__ st,
    float* __restrict__ out_points, int32_t* __restrict__ out_index) {
    __shared__ uint32_t wtot[GF_THREADS / 64];
    const float cx = centroid[0], cy = centroid[1], cz = centroid[2];
    const float thr = scal[3];
    const uint32_t block_off = st->use_b ? off_b[blockIdx.x] : off_a[blockIdx.x];
    const int w = wave_id(), l = lane_id();
    const int64_t seg = (int64_t)blockIdx.x * GF_TILE + (int64_t)w * (64 * GF_ROUNDS);
    float px[GF_ROUNDS], py[GF_ROUNDS], pz[GF_ROUNDS];
    uint32_t pos[GF_ROUNDS];
    uint32_t run = 0;
    const uint64_t lt = lanemask_lt();
#pragma unroll
    for (int r = 0; r < GF_ROUNDS; ++r) {
        const int64_t i = seg + r * 64 + l;
        bool keep = false;
        if (i < n) {
            pz[r] = zcol[i] - cz;                 // points = raw_points - centroid (float32)
            keep = pz[r] > thr;
            if (keep) {                           // x,y are only fetched for survivors
                px[r] = raw[3 * i + 0] - cx;
                py[r] = raw[3 * i + 1] - cy;
            }
        }
        const uint64_t m = __ballot(keep);
        pos[r] = keep ? run + (uint32_t)__popcll(m & lt) : 0xFFFFFFFFu;
        run += (uint32_t)__popcll(m);
    }
    if (l == 0) wtot[w] = run;
    __syncthreads();
    uint32_t woff = block_off;
    for (int w2 = 0; w2 < w; ++w2) woff += wtot[w2];
#pragma unroll
    for (int r = 0; r < GF_ROUNDS; ++r) {
        if (pos[r] != 0xFFFFFFFFu) {
            const int64_t o = (int64_t)woff + pos[r];
            out_points[3 * o + 0] = px[r];
            out_points[3 * o + 1] = py[r];
            out_points[3 * o + 2] = pz[r];
            if (out_index) out_index[o] = (int32_t)(seg + r * 64 + l);
        }
    }
}

// bounding box of the kept points: grid-stride over the compacted output, one set of atomics per
// workgroup
__global__ __launch_bounds__(256) void gf_aabb_k(const float* __restrict__ pts, const int64_t* __restrict__ count,
                                                 GfState* __restrict__ st) {
    __shared__ uint32_t sm[4][6];
    const int64_t total = 3 * (*count);
    uint32_t lo[3] = {0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu}, hi[3] = {0u, 0u, 0u};
    // flat float index e = 3*i + axis; a thread's stride (gridDim*256*... ) is a multiple of 3 so
    // every thread always sees the same axis
    const int64_t stride = (int64_t)gridDim.x * 256 * 3;
    for (int64_t e0 = ((int64_t)blockIdx.x * 256 + threadIdx.x) * 3; e0 < total; e0 += stride) {
        const float x = pts[e0], y = pts[e0 + 1], z = pts[e0 + 2];
        if (!(fabsf(x) < INFINITY && fabsf(y) < INFINITY && fabsf(z) < INFINITY)) continue;   // NaN/inf rows
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
        if (lane_id() == 0) { sm[wave_id()][a] = lo[a]; sm[wave_id()][3 + a] = hi[a]; }
    }
    __syncthreads();
    if (threadIdx.x < 6) {
        const int a = threadIdx.x;
        uint32_t v = sm[0][a];
        for (int w = 1; w < 4; ++w) v = (a < 3) ? (sm[w][a] < v ? sm[w][a] : v) : (sm[w][a] > v ? sm[w][a] : v);
        if (a < 3) { if (v != 0xFFFFFFFFu) atomicMin(&st->aabb[a], v); }
        else       { if (v != 0u) atomicMax(&st->aabb[a], v); }
    }
}

__global__ void gf_finalize_k(const GfState* __restrict__ st, float* __restrict__ out_aabb) {
    if (threadIdx.x < 6 && blockIdx.x == 0 && out_aabb) {
        const uint32_t k = st->aabb[threadIdx.x];
        const bool empty = (threadIdx.x < 3) ? (k == 0xFFFFFFFFu) : (k == 0u);
        out_aabb[threadIdx.x] = empty ? 0.0f : f32_unordered(k);
    }
}

struct GfWs {
    float*    centroid;
    float*    zcol;              // copy of the z column (written by the centroid pass)
    MsWs      ms;
    SelWs     sel;
    GfState*  st;
    uint32_t *cnt_a, *cnt_b, *scan_ws;
};
static void gf_plan(Arena& a, int64_t n, GfWs& w) {
    const int64_t nb = ceil_div(n > 0 ? n : 1, GF_TILE);
    w.centroid = a.take<float>(4);
    ms_plan(a, n, w.ms);
    w.zcol = a.take<float>(n > 0 ? n : 1);
    sel_plan(a, w.sel);
    w.st = a.take<GfState>(1);
    w.cnt_a = a.take<uint32_t>(nb + 8);
    w.cnt_b = a.take<uint32_t>(nb + 8);
    w.scan_ws = a.take<uint32_t>(scan_ws_u32(nb));
}

}  // namespace pch

using namespace pch;

extern "C" size_t pch_mean_seq_f32_ws_bytes(int64_t n) {
    if (n < 0) return 0;
    Arena a;
    MsWs w;
    ms_plan(a, n, w);
    return a.off;
}

extern "C" int pch_mean_seq_f32(const float* xyz, int64_t n, float* out_centroid, void* ws,
                                size_t ws_bytes, void* stream) {
    prof_begin_call();
    PCH_REQUIRE(n >= 0 && out_centroid && ws, "bad argument");
    PCH_REQUIRE(n == 0 || xyz, "null input");
    Arena a(ws, ws_bytes);
    MsWs w;
    ms_plan(a, n, w);
    if (a.overflow) { set_error("workspace too small: need %zu bytes", a.off); return PCH_ERR_WORKSPACE; }
    return mean_seq_launch(xyz, n, out_centroid, w, nullptr, (hipStream_t)stream);
}

extern "C" int pch_mean_seq_serial_f32(const float* xyz, int64_t n, float* out_centroid, void* stream) {
    prof_begin_call();
    PCH_REQUIRE(n >= 0 && out_centroid, "bad argument");
    PCH_REQUIRE(n == 0 || xyz, "null input");
    return mean_seq_serial_launch(xyz, n, out_centroid, (hipStream_t)stream);
}

extern "C" size_t pch_percentile_f32_ws_bytes(int64_t) {
    Arena a;
    SelWs w;
    sel_plan(a, w);
    return a.off;
}

extern "C" int pch_percentile_f32(const float* base, int64_t n, int64_t stride, const float* sub,
                                  double q_percent, float* out, void* ws, size_t ws_bytes,
                                  void* stream) {
    prof_begin_call();
    hipStream_t s = (hipStream_t)stream;
    PCH_REQUIRE(n >= 1, "percentile of an empty array (numpy raises IndexError)");
    PCH_REQUIRE(base && out && ws && stride >= 1, "bad argument");
    PCH_REQUIRE(q_percent >= 0.0 && q_percent <= 100.0, "Percentiles must be in the range [0, 100]");
    Arena a(ws, ws_bytes);
    SelWs w;
    sel_plan(a, w);
    if (a.overflow) { set_error("workspace too small: need %zu bytes", a.off); return PCH_ERR_WORKSPACE; }
    PCH_TRY(select_percentile(base, n, stride, sub, q_percent, 0.0f, 0.0f, w, s));
    PCH_HIP_TRY(hipMemcpyAsync(out, w.scal, sizeof(float), hipMemcpyDeviceToDevice, s));
    return PCH_OK;
}

extern "C" size_t pch_ground_filter_ws_bytes(int64_t n) {
    if (n < 0) return 0;
    Arena a;
    GfWs w;
    gf_plan(a, n, w);
    return a.off;
}

extern "C" int pch_ground_filter_f32(const float* raw, int64_t n, double pct, float offset,
                                     float fallback_offset, int64_t min_keep, float* out_points,
                                     int32_t* out_index, float* out_scalars, int64_t* out_count,
                                     float* out_aabb, void* ws, size_t ws_bytes, void* stream) {
    prof_begin_call();
    hipStream_t s = (hipStream_t)stream;
    PCH_REQUIRE(n >= 1 && n < (int64_t(1) << 31), "n out of range [1, 2^31) (numpy raises on empty input)");
    PCH_REQUIRE(raw && out_points && out_scalars && out_count && ws, "null buffer");
    PCH_REQUIRE(pct >= 0.0 && pct <= 100.0, "Percentiles must be in the range [0, 100]");
    Arena a(ws, ws_bytes);
    GfWs w;
    gf_plan(a, n, w);
    if (a.overflow) { set_error("workspace too small: need %zu bytes", a.off); return PCH_ERR_WORKSPACE; }
    const int64_t nb = ceil_div(n, GF_TILE);

    SideStream& ss = side_stream();
    if (ss.ok) {
        PCH_TRY(mean_seq_launch(raw, n, w.centroid, w.ms, w.zcol, s, ss.ev_fork));
        PCH_HIP_TRY(hipStreamWaitEvent(ss.s, ss.ev_fork, 0));
        PCH_TRY(select_passes(w.zcol, n, 1, pct, w.sel, ss.s));
        PCH_HIP_TRY(hipEventRecord(ss.ev_join, ss.s));
        PCH_HIP_TRY(hipStreamWaitEvent(s, ss.ev_join, 0));
    } else {
        PCH_TRY(mean_seq_launch(raw, n, w.centroid, w.ms, w.zcol, s));
        PCH_TRY(select_passes(w.zcol, n, 1, pct, w.sel, s));
    }
    PCH_TRY(select_lerp(n, w.centroid + 2, pct, offset, fallback_offset, w.sel, s));
    PCH_HIP_TRY(hipMemsetAsync(w.st, 0, sizeof(GfState), s));
    PCH_LAUNCH("gf_count", gf_count_k, dim3((unsigned)nb), dim3(GF_THREADS), 0, s, (const float*)w.zcol, n,
               (const float*)w.centroid, (const float*)w.sel.scal, w.cnt_a, w.cnt_b);
    PCH_TRY(scan_exclusive_u32(w.cnt_a, w.cnt_a, nb, w.scan_ws, &w.st->total_a, s));
    PCH_TRY(scan_exclusive_u32(w.cnt_b, w.cnt_b, nb, w.scan_ws, &w.st->total_b, s));
    PCH_LAUNCH("gf_decide", gf_decide_k, dim3(1), dim3(64), 0, s, w.st, (long long)min_keep,
               (const float*)w.centroid, w.sel.scal, out_scalars, out_count);
    PCH_LAUNCH("gf_scatter", gf_scatter_k, dim3((unsigned)nb), dim3(GF_THREADS), 0, s, raw,
               (const float*)w.zcol, n,
               (const float*)w.centroid, (const float*)w.sel.scal, (const uint32_t*)w.cnt_a,
               (const uint32_t*)w.cnt_b, w.st, out_points, out_index);
    if (out_aabb) {
        PCH_LAUNCH("gf_aabb", gf_aabb_k, dim3(512), dim3(256), 0, s, (const float*)out_points,
                   (const int64_t*)out_count, w.st);
        PCH_LAUNCH("gf_finalize", gf_finalize_k, dim3(1), dim3(64), 0, s, (const GfState*)w.st, out_aabb);
    }
    return PCH_OK;
}
