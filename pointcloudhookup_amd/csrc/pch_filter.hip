// Stage B: float32 centroid (numpy sequential-sum semantics), centring, numpy 'linear'
// percentile by radix select, and the order-preserving height-filter compaction.
// Reference: utils/tower_extraction.py:62-64 (centroid, centring), :82-89 (percentile filter).
#include "pch_prims.h"
#include "pch_mean.h"
#include "pch_lookback.h"

namespace pch {

// =====================================================================================
// B2: k-th order statistics of v[i] = base[i*stride] - sub by 12/12/8-bit radix select on the
// order-preserving uint32 image of the float.  NaN sorts last (as numpy's partition).
// x -> fl(x - sub) is monotone non-decreasing, so the k-th smallest of v is fl(k-th smallest of
// base - sub): the select runs on the raw values and the subtraction is applied to the two
// selected order statistics only.
// =====================================================================================
constexpr int SEL_BINS = 4096;
constexpr int SEL_REP0 = 4;             // histogram copies in the first pass
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
template <int ROUNDS>
__device__ __forceinline__ void sel_hist_add(uint32_t* h, bool active, uint32_t bin) {
    // up to ROUNDS rounds of 'merge every lane that shares the first pending lane's bin into one
    // LDS atomic' while many lanes are pending, plain atomics for the rest
    bool pending = active;
#pragma unroll
    for (int round = 0; round < ROUNDS; ++round) {
        const unsigned long long todo = __ballot(pending);
        if (__popcll(todo) < 8) break;
        const int leader = (int)__builtin_ctzll(todo);
        const uint32_t b0 = (uint32_t)__builtin_amdgcn_readlane((int)bin, leader);
        const unsigned long long same = __ballot(pending && bin == b0);
        if (lane_id() == leader) atomicAdd(&h[b0], (uint32_t)__popcll(same));
        pending = pending && bin != b0;
    }
    if (pending) atomicAdd(&h[bin], 1u);
}

// Source switch of the bracketed select (below): when *ok is set the kernel works on the candidate buffer
// (alt_base, *alt_n values, contiguous) instead of the column it was launched for - one launch serves either case.
struct SelGate { const uint32_t* ok; const float* alt_base; const uint32_t* alt_n; };
__device__ __forceinline__ void sel_gate(const SelGate& g, const float*& base, int64_t& n, int64_t& stride) {
    if (g.ok && *g.ok) { base = g.alt_base; n = (int64_t)*g.alt_n; stride = 1; }
}

template <int PASS>
__global__ __launch_bounds__(256) void sel_hist_k(const float* __restrict__ base_in, int64_t n,
                                                  int64_t stride, SelState* __restrict__ st,
                                                  uint32_t* __restrict__ hist, SelGate gate = SelGate{nullptr, nullptr, nullptr}) {
    const float* base = base_in;
    sel_gate(gate, base, n, stride);
    // pass 0 sees a handful of hot bins (z of a flat corridor): SEL_REP0 copies of the histogram, picked
    // by lane, divide the same-address serialisation of the LDS atomics
    constexpr int REP = PASS == 0 ? SEL_REP0 : 1;
    __shared__ uint32_t hh[REP][SEL_BINS];
    for (int j = threadIdx.x; j < REP * SEL_BINS; j += 256) (&hh[0][0])[j] = 0;
    __syncthreads();
    uint32_t* h = hh[lane_id() & (REP - 1)];
    const uint32_t prefix = st->prefix;
    unsigned long long nans = 0;
    auto take = [&](bool in, float v) {
        const uint32_t k = sel_key(v);
        if (PASS == 0) {
            if (in) atomicAdd(&h[k >> 20], 1u);
            nans += (in && v != v);
        } else if (PASS == 1) {
            sel_hist_add<1>(h, in && (k >> 20) == prefix, (k >> 8) & 0xFFFu);
        } else {
            sel_hist_add<1>(h, in && (k >> 8) == prefix, k & 0xFFu);
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
    for (int j = threadIdx.x; j < nb; j += 256) {
        uint32_t t = 0;
#pragma unroll
        for (int r = 0; r < REP; ++r) t += hh[r][j];
        if (t) atomicAdd(&hist[j], t);
    }
    if (PASS == 0) {
        nans = wave_reduce_add(nans);
        if (lane_id() == 0 && nans) atomicAdd(&st->nan_count, nans);
    }
}

// pass 4 (only when needed): smallest key strictly above v0key
__global__ __launch_bounds__(256) void sel_next_k(const float* __restrict__ base_in, int64_t n,
                                                  int64_t stride, SelState* __restrict__ st,
                                                  SelGate gate = SelGate{nullptr, nullptr, nullptr}) {
    const float* base = base_in;
    sel_gate(gate, base, n, stride);
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

// (the interpolation itself is numpy's _lerp in float32: a + (b-a)*t, and b - (b-a)*(1-t) where t >= 0.5,
// numpy/lib/_function_base_impl.py:4639-4660; NaN anywhere -> NaN.  scal: [0] = percentile,
// [1] = percentile + add1, [2] = percentile + add2 - see selx_lerp_k)

// ---- the same select with every pick folded into its consumer --------------------------------------
// A kernel of a few microseconds still costs a launch slot (~5 us each on this stream), and a percentile used
// to be 9 of them (27 with the sample bracket below).  Here the single-workgroup pick of pass p runs as the
// prologue of whatever needs its result - histogram pass p+1, the "next key" pass, the bracket sweep, the final
// interpolation: every workgroup repeats it from the finished histogram of pass p (16 KB, L2), workgroup 0 stores
// the new state for the kernels behind.  Histograms of the three passes live side by side and are cleared once.
// NR = 2 selects two ranks of the same data in the same sweeps (the two ends of the sample bracket).
struct SelRun {
    SelState st[4];                 // st[p]: state in front of histogram pass p; st[3]: after the last pick
    unsigned long long nan_count;
    uint32_t next_min, pad;
};
__device__ __forceinline__ uint32_t* selx_hist_of(uint32_t* hist, int run, int pass) {
    return hist + ((pass == 0 ? 0 : run) * 3 + pass) * SEL_BINS;        // pass 0 has no prefix: one histogram for all runs
}

template <int PASS>
__device__ __forceinline__ SelState sel_pick_block(const SelState& in, const uint32_t* __restrict__ hist) {
    __shared__ unsigned long long wsum[4];
    __shared__ int found_bin;
    __shared__ unsigned long long found_below, found_cnt;
    constexpr int nb = (PASS == 2) ? 256 : SEL_BINS;
    constexpr int per = nb / 256;                   // bins per thread (16 or 1)
    const unsigned long long rank = in.rank;
    unsigned long long loc[per];
    unsigned long long tsum = 0;
#pragma unroll
    for (int j = 0; j < per; ++j) { loc[j] = hist[threadIdx.x * per + j]; tsum += loc[j]; }
    const unsigned long long incl = wave_scan_incl(tsum);
    __syncthreads();                                // a previous call may still be reading the shared words
    if (lane_id() == 63) wsum[wave_id()] = incl;
    if (threadIdx.x == 0) found_bin = -1;
    __syncthreads();
    unsigned long long before = incl - tsum;
    for (int w = 0; w < wave_id(); ++w) before += wsum[w];
    if (rank >= before && rank < before + tsum) {   // the thread whose bins hold the rank
        unsigned long long b = before;
#pragma unroll
        for (int j = 0; j < per; ++j) {
            if (rank >= b && rank < b + loc[j]) { found_bin = threadIdx.x * per + j; found_below = b; found_cnt = loc[j]; }
            b += loc[j];
        }
    }
    __syncthreads();
    SelState out = in;
    const bool hit = found_bin >= 0;                // (an empty input never gets here)
    const uint32_t bin = hit ? (uint32_t)found_bin : 0u;
    const unsigned long long below = hit ? found_below : 0ull;
    out.less = in.less + below;
    out.rank = rank - below;
    if (PASS == 0) out.prefix = bin;
    else if (PASS == 1) out.prefix = (in.prefix << 12) | bin;
    else {
        out.v0key = (in.prefix << 8) | bin;
        out.need_next = (hit && rank + 1 < found_below + found_cnt) ? 0u : 1u;   // is rank+1 in this final bin too?
    }
    return out;
}

template <int PASS, int NR>
__global__ __launch_bounds__(256) void selx_hist_k(const float* __restrict__ base_in, int64_t n, int64_t stride,
                                                   SelRun* __restrict__ run, uint32_t* __restrict__ hist,
                                                   SelGate gate) {
    const float* base = base_in;
    sel_gate(gate, base, n, stride);
    // the grid is sized for the column the launch names; on the gated source (the bracket's candidates: ~1 % of it) most
    // workgroups have no tile - they leave before the pick and the 16 KB of LDS zeroing (workgroup 0 stays: it records
    // the pick)
    if (blockIdx.x != 0 && (int64_t)blockIdx.x * SEL_TILE >= n) return;
    constexpr int REP = PASS == 0 ? SEL_REP0 : 1;
    constexpr int NH = PASS == 0 ? 1 : NR;
    __shared__ uint32_t hh[NH][REP][SEL_BINS];
    uint32_t prefix[NR];
#pragma unroll
    for (int r = 0; r < NR; ++r) {
        prefix[r] = 0;
        if (PASS >= 1) {
            const SelState st = sel_pick_block<(PASS >= 1 ? PASS - 1 : 0)>(run[r].st[PASS >= 1 ? PASS - 1 : 0],
                                                                             selx_hist_of(hist, r, PASS >= 1 ? PASS - 1 : 0));
            if (blockIdx.x == 0 && threadIdx.x == 0) run[r].st[PASS] = st;
            prefix[r] = st.prefix;
        }
    }
    for (int j = threadIdx.x; j < NH * REP * SEL_BINS; j += 256) (&hh[0][0][0])[j] = 0;
    __syncthreads();
    const int rep = lane_id() & (REP - 1);
    unsigned long long nans = 0;
    auto take = [&](bool in, float v) {
        const uint32_t k = sel_key(v);
        if (PASS == 0) {
            if (in) atomicAdd(&hh[0][rep][k >> 20], 1u);
            nans += (in && v != v);
        } else {
#pragma unroll
            for (int r = 0; r < NR; ++r) {
                if (PASS == 1) sel_hist_add<1>(hh[r < NH ? r : 0][0], in && (k >> 20) == prefix[r], (k >> 8) & 0xFFFu);
                else sel_hist_add<1>(hh[r < NH ? r : 0][0], in && (k >> 8) == prefix[r], k & 0xFFu);
            }
        }
    };
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
    constexpr int nb = (PASS == 2) ? 256 : SEL_BINS;
#pragma unroll
    for (int r = 0; r < NH; ++r) {
        uint32_t* out = selx_hist_of(hist, r, PASS);
        for (int j = threadIdx.x; j < nb; j += 256) {
            uint32_t t = 0;
#pragma unroll
            for (int c = 0; c < REP; ++c) t += hh[r][c][j];
            if (t) atomicAdd(&out[j], t);
        }
    }
    if (PASS == 0) {
        nans = wave_reduce_add(nans);
        if (lane_id() == 0 && nans) atomicAdd(&run[0].nan_count, nans);
    }
}

// smallest key strictly above the selected one (only when rank+1 lies beyond its final bin); the last pick first
__global__ __launch_bounds__(256) void selx_next_k(const float* __restrict__ base_in, int64_t n, int64_t stride,
                                                   SelRun* __restrict__ run, uint32_t* __restrict__ hist, SelGate gate) {
    const float* base = base_in;
    sel_gate(gate, base, n, stride);
    if (blockIdx.x != 0 && (int64_t)blockIdx.x * SEL_TILE >= n) return;      // no tile (see selx_hist_k)
    const SelState st = sel_pick_block<2>(run->st[2], selx_hist_of(hist, 0, 2));
    if (blockIdx.x == 0 && threadIdx.x == 0) run->st[3] = st;
    if (st.need_next == 0) return;
    const uint32_t v0 = st.v0key;
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
    if (lane_id() == 0 && best != 0xFFFFFFFFu) atomicMin(&run->next_min, best);
}

// the final interpolation; picked = 0: the last pick has not run yet (there was no "next" pass)
__global__ __launch_bounds__(256) void selx_lerp_k(const SelRun* __restrict__ run, const uint32_t* __restrict__ hist2,
                                                   int picked, int same_index, float gamma,
                                                   const float* __restrict__ sub, float add1, float add2,
                                                   float* __restrict__ scal) {
    SelState st = run->st[3];
    if (!picked) st = sel_pick_block<2>(run->st[2], hist2);
    if (threadIdx.x != 0) return;
    const float c = sub ? *sub : 0.0f;
    const float a = sel_key_to_float(st.v0key) - c;
    float b = a;
    if (!same_index && st.need_next) b = sel_key_to_float(run->next_min) - c;
    const float diff = b - a;
    float r = a + diff * gamma;
    if (gamma >= 0.5f) r = b - diff * (1.0f - gamma);
    if (run->nan_count) r = __uint_as_float(0x7FC00000u);
    scal[0] = r;
    scal[1] = r + add1;
    scal[2] = r + add2;
}

// ---- bracketed select for large contiguous columns: ONE pass over the data instead of three ------------
// A sample (16 consecutive values out of every 1024) is selected exactly at two ranks around the wanted
// quantile; the keys L <= H found there bracket the wanted order statistics with overwhelming probability.
// One pass over the column counts the keys below L and collects the keys in [L, H] (a few per cent of the
// data); the exact three-pass select then runs on those candidates only, with the rank shifted by the count
// below L.  Whether the bracket really holds both order statistics is CHECKED on the device
// (sel_bracket_fix_k); if not, the same three passes read the whole column instead of the candidates.
// Either way the result is the exact order statistic - the sample only decides how much data is read.
constexpr int64_t SEL_BRACKET_MIN = int64_t(1) << 22;     // below this the three passes are cheap enough
constexpr int SEL_GROUP = 16, SEL_EVERY = 1024;

struct BrState {
    unsigned long long less;       // keys < L
    unsigned long long nan;        // NaN values seen by the bracket pass
    uint32_t count;                // candidates collected (keys in [L, H])
    uint32_t overflow;             // candidate buffer too small
    uint32_t ok;                   // 1: both order statistics lie inside the candidates
    uint32_t pad;
};

// four consecutive values per thread (SEL_GROUP is a multiple of 4, SEL_EVERY of 16: both sides 16-byte aligned when
// the column is); stride 3 reads the z of (n,3) rows (base = rows + 2)
__global__ void sel_sample_k(const float* __restrict__ base, int64_t stride, int64_t ns, float* __restrict__ sample) {
    const int64_t i = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) * 4;
    if (i >= ns) return;
    const int64_t src = (i / SEL_GROUP) * SEL_EVERY + (i % SEL_GROUP);
    float4 q;
    if (stride == 1 && (reinterpret_cast<uintptr_t>(base) & 15u) == 0) {
        q = *reinterpret_cast<const float4*>(base + src);
    } else {
        q.x = base[src * stride]; q.y = base[(src + 1) * stride];
        q.z = base[(src + 2) * stride]; q.w = base[(src + 3) * stride];
    }
    *reinterpret_cast<float4*>(sample + i) = q;
}
static_assert(SEL_GROUP % 4 == 0 && SEL_EVERY % 4 == 0, "sel_sample_k");

constexpr int SEL_STAGE = 2 * SEL_TILE;                  // candidates a workgroup stages in LDS before it reserves output

__global__ __launch_bounds__(256) void sel_bracket_k(const float* __restrict__ base, int64_t n,
                                                     const SelRun* __restrict__ lohi, uint32_t* __restrict__ hist,
                                                     int lo_is_min, int hi_is_max,
                                                     BrState* __restrict__ br, float* __restrict__ cand, uint32_t cap) {
    // candidates are staged in LDS and written out a few thousand at a time: one global atomic per flush (a
    // reservation per wave would be ~1.5 M atomics on one address for 100 M values - 13 ms, measured)
    __shared__ float stage[SEL_STAGE];
    __shared__ uint32_t nstage, gbase;
    // the last picks of the two sample selects; a bracket that reaches the bottom / top of the sample is open there
    const uint32_t Lk = sel_pick_block<2>(lohi[0].st[2], selx_hist_of(hist, 0, 2)).v0key;
    const uint32_t Hk = sel_pick_block<2>(lohi[1].st[2], selx_hist_of(hist, 1, 2)).v0key;
    const uint32_t L = lo_is_min ? 0u : Lk, H = hi_is_max ? 0xFFFFFFFFu : Hk;
    unsigned long long less = 0, nans = 0;
    const uint64_t lt = lanemask_lt();
    if (threadIdx.x == 0) nstage = 0;
    __syncthreads();
    auto flush = [&]() {                                   // workgroup-uniform call
        const uint32_t m = nstage;
        if (threadIdx.x == 0) gbase = m ? atomicAdd(&br->count, m) : 0u;
        __syncthreads();
        const uint32_t g0 = gbase;
        for (uint32_t i = threadIdx.x; i < m; i += 256) {
            if (g0 + i < cap) cand[g0 + i] = stage[i]; else br->overflow = 1u;
        }
        __syncthreads();
        if (threadIdx.x == 0) nstage = 0;
        __syncthreads();
    };
    auto take = [&](bool in, float v) {
        const uint32_t k = sel_key(v);
        less += (in && k < L) ? 1u : 0u;
        nans += (in && v != v) ? 1u : 0u;
        const bool keep = in && k >= L && k <= H;
        const unsigned long long m = __ballot(keep);
        if (m) {
            const int leader = (int)__builtin_ctzll(m);
            uint32_t at = 0;
            if (lane_id() == leader) at = atomicAdd(&nstage, (uint32_t)__popcll(m));      // LDS atomic
            at = (uint32_t)__builtin_amdgcn_readlane((int)at, leader);
            if (keep) stage[at + (uint32_t)__popcll(m & lt)] = v;                          // < SEL_STAGE: see the flush rule
        }
    };
    const int64_t span = (int64_t)gridDim.x * SEL_TILE;
    const bool vec = (reinterpret_cast<uintptr_t>(base) & 15u) == 0;
    for (int64_t t0 = (int64_t)blockIdx.x * SEL_TILE; t0 < n; t0 += span) {    // workgroup-uniform trip count
        __syncthreads();
        if (nstage > (uint32_t)(SEL_STAGE - SEL_TILE)) flush();                  // room for a whole tile of candidates
        if (vec && t0 + SEL_TILE <= n) {
            const float4* b4 = reinterpret_cast<const float4*>(base + t0);
            float4 q[4];
#pragma unroll
            for (int r = 0; r < 4; ++r) q[r] = b4[r * 256 + threadIdx.x];
#pragma unroll
            for (int r = 0; r < 4; ++r) { take(true, q[r].x); take(true, q[r].y); take(true, q[r].z); take(true, q[r].w); }
        } else {
            for (int r = 0; r < SEL_TILE / 256; ++r) {
                const int64_t i = t0 + r * 256 + threadIdx.x;
                const bool in = i < n;
                take(in, in ? base[i] : 0.0f);
            }
        }
    }
    __syncthreads();
    flush();
    less = wave_reduce_add(less);
    nans = wave_reduce_add(nans);
    if (lane_id() == 0) {
        if (less) atomicAdd(&br->less, less);
        if (nans) atomicAdd(&br->nan, nans);
    }
}

// decides whether the candidates hold both order statistics and prepares the state of whichever select runs next
__global__ void sel_bracket_fix_k(BrState* __restrict__ br, SelRun* __restrict__ run, unsigned long long k0, int same) {
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    const unsigned long long less = br->less, cnt = br->count;
    const bool ok = br->overflow == 0 && less <= k0 && (k0 - less) < cnt && (same || (k0 + 1 - less) < cnt);
    br->ok = ok ? 1u : 0u;
    SelState z;
    z.rank = ok ? k0 - less : k0;
    z.less = 0; z.prefix = 0; z.need_next = 0; z.v0key = 0; z.v1key = 0;
    z.nan_count = 0; z.next_min = 0; z.pad = 0;
    run->st[0] = z;
    run->nan_count = ok ? br->nan : 0ull;                  // (the full passes count NaN themselves)
}

// one launch in front of a select: clears the histograms (and the bracket state), sets the ranks of up to 3 runs
__global__ __launch_bounds__(256) void selx_init_k(uint32_t* __restrict__ hist, int nwords, SelRun* __restrict__ runs,
                                                   int nruns, unsigned long long r0, unsigned long long r1,
                                                   unsigned long long r2, BrState* __restrict__ br,
                                                   uint32_t* __restrict__ also_zero, int64_t also_words) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i < nwords) hist[i] = 0;
    // words of the caller that must be zero before a LATER kernel of the same call (the fused filter's sweep state and
    // look-back words: a fill of their own sat on the critical path between the interpolation and the sweep)
    for (int64_t j = i; j < also_words; j += (int64_t)gridDim.x * 256) also_zero[j] = 0u;
    if (i < nruns) {
        SelRun z;
        memset(&z, 0, sizeof(z));
        z.st[0].rank = i == 0 ? r0 : (i == 1 ? r1 : r2);
        z.next_min = 0xFFFFFFFFu;
        runs[i] = z;
    }
    if (i == 0 && br) { BrState b; memset(&b, 0, sizeof(b)); *br = b; }
}

struct SelWs {
    SelState* st;
    uint32_t* hist;
    float*    scal;     // 4 floats
    // bracketed select (n >= SEL_BRACKET_MIN)
    SelRun*   run;      // [0], [1]: the two ends of the sample bracket, [2]: the select proper
    uint32_t* xhist;    // [3 runs][3 passes][SEL_BINS]
    BrState*  br;
    float    *sample, *cand;
    int64_t   ns;       // sample size
    uint32_t  cap;      // candidate capacity
    uint32_t* also_zero;   // optional: words sel_init clears for the caller (see selx_init_k)
    int64_t   also_words;
};
static void sel_plan(Arena& a, SelWs& w, int64_t n = 0) {
    w.st = a.take<SelState>(1);
    w.hist = a.take<uint32_t>(SEL_BINS);
    w.scal = a.take<float>(4);
    w.run = a.take<SelRun>(3);
    w.xhist = a.take<uint32_t>(9 * SEL_BINS);
    w.br = nullptr; w.sample = w.cand = nullptr; w.ns = 0; w.cap = 0;
    w.also_zero = nullptr; w.also_words = 0;
    if (n >= SEL_BRACKET_MIN) {
        w.br = a.take<BrState>(1);
        w.ns = SEL_GROUP * (n / SEL_EVERY);
        w.sample = a.take<float>(w.ns);
        w.cap = (uint32_t)(n / 16) + 1024u;
        w.cand = a.take<float>(w.cap);
    }
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

// three histogram passes (+ the "next key" pass) on one SelRun; run and histograms are prepared by the caller
static int select_rounds(const float* base, int64_t n, int64_t stride, bool with_next, SelRun* run, uint32_t* hist,
                         hipStream_t s, SelGate gate, int64_t grid_n) {
    int64_t gb = ceil_div(grid_n, SEL_TILE);
    if (gb > 2048) gb = 2048;
    if (gb < 1) gb = 1;
    const dim3 grid((unsigned)gb), blk(256);
    PCH_LAUNCH("sel_hist0", (selx_hist_k<0, 1>), grid, blk, 0, s, base, n, stride, run, hist, gate);
    PCH_LAUNCH("sel_hist1", (selx_hist_k<1, 1>), grid, blk, 0, s, base, n, stride, run, hist, gate);
    PCH_LAUNCH("sel_hist2", (selx_hist_k<2, 1>), grid, blk, 0, s, base, n, stride, run, hist, gate);
    if (with_next) PCH_LAUNCH("sel_next", selx_next_k, grid, blk, 0, s, base, n, stride, run, hist, gate);
    return PCH_OK;
}

// Bracketed select (columns of >= SEL_BRACKET_MIN values with a sample buffer planned): is it?
static bool select_is_bracketed(const SelWs& w, int64_t n, int64_t stride) {
    return w.sample && stride == 1 && n >= SEL_BRACKET_MIN;
}
struct SelBracket { int64_t r_lo, r_hi; };
static SelBracket select_bracket_ranks(const SelWs& w, int64_t n, const PctIndex& pi) {
    // bracket from the sample: ranks around k0 * ns / n, six sigma for ns/16 independent draws + 0.2 %
    const int64_t ns = w.ns;
    const double p = (double)pi.k0 / (double)(n > 1 ? n - 1 : 1);
    const double sigma = sqrt(p * (1.0 - p) * (double)ns * 16.0);
    const int64_t margin = (int64_t)(6.0 * sigma) + ns / 500 + 16;
    const int64_t mid = (int64_t)(p * (double)(ns - 1));
    SelBracket r;
    r.r_lo = mid - margin < 0 ? 0 : mid - margin;
    r.r_hi = mid + margin > ns - 1 ? ns - 1 : mid + margin;
    return r;
}
// First half of the bracketed select: the sample and the two order statistics of it that bracket the ranks.  It
// needs the VALUES only (src / src_stride: the column itself, or the z of the (n,3) rows it will be copied from), so the
// fused filter runs it beside the centroid summary, before the column exists.
static int select_sample_passes(const float* src, int64_t src_stride, int64_t n, double q_percent, SelWs& w,
                                hipStream_t s) {
    const PctIndex pi = pct_index(n, q_percent);
    const SelGate always = {nullptr, nullptr, nullptr};
    const SelBracket br = select_bracket_ranks(w, n, pi);
    const int64_t ns = w.ns;
    PCH_LAUNCH("sel_init", selx_init_k, dim3((unsigned)ceil_div(9 * SEL_BINS, 256)), dim3(256), 0, s, w.xhist,
               9 * SEL_BINS, w.run, 3, (unsigned long long)br.r_lo, (unsigned long long)br.r_hi, 0ull, w.br,
               w.also_zero, w.also_words);
    PCH_LAUNCH("sel_sample", sel_sample_k, dim3((unsigned)ceil_div(ns, 1024)), dim3(256), 0, s, src, src_stride, ns,
               w.sample);
    int64_t gs = ceil_div(ns, SEL_TILE);                   // both ends of the bracket in the same three sweeps
    if (gs > 2048) gs = 2048;
    if (gs < 1) gs = 1;
    const dim3 grid((unsigned)gs), blk(256);
    PCH_LAUNCH("sel_hist0", (selx_hist_k<0, 2>), grid, blk, 0, s, (const float*)w.sample, ns, (int64_t)1, w.run, w.xhist, always);
    PCH_LAUNCH("sel_hist1", (selx_hist_k<1, 2>), grid, blk, 0, s, (const float*)w.sample, ns, (int64_t)1, w.run, w.xhist, always);
    PCH_LAUNCH("sel_hist2", (selx_hist_k<2, 2>), grid, blk, 0, s, (const float*)w.sample, ns, (int64_t)1, w.run, w.xhist, always);
    return PCH_OK;
}

// the histogram / next passes (need only the raw values) ...
// sample_done: select_sample_passes already ran for this column (and is ordered in front of `s`)
static int select_passes(const float* base, int64_t n, int64_t stride, double q_percent, SelWs& w,
                         hipStream_t s, bool sample_done = false) {
    const PctIndex pi = pct_index(n, q_percent);
    const SelGate always = {nullptr, nullptr, nullptr};
    SelRun* fin = w.run + 2;
    uint32_t* hfin = w.xhist + 6 * SEL_BINS;
    if (!select_is_bracketed(w, n, stride)) {
        PCH_LAUNCH("sel_init", selx_init_k, dim3((unsigned)ceil_div(9 * SEL_BINS, 256)), dim3(256), 0, s, w.xhist,
                   9 * SEL_BINS, w.run, 3, 0ull, 0ull, (unsigned long long)pi.k0, (BrState*)nullptr, w.also_zero,
                   w.also_words);
        return select_rounds(base, n, stride, !pi.same, fin, hfin, s, always, n);
    }
    if (!sample_done) PCH_TRY(select_sample_passes(base, 1, n, q_percent, w, s));
    const SelBracket br = select_bracket_ranks(w, n, pi);
    const int64_t ns = w.ns;
    // one resident round of workgroups: 32 KB of LDS each, five per CU (a 2048-workgroup grid ran as 1280 + 768:
    // 0.191 -> 0.177 ms).  Tried and dropped (round 3): one ballot per four values and 32-bit counters (no change: the
    // sweep is not bound by its arithmetic); per-wave staging without barriers + register prefetch (0.194 ms);
    // 2 KB of staging per wave for 8 workgroups per CU (0.266 ms: more waves streaming shorter runs each)
    int64_t gb = ceil_div(n, SEL_TILE);
    {
        static int per_round[PCH_MAX_DEVICES] = {};
        const int slot = current_device_slot();
        if (!per_round[slot]) {
            int dev = 0, cus = 256;
            if (hipGetDevice(&dev) == hipSuccess) (void)hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev);
            per_round[slot] = cus * 5;
        }
        if (gb > per_round[slot]) gb = per_round[slot];
    }
    PCH_LAUNCH("sel_bracket", sel_bracket_k, dim3((unsigned)gb), dim3(256), 0, s, base, n, (const SelRun*)w.run, w.xhist,
               br.r_lo == 0 ? 1 : 0, br.r_hi == ns - 1 ? 1 : 0, w.br, w.cand, w.cap);
    PCH_LAUNCH("sel_bracket_fix", sel_bracket_fix_k, dim3(1), dim3(64), 0, s, w.br, fin, (unsigned long long)pi.k0,
               pi.same);
    // exact select: on the candidates when the bracket holds, else over the whole column - the same launches either
    // way, the kernels pick their source from br->ok
    const SelGate src = {&w.br->ok, w.cand, &w.br->count};
    return select_rounds(base, n, stride, !pi.same, fin, hfin, s, src, n);
}
// ... and the final interpolation, which is where `sub` (the centroid) enters
static int select_lerp(int64_t n, const float* sub, double q_percent, float add1, float add2, SelWs& w,
                       hipStream_t s) {
    const PctIndex pi = pct_index(n, q_percent);
    PCH_LAUNCH("sel_lerp", selx_lerp_k, dim3(1), dim3(256), 0, s, (const SelRun*)(w.run + 2),
               (const uint32_t*)(w.xhist + 8 * SEL_BINS), pi.same ? 0 : 1, pi.same, pi.gamma, sub, add1, add2, w.scal);
    return PCH_OK;
}
static int select_percentile(const float* base, int64_t n, int64_t stride, const float* sub,
                             double q_percent, float add1, float add2, SelWs& w, hipStream_t s) {
    PCH_TRY(select_passes(base, n, stride, q_percent, w, s));
    return select_lerp(n, sub, q_percent, add1, add2, w, s);
}

// per-thread side stream + events: the select passes only need the raw z values, so they run
// beside the centroid passes (the sample half beside the summary, the sweep beside the 3-wave walk)
struct SideStream { hipStream_t s; hipEvent_t ev_fork, ev_join, ev_start; bool ok; };
struct SideStreams {                                    // one per device this thread has used; gone with the thread
    SideStream ss[PCH_MAX_DEVICES];
    SideStreams() { for (auto& x : ss) x = {nullptr, nullptr, nullptr, nullptr, false}; }
    ~SideStreams() {
        if (!may_release_hip_objects()) return;
        for (auto& x : ss)
            if (x.ok) {
                (void)hipEventDestroy(x.ev_fork); (void)hipEventDestroy(x.ev_join);
                (void)hipEventDestroy(x.ev_start);
                (void)hipStreamDestroy(x.s);
            }
    }
};
static SideStream& side_stream() {
    static thread_local SideStreams all;
    SideStream& ss = all.ss[current_device_slot()];
    if (!ss.ok) {
        if (hipStreamCreateWithFlags(&ss.s, hipStreamNonBlocking) == hipSuccess &&
            hipEventCreateWithFlags(&ss.ev_fork, hipEventDisableTiming) == hipSuccess &&
            hipEventCreateWithFlags(&ss.ev_join, hipEventDisableTiming) == hipSuccess &&
            hipEventCreateWithFlags(&ss.ev_start, hipEventDisableTiming) == hipSuccess)
            ss.ok = true;
    }
    return ss;
}

// =====================================================================================
// B2/B3: keep = (z - cz) > thr, order preserving, in ONE sweep over the z column: every
// workgroup counts its tile, publishes the count and obtains its output offset by a decoupled
// look-back over the tiles in front of it (status word = 2-bit flag + 32-bit value, one 64-bit
// relaxed atomic), then writes the centred survivors and folds their bounding box into one of
// 64 slot sets.  The sweep runs with the first threshold (offset); only if that keeps fewer than
// min_keep points (utils/tower_extraction.py:87-89) a second sweep with the fallback threshold
// overwrites the output - its workgroups return at once otherwise (the last tile of the first sweep
// sets the switch).
// =====================================================================================
constexpr int GF_THREADS = 256;
constexpr int GF_ROUNDS  = 64;
constexpr int GF_TILE    = GF_THREADS * GF_ROUNDS;   // 16 384 points per workgroup: few enough tiles for the look-back
constexpr int GF_SLOTS   = 64;

struct GfState {
    uint32_t total[2];                     // kept with threshold A (offset) / B (fallback)
    uint32_t use_b;
    uint32_t failed;                       // a look-back wait ran out of its budget (pch_lookback.h): count -> -1
    uint32_t ticket[2];                    // next logical tile of either sweep (order of arrival, see gf_compact_k)
    uint32_t pad2[2];
    uint32_t slots[2][GF_SLOTS][6];        // per sweep: ~min xyz (complemented) / max xyz as ordered uint32,
                                           // all folded with atomicMax so that 0 is the neutral start
};

// ---- the candidate rows of the summary pass (MsCand, pch_mean.h) -------------------------------------------------
// May the sweep read the candidate slots instead of the tile?  Only if no survivor can be missing from them.  A row
// survives iff fl(z - cz) > thr; the slots hold every row with z > tcand.  If tcand <= thr + cz (as real numbers: the
// double sum is exact while the two exponents are within 29 bits, and the test takes one step below it for when they
// are not), a row with z <= tcand has z - cz <= thr, and rounding is monotone with fl(thr) = thr, so
// fl(z - cz) <= thr: it does not survive.  The same test, on the same device words, decides in gf_cand_k (go) and in
// gf_compact_k (stand down), for both thresholds at once; a NaN anywhere fails it.
__device__ __forceinline__ bool gf_cand_ok(const float* __restrict__ tcand, const float* __restrict__ centroid,
                                           const float* __restrict__ scal) {
    if (!tcand) return false;
    if (reinterpret_cast<const uint32_t*>(tcand)[1] != 0u) return false;      // a slot overflowed (pch_mean.h)
    const double cz = (double)centroid[2];
    const double ta = (double)scal[1] + cz, tb = (double)scal[2] + cz;
    const double t = (double)*tcand;
    // the two sums are exact in double unless the exponents of threshold and centroid lie more than 29 bits apart;
    // one step below the (then rounded) sum is at or below the real sum either way
    return t <= nextafter(ta, -INFINITY) && t <= nextafter(tb, -INFINITY);
}

template <int WHICH>
__global__ __launch_bounds__(GF_THREADS) void gf_compact_k(
    const float* __restrict__ raw, const float* __restrict__ zcol, int64_t n,
    const float* __restrict__ centroid, const float* __restrict__ scal, GfState* __restrict__ st,
    uint64_t* __restrict__ status, long long min_keep, float* __restrict__ out_points,
    int32_t* __restrict__ out_index, const float* __restrict__ tcand) {
    if (gf_cand_ok(tcand, centroid, scal)) return;         // gf_cand_k does this sweep from the candidate slots
    __shared__ uint32_t wtot[GF_THREADS / 64];
    __shared__ uint32_t excl_sh;
    __shared__ uint32_t box[GF_THREADS / 64][6];
    __shared__ unsigned long long masks[GF_THREADS / 64][GF_ROUNDS];      // survivors of every 64-point round
    __shared__ uint32_t cum[GF_THREADS / 64][GF_ROUNDS + 1];
    __shared__ uint32_t tile_sh;
    if (WHICH == 1 && st->use_b == 0) return;
    // The look-back below waits for every tile in front of this one, so the tile order must be an
    // order in which workgroups START: HIP promises nothing about blockIdx dispatch order, and a
    // resident workgroup spinning on a tile that was never scheduled would hang the GPU.  The logical
    // tile is therefore a ticket drawn on arrival (as rocPRIM's look-back scan does): every tile with
    // a smaller ticket belongs to a workgroup that is already running.
    if (threadIdx.x == 0) tile_sh = atomicAdd(&st->ticket[WHICH], 1u);
    __syncthreads();
    const int64_t tile = tile_sh;
    const float cx = centroid[0], cy = centroid[1], cz = centroid[2];
    const float thr = scal[1 + WHICH];
    const int w = wave_id(), l = lane_id();
    const int64_t seg = tile * GF_TILE + (int64_t)w * (64 * GF_ROUNDS);
    // ---- sweep over the z column: survivor masks and the tile's count
    uint32_t run = 0;
#pragma unroll 8
    for (int r = 0; r < GF_ROUNDS; ++r) {
        const int64_t i = seg + r * 64 + l;
        const float z = i < n ? (zcol ? zcol[i] : raw[3 * i + 2]) : 0.0f;     // no column given: the z of the rows
        const bool keep = i < n && (z - cz) > thr;       // points = raw_points - centroid (float32)
        const unsigned long long m = __ballot(keep);
        if (l == 0) masks[w][r] = m;
        run += (uint32_t)__popcll(m);
    }
    if (l == 0) wtot[w] = run;
    {   // survivors in the rounds in front of round l (cum[w][64] = the wave's total)
        __builtin_amdgcn_wave_barrier();
        const uint32_t c = (uint32_t)__popcll(masks[w][l]);
        const uint32_t incl = wave_scan_incl(c);
        cum[w][l] = incl - c;
        if (l == 63) cum[w][64] = incl;
    }
    __syncthreads();
    const uint32_t T = wtot[0] + wtot[1] + wtot[2] + wtot[3];
    if (w == 0) {
        const uint32_t e0 = gf_lookback(status, tile, T);
        const bool lb_failed = e0 == GF_LB_FAILED;
        const uint32_t e = lb_failed ? 0u : e0;              // prefix 0 keeps the writes below inside the output
        if (l == 0) {
            excl_sh = e;
            if (lb_failed) {
                atomicOr(&st->failed, 1u);
            } else if (tile == (int64_t)gridDim.x - 1) {         // the last tile knows the total: it also decides
                st->total[WHICH] = e + T;                     // whether the fallback threshold applies
                if (WHICH == 0) st->use_b = ((long long)(e + T) < min_keep) ? 1u : 0u;
            }
        }
    }
    __syncthreads();
    if (T == 0) return;                                       // workgroup-uniform
    // ---- survivors: the row (one 12-byte load) is only fetched for them.  The wave's survivors are taken
    // DENSELY - lane j of pass d handles survivor 64 d + j of the wave, found through the per-round counts
    // and the position of its bit in the round's mask - so that every lane has a row in flight and
    // consecutive lanes write consecutive output rows (a pass over the 64 rounds with the ~10 % surviving
    // lanes active would leave most lanes idle and scatter the writes)
    struct Row3 { float x, y, z; };
    uint32_t woff = excl_sh;
    for (int w2 = 0; w2 < w; ++w2) woff += wtot[w2];
    uint32_t lo[3] = {0u, 0u, 0u}, hi[3] = {0u, 0u, 0u};      // lo holds ~ordered(min)
    const uint32_t nw = wtot[w];
#pragma unroll 2
    for (uint32_t j = l; j < nw; j += 64) {
        int r = 0;                                             // largest round with cum[r] <= j
#pragma unroll
        for (int step = 32; step > 0; step >>= 1)
            if (cum[w][r + step] <= j) r += step;
        unsigned long long m = masks[w][r];
        uint32_t k = j - cum[w][r];                            // the k-th set bit of that round's mask
        int bit = 0;
        uint32_t word = (uint32_t)m;
        {
            const uint32_t c = (uint32_t)__popc(word);
            if (k >= c) { k -= c; word = (uint32_t)(m >> 32); bit = 32; }
        }
#pragma unroll
        for (int half = 16; half > 0; half >>= 1) {
            const uint32_t c = (uint32_t)__popc(word & ((1u << half) - 1u));
            if (k >= c) { k -= c; word >>= half; bit += half; }
        }
        const int64_t i = seg + r * 64 + bit;
        const Row3 q = reinterpret_cast<const Row3*>(raw)[i];
        const float v[3] = {q.x - cx, q.y - cy, q.z - cz};
        const int64_t o = (int64_t)woff + j;
        out_points[3 * o + 0] = v[0];
        out_points[3 * o + 1] = v[1];
        out_points[3 * o + 2] = v[2];
        if (out_index) out_index[o] = (int32_t)i;
        if (fabsf(v[0]) < INFINITY && fabsf(v[1]) < INFINITY && fabsf(v[2]) < INFINITY) {   // NaN/inf rows
#pragma unroll
            for (int a = 0; a < 3; ++a) {
                const uint32_t kk = f32_ordered(v[a]);
                lo[a] = ~kk > lo[a] ? ~kk : lo[a];
                hi[a] = kk > hi[a] ? kk : hi[a];
            }
        }
    }
#pragma unroll
    for (int a = 0; a < 3; ++a) {
        lo[a] = wave_reduce_max(lo[a]);
        hi[a] = wave_reduce_max(hi[a]);
        if (l == 0) { box[w][a] = lo[a]; box[w][3 + a] = hi[a]; }
    }
    __syncthreads();
    if (threadIdx.x < 6) {
        const int a = threadIdx.x;
        uint32_t v = box[0][a];
        for (int w2 = 1; w2 < GF_THREADS / 64; ++w2) v = box[w2][a] > v ? box[w2][a] : v;
        if (v) atomicMax(&st->slots[WHICH][tile % GF_SLOTS][a], v);
    }
}

// The same sweep from the candidate slots (MsCand, pch_mean.h: four planes per 1024-row block): a workgroup takes 64
// slots, a wave 16 of them, round-major.  ~12 % of the rows are candidates and they lie contiguously, so the sweep
// reads 4 B per candidate to count and 16 B per candidate for the rows, instead of the z column plus one memory
// line per survivor.
constexpr int GF_CBLK = 1024;                          // rows per block of the summary
constexpr int GF_CSLOT = MS_CAND_SLOT;                 // candidate rows a slot holds
constexpr int GF_CT_BLKS = 64;                         // slots per workgroup: 65 536 rows of the tile (32 measured no
                                                       // faster; the ticket word hands out ~88 tiles per microsecond)
constexpr int GF_CW_BLKS = GF_CT_BLKS / (GF_THREADS / 64);     // 16 slots per wave

template <int WHICH>
__global__ __launch_bounds__(GF_THREADS, 6) void gf_cand_k(     // six waves per SIMD: every tile of a 100 M-point call resident
    const float* __restrict__ slots, const uint32_t* __restrict__ counts, const float* __restrict__ tcand,
    int64_t nblk, const float* __restrict__ centroid, const float* __restrict__ scal, GfState* __restrict__ st,
    uint64_t* __restrict__ status, long long min_keep, float* __restrict__ out_points,
    int32_t* __restrict__ out_index) {
    constexpr int NR = GF_CSLOT / 64;                      // rounds of 64 rows in a slot
    constexpr int64_t SLOT_WORDS = 4 * GF_CSLOT;           // planes x | y | z | row-in-block (MsCand, pch_mean.h)
    constexpr int NE = GF_CW_BLKS * NR;                    // (slot, round) pairs of a wave
    __shared__ uint32_t wtot[GF_THREADS / 64];
    __shared__ uint32_t excl_sh;
    __shared__ uint32_t box[GF_THREADS / 64][6];
    __shared__ unsigned long long masks[GF_THREADS / 64][NE];        // survivors of every (slot, round): 4 KB
    __shared__ uint32_t cum[GF_THREADS / 64][NE];                     // survivors of the wave in front of that pair
    __shared__ uint32_t tile_sh;
    static_assert(NE == 128, "two (slot, round) pairs per lane in the prefix below");
    if (!gf_cand_ok(tcand, centroid, scal)) return;        // gf_compact_k runs instead
    if (WHICH == 1 && st->use_b == 0) return;
#ifdef PCH_GF_STAMPS
    const unsigned long long ts0 = wall_clock64();
    unsigned long long ts1 = 0, ts2 = 0, ts3 = 0;
#endif
    if (threadIdx.x == 0) tile_sh = atomicAdd(&st->ticket[WHICH], 1u);     // tile order = order of arrival (look-back)
    const int w = wave_id(), l = lane_id();
    masks[w][2 * l] = 0ull;
    masks[w][2 * l + 1] = 0ull;
    __syncthreads();
    const int64_t tile = tile_sh;
    const float cx = centroid[0], cy = centroid[1], cz = centroid[2];
    const float thr = scal[1 + WHICH];
    const int64_t b0 = tile * GF_CT_BLKS + (int64_t)w * GF_CW_BLKS;
    // Candidates come in runs: blocks next to a tower hold hundreds, most blocks fewer than 64 or none.  A slot after
    // the other is a chain of dependent memory round trips (the kernel waited 86 % of its wave cycles that way), so
    // the wave goes ROUND-major: round r of all its 16 slots at once - 16 independent loads in flight, of the z PLANE
    // alone in the counting phase (a quarter of the slot's bytes) - and usually one or two rounds are all there is.
    uint32_t ncs[GF_CW_BLKS];
    uint32_t maxnc = 0;
#pragma unroll
    for (int kb = 0; kb < GF_CW_BLKS; ++kb) {
        ncs[kb] = (b0 + kb) < nblk ? counts[b0 + kb] : 0u;
        maxnc = ncs[kb] > maxnc ? ncs[kb] : maxnc;
    }
    maxnc = (uint32_t)__builtin_amdgcn_readfirstlane((int)maxnc);      // (every lane read the same words)
    uint32_t run = 0;
#pragma unroll 1
    for (int r = 0; r < NR; ++r) {
        if ((uint32_t)(r * 64) >= maxnc) break;            // wave-uniform
        const uint32_t i = r * 64 + l;
        float z[GF_CW_BLKS];
#pragma unroll
        for (int kb = 0; kb < GF_CW_BLKS; ++kb) {
            z[kb] = 0.0f;
            if (i < ncs[kb]) z[kb] = slots[(b0 + kb) * SLOT_WORDS + 2 * GF_CSLOT + i];
        }
#pragma unroll
        for (int kb = 0; kb < GF_CW_BLKS; ++kb) {
            const bool keep = i < ncs[kb] && (z[kb] - cz) > thr;        // points = raw_points - centroid (float32)
            const unsigned long long m = __ballot(keep);
            if (l == 0 && m) masks[w][kb * NR + r] = m;
            run += (uint32_t)__popcll(m);
        }
    }
    if (l == 0) wtot[w] = run;
    __syncthreads();
#ifdef PCH_GF_STAMPS
    ts1 = wall_clock64();
#endif
    const uint32_t T = wtot[0] + wtot[1] + wtot[2] + wtot[3];
    // eight slots at a time: with all sixteen rows in registers the kernel needs 130 VGPRs, and at three waves per
    // SIMD only half of the tiles are resident - two rounds of workgroups that each wait on the look-back
    constexpr int HB = GF_CW_BLKS / 2;
    const uint64_t lt = lanemask_lt();
    uint32_t lo[3] = {0u, 0u, 0u}, hi[3] = {0u, 0u, 0u};  // lo holds ~ordered(min)
    auto load_half = [&](int r, int half, float4 (&q)[HB], unsigned long long (&ms)[HB]) {
        const uint32_t i = r * 64 + l;
#pragma unroll
        for (int k = 0; k < HB; ++k) {                     // the four planes, survivors only
            const int kb = half * HB + k;
            ms[k] = masks[w][kb * NR + r];
            q[k].x = q[k].y = q[k].z = q[k].w = 0.0f;
            if ((ms[k] >> l) & 1ull) {
                const float* __restrict__ sp = slots + (b0 + kb) * SLOT_WORDS + i;
                q[k].x = sp[0]; q[k].y = sp[GF_CSLOT]; q[k].z = sp[2 * GF_CSLOT]; q[k].w = sp[3 * GF_CSLOT];
            }
        }
    };
    auto emit_half = [&](int r, int half, const float4 (&q)[HB], const unsigned long long (&ms)[HB], uint32_t woff) {
#pragma unroll
        for (int k = 0; k < HB; ++k) {
            const int kb = half * HB + k;
            const unsigned long long m = ms[k];
            if ((m >> l) & 1ull) {
                const float4 c4 = q[k];
                const float v[3] = {c4.x - cx, c4.y - cy, c4.z - cz};
                const int64_t o = (int64_t)woff + cum[w][kb * NR + r] + (uint32_t)__popcll(m & lt);
                out_points[3 * o + 0] = v[0];
                out_points[3 * o + 1] = v[1];
                out_points[3 * o + 2] = v[2];
                if (out_index) out_index[o] = (int32_t)((b0 + kb) * GF_CBLK + (int64_t)__float_as_uint(c4.w));
                if (fabsf(v[0]) < INFINITY && fabsf(v[1]) < INFINITY && fabsf(v[2]) < INFINITY) {   // NaN/inf rows
#pragma unroll
                    for (int a = 0; a < 3; ++a) {
                        const uint32_t kk = f32_ordered(v[a]);
                        lo[a] = ~kk > lo[a] ? ~kk : lo[a];
                        hi[a] = kk > hi[a] ? kk : hi[a];
                    }
                }
            }
        }
    };
    // the rows of the first half-batch are requested BEFORE the look-back: the loads need no output offset (only the
    // stores do), and a tile waits ~30 us for the tiles in front of it with nothing of its own in flight otherwise
    float4 q0[HB];
    unsigned long long ms0[HB];
    if (T != 0 && maxnc != 0) load_half(0, 0, q0, ms0);
    if (w == 0) {
        const uint32_t e0 = gf_lookback(status, tile, T);
        const bool lb_failed = e0 == GF_LB_FAILED;
        const uint32_t e = lb_failed ? 0u : e0;              // prefix 0 keeps the writes below inside the output
        if (l == 0) {
            excl_sh = e;
            if (lb_failed) {
                atomicOr(&st->failed, 1u);
            } else if (tile == (int64_t)gridDim.x - 1) {     // the last tile knows the total: it also decides
                st->total[WHICH] = e + T;                    // whether the fallback threshold applies
                if (WHICH == 0) st->use_b = ((long long)(e + T) < min_keep) ? 1u : 0u;
            }
        }
#ifdef PCH_GF_STAMPS
        ts2 = wall_clock64();
#endif
    }
    {   // survivors of the wave in front of every (slot, round) pair, in file order = slot-major
        const uint32_t c0 = (uint32_t)__popcll(masks[w][2 * l]), c1 = (uint32_t)__popcll(masks[w][2 * l + 1]);
        const uint32_t incl = wave_scan_incl(c0 + c1);
        cum[w][2 * l] = incl - c0 - c1;
        cum[w][2 * l + 1] = incl - c1;
    }
    __syncthreads();
    if (T == 0) return;                                    // workgroup-uniform
    uint32_t woff = excl_sh;
    for (int w2 = 0; w2 < w; ++w2) woff += wtot[w2];
    if (maxnc != 0) {
        emit_half(0, 0, q0, ms0, woff);
        {
            float4 q[HB];
            unsigned long long ms[HB];
            load_half(0, 1, q, ms);
            emit_half(0, 1, q, ms, woff);
        }
#pragma unroll 1
        for (int r = 1; r < NR; ++r) {
            if ((uint32_t)(r * 64) >= maxnc) break;
#pragma unroll
            for (int half = 0; half < 2; ++half) {
                float4 q[HB];
                unsigned long long ms[HB];
                load_half(r, half, q, ms);
                emit_half(r, half, q, ms, woff);
            }
        }
    }
#pragma unroll
    for (int a = 0; a < 3; ++a) {
        lo[a] = wave_reduce_max(lo[a]);
        hi[a] = wave_reduce_max(hi[a]);
        if (l == 0) { box[w][a] = lo[a]; box[w][3 + a] = hi[a]; }
    }
    __syncthreads();
    if (threadIdx.x < 6) {
        const int a = threadIdx.x;
        uint32_t v = box[0][a];
        for (int w2 = 1; w2 < GF_THREADS / 64; ++w2) v = box[w2][a] > v ? box[w2][a] : v;
        if (v) atomicMax(&st->slots[WHICH][tile % GF_SLOTS][a], v);
    }
#ifdef PCH_GF_STAMPS
    ts3 = wall_clock64();
    if (WHICH == 0 && threadIdx.x == 0 && (tile % 190 == 0 || tile == gridDim.x - 1))
        printf("gf_cand tile %4lld (block %4u): start %8.2f us, count %6.2f, look-back %6.2f, rows %6.2f, end %8.2f us\n",
               (long long)tile, blockIdx.x, (double)(ts0 % 100000000ull) / 100.0, (double)(ts1 - ts0) / 100.0,
               (double)(ts2 - ts1) / 100.0, (double)(ts3 - ts2) / 100.0, (double)(ts3 % 100000000ull) / 100.0);
#endif
}

// publishes the scalars, the count and the bounding box of the sweep that counts
__global__ void gf_finalize_k(const GfState* __restrict__ st, const float* __restrict__ centroid,
                              const float* __restrict__ scal, float* __restrict__ out_scalars,
                              int64_t* __restrict__ out_count, float* __restrict__ out_aabb) {
    if (blockIdx.x != 0) return;
    const uint32_t use_b = st->use_b;
    if (threadIdx.x == 0) {
        out_scalars[0] = centroid[0];
        out_scalars[1] = centroid[1];
        out_scalars[2] = centroid[2];
        out_scalars[3] = scal[0];
        out_scalars[4] = scal[1 + use_b];
        out_scalars[5] = use_b ? 1.0f : 0.0f;
        out_scalars[6] = __uint_as_float(st->total[0]);   // kept at the first threshold: the uint32 count's BITS
        out_scalars[7] = 0.0f;
        *out_count = st->failed ? (int64_t)-1 : (int64_t)st->total[use_b];
    }
    if (threadIdx.x < 6 && out_aabb) {
        const int a = threadIdx.x;
        uint32_t v = 0;
        for (int k = 0; k < GF_SLOTS; ++k) { const uint32_t u = st->slots[use_b][k][a]; v = u > v ? u : v; }
        out_aabb[a] = v == 0u ? 0.0f : f32_unordered(a < 3 ? ~v : v);
    }
}

struct GfWs {
    float*    centroid;
    float*    zcol;              // copy of the z column (written by the centroid pass)
    MsWs      ms;
    SelWs     sel;
    GfState*  st;
    uint64_t* status;            // [2][tiles] look-back words of the two sweeps
    size_t    clear_bytes;       // st .. end of status: zeroed before the sweeps
    MsCand    cand;              // candidate rows emitted by the summary pass (see gf_cand_k)
};
static void gf_plan(Arena& a, int64_t n, GfWs& w) {
    const int64_t nb = ceil_div(n > 0 ? n : 1, GF_TILE);
    w.centroid = a.take<float>(4);
    ms_plan(a, n, w.ms);
    w.zcol = a.take<float>(n > 0 ? n : 1);
    sel_plan(a, w.sel, n);
    const size_t st_off = a.off;
    w.st = a.take<GfState>(1);
    w.status = a.take<uint64_t>(2 * nb);
    w.clear_bytes = a.off - st_off;
    const int64_t nblk = ceil_div(n > 0 ? n : 1, GF_CBLK);
    w.cand.tcand = a.take<float>(4);
    w.cand.counts = a.take<uint32_t>(nblk);
    w.cand.zsample = a.take<float>(nblk);
    w.cand.slots = a.take<float>(nblk * 4 * MS_CAND_SLOT); // 8 KB per 1024-row block: four planes (MsCand)
    w.cand.pct = 0.0;
    w.cand.add = 0.0f;
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
    PCH_DEVICE_GUARD(xyz ? (const void*)xyz : (const void*)out_centroid);
    PCH_REQUIRE(n >= 0 && out_centroid && ws, "bad argument");
    PCH_REQUIRE(n == 0 || xyz, "null input");
    Arena a(ws, ws_bytes);
    MsWs w;
    ms_plan(a, n, w);
    if (a.overflow) { set_error("workspace too small: need %zu bytes", a.off); return PCH_ERR_WORKSPACE; }
    return mean_seq_launch(xyz, n, out_centroid, w, nullptr, (hipStream_t)stream);
}

extern "C" int pch_mean_seq_partial_f32(const float* xyz, int64_t n, const float* sum_in3, int64_t total_n,
                                        float* out3, float* zcol_out, int32_t phase, void* ws, size_t ws_bytes,
                                        void* stream) {
    PCH_DEVICE_GUARD(ws);
    PCH_REQUIRE(n >= 0 && total_n >= 0 && ws && phase >= 0 && phase <= 2, "bad argument");
    PCH_REQUIRE(phase == MS_PHASE_TABLES || out3, "null output");
    PCH_REQUIRE(n == 0 || xyz, "null input");
    Arena a(ws, ws_bytes, phase == MS_PHASE_WALK);       // the walk continues the tables of the call before
    MsWs w;
    ms_plan(a, n, w);
    if (a.overflow) { set_error("workspace too small: need %zu bytes", a.off); return PCH_ERR_WORKSPACE; }
    return mean_seq_launch(xyz, n, out3, w, zcol_out, (hipStream_t)stream, nullptr, sum_in3,
                           total_n > 0 ? total_n : MS_NO_DIVIDE, phase);
}

extern "C" int pch_mean_seq_serial_f32(const float* xyz, int64_t n, float* out_centroid, void* stream) {
    PCH_DEVICE_GUARD(xyz ? (const void*)xyz : (const void*)out_centroid);
    PCH_REQUIRE(n >= 0 && out_centroid, "bad argument");
    PCH_REQUIRE(n == 0 || xyz, "null input");
    return mean_seq_serial_launch(xyz, n, out_centroid, (hipStream_t)stream);
}

extern "C" size_t pch_percentile_f32_ws_bytes(int64_t n) {
    Arena a;
    SelWs w;
    sel_plan(a, w, n);
    return a.off;
}

extern "C" int pch_percentile_f32(const float* base, int64_t n, int64_t stride, const float* sub,
                                  double q_percent, float* out, void* ws, size_t ws_bytes,
                                  void* stream) {
    PCH_DEVICE_GUARD(base);
    hipStream_t s = (hipStream_t)stream;
    PCH_REQUIRE(n >= 1, "percentile of an empty array (numpy raises IndexError)");
    PCH_REQUIRE(base && out && ws && stride >= 1, "bad argument");
    PCH_REQUIRE(q_percent >= 0.0 && q_percent <= 100.0, "Percentiles must be in the range [0, 100]");
    Arena a(ws, ws_bytes);
    SelWs w;
    sel_plan(a, w, n);
    if (a.overflow) { set_error("workspace too small: need %zu bytes", a.off); return PCH_ERR_WORKSPACE; }
    PCH_TRY(select_percentile(base, n, stride, sub, q_percent, 0.0f, 0.0f, w, s));
    PCH_HIP_TRY(hipMemcpyAsync(out, w.scal, sizeof(float), hipMemcpyDeviceToDevice, s));
    return PCH_OK;
}

// ---- the select passes one by one, for a percentile over values that are spread over several GPUs
// (tiles.shared_percentile: every rank histograms its part, the histograms are all-reduced, the host picks
// the bin - same arithmetic, same result as pch_percentile_f32 over the concatenation)
extern "C" int pch_select_hist_f32(const float* base, int64_t n, int64_t stride, int32_t pass, uint32_t prefix,
                                   uint32_t* out_hist, unsigned long long* out_nan, void* ws, size_t ws_bytes,
                                   void* stream) {
    PCH_DEVICE_GUARD(out_hist);
    hipStream_t s = (hipStream_t)stream;
    PCH_REQUIRE(n >= 0 && stride >= 1 && pass >= 0 && pass <= 2 && out_hist && ws, "bad argument");
    PCH_REQUIRE(n == 0 || base, "null input");
    Arena a(ws, ws_bytes);
    SelWs w;
    sel_plan(a, w);
    if (a.overflow) { set_error("workspace too small: need %zu bytes", a.off); return PCH_ERR_WORKSPACE; }
    SelState h;
    memset(&h, 0, sizeof(h));
    h.prefix = prefix;
    PCH_HIP_TRY(hipMemcpyAsync(w.st, &h, sizeof(h), hipMemcpyHostToDevice, s));
    PCH_HIP_TRY(hipMemsetAsync(w.hist, 0, sizeof(uint32_t) * SEL_BINS, s));
    if (n > 0) {
        int64_t gb = ceil_div(n, SEL_TILE);
        if (gb > 2048) gb = 2048;
        const dim3 grid((unsigned)gb), blk(256);
        if (pass == 0) PCH_LAUNCH("sel_hist0", sel_hist_k<0>, grid, blk, 0, s, base, n, stride, w.st, w.hist);
        else if (pass == 1) PCH_LAUNCH("sel_hist1", sel_hist_k<1>, grid, blk, 0, s, base, n, stride, w.st, w.hist);
        else PCH_LAUNCH("sel_hist2", sel_hist_k<2>, grid, blk, 0, s, base, n, stride, w.st, w.hist);
    }
    PCH_HIP_TRY(hipMemcpyAsync(out_hist, w.hist, sizeof(uint32_t) * SEL_BINS, hipMemcpyDeviceToDevice, s));
    if (out_nan)
        PCH_HIP_TRY(hipMemcpyAsync(out_nan, &w.st->nan_count, sizeof(unsigned long long), hipMemcpyDeviceToDevice, s));
    PCH_HIP_TRY(hipStreamSynchronize(s));               // `h` lives on this stack frame
    return PCH_OK;
}

extern "C" int pch_select_min_above_f32(const float* base, int64_t n, int64_t stride, uint32_t key,
                                        uint32_t* out_key, void* ws, size_t ws_bytes, void* stream) {
    PCH_DEVICE_GUARD(out_key);
    hipStream_t s = (hipStream_t)stream;
    PCH_REQUIRE(n >= 0 && stride >= 1 && out_key && ws, "bad argument");
    PCH_REQUIRE(n == 0 || base, "null input");
    Arena a(ws, ws_bytes);
    SelWs w;
    sel_plan(a, w);
    if (a.overflow) { set_error("workspace too small: need %zu bytes", a.off); return PCH_ERR_WORKSPACE; }
    SelState h;
    memset(&h, 0, sizeof(h));
    h.need_next = 1;
    h.v0key = key;
    h.next_min = 0xFFFFFFFFu;
    PCH_HIP_TRY(hipMemcpyAsync(w.st, &h, sizeof(h), hipMemcpyHostToDevice, s));
    if (n > 0) {
        int64_t gb = ceil_div(n, SEL_TILE);
        if (gb > 2048) gb = 2048;
        PCH_LAUNCH("sel_next", sel_next_k, dim3((unsigned)gb), dim3(256), 0, s, base, n, stride, w.st);
    }
    PCH_HIP_TRY(hipMemcpyAsync(out_key, &w.st->next_min, sizeof(uint32_t), hipMemcpyDeviceToDevice, s));
    PCH_HIP_TRY(hipStreamSynchronize(s));
    return PCH_OK;
}

// ---- keep = (z - cz) > threshold with GIVEN centroid and threshold (the shared values of a tiled run);
// same sweep as the fused filter
extern "C" size_t pch_filter_gt_ws_bytes(int64_t n) {
    if (n < 0) return 0;
    Arena a;
    a.take<float>(8);
    a.take<GfState>(1);
    a.take<uint64_t>(2 * ceil_div(n > 0 ? n : 1, GF_TILE));
    return a.off;
}

namespace pch {
__global__ void gf_count_out_k(const GfState* __restrict__ st, int64_t* __restrict__ out_count, float* __restrict__ out_aabb) {
    if (threadIdx.x == 0) *out_count = st->failed ? (int64_t)-1 : (int64_t)st->total[0];
    if (threadIdx.x < 6 && out_aabb) {
        const int a = threadIdx.x;
        uint32_t v = 0;
        for (int k = 0; k < GF_SLOTS; ++k) { const uint32_t u = st->slots[0][k][a]; v = u > v ? u : v; }
        out_aabb[a] = v == 0u ? 0.0f : f32_unordered(a < 3 ? ~v : v);
    }
}
}  // namespace pch

extern "C" int pch_filter_gt_f32(const float* raw, int64_t n, const float* centroid3_host, float threshold,
                                 float* out_points, int32_t* out_index, int64_t* out_count, float* out_aabb,
                                 void* ws, size_t ws_bytes, void* stream) {
    PCH_DEVICE_GUARD(out_count);
    hipStream_t s = (hipStream_t)stream;
    PCH_REQUIRE(n >= 0 && n < (int64_t(1) << 31) && centroid3_host && out_count && ws, "bad argument");
    if (n == 0) {
        PCH_HIP_TRY(hipMemsetAsync(out_count, 0, sizeof(int64_t), s));
        return PCH_OK;
    }
    PCH_REQUIRE(raw && out_points, "null buffer");
    Arena a(ws, ws_bytes);
    float* scal = a.take<float>(8);                  // [0..2] centroid, [5] threshold (gf_compact_k<0> reads scal[4 + 1])
    const size_t st_off = a.off;
    GfState* st = a.take<GfState>(1);
    const int64_t nb = ceil_div(n, GF_TILE);
    uint64_t* status = a.take<uint64_t>(2 * nb);
    if (a.overflow) { set_error("workspace too small: need %zu bytes", a.off); return PCH_ERR_WORKSPACE; }
    float hs[8] = {centroid3_host[0], centroid3_host[1], centroid3_host[2], 0.0f, 0.0f, threshold, threshold, 0.0f};
    PCH_HIP_TRY(hipMemcpyAsync(scal, hs, sizeof(hs), hipMemcpyHostToDevice, s));
    PCH_HIP_TRY(hipMemsetAsync(st, 0, a.off - st_off, s));
    // the sweep reads z straight from the rows (a pass that first copied the z column out - 16 B per point of traffic
    // for a 12 B per point sweep - is gone)
    PCH_LAUNCH("gf_compact", gf_compact_k<0>, dim3((unsigned)nb), dim3(GF_THREADS), 0, s, raw, (const float*)nullptr, n,
               (const float*)scal, (const float*)(scal + 4), st, status, (long long)0, out_points, out_index,
               (const float*)nullptr);
    PCH_LAUNCH("gf_count_out", gf_count_out_k, dim3(1), dim3(64), 0, s, (const GfState*)st, out_count, out_aabb);
    PCH_HIP_TRY(hipStreamSynchronize(s));               // `hs` lives on this stack frame
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
    PCH_DEVICE_GUARD(raw);
    return pch::ground_filter_run(raw, n, pct, offset, fallback_offset, min_keep, out_points, out_index, out_scalars,
                                  out_count, out_aabb, ws, ws_bytes, (hipStream_t)stream, nullptr);
}

int pch::ground_filter_run(const float* raw, int64_t n, double pct, float offset, float fallback_offset,
                           int64_t min_keep, float* out_points, int32_t* out_index, float* out_scalars,
                           int64_t* out_count, float* out_aabb, void* ws, size_t ws_bytes, hipStream_t s,
                           const GfEarly* early) {
    PCH_REQUIRE(n >= 1 && n < (int64_t(1) << 31), "n out of range [1, 2^31) (numpy raises on empty input)");
    PCH_REQUIRE(raw && out_points && out_scalars && out_count && ws, "null buffer");
    PCH_REQUIRE(pct >= 0.0 && pct <= 100.0, "Percentiles must be in the range [0, 100]");
    Arena a(ws, ws_bytes);
    GfWs w;
    gf_plan(a, n, w);
    if (a.overflow) { set_error("workspace too small: need %zu bytes", a.off); return PCH_ERR_WORKSPACE; }
    const int64_t nb = ceil_div(n, GF_TILE);

    // candidate rows for the sweep: raw z above (estimated percentile + the smaller offset - 0.5 m), emitted by the summary
    w.cand.pct = pct;
    w.cand.add = (offset < fallback_offset ? offset : fallback_offset) - 0.5f;
    bool cand_made = false;
    // the sweep's state and look-back words are cleared by the select's init kernel (first thing on the side stream)
    w.sel.also_zero = reinterpret_cast<uint32_t*>(w.st);
    w.sel.also_words = (int64_t)(w.clear_bytes / 4);
    SideStream& ss = side_stream();
    if (ss.ok) {
        // Two strands.  `s`: summary, level 2, walk - no cross-stream hop inside the chain.  Side stream: at once the
        // sample half of the percentile (it reads the z of the raw rows, so it does not wait for the z column and
        // runs beside the summary), then - behind the summary - the percentile's sweep and exact select; joined in
        // front of the interpolation.  With the sample half out of the way both strands end within a few us of each
        // other (measured: the other assignment, select on `s` and walk on the side stream, loses 45 us to the hops).
        const bool sample_early = select_is_bracketed(w.sel, n, 1);
        if (sample_early) {
            PCH_HIP_TRY(hipEventRecord(ss.ev_start, s));            // everything the caller queued before this call
            PCH_HIP_TRY(hipStreamWaitEvent(ss.s, ss.ev_start, 0));   // (the rows; the previous user of the workspace)
            PCH_TRY(select_sample_passes(raw + 2, 3, n, pct, w.sel, ss.s));
        }
        PCH_TRY(mean_seq_launch(raw, n, w.centroid, w.ms, w.zcol, s, ss.ev_fork, nullptr, MS_DIVIDE_BY_N, MS_PHASE_BOTH,
                                &w.cand, &cand_made));
        PCH_HIP_TRY(hipStreamWaitEvent(ss.s, ss.ev_fork, 0));
        PCH_TRY(select_passes(w.zcol, n, 1, pct, w.sel, ss.s, sample_early));
        PCH_HIP_TRY(hipEventRecord(ss.ev_join, ss.s));
        PCH_HIP_TRY(hipStreamWaitEvent(s, ss.ev_join, 0));
    } else {
        PCH_TRY(mean_seq_launch(raw, n, w.centroid, w.ms, w.zcol, s, nullptr, nullptr, MS_DIVIDE_BY_N, MS_PHASE_BOTH,
                                &w.cand, &cand_made));
        PCH_TRY(select_passes(w.zcol, n, 1, pct, w.sel, s));
    }
    PCH_TRY(select_lerp(n, w.centroid + 2, pct, offset, fallback_offset, w.sel, s));
    const dim3 grid((unsigned)nb), blk(GF_THREADS);
    const float* tcand = cand_made ? (const float*)w.cand.tcand : (const float*)nullptr;
    // Sweep A (first threshold) from the candidate slots when the device-side guard allows it (gf_cand_ok), else over
    // the tile - one of the two kernels returns at once; same tiles, same tickets, same look-back words.  Then the
    // results are published ONCE ALREADY (gf_finalize; and queued for the host if the caller asked): they are final
    // unless the last tile of sweep A raised use_b.  Sweep B (fallback threshold) and the second gf_finalize follow;
    // without use_b they return at once, and the host is by then preparing the next stage.
    const int64_t nblk = ceil_div(n, GF_CBLK);
    const dim3 cgrid((unsigned)ceil_div(nblk, GF_CT_BLKS));
    if (cand_made)
        PCH_LAUNCH("gf_cand", gf_cand_k<0>, cgrid, blk, 0, s, (const float*)w.cand.slots, (const uint32_t*)w.cand.counts,
                   tcand, nblk, (const float*)w.centroid, (const float*)w.sel.scal, w.st, w.status, (long long)min_keep,
                   out_points, out_index);
    PCH_LAUNCH("gf_compact", gf_compact_k<0>, grid, blk, 0, s, raw, (const float*)w.zcol, n,
               (const float*)w.centroid, (const float*)w.sel.scal, w.st, w.status, (long long)min_keep, out_points,
               out_index, tcand);
    if (early) {
        PCH_LAUNCH("gf_finalize", gf_finalize_k, dim3(1), dim3(64), 0, s, (const GfState*)w.st,
                   (const float*)w.centroid, (const float*)w.sel.scal, out_scalars, out_count, out_aabb);
        PCH_TRY(peek_enqueue(early->dev, early->bytes, s));
    }
    if (cand_made)
        PCH_LAUNCH("gf_cand_fb", gf_cand_k<1>, cgrid, blk, 0, s, (const float*)w.cand.slots,
                   (const uint32_t*)w.cand.counts, tcand, nblk, (const float*)w.centroid, (const float*)w.sel.scal, w.st,
                   w.status + nb, (long long)min_keep, out_points, out_index);
    PCH_LAUNCH("gf_compact_fb", gf_compact_k<1>, grid, blk, 0, s, raw, (const float*)w.zcol, n,
               (const float*)w.centroid, (const float*)w.sel.scal, w.st, w.status + nb, (long long)min_keep, out_points,
               out_index, tcand);
    PCH_LAUNCH("gf_finalize", gf_finalize_k, dim3(1), dim3(64), 0, s, (const GfState*)w.st,
               (const float*)w.centroid, (const float*)w.sel.scal, out_scalars, out_count, out_aabb);
    return PCH_OK;
}
