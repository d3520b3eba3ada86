// Stage D0: group point rows by cluster label with one stable radix sort instead of the
// reference's K boolean masks over all N_f points (utils/tower_extraction.py:125,131-134),
// plus one bounding box per cluster.
#include "pch_prims.h"

namespace pch {

constexpr int SG_THREADS = 256;

__global__ __launch_bounds__(SG_THREADS) void sg_keys_k(const int32_t* __restrict__ labels, int64_t n,
                                                        int32_t nclusters, uint64_t* __restrict__ keys,
                                                        uint32_t* __restrict__ vals) {
    const int64_t i = (int64_t)blockIdx.x * SG_THREADS + threadIdx.x;
    if (i >= n) return;
    const int32_t l = labels[i];
    keys[i] = (l < 0 || l >= nclusters) ? (uint64_t)nclusters : (uint64_t)l;   // noise last
    vals[i] = (uint32_t)i;
}

// offsets[k] = first sorted position whose key >= k  (k = 0..nclusters); perm = sorted rows
__global__ __launch_bounds__(SG_THREADS) void sg_offsets_k(const uint64_t* __restrict__ keys, int64_t n,
                                                           int32_t nclusters, int64_t* __restrict__ offsets) {
    const int64_t k = (int64_t)blockIdx.x * SG_THREADS + threadIdx.x;
    if (k > nclusters) return;
    int64_t lo = 0, hi = n;
    while (lo < hi) { const int64_t mid = (lo + hi) >> 1; if (keys[mid] < (uint64_t)k) lo = mid + 1; else hi = mid; }
    offsets[k] = lo;
}

__global__ __launch_bounds__(SG_THREADS) void sg_perm_k(const uint32_t* __restrict__ vals, int64_t n,
                                                        int32_t* __restrict__ perm) {
    const int64_t i = (int64_t)blockIdx.x * SG_THREADS + threadIdx.x;
    if (i < n) perm[i] = (int32_t)vals[i];
}

// one workgroup per cluster: min / max of its points
__global__ __launch_bounds__(SG_THREADS) void sg_stats_k(const float* __restrict__ xyz,
                                                         const int32_t* __restrict__ perm,
                                                         const int64_t* __restrict__ offsets,
                                                         float* __restrict__ stats) {
    __shared__ float sm[SG_THREADS / 64][6];
    const int k = blockIdx.x;
    const int64_t s = offsets[k], e = offsets[k + 1];
    float lo[3] = {INFINITY, INFINITY, INFINITY}, hi[3] = {-INFINITY, -INFINITY, -INFINITY};
    for (int64_t j = s + threadIdx.x; j < e; j += SG_THREADS) {
        const int64_t p = perm[j];
#pragma unroll
        for (int a = 0; a < 3; ++a) {
            const float v = xyz[3 * p + a];
            lo[a] = fminf(lo[a], v);
            hi[a] = fmaxf(hi[a], v);
        }
    }
#pragma unroll
    for (int a = 0; a < 3; ++a) { lo[a] = wave_reduce_min(lo[a]); hi[a] = wave_reduce_max(hi[a]); }
    if (lane_id() == 0) {
#pragma unroll
        for (int a = 0; a < 3; ++a) { sm[wave_id()][a] = lo[a]; sm[wave_id()][3 + a] = hi[a]; }
    }
    __syncthreads();
    if (threadIdx.x < 6) {
        const int a = threadIdx.x;
        float v = sm[0][a];
        for (int w = 1; w < SG_THREADS / 64; ++w) v = (a < 3) ? fminf(v, sm[w][a]) : fmaxf(v, sm[w][a]);
        stats[8 * (int64_t)k + a] = v;
    }
    if (threadIdx.x == 6) stats[8 * (int64_t)k + 6] = 0.0f;
    if (threadIdx.x == 7) stats[8 * (int64_t)k + 7] = 0.0f;
}

struct SgWs {
    uint64_t *k0, *k1;
    uint32_t *v0, *v1, *radix_ws;
};
static void sg_plan(Arena& a, int64_t n, SgWs& w) {
    const int64_t nn = n > 0 ? n : 1;
    w.k0 = a.take<uint64_t>(nn);
    w.k1 = a.take<uint64_t>(nn);
    w.v0 = a.take<uint32_t>(nn);
    w.v1 = a.take<uint32_t>(nn);
    w.radix_ws = a.take<uint32_t>(radix_ws_u32(nn));
}

}  // namespace pch

using namespace pch;

extern "C" size_t pch_segment_by_label_ws_bytes(int64_t n, int32_t) {
    if (n < 0) return 0;
    Arena a;
    SgWs w;
    sg_plan(a, n, w);
    return a.off;
}

extern "C" int pch_segment_by_label(const int32_t* labels, const float* xyz, int64_t n,
                                    int32_t nclusters, int32_t* out_perm, int64_t* out_offsets,
                                    float* out_stats, void* ws, size_t ws_bytes, void* stream) {
    prof_begin_call();
    hipStream_t s = (hipStream_t)stream;
    PCH_REQUIRE(n >= 0 && n < (int64_t(1) << 31) && nclusters >= 0, "bad size");
    PCH_REQUIRE(out_offsets != nullptr, "out_offsets is null");
    if (n == 0) {
        PCH_HIP_TRY(hipMemsetAsync(out_offsets, 0, sizeof(int64_t) * ((size_t)nclusters + 1), s));
        return PCH_OK;
    }
    PCH_REQUIRE(labels && out_perm && ws, "null buffer");
    PCH_REQUIRE(!out_stats || xyz, "stats requested without coordinates");
    Arena a(ws, ws_bytes);
    SgWs w;
    sg_plan(a, n, w);
    if (a.overflow) { set_error("workspace too small: need %zu bytes", a.off); return PCH_ERR_WORKSPACE; }
    const unsigned gn = (unsigned)ceil_div(n, SG_THREADS);
    PCH_LAUNCH("seg_keys", sg_keys_k, dim3(gn), dim3(SG_THREADS), 0, s, labels, n, nclusters, w.k0, w.v0);
    const int nbits = bits_for((uint64_t)nclusters + 1);
    PCH_TRY(radix_sort_pairs(w.k0, w.v0, w.k1, w.v1, n, nbits, w.radix_ws, s));
    const bool in1 = radix_sort_result_buffer(nbits) == 1;
    const uint64_t* ks = in1 ? w.k1 : w.k0;
    const uint32_t* vs = in1 ? w.v1 : w.v0;
    PCH_LAUNCH("seg_perm", sg_perm_k, dim3(gn), dim3(SG_THREADS), 0, s, vs, n, out_perm);
    PCH_LAUNCH("seg_offsets", sg_offsets_k, dim3((unsigned)ceil_div((int64_t)nclusters + 1, SG_THREADS)),
               dim3(SG_THREADS), 0, s, ks, n, nclusters, out_offsets);
    if (out_stats && nclusters > 0)
        PCH_LAUNCH("seg_stats", sg_stats_k, dim3((unsigned)nclusters), dim3(SG_THREADS), 0, s, xyz,
                   (const int32_t*)out_perm, (const int64_t*)out_offsets, out_stats);
    return PCH_OK;
}
