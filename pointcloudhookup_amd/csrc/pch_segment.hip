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

// bounding box per cluster: waves walk the label-sorted rows (1024 per wave); a wave almost
// always sees a single label, reduces in registers and issues six ordered-uint atomics
__global__ __launch_bounds__(SG_THREADS) void sg_stats_init_k(uint32_t* __restrict__ acc, int32_t nclusters) {
    const int i = blockIdx.x * SG_THREADS + threadIdx.x;
    if (i < 6 * nclusters) acc[i] = (i % 6) < 3 ? 0xFFFFFFFFu : 0u;
}

__device__ __forceinline__ void sg_flush(uint32_t* __restrict__ acc, int cur, const uint32_t (&lo)[3],
                                         const uint32_t (&hi)[3]) {
    const int l = lane_id();
    if (cur >= 0 && l < 6) {
        uint32_t v = 0;
#pragma unroll
        for (int a = 0; a < 3; ++a) { if (l == a) v = lo[a]; if (l == 3 + a) v = hi[a]; }
        if (l < 3) atomicMin(&acc[6 * (int64_t)cur + l], v);
        else       atomicMax(&acc[6 * (int64_t)cur + l], v);
    }
}

__global__ __launch_bounds__(SG_THREADS) void sg_stats_k(const float* __restrict__ xyz,
                                                         const int32_t* __restrict__ perm,
                                                         const uint64_t* __restrict__ keys, int64_t n,
                                                         int32_t nclusters, uint32_t* __restrict__ acc) {
    const int64_t w = (int64_t)blockIdx.x * (SG_THREADS / 64) + wave_id();
    const int l = lane_id();
    int cur = -1;
    uint32_t lo[3] = {0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu}, hi[3] = {0u, 0u, 0u};
    for (int r = 0; r < 16; ++r) {
        const int64_t j = w * 1024 + r * 64 + l;
        int lab = -1;
        uint32_t k[3] = {0, 0, 0};
        if (j < n) {
            const uint64_t key = keys[j];
            if (key < (uint64_t)nclusters) {               // noise rows carry key == nclusters
                lab = (int)key;
                const int64_t p = perm[j];
#pragma unroll
                for (int a = 0; a < 3; ++a) k[a] = f32_ordered(xyz[3 * p + a]);
            }
        }
        // rows are sorted by label: handle the labels present in this round one at a time
        unsigned long long todo = __ballot(lab >= 0);
        while (todo) {
            const int lead = (int)__builtin_ctzll(todo);
            const int L = __builtin_amdgcn_readlane(lab, lead);
            const unsigned long long same = __ballot(lab == L);
            todo &= ~same;
            if (L != cur) {                                // a new label starts: flush the previous one
                sg_flush(acc, cur, lo, hi);
#pragma unroll
                for (int a = 0; a < 3; ++a) { lo[a] = 0xFFFFFFFFu; hi[a] = 0u; }
                cur = L;
            }
#pragma unroll
            for (int a = 0; a < 3; ++a) {
                const uint32_t mn = wave_reduce_min(lab == L ? k[a] : 0xFFFFFFFFu);
                const uint32_t mx = wave_reduce_max(lab == L ? k[a] : 0u);
                lo[a] = mn < lo[a] ? mn : lo[a];
                hi[a] = mx > hi[a] ? mx : hi[a];
            }
        }
    }
    sg_flush(acc, cur, lo, hi);
}

__global__ __launch_bounds__(SG_THREADS) void sg_stats_out_k(const uint32_t* __restrict__ acc, int32_t nclusters,
                                                             float* __restrict__ stats) {
    const int i = blockIdx.x * SG_THREADS + threadIdx.x;
    if (i >= 8 * nclusters) return;
    const int k = i >> 3, a = i & 7;
    float v = 0.0f;
    if (a < 6) {
        const uint32_t u = acc[6 * (int64_t)k + a];
        const bool empty = a < 3 ? (u == 0xFFFFFFFFu) : (u == 0u);
        v = empty ? (a < 3 ? INFINITY : -INFINITY) : f32_unordered(u);
    }
    stats[i] = v;
}

struct SgWs {
    uint64_t *k0, *k1;
    uint32_t *v0, *v1, *radix_ws, *acc;
};
static void sg_plan(Arena& a, int64_t n, int32_t nclusters, SgWs& w) {
    const int64_t nn = n > 0 ? n : 1;
    w.acc = a.take<uint32_t>(6 * (size_t)(nclusters > 0 ? nclusters : 1));
    w.k0 = a.take<uint64_t>(nn);
    w.k1 = a.take<uint64_t>(nn);
    w.v0 = a.take<uint32_t>(nn);
    w.v1 = a.take<uint32_t>(nn);
    w.radix_ws = a.take<uint32_t>(radix_ws_u32(nn));
}

}  // namespace pch

using namespace pch;

extern "C" size_t pch_segment_by_label_ws_bytes(int64_t n, int32_t nclusters) {
    if (n < 0) return 0;
    Arena a;
    SgWs w;
    sg_plan(a, n, nclusters, w);
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
    sg_plan(a, n, nclusters, w);
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
    if (out_stats && nclusters > 0) {
        PCH_LAUNCH("seg_stats_init", sg_stats_init_k, dim3((unsigned)ceil_div(6 * (int64_t)nclusters, SG_THREADS)),
                   dim3(SG_THREADS), 0, s, w.acc, nclusters);
        PCH_LAUNCH("seg_stats", sg_stats_k, dim3((unsigned)ceil_div(n, 1024 * (SG_THREADS / 64))), dim3(SG_THREADS),
                   0, s, xyz, (const int32_t*)out_perm, ks, n, nclusters, w.acc);
        PCH_LAUNCH("seg_stats_out", sg_stats_out_k, dim3((unsigned)ceil_div(8 * (int64_t)nclusters, SG_THREADS)),
                   dim3(SG_THREADS), 0, s, (const uint32_t*)w.acc, nclusters, out_stats);
    }
    return PCH_OK;
}
