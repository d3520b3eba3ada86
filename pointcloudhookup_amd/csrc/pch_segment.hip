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

// word i of the caller's float [K,8] stats table, in place: the ordered-uint box table stage C left there (DbBoxOut)
// -> the bit patterns of min xyz, max xyz, 0, 0.  One word per thread (a row-wide version of this crashes the
// gfx950 instruction selector of ROCm 7.2's clang).
__device__ __forceinline__ void sg_decode_word(uint32_t* enc, int64_t i) {
    const int a = (int)(i & 7);
    const uint32_t u = enc[i];
    uint32_t v = 0u;
    if (a < 3) v = u == 0u ? 0x7F800000u : __float_as_uint(f32_unordered(~u));           // +inf: no point
    else if (a < 6) v = u == 0u ? 0xFF800000u : __float_as_uint(f32_unordered(u));       // -inf
    enc[i] = v;
}

// offsets[k] = first sorted position whose key >= k  (k = 0..nclusters); perm = sorted rows
__global__ __launch_bounds__(SG_THREADS) void sg_offsets_k(const uint64_t* __restrict__ keys, int64_t n,
                                                           int32_t nclusters, int64_t* __restrict__ offsets,
                                                           uint32_t* enc_stats) {
    const int64_t k = (int64_t)blockIdx.x * SG_THREADS + threadIdx.x;
    if (enc_stats && k < 8 * (int64_t)nclusters) sg_decode_word(enc_stats, k);
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

// bounding box per cluster: waves walk the rows in file order (1024 per wave, coalesced reads);
// neighbouring rows mostly share a label, which a wave reduces in registers before it issues six
// ordered-uint atomics
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
                                                         const int32_t* __restrict__ labels, int64_t n,
                                                         int32_t nclusters, uint32_t* __restrict__ acc) {
    const int64_t w = (int64_t)blockIdx.x * (SG_THREADS / 64) + wave_id();
    const int l = lane_id();
    int cur = -1;
    uint32_t lo[3] = {0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu}, hi[3] = {0u, 0u, 0u};
    struct Row3 { float x, y, z; };
    int lab16[16];
#pragma unroll
    for (int r = 0; r < 16; ++r) {                         // rows in file order: coalesced label reads
        const int64_t j = w * 1024 + r * 64 + l;
        const int v = j < n ? labels[j] : -1;
        lab16[r] = (v >= 0 && v < nclusters) ? v : -1;
    }
    for (int r = 0; r < 16; ++r) {
        const int64_t j = w * 1024 + r * 64 + l;
        const int lab = lab16[r];
        uint32_t k[3] = {0, 0, 0};
        if (lab >= 0) {
            const Row3 q = reinterpret_cast<const Row3*>(xyz)[j];
            k[0] = f32_ordered(q.x); k[1] = f32_ordered(q.y); k[2] = f32_ordered(q.z);
        }
        // neighbouring rows mostly share a label: handle the labels present in this round one at a time
        unsigned long long todo = __ballot(lab >= 0);
        while (todo) {
            const int lead = (int)__builtin_ctzll(todo);
            const int L = __builtin_amdgcn_readlane(lab, lead);
            const unsigned long long same = __ballot(lab == L);
            todo &= ~same;
            if (L != cur) {                                // another label: flush the one carried so far
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

// ---- one-pass counting sort for at most 255 clusters (bins 0..K-1, noise = bin K) ---------
constexpr int SL_ROUNDS = 8;
constexpr int SL_TILE   = SG_THREADS * SL_ROUNDS;          // 2048 labels per workgroup

__device__ __forceinline__ uint32_t sl_bin(int32_t l, int32_t nclusters) {
    return (l < 0 || l >= nclusters) ? (uint32_t)nclusters : (uint32_t)l;     // noise last
}

__global__ __launch_bounds__(SG_THREADS) void sl_hist_k(const int32_t* __restrict__ labels, int64_t n,
                                                        int32_t nclusters, uint32_t* __restrict__ hist,
                                                        int64_t nb, uint32_t* __restrict__ zero_ws,
                                                        int64_t zero_words) {
    __shared__ uint32_t h[256];
    {   // the single-pass scan of the count table that follows needs its status words zeroed (pch_prims.h)
        const int64_t g = (int64_t)blockIdx.x * SG_THREADS + threadIdx.x;
        if (g < zero_words) zero_ws[g] = 0u;
    }
    h[threadIdx.x] = 0;
    __syncthreads();
    const int64_t base = (int64_t)blockIdx.x * SL_TILE;
    int32_t v[SL_ROUNDS];
#pragma unroll
    for (int r = 0; r < SL_ROUNDS; ++r) {
        const int64_t i = base + r * SG_THREADS + threadIdx.x;
        v[r] = i < n ? labels[i] : -1;
    }
#pragma unroll
    for (int r = 0; r < SL_ROUNDS; ++r) {
        const int64_t i = base + r * SG_THREADS + threadIdx.x;
        // rows of a wave mostly share a bin: one merged add for the first active lane's bin
        const bool in = i < n;
        const uint32_t b = sl_bin(v[r], nclusters);
        const unsigned long long act = __ballot(in);
        if (act) {
            const uint32_t b0 = (uint32_t)__builtin_amdgcn_readlane((int)b, (int)__builtin_ctzll(act));
            const unsigned long long same = __ballot(in && b == b0);
            if (lane_id() == (int)__builtin_ctzll(act)) atomicAdd(&h[b0], (uint32_t)__popcll(same));
            if (in && b != b0) atomicAdd(&h[b], 1u);
        }
    }
    __syncthreads();
    if ((int)threadIdx.x <= nclusters) hist[(int64_t)threadIdx.x * nb + blockIdx.x] = h[threadIdx.x];
}

__global__ __launch_bounds__(SG_THREADS) void sl_scatter_k(const int32_t* __restrict__ labels, int64_t n,
                                                           int32_t nclusters,
                                                           const uint32_t* __restrict__ offs, int64_t nb,
                                                           int32_t* __restrict__ perm) {
    constexpr int WAVES = SG_THREADS / 64;
    __shared__ uint32_t cnt[WAVES][256];
    const int w = wave_id(), l = lane_id();
    for (int j = threadIdx.x; j < WAVES * 256; j += SG_THREADS) (&cnt[0][0])[j] = 0;
    __syncthreads();
    const int64_t seg = (int64_t)blockIdx.x * SL_TILE + (int64_t)w * (64 * SL_ROUNDS);
    uint32_t bin[SL_ROUNDS], rank[SL_ROUNDS];
    const uint64_t lt = lanemask_lt();
#pragma unroll
    for (int r = 0; r < SL_ROUNDS; ++r) {
        const int64_t i = seg + r * 64 + l;
        bin[r] = sl_bin(i < n ? labels[i] : -1, nclusters);
    }
#pragma unroll
    for (int r = 0; r < SL_ROUNDS; ++r) {
        const bool valid = seg + r * 64 + l < n;
        const uint32_t d = bin[r];
        uint64_t peers = __ballot(valid);
#pragma unroll
        for (int b = 0; b < 8; ++b) {
            const bool bit = (d >> b) & 1u;
            const uint64_t m = __ballot(bit);
            peers &= bit ? m : ~m;
        }
        const uint32_t prior = cnt[w][d];
        const uint32_t rk = (uint32_t)__popcll(peers & lt);
        __builtin_amdgcn_wave_barrier();
        if (valid && rk == 0) cnt[w][d] = prior + (uint32_t)__popcll(peers);
        __builtin_amdgcn_wave_barrier();
        rank[r] = prior + rk;
    }
    __syncthreads();
    if ((int)threadIdx.x <= nclusters) {
        const int d = threadIdx.x;
        uint32_t run = offs[(int64_t)d * nb + blockIdx.x];
#pragma unroll
        for (int w2 = 0; w2 < WAVES; ++w2) {
            const uint32_t c = cnt[w2][d];
            cnt[w2][d] = run;
            run += c;
        }
    }
    __syncthreads();
#pragma unroll
    for (int r = 0; r < SL_ROUNDS; ++r) {
        const int64_t i = seg + r * 64 + l;
        if (i < n) perm[cnt[w][bin[r]] + rank[r]] = (int32_t)i;
    }
}

// offsets[k] = start of bin k = scanned count of (bin k, tile 0); offsets[K] = start of the noise rows
__global__ __launch_bounds__(SG_THREADS) void sl_offsets_k(const uint32_t* __restrict__ offs, int64_t nb,
                                                           int32_t nclusters, int64_t* __restrict__ offsets,
                                                           uint32_t* enc_stats, const uint32_t* __restrict__ scan_total) {
    const int k = blockIdx.x * SG_THREADS + threadIdx.x;
    // a single-pass scan whose bounded look-back wait gave up (pch_lookback.h) raised its total word to 0xFFFFFFFF:
    // perm and offsets are then undefined, and every offset reads -1 (include/pch_hip.h, PCH_ERR_TIMEOUT)
    const bool failed = scan_total && *scan_total == 0xFFFFFFFFu;
    if (k <= nclusters) offsets[k] = failed ? int64_t(-1) : (int64_t)offs[(int64_t)k * nb];
    if (enc_stats && k < 8 * nclusters) sg_decode_word(enc_stats, k);
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
    uint32_t *table, *table_scan;      // one-pass path: [bins][tiles] counts + scan scratch
};
static void sg_plan(Arena& a, int64_t n, int32_t nclusters, SgWs& w) {
    const int64_t nn = n > 0 ? n : 1;
    w.acc = a.take<uint32_t>(6 * (size_t)(nclusters > 0 ? nclusters : 1));
    if (nclusters < 256) {                                 // labels + noise fit one 8-bit digit
        const int64_t table = ((int64_t)nclusters + 1) * ceil_div(nn, SL_TILE);
        w.table = a.take<uint32_t>(table);
        w.table_scan = a.take<uint32_t>(scan1_pays(table) ? scan1_ws_u32(table) : scan_ws_u32(table));
        w.k0 = w.k1 = nullptr; w.v0 = w.v1 = w.radix_ws = nullptr;
    } else {
        w.table = w.table_scan = nullptr;
        w.k0 = a.take<uint64_t>(nn);
        w.k1 = a.take<uint64_t>(nn);
        w.v0 = a.take<uint32_t>(nn);
        w.v1 = a.take<uint32_t>(nn);
        w.radix_ws = a.take<uint32_t>(radix_ws_u32(nn));
    }
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
    PCH_DEVICE_GUARD(labels ? (const void*)labels : (const void*)out_offsets);
    return segment_run(labels, xyz, n, nclusters, out_perm, out_offsets, out_stats, ws, ws_bytes, (hipStream_t)stream,
                       false);
}

int pch::segment_run(const int32_t* labels, const float* xyz, int64_t n, int32_t nclusters, int32_t* out_perm,
                     int64_t* out_offsets, float* out_stats, void* ws, size_t ws_bytes, hipStream_t s,
                     bool stats_encoded) {
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
    if (nclusters < 256) {
        const int64_t nb = ceil_div(n, SL_TILE);
        const int64_t table = ((int64_t)nclusters + 1) * nb;
        const bool one = scan1_pays(table);
        const int64_t zero_words = one ? (int64_t)scan1_ws_u32(table) : 0;
        PCH_REQUIRE(zero_words <= nb * SG_THREADS, "count table too large for the fused zeroing");   // (K+1)/1024 < 256
        PCH_LAUNCH("seg_hist", sl_hist_k, dim3((unsigned)nb), dim3(SG_THREADS), 0, s, labels, n, nclusters,
                   w.table, nb, w.table_scan, zero_words);
        // the single-pass scan reports a look-back time-out through its total word: the spare word behind the ticket
        // (scan1_ws_u32 rounds 2 * tiles + 1 up), zeroed with the rest of the scratch by sl_hist_k
        const uint32_t* scan_total = one ? w.table_scan + 2 * ceil_div(table, (int64_t)SCAN_TILE) + 1 : nullptr;
        if (one) PCH_TRY(scan1_exclusive_u32(w.table, w.table, table, w.table_scan, const_cast<uint32_t*>(scan_total), s));
        else PCH_TRY(scan_exclusive_u32(w.table, w.table, table, w.table_scan, nullptr, s));
        PCH_LAUNCH("seg_scatter", sl_scatter_k, dim3((unsigned)nb), dim3(SG_THREADS), 0, s, labels, n, nclusters,
                   (const uint32_t*)w.table, nb, out_perm);
        PCH_LAUNCH("seg_offsets", sl_offsets_k,
                   dim3((unsigned)ceil_div(stats_encoded ? 8 * (int64_t)nclusters + 1 : (int64_t)nclusters + 1, SG_THREADS)),
                   dim3(SG_THREADS), 0, s, (const uint32_t*)w.table, nb, nclusters, out_offsets,
                   stats_encoded ? reinterpret_cast<uint32_t*>(out_stats) : (uint32_t*)nullptr, scan_total);
    } else {
        PCH_LAUNCH("seg_keys", sg_keys_k, dim3(gn), dim3(SG_THREADS), 0, s, labels, n, nclusters, w.k0, w.v0);
        const int nbits = bits_for((uint64_t)nclusters + 1);
        PCH_TRY(radix_sort_pairs(w.k0, w.v0, w.k1, w.v1, n, nbits, w.radix_ws, s));
        const bool in1 = radix_sort_result_buffer(nbits) == 1;
        const uint64_t* ks = in1 ? w.k1 : w.k0;
        const uint32_t* vs = in1 ? w.v1 : w.v0;
        PCH_LAUNCH("seg_perm", sg_perm_k, dim3(gn), dim3(SG_THREADS), 0, s, vs, n, out_perm);
        PCH_LAUNCH("seg_offsets", sg_offsets_k,
                   dim3((unsigned)ceil_div(stats_encoded ? 8 * (int64_t)nclusters + 1 : (int64_t)nclusters + 1, SG_THREADS)),
                   dim3(SG_THREADS), 0, s, ks, n, nclusters, out_offsets,
                   stats_encoded ? reinterpret_cast<uint32_t*>(out_stats) : (uint32_t*)nullptr);
    }
    if (out_stats && nclusters > 0 && !stats_encoded) {
        PCH_LAUNCH("seg_stats_init", sg_stats_init_k, dim3((unsigned)ceil_div(6 * (int64_t)nclusters, SG_THREADS)),
                   dim3(SG_THREADS), 0, s, w.acc, nclusters);
        PCH_LAUNCH("seg_stats", sg_stats_k, dim3((unsigned)ceil_div(n, 1024 * (SG_THREADS / 64))), dim3(SG_THREADS),
                   0, s, xyz, labels, n, nclusters, w.acc);
        PCH_LAUNCH("seg_stats_out", sg_stats_out_k, dim3((unsigned)ceil_div(8 * (int64_t)nclusters, SG_THREADS)),
                   dim3(SG_THREADS), 0, s, (const uint32_t*)w.acc, nclusters, out_stats);
    }
    return PCH_OK;
}
