// Device-wide exclusive scan and stable LSD radix sort, written for 64-lane wavefronts.
#include "pch_prims.h"
#include "pch_lookback.h"

namespace pch {

// =====================================================================================
// exclusive scan (reduce-then-scan, three launches; every element is read twice and
// written once -> 12 B/element of HBM traffic)
// =====================================================================================
constexpr int SC_THREADS = 256;
constexpr int SC_ITEMS   = 8;
constexpr int SC_TILE    = SC_THREADS * SC_ITEMS;   // 2048 elements per workgroup
static_assert(SC_TILE == SCAN_TILE, "pch_prims.h");

// POPC: the scanned values are the population counts of the input words (row bitmap -> ranks), taken on load
template <bool POPC>
__device__ __forceinline__ void sc_load8(const uint32_t* in, int64_t base, int64_t n,
                                         uint32_t (&v)[SC_ITEMS]) {
    if (base + SC_ITEMS <= n) {
        const uint4 a = *reinterpret_cast<const uint4*>(in + base);
        const uint4 b = *reinterpret_cast<const uint4*>(in + base + 4);
        v[0] = a.x; v[1] = a.y; v[2] = a.z; v[3] = a.w;
        v[4] = b.x; v[5] = b.y; v[6] = b.z; v[7] = b.w;
    } else {
#pragma unroll
        for (int k = 0; k < SC_ITEMS; ++k) v[k] = (base + k < n) ? in[base + k] : 0u;
    }
    if (POPC) {
#pragma unroll
        for (int k = 0; k < SC_ITEMS; ++k) v[k] = (uint32_t)__popc(v[k]);
    }
}

template <bool POPC>
__global__ __launch_bounds__(SC_THREADS) void scan_reduce_k(const uint32_t* __restrict__ in,
                                                            uint32_t* __restrict__ bsum,
                                                            int64_t n) {
    __shared__ uint32_t wsum[SC_THREADS / 64];
    const int64_t base = (int64_t)blockIdx.x * SC_TILE + (int64_t)threadIdx.x * SC_ITEMS;
    uint32_t v[SC_ITEMS];
    sc_load8<POPC>(in, base, n, v);
    uint32_t s = 0;
#pragma unroll
    for (int k = 0; k < SC_ITEMS; ++k) s += v[k];
    s = wave_reduce_add(s);
    if (lane_id() == 0) wsum[wave_id()] = s;
    __syncthreads();
    if (threadIdx.x == 0) bsum[blockIdx.x] = wsum[0] + wsum[1] + wsum[2] + wsum[3];
}

// single workgroup: in-place exclusive scan of the block sums
__global__ __launch_bounds__(1024) void scan_bsums_k(uint32_t* __restrict__ bsum, int64_t nb,
                                                     uint32_t* __restrict__ total) {
    __shared__ uint32_t wsum[16];
    __shared__ uint32_t carry_s;
    if (threadIdx.x == 0) carry_s = 0;
    __syncthreads();
    for (int64_t base = 0; base < nb; base += 1024) {
        const int64_t i = base + threadIdx.x;
        const uint32_t v = (i < nb) ? bsum[i] : 0u;
        uint32_t incl = wave_scan_incl(v);
        if (lane_id() == 63) wsum[wave_id()] = incl;
        __syncthreads();
        if (wave_id() == 0) {
            uint32_t w = (lane_id() < 16) ? wsum[lane_id()] : 0u;
            uint32_t wi = wave_scan_incl(w);
            if (lane_id() < 16) wsum[lane_id()] = wi - w;     // exclusive wave offsets
        }
        __syncthreads();
        const uint32_t carry = carry_s;
        const uint32_t excl = carry + wsum[wave_id()] + incl - v;
        if (i < nb) bsum[i] = excl;
        __syncthreads();
        if (threadIdx.x == 1023) carry_s = excl + v;
        __syncthreads();
    }
    if (threadIdx.x == 0 && total) *total = carry_s;
}

template <bool POPC>
__global__ __launch_bounds__(SC_THREADS) void scan_apply_k(const uint32_t* in,
                                                           uint32_t* out,
                                                           const uint32_t* __restrict__ bsum,
                                                           int64_t n) {
    __shared__ uint32_t wsum[SC_THREADS / 64];
    const int64_t base = (int64_t)blockIdx.x * SC_TILE + (int64_t)threadIdx.x * SC_ITEMS;
    uint32_t v[SC_ITEMS];
    sc_load8<POPC>(in, base, n, v);
    uint32_t s = 0;
#pragma unroll
    for (int k = 0; k < SC_ITEMS; ++k) { uint32_t t = v[k]; v[k] = s; s += t; }
    const uint32_t incl = wave_scan_incl(s);
    if (lane_id() == 63) wsum[wave_id()] = incl;
    __syncthreads();
    uint32_t woff = 0;
    for (int w = 0; w < wave_id(); ++w) woff += wsum[w];
    const uint32_t off = bsum[blockIdx.x] + woff + incl - s;
    if (base + SC_ITEMS <= n) {
        uint4 a, b;
        a.x = v[0] + off; a.y = v[1] + off; a.z = v[2] + off; a.w = v[3] + off;
        b.x = v[4] + off; b.y = v[5] + off; b.z = v[6] + off; b.w = v[7] + off;
        *reinterpret_cast<uint4*>(out + base) = a;
        *reinterpret_cast<uint4*>(out + base + 4) = b;
    } else {
#pragma unroll
        for (int k = 0; k < SC_ITEMS; ++k)
            if (base + k < n) out[base + k] = v[k] + off;
    }
}

// =====================================================================================
// single-pass exclusive scan: one launch, every element read once and written once.  Tiles are handed out by a
// ticket (the look-back waits for the tiles in FRONT, so the order must be an order of arrival), a tile's prefix
// comes from the decoupled look-back of pch_lookback.h (bounded wait).  The caller provides scan1_ws_u32(n) words
// that are ZERO when the kernel starts - status words + ticket; callers fold that zeroing into a fill or a kernel
// they run anyway, which is what makes this one launch instead of three.
// A tile whose wait ran out of its budget scans with prefix 0 and raises *total to 0xFFFFFFFF (atomicMax).
// =====================================================================================
template <bool POPC>
__global__ __launch_bounds__(SC_THREADS) void scan1_k(const uint32_t* in, uint32_t* out, int64_t n,
                                                      uint64_t* __restrict__ status,
                                                      uint32_t* __restrict__ ticket,
                                                      uint32_t* __restrict__ total) {
    __shared__ uint32_t wsum[SC_THREADS / 64];
    __shared__ uint32_t tile_sh, excl_sh;
    if (threadIdx.x == 0) tile_sh = atomicAdd(ticket, 1u);
    __syncthreads();
    const int64_t tile = tile_sh;
    const int64_t base = tile * SC_TILE + (int64_t)threadIdx.x * SC_ITEMS;
    uint32_t v[SC_ITEMS];
    sc_load8<POPC>(in, base, n, v);
    uint32_t s = 0;
#pragma unroll
    for (int k = 0; k < SC_ITEMS; ++k) { uint32_t t = v[k]; v[k] = s; s += t; }
    const uint32_t incl = wave_scan_incl(s);
    if (lane_id() == 63) wsum[wave_id()] = incl;
    __syncthreads();
    const uint32_t T = wsum[0] + wsum[1] + wsum[2] + wsum[3];
    if (wave_id() == 0) {
        uint32_t e = gf_lookback(status, tile, T);
        if (e == GF_LB_FAILED) {
            e = 0u;
            if (lane_id() == 0 && total) atomicMax(total, 0xFFFFFFFFu);
        } else if (lane_id() == 0 && total && tile == (int64_t)gridDim.x - 1) {
            atomicMax(total, e + T);
        }
        if (lane_id() == 0) excl_sh = e;
    }
    __syncthreads();
    uint32_t woff = 0;
    for (int w = 0; w < wave_id(); ++w) woff += wsum[w];
    const uint32_t off = excl_sh + woff + incl - s;
    if (base + SC_ITEMS <= n) {
        uint4 a, b;
        a.x = v[0] + off; a.y = v[1] + off; a.z = v[2] + off; a.w = v[3] + off;
        b.x = v[4] + off; b.y = v[5] + off; b.z = v[6] + off; b.w = v[7] + off;
        *reinterpret_cast<uint4*>(out + base) = a;
        *reinterpret_cast<uint4*>(out + base + 4) = b;
    } else {
#pragma unroll
        for (int k = 0; k < SC_ITEMS; ++k)
            if (base + k < n) out[base + k] = v[k] + off;
    }
}

size_t scan1_ws_u32(int64_t n) {                       // status words (2 per tile) + ticket, rounded to 16 bytes
    const int64_t nt = ceil_div(n > 0 ? n : 1, SC_TILE);
    return (size_t)((2 * nt + 2 + 3) & ~int64_t(3));
}

// One ticket per 2048-element tile on ONE address: that word hands out ~88 tickets per microsecond, i.e. at most
// ~1.4 TB/s of scan traffic - measured 86.7 us for 9.8 M elements against 31 us for the three-launch scan, but 6.4
// and 13 us against 14 and 16 us for 0.3 M and 1.1 M elements.  Callers pick by size (scan1_pays).
bool scan1_pays(int64_t n) { return n <= (int64_t(1) << 20) + (int64_t(1) << 18); }

template <bool POPC>
static int scan1_launch(const uint32_t* in, uint32_t* out, int64_t n, uint32_t* zeroed_ws, uint32_t* total,
                        hipStream_t s) {
    if (n <= 0) return PCH_OK;                         // `total` is zero already (it is part of the caller's fill)
    const int64_t nt = ceil_div(n, SC_TILE);
    uint64_t* status = reinterpret_cast<uint64_t*>(zeroed_ws);
    uint32_t* ticket = zeroed_ws + 2 * nt;
    PCH_LAUNCH("scan1", scan1_k<POPC>, dim3((unsigned)nt), dim3(SC_THREADS), 0, s, in, out, n, status, ticket, total);
    return PCH_OK;
}
int scan1_exclusive_u32(const uint32_t* in, uint32_t* out, int64_t n, uint32_t* zeroed_ws, uint32_t* total,
                        hipStream_t s) {
    return scan1_launch<false>(in, out, n, zeroed_ws, total, s);
}
int scan1_exclusive_popc_u32(const uint32_t* bits, uint32_t* out, int64_t n, uint32_t* zeroed_ws, uint32_t* total,
                             hipStream_t s) {
    return scan1_launch<true>(bits, out, n, zeroed_ws, total, s);
}

size_t scan_ws_u32(int64_t n) {
    int64_t nb = ceil_div(n > 0 ? n : 1, SC_TILE);
    return (size_t)nb + 64;
}

template <bool POPC>
static int scan_launch(const uint32_t* in, uint32_t* out, int64_t n, uint32_t* ws, uint32_t* total, hipStream_t s) {
    if (n <= 0) {
        if (total) PCH_HIP_TRY(hipMemsetAsync(total, 0, sizeof(uint32_t), s));
        return PCH_OK;
    }
    const int64_t nb = ceil_div(n, SC_TILE);
    PCH_LAUNCH("scan_reduce", scan_reduce_k<POPC>, dim3((unsigned)nb), dim3(SC_THREADS), 0, s, in, ws, n);
    PCH_LAUNCH("scan_bsums", scan_bsums_k, dim3(1), dim3(1024), 0, s, ws, nb, total);
    PCH_LAUNCH("scan_apply", scan_apply_k<POPC>, dim3((unsigned)nb), dim3(SC_THREADS), 0, s, in, out,
               (const uint32_t*)ws, n);
    return PCH_OK;
}

int scan_exclusive_u32(const uint32_t* in, uint32_t* out, int64_t n, uint32_t* ws,
                       uint32_t* total, hipStream_t s) {
    return scan_launch<false>(in, out, n, ws, total, s);
}

int scan_tile_sums_u32(uint32_t* tile_sums, int64_t nb, uint32_t* total, hipStream_t s) {
    PCH_LAUNCH("scan_bsums", scan_bsums_k, dim3(1), dim3(1024), 0, s, tile_sums, nb, total);
    return PCH_OK;
}

int scan_exclusive_popc_u32(const uint32_t* bits, uint32_t* out, int64_t n, uint32_t* ws,
                            uint32_t* total, hipStream_t s) {
    return scan_launch<true>(bits, out, n, ws, total, s);
}

// =====================================================================================
// LSD radix sort, 8 bits per pass.  Per pass: digit histogram per workgroup tile ->
// exclusive scan of the [digit][tile] table -> stable scatter.  Stability inside a tile
// comes from wave-level digit matching (8 ballots) + per-wave digit counters in LDS.
// =====================================================================================
constexpr int RS_THREADS = 256;
constexpr int RS_ROUNDS  = 8;
constexpr int RS_TILE    = RS_THREADS * RS_ROUNDS;   // 2048 keys per workgroup
constexpr int RS_WAVES   = RS_THREADS / 64;

__global__ __launch_bounds__(RS_THREADS) void rs_hist_k(const uint64_t* __restrict__ keys,
                                                        int64_t n, int shift, uint32_t mask,
                                                        uint32_t* __restrict__ hist, int64_t nb) {
    __shared__ uint32_t h[256];
    h[threadIdx.x] = 0;
    __syncthreads();
    const int64_t base = (int64_t)blockIdx.x * RS_TILE;
#pragma unroll
    for (int r = 0; r < RS_ROUNDS; ++r) {
        const int64_t i = base + r * RS_THREADS + threadIdx.x;
        if (i < n) atomicAdd(&h[(uint32_t)(keys[i] >> shift) & mask], 1u);
    }
    __syncthreads();
    hist[(int64_t)threadIdx.x * nb + blockIdx.x] = h[threadIdx.x];
}

__global__ __launch_bounds__(RS_THREADS) void rs_scatter_k(
    const uint64_t* __restrict__ kin, const uint32_t* __restrict__ vin,
    uint64_t* __restrict__ kout, uint32_t* __restrict__ vout, int64_t n, int shift,
    uint32_t mask, const uint32_t* __restrict__ offs, int64_t nb) {
    __shared__ uint32_t cnt[RS_WAVES][256];
    const int w = wave_id(), l = lane_id();
    for (int j = threadIdx.x; j < RS_WAVES * 256; j += RS_THREADS) (&cnt[0][0])[j] = 0;
    __syncthreads();

    const int64_t seg = (int64_t)blockIdx.x * RS_TILE + (int64_t)w * (64 * RS_ROUNDS);
    uint64_t key[RS_ROUNDS];
    uint32_t val[RS_ROUNDS];
    uint32_t rank[RS_ROUNDS];
    const uint64_t lt = lanemask_lt();
#pragma unroll
    for (int r = 0; r < RS_ROUNDS; ++r) {
        const int64_t i = seg + r * 64 + l;
        const bool valid = i < n;
        key[r] = valid ? kin[i] : ~0ull;
        val[r] = valid ? vin[i] : 0u;
        const uint32_t d = (uint32_t)(key[r] >> shift) & mask;
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
    {
        const int d = threadIdx.x;
        uint32_t run = offs[(int64_t)d * nb + blockIdx.x];
#pragma unroll
        for (int w2 = 0; w2 < RS_WAVES; ++w2) {
            const uint32_t c = cnt[w2][d];
            cnt[w2][d] = run;
            run += c;
        }
    }
    __syncthreads();
#pragma unroll
    for (int r = 0; r < RS_ROUNDS; ++r) {
        const int64_t i = seg + r * 64 + l;
        if (i < n) {
            const uint32_t d = (uint32_t)(key[r] >> shift) & mask;
            const uint32_t pos = cnt[w][d] + rank[r];
            kout[pos] = key[r];
            vout[pos] = val[r];
        }
    }
}

static inline int rs_passes(int nbits) { return nbits <= 0 ? 0 : (nbits + 7) / 8; }

size_t radix_ws_u32(int64_t n) {
    const int64_t nb = ceil_div(n > 0 ? n : 1, RS_TILE);
    const int64_t table = 256 * nb;
    return (size_t)table + scan_ws_u32(table) + 64;
}

int radix_sort_result_buffer(int nbits) { return rs_passes(nbits) & 1; }

int radix_sort_pairs(uint64_t* k0, uint32_t* v0, uint64_t* k1, uint32_t* v1, int64_t n,
                     int nbits, uint32_t* ws, hipStream_t s) {
    if (n <= 0) return PCH_OK;
    const int64_t nb = ceil_div(n, RS_TILE);
    const int64_t table = 256 * nb;
    uint32_t* hist = ws;
    uint32_t* scan_ws = ws + table;
    uint64_t* kin = k0; uint32_t* vin = v0;
    uint64_t* kout = k1; uint32_t* vout = v1;
    const int passes = rs_passes(nbits);
    for (int p = 0; p < passes; ++p) {
        const int shift = 8 * p;
        const int bits = (nbits - shift) < 8 ? (nbits - shift) : 8;
        const uint32_t mask = (1u << bits) - 1u;
        PCH_LAUNCH("radix_hist", rs_hist_k, dim3((unsigned)nb), dim3(RS_THREADS), 0, s,
                   (const uint64_t*)kin, n, shift, mask, hist, nb);
        PCH_TRY(scan_exclusive_u32(hist, hist, table, scan_ws, nullptr, s));
        PCH_LAUNCH("radix_scatter", rs_scatter_k, dim3((unsigned)nb), dim3(RS_THREADS), 0, s,
                   (const uint64_t*)kin, (const uint32_t*)vin, kout, vout, n, shift, mask,
                   (const uint32_t*)hist, nb);
        uint64_t* tk = kin; kin = kout; kout = tk;
        uint32_t* tv = vin; vin = vout; vout = tv;
    }
    return PCH_OK;
}

}  // namespace pch
