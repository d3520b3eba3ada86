// Viewer-side helpers the reference runs on the host over the whole cloud at every redraw
// (SURVEY.md section 8f-3): the axis-aligned crop of a tower's box and the random preview decimation.
//   crop     : points[(x>=x0)&(x<=x1)&(y>=y0)&(y<=y1)&(z>=z0)&(z<=z1)]     test/kuangxuan.py:69-79
//   decimate : points[np.random.choice(len(points), k, replace=False)]      pyGUI_towers_test.py:174-177,
//                                                                           ui/vtk_widget.py:115-118
#include "pch_common.h"
#include "pch_lookback.h"

namespace pch {

constexpr int CR_THREADS = 256;
constexpr int CR_ROUNDS  = 8;
constexpr int CR_TILE    = CR_THREADS * CR_ROUNDS;          // 2048 rows per workgroup

struct CropBox { double lo[3], hi[3]; };
struct Row64 { double x, y, z; };

struct CropState { uint32_t ticket, total, pad[2]; };

// order-preserving compaction in one sweep: tiles are taken by ticket (order of arrival), the output
// offset of a tile comes from a decoupled look-back over the tiles in front of it
__global__ __launch_bounds__(CR_THREADS) void crop_aabb_k(const double* __restrict__ xyz, int64_t n, CropBox box,
                                                          CropState* __restrict__ st, uint64_t* __restrict__ status,
                                                          double* __restrict__ out_points,
                                                          int64_t* __restrict__ out_index,
                                                          int64_t* __restrict__ out_count) {
    __shared__ uint32_t wtot[CR_THREADS / 64];
    __shared__ uint32_t tile_sh, excl_sh;
    if (threadIdx.x == 0) tile_sh = atomicAdd(&st->ticket, 1u);
    __syncthreads();
    const int64_t tile = tile_sh;
    const int w = wave_id(), l = lane_id();
    const int64_t seg = tile * CR_TILE + (int64_t)w * (64 * CR_ROUNDS);
    const Row64* __restrict__ rows = reinterpret_cast<const Row64*>(xyz);
    Row64 q[CR_ROUNDS];
    unsigned long long m[CR_ROUNDS];
    uint32_t run = 0;
#pragma unroll
    for (int r = 0; r < CR_ROUNDS; ++r) {
        const int64_t i = seg + r * 64 + l;
        q[r] = rows[i < n ? i : 0];
    }
#pragma unroll
    for (int r = 0; r < CR_ROUNDS; ++r) {
        const int64_t i = seg + r * 64 + l;
        const bool keep = i < n && q[r].x >= box.lo[0] && q[r].x <= box.hi[0] && q[r].y >= box.lo[1] &&
                          q[r].y <= box.hi[1] && q[r].z >= box.lo[2] && q[r].z <= box.hi[2];
        m[r] = __ballot(keep);
        run += (uint32_t)__popcll(m[r]);
    }
    if (l == 0) wtot[w] = run;
    __syncthreads();
    const uint32_t T = wtot[0] + wtot[1] + wtot[2] + wtot[3];
    if (w == 0) {
        const uint32_t e0 = gf_lookback(status, tile, T);
        const bool lb_failed = e0 == GF_LB_FAILED;
        const uint32_t e = lb_failed ? 0u : e0;              // prefix 0 keeps the writes below inside the output
        if (l == 0) {
            excl_sh = e;
            // out_count starts at 0: the last tile adds the total (< 2^62), a tile whose wait ran out of its budget
            // sets the sign bit - idempotent, so the word reads negative iff some tile failed, however many did
            // (every tile behind a poisoned one fails too) and in whatever order the two happen
            unsigned long long* oc = reinterpret_cast<unsigned long long*>(out_count);
            if (lb_failed) atomicOr(oc, 1ull << 63);
            else if (tile == (int64_t)gridDim.x - 1) atomicAdd(oc, (unsigned long long)e + T);
        }
    }
    __syncthreads();
    if (T == 0) return;
    uint32_t woff = excl_sh;
    for (int w2 = 0; w2 < w; ++w2) woff += wtot[w2];
    const uint64_t lt = lanemask_lt();
#pragma unroll
    for (int r = 0; r < CR_ROUNDS; ++r) {
        if ((m[r] >> l) & 1ull) {
            const int64_t o = (int64_t)woff + (uint32_t)__popcll(m[r] & lt);
            reinterpret_cast<Row64*>(out_points)[o] = q[r];
            if (out_index) out_index[o] = seg + r * 64 + l;
        }
        woff += (uint32_t)__popcll(m[r]);
    }
}

// ---- seeded sampling without replacement: a keyed bijection of [0, 2^b) (four Feistel rounds) walked
// until it lands inside [0, n) maps 0..k-1 to k distinct rows
__device__ __forceinline__ uint32_t dc_round(uint32_t x, uint32_t key) {
    x ^= key;
    x *= 0x9E3779B1u;
    x ^= x >> 15;
    x *= 0x85EBCA77u;
    x ^= x >> 13;
    return x;
}
__device__ __forceinline__ uint64_t dc_permute(uint64_t v, int half_bits, const uint32_t (&keys)[4]) {
    const uint32_t mask = half_bits >= 32 ? 0xFFFFFFFFu : ((1u << half_bits) - 1u);
    uint32_t L = (uint32_t)(v >> half_bits) & mask, R = (uint32_t)v & mask;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const uint32_t t = L ^ (dc_round(R, keys[r]) & mask);
        L = R;
        R = t;
    }
    return ((uint64_t)L << half_bits) | R;
}
__global__ void decimate_k(const double* __restrict__ xyz, int64_t n, int64_t k, int half_bits, uint64_t seed,
                           double* __restrict__ out_points, int64_t* __restrict__ out_index) {
    const int64_t j = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= k) return;
    const uint32_t keys[4] = {(uint32_t)seed, (uint32_t)(seed >> 32), (uint32_t)seed * 0x632BE5ABu + 0x9E3779B9u,
                              (uint32_t)(seed >> 32) * 0x85EBCA6Bu + 0xC2B2AE35u};
    uint64_t v = (uint64_t)j;
    do { v = dc_permute(v, half_bits, keys); } while (v >= (uint64_t)n);     // cycle walking: expected < 4 steps
    if (out_index) out_index[j] = (int64_t)v;
    if (out_points) reinterpret_cast<Row64*>(out_points)[j] = reinterpret_cast<const Row64*>(xyz)[v];
}

}  // namespace pch

using namespace pch;

extern "C" size_t pch_crop_aabb_ws_bytes(int64_t n) {
    if (n < 0) return 0;
    Arena a;
    a.take<CropState>(1);
    a.take<uint64_t>(ceil_div(n > 0 ? n : 1, CR_TILE));
    return a.off;
}

extern "C" int pch_crop_aabb_f64(const double* xyz, int64_t n, const double* min3_host, const double* max3_host,
                                 double* out_points, int64_t* out_index, int64_t* out_count, void* ws,
                                 size_t ws_bytes, void* stream) {
    PCH_DEVICE_GUARD(out_count);
    hipStream_t s = (hipStream_t)stream;
    PCH_REQUIRE(n >= 0 && n < (int64_t(1) << 32) && min3_host && max3_host && out_count, "bad argument");
    if (n == 0) {
        PCH_HIP_TRY(hipMemsetAsync(out_count, 0, sizeof(int64_t), s));
        return PCH_OK;
    }
    PCH_REQUIRE(xyz && out_points && ws, "null buffer");
    Arena a(ws, ws_bytes);
    CropState* st = a.take<CropState>(1);
    const int64_t nt = ceil_div(n, CR_TILE);
    uint64_t* status = a.take<uint64_t>(nt);
    if (a.overflow) { set_error("workspace too small: need %zu bytes", a.off); return PCH_ERR_WORKSPACE; }
    PCH_HIP_TRY(hipMemsetAsync(ws, 0, a.off, s));
    PCH_HIP_TRY(hipMemsetAsync(out_count, 0, sizeof(int64_t), s));
    CropBox box;
    for (int k = 0; k < 3; ++k) { box.lo[k] = min3_host[k]; box.hi[k] = max3_host[k]; }
    PCH_LAUNCH("crop_aabb", crop_aabb_k, dim3((unsigned)nt), dim3(CR_THREADS), 0, s, xyz, n, box, st, status,
               out_points, out_index, out_count);
    return PCH_OK;
}

// ---- self-test of the bounded look-back wait (pch_lookback.h): eight tiles in ticket order, the workgroup that
// draws ticket 1 leaves WITHOUT publishing - what a lost or never-scheduled tile looks like to the tiles behind
// it.  Tickets 2..7 must give up after `budget` ticks (the first by its own clock, the rest on its poisoned word),
// poison their words and mark the count word exactly like crop_aabb_k / vx_finish_k do (sign bit, idempotent;
// the last ticket would add the total); the grid drains by construction.  Never part of the data path.
namespace pch {
__global__ __launch_bounds__(64) void lb_selftest_k(uint64_t* __restrict__ status, uint32_t* __restrict__ state,
                                                    unsigned long long* __restrict__ count,
                                                    unsigned long long budget) {
    __shared__ uint32_t tile_sh;
    if (threadIdx.x == 0) tile_sh = atomicAdd(&state[0], 1u);
    __syncthreads();
    const int64_t tile = tile_sh;
    if (tile == 1) return;                               // the tile that never publishes
    const uint32_t e = gf_lookback(status, tile, 1u, false, budget);
    if (threadIdx.x == 0) {
        if (e == GF_LB_FAILED) { atomicAdd(&state[1], 1u); atomicOr(count, 1ull << 63); }
        else {
            atomicAdd(&state[2], 1u);
            if (tile == (int64_t)gridDim.x - 1) atomicAdd(count, (unsigned long long)e + 1u);
        }
    }
}
}  // namespace pch

extern "C" int pch_selftest_lookback_timeout(int budget_ms, void* dev_scratch, size_t scratch_bytes, void* stream) {
    PCH_DEVICE_GUARD(dev_scratch);
    hipStream_t s = (hipStream_t)stream;
    PCH_REQUIRE(budget_ms >= 1 && budget_ms <= 2000 && dev_scratch && scratch_bytes >= 256, "bad argument");
    uint64_t* status = static_cast<uint64_t*>(dev_scratch);
    uint32_t* state = reinterpret_cast<uint32_t*>(status + 8);                 // 8 status words, then 4 state words,
    unsigned long long* count = reinterpret_cast<unsigned long long*>(status + 10);   // then the published count word
    PCH_HIP_TRY(hipMemsetAsync(dev_scratch, 0, 256, s));
    PCH_LAUNCH("lb_selftest", lb_selftest_k, dim3(8), dim3(64), 0, s, status, state, count,
               (unsigned long long)budget_ms * 100000ull);
    uint32_t st[6];                                      // state[0..3] + the 64-bit count word
    PCH_TRY(peek_enqueue(state, sizeof(st), s));
    PCH_TRY(peek_wait(st, sizeof(st)));
    long long published;
    memcpy(&published, &st[4], sizeof(published));
    if (st[0] != 8 || st[1] != 6 || st[2] != 1) {       // ticket 0 succeeds, 1 leaves, 2..7 give up
        set_error("look-back self-test: tickets %u, gave up %u, succeeded %u (expected 8 / 6 / 1)", st[0], st[1], st[2]);
        return PCH_ERR_HIP;
    }
    if (published >= 0) {                                // six failure marks must still read as a failure
        set_error("look-back self-test: the count word reads %lld after six failed tiles (must be negative)", published);
        return PCH_ERR_HIP;
    }
    set_error("look-back wait ran out of its %d ms budget (self-test: expected)", budget_ms);
    return PCH_ERR_TIMEOUT;
}

extern "C" int pch_decimate_f64(const double* xyz, int64_t n, int64_t k, uint64_t seed, double* out_points,
                                int64_t* out_index, void* stream) {
    PCH_DEVICE_GUARD(out_points ? (const void*)out_points : (const void*)out_index);
    PCH_REQUIRE(n >= 0 && k >= 0 && k <= n, "Cannot take a larger sample than population when 'replace=False'");
    if (k == 0) return PCH_OK;
    PCH_REQUIRE(out_points || out_index, "no output buffer");
    PCH_REQUIRE(!out_points || xyz, "null input");
    int bits = bits_for((uint64_t)n);
    if (bits < 2) bits = 2;
    const int half_bits = (bits + 1) / 2;
    PCH_LAUNCH("decimate", decimate_k, dim3((unsigned)ceil_div(k, 256)), dim3(256), 0, (hipStream_t)stream, xyz, n, k,
               half_bits, (uint64_t)seed, out_points, out_index);
    return PCH_OK;
}
