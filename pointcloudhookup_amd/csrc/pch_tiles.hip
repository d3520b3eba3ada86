// Cross-tile label reconciliation, device side (BASELINE config 4; generalises the per-chunk label offsets of
// utils/tower_extraction.py:113-116 and the 30 m duplicate rule :153-162 to tiles that share points).
//
// pch_strip_lattice_reps_f32: for up to two strips [x_from, x_to) of a tile, one representative per LATTICE cell that
// holds a core point of the strip: the smallest row among them.  The lattice is the same on every rank - cell =
// floor(double(coordinate) * (1 / side)) per axis, anchored at the origin of the shared (centred) frame, NOT at the
// tile's own box - so two tiles that both hold a strip's points with the same core flags name the same rows
// (pointcloudhookup_amd/tiles.py: the join on the row is the cross-tile link).  Two points of one cell are closer than
// eps (side = eps / sqrt(3) * (1 - 2^-16)), so the core points of a cell are one cluster in either tile.
//
//   sl_insert_k  points-parallel: a row of a strip that is core and labelled hashes its (strip, cell) key into an open-
//                addressing table (64-bit CAS on the key word) and folds its LOCAL row into the slot with atomicMin -
//                local rows ascend with the global ones, so the smallest local row is the smallest global row.
//                Everything else leaves after one 4-byte and two 1-byte loads.
//   sl_emit_k    table-parallel: every occupied slot appends (global row, label) of its row to its strip's output.
#include "pch_common.h"

namespace pch {

constexpr int SL2_THREADS = 256;
constexpr unsigned long long SL2_EMPTY = ~0ull;

struct SlStrips { float a[2], b[2]; int n; };

__device__ __forceinline__ uint32_t sl2_hash(unsigned long long k, uint32_t mask) {
    k ^= k >> 33; k *= 0xff51afd7ed558ccdull; k ^= k >> 33; k *= 0xc4ceb9fe1a85ec53ull; k ^= k >> 33;
    return (uint32_t)k & mask;
}

__global__ __launch_bounds__(SL2_THREADS) void sl_insert_k(const float* __restrict__ xyz, const int32_t* __restrict__ labels,
                                                           const uint8_t* __restrict__ core, int64_t n, SlStrips st,
                                                           double inv_side, unsigned long long* __restrict__ keys,
                                                           uint32_t* __restrict__ minrow, uint32_t mask,
                                                           uint32_t* __restrict__ flags) {
    const int64_t i = (int64_t)blockIdx.x * SL2_THREADS + threadIdx.x;
    if (i >= n) return;
    const float x = xyz[3 * i];
    int s = -1;
    if (st.n > 0 && x >= st.a[0] && x < st.b[0]) s = 0;
    else if (st.n > 1 && x >= st.a[1] && x < st.b[1]) s = 1;
    if (s < 0 || !core[i] || labels[i] < 0) return;
    // 21 bits per axis around the origin: +-2^20 cells (+-4.8e6 m at eps = 8 m); beyond that the call fails
    const double cx = floor((double)x * inv_side), cy = floor((double)xyz[3 * i + 1] * inv_side),
                 cz = floor((double)xyz[3 * i + 2] * inv_side);
    const double lim = 1048576.0;
    if (!(cx >= -lim && cx < lim && cy >= -lim && cy < lim && cz >= -lim && cz < lim)) { atomicOr(flags, 1u); return; }
    const unsigned long long key = ((unsigned long long)s << 63) | ((unsigned long long)((long long)cx + 1048576ll) << 42) |
                                   ((unsigned long long)((long long)cy + 1048576ll) << 21) |
                                   (unsigned long long)((long long)cz + 1048576ll);
    uint32_t slot = sl2_hash(key, mask);
    for (uint32_t probe = 0; probe <= mask; ++probe) {
        const unsigned long long old = atomicCAS(&keys[slot], SL2_EMPTY, key);
        if (old == SL2_EMPTY || old == key) { atomicMin(&minrow[slot], (uint32_t)i); return; }
        slot = (slot + 1) & mask;
    }
    atomicOr(flags, 2u);                                  // table full
}

__global__ __launch_bounds__(SL2_THREADS) void sl_emit_k(const unsigned long long* __restrict__ keys,
                                                         const uint32_t* __restrict__ minrow, uint32_t nslots,
                                                         const int64_t* __restrict__ rows, const int32_t* __restrict__ labels,
                                                         int32_t cap, int64_t* __restrict__ out_rows,
                                                         int32_t* __restrict__ out_labels, int32_t* __restrict__ out_count) {
    const uint32_t j = blockIdx.x * SL2_THREADS + threadIdx.x;
    if (j >= nslots) return;
    const unsigned long long k = keys[j];
    if (k == SL2_EMPTY) return;
    const int s = (int)(k >> 63);
    const int32_t at = atomicAdd(&out_count[s], 1);
    if (at < cap) {
        const uint32_t r = minrow[j];
        out_rows[(int64_t)s * cap + at] = rows[r];
        out_labels[(int64_t)s * cap + at] = labels[r];
    }
}

static uint32_t sl2_slots(int32_t cap) {                  // power of two >= 4 x (2 strips x cap): load factor <= 1/4
    uint32_t h = 1024;
    while (h < 8u * (uint32_t)(cap > 0 ? cap : 1)) h <<= 1;
    return h;
}

}  // namespace pch

using namespace pch;

extern "C" size_t pch_strip_lattice_reps_ws_bytes(int32_t cap) {
    if (cap < 0) return 0;
    Arena a;
    const uint32_t h = sl2_slots(cap);
    a.take<unsigned long long>(h);
    a.take<uint32_t>(h);
    a.take<uint32_t>(64);
    return a.off;
}

extern "C" int pch_strip_lattice_reps_f32(const float* xyz, const int64_t* rows, const int32_t* labels,
                                          const uint8_t* core, int64_t n, int32_t nstrips, const float* strips_host,
                                          double eps, int32_t cap, int64_t* out_rows, int32_t* out_labels,
                                          int32_t* out_count, void* ws, size_t ws_bytes, void* stream) {
    PCH_DEVICE_GUARD(out_count);
    hipStream_t s = (hipStream_t)stream;
    PCH_REQUIRE(n >= 0 && n < (int64_t(1) << 32) - 1 && nstrips >= 0 && nstrips <= 2 && cap >= 0 && out_count, "bad argument");
    PCH_REQUIRE(nstrips == 0 || strips_host, "strips_host is null");
    PCH_REQUIRE(eps > 0.0, "eps must be > 0");
    PCH_HIP_TRY(hipMemsetAsync(out_count, 0, 4 * sizeof(int32_t), s));     // [0], [1]: cells per strip; [2]: flags
    if (n == 0 || nstrips == 0) return PCH_OK;
    PCH_REQUIRE(xyz && rows && labels && core && ws && (cap == 0 || (out_rows && out_labels)), "null buffer");
    Arena a(ws, ws_bytes);
    const uint32_t h = sl2_slots(cap);
    unsigned long long* keys = a.take<unsigned long long>(h);
    uint32_t* minrow = a.take<uint32_t>(h);
    a.take<uint32_t>(64);
    if (a.overflow) { set_error("workspace too small: need %zu bytes", a.off); return PCH_ERR_WORKSPACE; }
    SlStrips st;
    st.n = nstrips;
    for (int k = 0; k < 2; ++k) {
        st.a[k] = k < nstrips ? strips_host[2 * k] : 0.0f;
        st.b[k] = k < nstrips ? strips_host[2 * k + 1] : 0.0f;
    }
    PCH_HIP_TRY(hipMemsetAsync(keys, 0xFF, sizeof(unsigned long long) * h, s));
    PCH_HIP_TRY(hipMemsetAsync(minrow, 0xFF, sizeof(uint32_t) * h, s));
    const double side = eps / sqrt(3.0) * (1.0 - ldexp(1.0, -16));
    PCH_LAUNCH("sl_insert", sl_insert_k, dim3((unsigned)ceil_div(n, SL2_THREADS)), dim3(SL2_THREADS), 0, s, xyz, labels,
               core, n, st, 1.0 / side, keys, minrow, h - 1, reinterpret_cast<uint32_t*>(out_count + 2));
    PCH_LAUNCH("sl_emit", sl_emit_k, dim3((unsigned)ceil_div((int64_t)h, SL2_THREADS)), dim3(SL2_THREADS), 0, s,
               (const unsigned long long*)keys, (const uint32_t*)minrow, h, rows, labels, cap, out_rows, out_labels,
               out_count);
    return PCH_OK;
}
