// Shared host/device helpers for libpch_hip.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stddef.h>
#include <stdio.h>
#include <string.h>

#include "../../include/pch_hip.h"

namespace pch {

// ---------------------------------------------------------------- errors
void set_error(const char* fmt, ...);

#define PCH_HIP_TRY(expr)                                                         \
    do {                                                                          \
        hipError_t _e = (expr);                                                   \
        if (_e != hipSuccess) {                                                   \
            ::pch::set_error("%s failed: %s (%s:%d)", #expr, hipGetErrorString(_e), \
                             __FILE__, __LINE__);                                 \
            return PCH_ERR_HIP;                                                   \
        }                                                                         \
    } while (0)

#define PCH_TRY(expr)                  \
    do {                               \
        int _rc = (expr);              \
        if (_rc != PCH_OK) return _rc; \
    } while (0)

#define PCH_REQUIRE(cond, msg)                                   \
    do {                                                         \
        if (!(cond)) {                                           \
            ::pch::set_error("%s: %s", __func__, msg);           \
            return PCH_ERR_ARG;                                  \
        }                                                        \
    } while (0)

// ---------------------------------------------------------------- device guard
// Every entry point runs on the device that owns its buffers, whatever device is current in the
// calling thread: the guard looks the pointer up (hipPointerGetAttributes), switches with
// hipSetDevice and restores the caller's device on exit.  `rc` is PCH_OK, PCH_ERR_ARG (host
// pointer) or PCH_ERR_HIP.
constexpr int PCH_MAX_DEVICES = 16;
struct DeviceGuard {
    int rc, prev, dev;
    explicit DeviceGuard(const void* device_ptr);
    ~DeviceGuard();
    DeviceGuard(const DeviceGuard&) = delete;
    DeviceGuard& operator=(const DeviceGuard&) = delete;
};
// device the calling thread currently has selected, clamped to [0, PCH_MAX_DEVICES)
int current_device_slot();
// Per-thread HIP objects (events, pinned peek buffers, side streams) are released by thread-local destructors.
// For the main thread those run during process exit, when the HIP runtime (owned by torch's libamdhip64) may
// already be tearing down: calling into it then is undefined.  False on the main thread and once exit has begun;
// the destructors then leave the objects to the operating system.
bool may_release_hip_objects();
#define PCH_DEVICE_GUARD(ptr)                 \
    ::pch::DeviceGuard _pch_guard(ptr);       \
    if (_pch_guard.rc != PCH_OK) return _pch_guard.rc

// ---------------------------------------------------------------- workspace arena
// The same plan function is run once with base == nullptr (size query) and once with the
// caller's buffer, so *_ws_bytes() can never disagree with the real carve-up.
//
// A workspace whose contents a LATER call continues (pch_dbscan_relabel_i32 / pch_dbscan_first_core_rows_i32 on
// the grid pch_dbscan_f32 left behind) is remembered per thread.  Every entry point carves its workspace through
// an Arena, and carving a range that overlaps the remembered one drops it (ws_touched, pch_dbscan.hip): the
// continuation then fails with PCH_ERR_ARG instead of following overwritten indices on the device.  The
// continuation itself builds its Arena with continues = true.
void ws_touched(const void* base, size_t bytes);
struct Arena {
    char*  base;
    size_t cap;
    size_t off;
    bool   overflow;
    explicit Arena(void* b = nullptr, size_t c = 0, bool continues = false)
        : base(static_cast<char*>(b)), cap(c), off(0), overflow(false) {
        if (b && !continues) ws_touched(b, c);
    }
    template <typename T>
    T* take(size_t count) {
        size_t bytes = (count * sizeof(T) + 255) & ~size_t(255);
        size_t at = off;
        off += bytes;
        if (!base) return nullptr;
        if (off > cap) { overflow = true; return nullptr; }
        return reinterpret_cast<T*>(base + at);
    }
    size_t mark() const { return off; }
    void   reset(size_t m) { off = m; }
};

// ---------------------------------------------------------------- small device -> host reads
// A few words the host needs to size the next launch, without draining the stream: the copy lands
// in a pinned per-thread, per-device buffer and the host waits for an event recorded right behind it, so
// kernels enqueued after the copy keep the GPU busy meanwhile.  (pch_core.hip)
struct HostPeek {
    void*      pinned;          // 256 bytes of pinned host memory
    hipEvent_t ev;
    bool       ok;
};
HostPeek& host_peek();
// enqueue: copy `bytes` (<= 256) from dev to the pinned buffer and record the event
int peek_enqueue(const void* dev, size_t bytes, hipStream_t s);
// wait for the copy, then memcpy to dst
int peek_wait(void* dst, size_t bytes);

// By-product of stage C for the grouping stage (D0): per-cluster bounding boxes accumulated by the kernels that
// make the labels.  acc: [cap][8] uint32 (device; zeroed by the run), row k = {max ~ordered(min xyz), max ordered(max
// xyz), 0, 0} - decoded in place by the grouping stage (segment_run, stats_encoded).  done: set by the run when the
// table was filled (not on the per-chunk fallback for grids beyond the 64-bit key).
struct DbBoxOut { uint32_t* acc; int32_t cap; bool done; };
// stage B (pch_filter.hip): pch_ground_filter_f32's body.  early (optional): `bytes` device bytes at `dev` - they must
// cover out_scalars, out_count and out_aabb - are queued for the host (peek_enqueue) directly behind the FIRST sweep and
// its gf_finalize, i.e. in front of the launches of the fallback threshold, which return at once unless fewer than
// min_keep rows survived.  The caller waits for that copy (peek_wait); if out_scalars[5] (fallback used) is set in
// it, the values are not final yet and it queues and waits for a second copy, which then lies behind everything.
struct GfEarly { const void* dev; size_t bytes; };
int ground_filter_run(const float* raw, int64_t n, double pct, float offset, float fallback_offset, int64_t min_keep,
                      float* out_points, int32_t* out_index, float* out_scalars, int64_t* out_count, float* out_aabb,
                      void* ws, size_t ws_bytes, hipStream_t s, const GfEarly* early);
// stage C with an early host copy of the cluster count (pch_dbscan.hip); k_host and box may be null
int dbscan_run(const float* xyz, int64_t n, double eps, int32_t min_samples, int64_t chunk_size,
               const float* aabb_host, int32_t* labels, uint8_t* core, int32_t* out_nclusters,
               void* ws, size_t ws_bytes, hipStream_t s, int32_t* k_host, DbBoxOut* box = nullptr);
// stage D0 (pch_segment.hip); stats_encoded: out_stats already holds the table a DbBoxOut run left there
int segment_run(const int32_t* labels, const float* xyz, int64_t n, int32_t nclusters, int32_t* out_perm,
                int64_t* out_offsets, float* out_stats, void* ws, size_t ws_bytes, hipStream_t s, bool stats_encoded);

// ---------------------------------------------------------------- profiling
// When enabled every PCH_LAUNCH is bracketed by hipEvents recorded on the launch stream.
bool prof_enabled();
bool prof_wanted(const char* name);             // name filter (pch_set_profiling_filter)
void prof_pre(const char* name, hipStream_t s);
void prof_post(hipStream_t s);

#define PCH_LAUNCH(name, kernel, grid, block, shmem, stream, ...)                  \
    do {                                                                           \
        const bool _prof = ::pch::prof_enabled() && ::pch::prof_wanted(name);            \
        if (_prof) ::pch::prof_pre(name, stream);                                  \
        hipLaunchKernelGGL(kernel, grid, block, shmem, stream, __VA_ARGS__);       \
        if (_prof) ::pch::prof_post(stream);                                       \
        hipError_t _le = hipGetLastError();                                        \
        if (_le != hipSuccess) {                                                   \
            ::pch::set_error("launch %s failed: %s", name, hipGetErrorString(_le)); \
            return PCH_ERR_HIP;                                                    \
        }                                                                          \
    } while (0)

static inline int64_t ceil_div(int64_t a, int64_t b) { return (a + b - 1) / b; }
static inline int bits_for(uint64_t count) {   // bits needed to store values in [0, count)
    int b = 0;
    while (b < 64 && (uint64_t(1) << b) < count) ++b;
    return b;
}

// ---------------------------------------------------------------- device helpers
#ifdef __HIPCC__
constexpr int WAVE = 64;

__device__ __forceinline__ int lane_id() { return threadIdx.x & 63; }
__device__ __forceinline__ int wave_id() { return threadIdx.x >> 6; }

// order-preserving float <-> uint transforms (for atomicMin/Max and radix select)
__device__ __forceinline__ uint32_t f32_ordered(float f) {
    uint32_t u = __float_as_uint(f);
    return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}
__device__ __forceinline__ float f32_unordered(uint32_t k) {
    uint32_t u = (k & 0x80000000u) ? (k & 0x7fffffffu) : ~k;
    return __uint_as_float(u);
}
__device__ __forceinline__ uint64_t f64_ordered(double d) {
    uint64_t u = (uint64_t)__double_as_longlong(d);
    return (u & 0x8000000000000000ull) ? ~u : (u | 0x8000000000000000ull);
}
__device__ __forceinline__ double f64_unordered(uint64_t k) {
    uint64_t u = (k & 0x8000000000000000ull) ? (k & 0x7fffffffffffffffull) : ~k;
    return __longlong_as_double((long long)u);
}

__device__ __forceinline__ uint64_t lanemask_lt() {
    return (uint64_t(1) << lane_id()) - 1;
}

template <typename T>
__device__ __forceinline__ T wave_reduce_add(T v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
template <typename T>
__device__ __forceinline__ T wave_reduce_min(T v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) { T w = __shfl_xor(v, o, 64); v = w < v ? w : v; }
    return v;
}
template <typename T>
__device__ __forceinline__ T wave_reduce_max(T v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) { T w = __shfl_xor(v, o, 64); v = w > v ? w : v; }
    return v;
}
// inclusive wave scan
template <typename T>
__device__ __forceinline__ T wave_scan_incl(T v) {
    const int l = lane_id();
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) { T w = __shfl_up(v, o, 64); if (l >= o) v += w; }
    return v;
}
// Rank of this lane among the VALID lanes of its wave that hold the same BITS-bit digit (lower lanes first), and the
// number of such lanes: the stable in-wave ranking of every LSD / MSD pass.  What is accumulated is the set of lanes
// that DIFFER from this one in some bit: per bit one sign-extended bit field B (0 / -1), one ballot m, and (m ^ B)
// or-ed into each 32-bit half - five vector instructions per bit (the `bit ? m : ~m` form compiles to nine or ten).
template <int BITS>
__device__ __forceinline__ uint32_t wave_match(uint32_t d, bool valid, uint32_t& peers_out) {
    uint32_t nlo = 0, nhi = 0;
#pragma unroll
    for (int b = 0; b < BITS; ++b) {
        const int32_t B = ((int32_t)(d << (31 - b))) >> 31;
        const uint64_t m = __ballot(B != 0);
        nlo |= (uint32_t)m ^ (uint32_t)B;
        nhi |= (uint32_t)(m >> 32) ^ (uint32_t)B;
    }
    const uint64_t v = __ballot(valid);
    const uint32_t plo = (uint32_t)v & ~nlo, phi = (uint32_t)(v >> 32) & ~nhi;
    peers_out = (uint32_t)__popc(plo) + (uint32_t)__popc(phi);
    return __builtin_amdgcn_mbcnt_hi(phi, __builtin_amdgcn_mbcnt_lo(plo, 0u));
}
#endif  // __HIPCC__

}  // namespace pch
