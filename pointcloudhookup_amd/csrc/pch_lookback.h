// Decoupled look-back over an ordered sequence of tiles (single-pass prefix sum across workgroups).
#pragma once
#include "pch_common.h"

namespace pch {

constexpr int GF_LOOK = 4;                        // 64-tile windows fetched per look-back round trip
constexpr uint64_t GF_FLAG_AGG = 1ull << 62, GF_FLAG_INCL = 2ull << 62, GF_FLAG_POISON = 3ull << 62;
// BOUNDED WAIT.  A tile in front normally publishes within microseconds.  The wait is nevertheless bounded by a
// wall-clock budget (wall_clock64: the 100 MHz constant counter), so that the grid ALWAYS drains: a tile that has
// waited longer poisons its own status word - every tile behind it then gives up at once instead of waiting out
// its own budget - and returns GF_LB_FAILED.  The caller must then (a) use prefix 0 (its writes stay inside the
// output, whose capacity covers every tile's count) and (b) raise the call's failure word, which the host turns
// into PCH_ERR_TIMEOUT.  Seconds, not milliseconds: several processes may share the GPU and be time-sliced.
constexpr unsigned long long GF_LB_BUDGET = 400000000ull;        // 4 s of the 100 MHz counter
constexpr uint32_t GF_LB_FAILED = 0xFFFFFFFFu;                   // not a prefix: counts stay below 2^31

// Exclusive prefix of tile b (called by ONE whole wave, all 64 lanes); T = this tile's count.
// status: one zeroed 64-bit word per tile (2-bit flag + 32-bit value, one relaxed agent-scope atomic).
// FORWARD PROGRESS: the caller waits for every tile in front of b, so b must be an order in which
// workgroups START (a ticket drawn on arrival), never blockIdx - HIP promises no dispatch order.
// gf_announce (one lane) may publish the tile's count as soon as it is known - long before the tile needs its
// own prefix - so that tiles behind it do not wait; gf_lookback(..., announced = true) then skips that store.
__device__ __forceinline__ void gf_announce(uint64_t* __restrict__ status, int64_t b, uint32_t T) {
    __hip_atomic_store(&status[b], (b == 0 ? GF_FLAG_INCL : GF_FLAG_AGG) | T, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ uint32_t gf_lookback(uint64_t* __restrict__ status, int64_t b, uint32_t T,
                                                bool announced = false, unsigned long long budget = GF_LB_BUDGET) {
    const int l = lane_id();
    if (b == 0) {
        if (l == 0 && !announced) __hip_atomic_store(&status[0], GF_FLAG_INCL | T, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        return 0;
    }
    if (l == 0 && !announced) __hip_atomic_store(&status[b], GF_FLAG_AGG | T, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    uint32_t excl = 0;
    bool done = false;
    unsigned long long t_first_miss = 0;                // set at the first unsuccessful poll only
#ifdef PCH_LB_COUNT
    extern __device__ unsigned long long g_lb_polls, g_lb_windows;
#endif
    for (int64_t j = b - 1; !done; j -= 64 * GF_LOOK) {
#ifdef PCH_LB_COUNT
        if (l == 0) atomicAdd(&g_lb_windows, 1ull);
#endif  // windows [j-64k-63, j-64k]: lane l looks at tile j-64k-l
        uint64_t v[GF_LOOK];
        bool failed = false;
        do {                                            // tiles in front drew their ticket earlier: they run and publish
            bool missing = false, poisoned = false;
#pragma unroll
            for (int k = 0; k < GF_LOOK; ++k) {
                const int64_t idx = j - 64 * k - l;
                v[k] = idx >= 0 ? __hip_atomic_load(&status[idx], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)
                                : GF_FLAG_INCL;         // in front of tile 0: prefix 0
                missing |= (v[k] >> 62) == 0;
                poisoned |= (v[k] >> 62) == 3;
            }
            if (__ballot(poisoned) != 0) { failed = true; break; }
            if (__ballot(missing) == 0) break;
#ifdef PCH_LB_COUNT
            if (l == 0) atomicAdd(&g_lb_polls, 1ull);
#endif
            const unsigned long long now = wall_clock64();          // wave-uniform (scalar) read
            if (t_first_miss == 0) t_first_miss = now | 1ull;
            else if (now - t_first_miss > budget) { failed = true; break; }
            __builtin_amdgcn_s_sleep(1);
        } while (true);
        if (failed) {
            if (l == 0) __hip_atomic_store(&status[b], GF_FLAG_POISON, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            return GF_LB_FAILED;
        }
#pragma unroll
        for (int k = 0; k < GF_LOOK; ++k) {
            if (done) break;
            const unsigned long long incl = __ballot((v[k] >> 62) == 2);
            if (incl) {                                 // nearest tile with a full prefix ends the walk
                const int first = (int)__builtin_ctzll(incl);
                excl += wave_reduce_add(l <= first ? (uint32_t)v[k] : 0u);
                done = true;
            } else {
                excl += wave_reduce_add((uint32_t)v[k]);
            }
        }
    }
    if (l == 0) __hip_atomic_store(&status[b], GF_FLAG_INCL | (uint64_t)(excl + T), __ATOMIC_RELAXED,
                                   __HIP_MEMORY_SCOPE_AGENT);
    return excl;
}


}  // namespace pch
