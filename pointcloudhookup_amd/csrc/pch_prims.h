// Device-wide primitives shared by the stage kernels: exclusive scan and LSD radix sort.
#pragma once
#include "pch_common.h"

namespace pch {

// ---- exclusive scan of uint32 (n < 2^31).  `out` may alias `in`.
// ws: scan_ws_u32(n) uint32 words.  total (optional, device) receives the grand total.
size_t scan_ws_u32(int64_t n);
int scan_exclusive_u32(const uint32_t* in, uint32_t* out, int64_t n, uint32_t* ws,
                       uint32_t* total, hipStream_t s);
// the same over popcount(bits[i]) (ranks of the set bits of a bitmap); `out` must NOT alias `bits`
int scan_exclusive_popc_u32(const uint32_t* bits, uint32_t* out, int64_t n, uint32_t* ws,
                            uint32_t* total, hipStream_t s);

// ---- the middle step alone, for callers whose own kernels do the tile sums and the tile scans (SCAN_TILE elements
// per workgroup): in-place exclusive scan of nb tile sums (one workgroup), grand total to `total` (optional, device)
constexpr int SCAN_TILE = 2048;
int scan_tile_sums_u32(uint32_t* tile_sums, int64_t nb, uint32_t* total, hipStream_t s);

// ---- the same in ONE launch (decoupled look-back).  zeroed_ws: scan1_ws_u32(n) uint32 words, 8-byte aligned, that
// are ZERO when the kernel starts (the caller folds that into a fill or a kernel it runs anyway); `total`
// (optional, device) must be ZERO too and receives the grand total - or 0xFFFFFFFF if a tile's bounded wait gave
// up (the outputs are then undefined).  `out` may alias `in` (not for the popcount form).
size_t scan1_ws_u32(int64_t n);
bool   scan1_pays(int64_t n);          // the one-launch form is the faster one only up to ~1.3 M elements
int scan1_exclusive_u32(const uint32_t* in, uint32_t* out, int64_t n, uint32_t* zeroed_ws, uint32_t* total,
                        hipStream_t s);
int scan1_exclusive_popc_u32(const uint32_t* bits, uint32_t* out, int64_t n, uint32_t* zeroed_ws, uint32_t* total,
                             hipStream_t s);

// ---- stable LSD radix sort of (uint64 key, uint32 value) pairs on key bits [0, nbits).
// Buffers ping-pong between (k0,v0) and (k1,v1); the sorted result lands in buffer
// radix_sort_result_buffer(nbits) (0 or 1).  ws: radix_ws_u32(n) uint32 words.
size_t radix_ws_u32(int64_t n);
int    radix_sort_result_buffer(int nbits);
int radix_sort_pairs(uint64_t* k0, uint32_t* v0, uint64_t* k1, uint32_t* v1, int64_t n,
                     int nbits, uint32_t* ws, hipStream_t s);

}  // namespace pch
