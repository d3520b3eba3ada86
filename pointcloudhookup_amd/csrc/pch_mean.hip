// B1: np.mean(raw, axis=0) on a C-order (n,3) float32 array == SEQUENTIAL float32 running sum
// per column, then one float32 division by float32(n)  (utils/tower_extraction.py:63).
//
// The sequential sum is reproduced bit for bit, in parallel.  While the running sum
// s = sigma*m*u keeps its sign and stays inside one binade (m in [2^23,2^24), u = ulp(s)),
//     fl(s + a) = sigma * (m + rne(sigma*a/u)) * u,
// i.e. the sum is INTEGER addition of per-element increments d_i = rne(a_i/u) that do not depend
// on m - unless a_i/u has fraction exactly 1/2 (a "tie", resolved towards the even mantissa).
//
//   ms_summary_k : one wave per 1024-point block (read ONCE, staged in LDS; the z column is also
//                  copied out for the percentile passes).  Per column and for the 24 useful
//                  candidate binades E = emax+1 .. emax+24 (E <= emax: an element as large as
//                  the sum; E > emax+24: every increment is 0) it stores the signed increment
//                  sum S_E, one bound A0 >= sum |a/ulp(2^(emax+1))| (so that
//                  sum|d_i| <= (A0 >> j) + 1024) and a tie bit per candidate (an element ties at
//                  exactly one candidate: the one just above its lowest set bit).
//   ms_level2_k  : rows for 64 blocks (65 536 points): child rows added at equal absolute binade.
//   ms_walk_k    : one workgroup per column walks level-2 rows 64 at a time (prefix scan of the
//                  increments + per-row certificate that no prefix can leave the binade), opens
//                  the children of a row that fails, and adds a child that fails exactly: its four
//                  waves (which all run the same walk) chain a quarter of the block each, from 64
//                  neighbouring start values, and the quarters are joined by look-ups (ms_blocks_exact).
// Order inside a certified block is irrelevant, hence the result equals the sequential sum.
#include "pch_mean.h"

namespace pch {

constexpr int MSB       = 1024;    // points per summary block (one wave, 16 per lane)
constexpr int MS_PER    = 16;
constexpr int MS_CAND   = 24;
constexpr int MS_ROW2   = 3 * MS_CAND;      // level-2 row: N0[24], N1[24] (net per incoming parity), A[24]
constexpr int MS_WAVES  = 4;
constexpr uint32_t MS_NONFINITE = 1u, MS_ALLZERO = 2u;
constexpr uint32_t MS_TIGHT = 4u;   // level-1 record: pad[0..3] hold two 64-bit bounds of the running prefix (ms_summary_k)
constexpr int MS_FIX_FROM = 11;     // sparse-tie fix-up for candidates >= this ...
constexpr int MS_FIX_LOW  = 6;      // ... and from this one on where the prefix is mostly cancellation (ms_summary_k)
constexpr int MS_FIX_MAX  = 16;     // ... holding at most this many tie elements

// mask: candidates j whose S / fix entries were computed (bit j); a candidate outside the mask is "unknown" to the walk
struct MsHdr { int emax; uint32_t tie; uint32_t flags; uint32_t mask; };
constexpr uint32_t MS_ALLCAND = (1u << MS_CAND) - 1u;
constexpr int MS_WIN = 4;           // predicted window: candidates jp-1 .. jp+2

// level-1 record of one (column, block): three 128-byte lines, assembled in LDS and written by one
// store instruction (whole lines: no read-modify-write in HBM), read by the lane that owns the
// block in ms_level2_k / ms_walk_k
struct alignas(128) MsRec {
    MsHdr     h;
    long long A0;                        // upper bound of sum |a / ulp(2^(emax+1))|
    long long S[MS_CAND];                // S[j] = sum_i rne(a_i / 2^j ulps)
    uint32_t  fix[MS_CAND];              // sparse-tie adjustments (adj0 & 0xFFFF) | adj1 << 16
    uint32_t  pad[18];
};
static_assert(sizeof(MsRec) == 384 && offsetof(MsRec, A0) == 16 && offsetof(MsRec, S) == 24 &&
              offsetof(MsRec, fix) == 24 + 8 * MS_CAND, "three cache lines per record; ms_level2_k reads it by word offsets");

// Level-2 table layout: candidate-major planes [column][slot][row], so that lanes that walk
// consecutive rows at (mostly) the same candidate read consecutive words:
// slots 0..23 = N0_j, 24..47 = N1_j, 48..71 = A_j (MS_ROW2 planes)
__host__ __device__ __forceinline__ int64_t ms_at(int c, int slot, int64_t blk, int64_t nblk, int planes) {
    return ((int64_t)c * planes + slot) * nblk + blk;
}

// ---- wave-wide reductions without the LDS crossbar (all 64 lanes must be active) ------------
// Inside a row of 16 lanes: DPP rotations folded into the ALU instruction; across rows and halves:
// the gfx950 lane-swap instructions (v_permlane16_swap / v_permlane32_swap) exchange the odd rows
// (upper half) of one register with the even rows (lower half) of another, which is exactly one
// step of a transposed butterfly: two registers in, op(a', b') holds the pair results of the
// first register in the even rows (lower half) and of the second one in the odd rows (upper half).
template <int CTRL>
__device__ __forceinline__ uint32_t ms_dpp(uint32_t v) {
    return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, CTRL, 0xF, 0xF, false);
}
constexpr int MS_ROR8 = 0x128, MS_ROR4 = 0x124, MS_ROR2 = 0x122, MS_ROR1 = 0x121;   // rotate inside a row of 16
constexpr int MS_HALF_MIRROR = 0x141;                  // lane i <-> 7 - i inside every 8 lanes
constexpr int MS_QUAD_X1 = 0xB1, MS_QUAD_X2 = 0x4E;    // quad_perm [1,0,3,2] / [2,3,0,1]: lane ^ 1, lane ^ 2

template <typename Op>
__device__ __forceinline__ uint32_t ms_wave_all(uint32_t v, Op op) {     // every lane gets op over the wave
    v = op(v, ms_dpp<MS_ROR8>(v));
    v = op(v, ms_dpp<MS_ROR4>(v));
    v = op(v, ms_dpp<MS_ROR2>(v));
    v = op(v, ms_dpp<MS_ROR1>(v));
    const auto r = __builtin_amdgcn_permlane16_swap(v, v, false, false);
    v = op(r[0], r[1]);
    const auto q = __builtin_amdgcn_permlane32_swap(v, v, false, false);
    return op(q[0], q[1]);
}
template <int CTRL>
__device__ __forceinline__ long long ms_add_ror64(long long v) {         // v + (v of the lane the DPP control names)
    const uint32_t lo = ms_dpp<CTRL>((uint32_t)v), hi = ms_dpp<CTRL>((uint32_t)((unsigned long long)v >> 32));
    return v + (long long)(((unsigned long long)hi << 32) | lo);
}

// x/2^j is a rounding tie iff the lowest set bit of x is 2^(j-1): returns 1 << j for that j
// (0 when x == 0 or x has bits below 2^-1; bits >= MS_CAND are masked off by the caller).
// |x| < 2^23, so 2x converts to int exactly when it is an integer.
__device__ __forceinline__ uint32_t ms_tie_bit(float x) {
    const float y = x + x;
    const int Y = (int)y;
    return ((float)Y == y) ? (uint32_t)(Y & -Y) : 0u;
}

// ---- which candidates will the walk ask for? ------------------------------------------------------------------
// The walk reads a block's table at ONE candidate: the binade of the float32 running sum when it gets there.  That
// binade follows from the prefix of the column, which a block cannot know - but it can be ESTIMATED cheaply: one
// sampled row per block gives the sum of every 65 536-point row (ms_sample_k), a prefix over the rows (ms_prefix_k)
// the running sum in front of every row, and the float32 sum stops growing roughly 25 binades above the typical
// element (increments round to zero).  ms_summary_k then computes a window of MS_WIN candidates around
// the estimate instead of all 24 (2.4x fewer vector instructions: the kernel is bound by them).  A wrong estimate
// costs time, never exactness: a candidate that was not computed is "unknown" to the walk, which then adds the
// block element by element (ms_blocks_exact).  Blocks with both signs, non-finite values or a prefix dominated by
// cancellation keep all 24 candidates.
struct MsPred { double pre, rowsum, meanabs; };          // per (column, level-2 row)

__global__ __launch_bounds__(64) void ms_sample_k(const float* __restrict__ xyz, int64_t n, int64_t nb, int64_t nb2,
                                                  MsPred* __restrict__ pred, float* __restrict__ zsample) {
    const int64_t row = blockIdx.x;
    const int l = lane_id();
    const int64_t blk = row * 64 + l;
    double v[3] = {0.0, 0.0, 0.0}, av[3] = {0.0, 0.0, 0.0};
    int have = 0;
    if (blk < nb) {
        const int64_t p0 = blk * MSB;
        const int64_t cnt = (n - p0) < MSB ? (n - p0) : MSB;
        const int64_t p = p0 + (int64_t)(((uint32_t)blk * 2654435761u) >> 22) % cnt;   // one pseudo-random row of the block
        struct Row3 { float x, y, z; };
        const Row3 q = reinterpret_cast<const Row3*>(xyz)[p];
        v[0] = q.x; v[1] = q.y; v[2] = q.z;
        have = 1;
        if (zsample) zsample[blk] = q.z;
    }
    const int64_t first = row * 64 * MSB;
    const double cnt_row = (double)((n - first) < (int64_t)64 * MSB ? (n - first) : (int64_t)64 * MSB);
    const double ns = (double)__popcll(__ballot(have != 0));
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        const bool fin = fabs(v[c]) < 1.0e300;             // NaN / inf samples: no estimate from them
        av[c] = fin ? fabs(v[c]) : 0.0;
        v[c] = fin ? v[c] : 0.0;
        double sv = v[c], sa = av[c];
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) { sv += __shfl_xor(sv, o, 64); sa += __shfl_xor(sa, o, 64); }
        if (l == 0) {
            MsPred o;
            o.pre = 0.0;
            o.rowsum = ns > 0.0 ? sv / ns * cnt_row : 0.0;
            o.meanabs = ns > 0.0 ? sa / ns : 0.0;
            pred[(int64_t)c * nb2 + row] = o;
        }
    }
}

// exclusive prefix of the row sums per column (one 1024-thread workgroup per column: a block scan per 1024 rows),
// starting from the incoming running sum
// Block 3 (when launched): a deliberately LOW estimate of the height filter's raw-z threshold from the sampled z values
// (one per block): the lower edge of the 2^-11-wide key bin that holds the sample's pct-quantile, found by two LDS
// histogram passes over the order-preserving keys, plus `add`.  Only an estimate: what it is used for is checked
// against the exact threshold later (gf_cand_k).
__device__ void ms_zestimate(const float* __restrict__ zs, int64_t ns, double pct, float add, float* __restrict__ tcand) {
    __shared__ uint32_t hist[2048];
    __shared__ uint32_t wsum[16];
    __shared__ uint32_t pick_bin, pick_below, total_sh;
    const int l = lane_id(), w = wave_id();
    // most samples share a handful of bins (flat ground): lanes that hold the leader's bin add ONCE for all of them
    auto add_bin = [&](bool in, uint32_t bin) {
        const unsigned long long act = __ballot(in);
        if (!act) return;
        const int lead = (int)__builtin_ctzll(act);
        const uint32_t b0 = (uint32_t)__builtin_amdgcn_readlane((int)bin, lead);
        const unsigned long long same = __ballot(in && bin == b0);
        if (l == lead) atomicAdd(&hist[b0], (uint32_t)__popcll(same));
        if (in && bin != b0) atomicAdd(&hist[bin], 1u);
    };
    // bin that holds `rank` and the count below it: every thread owns two bins, block-wide exclusive scan
    auto scan_pick = [&](uint32_t rank) {
        __syncthreads();
        const uint32_t h0 = hist[2 * threadIdx.x], h1 = hist[2 * threadIdx.x + 1];
        const uint32_t incl = wave_scan_incl(h0 + h1);
        if (l == 63) wsum[w] = incl;
        __syncthreads();
        uint32_t before = incl - (h0 + h1);
        for (int k = 0; k < w; ++k) before += wsum[k];
        if (rank >= before && rank < before + h0) { pick_bin = 2 * threadIdx.x; pick_below = before; }
        else if (rank >= before + h0 && rank < before + h0 + h1) { pick_bin = 2 * threadIdx.x + 1; pick_below = before + h0; }
        __syncthreads();
    };
    for (int j = threadIdx.x; j < 2048; j += 1024) hist[j] = 0;
    if (threadIdx.x == 0) { total_sh = 0; pick_bin = 2047; pick_below = 0; }
    __syncthreads();
    // every 16th sample is plenty for an estimate (6 k values at 100 M points, and the margin is 0.5 m); 8 loads in flight
    constexpr int EST_STRIDE = 16, EST_BATCH = 8;
    const int64_t ne = (ns + EST_STRIDE - 1) / EST_STRIDE;
    uint32_t mine = 0;
    for (int64_t i0 = 0; i0 < ne; i0 += 1024 * EST_BATCH) {   // workgroup-uniform trip count (ballots inside)
        float zb[EST_BATCH];
#pragma unroll
        for (int u = 0; u < EST_BATCH; ++u) {
            const int64_t i = i0 + u * 1024 + threadIdx.x;
            zb[u] = i < ne ? zs[i * EST_STRIDE] : NAN;
        }
#pragma unroll
        for (int u = 0; u < EST_BATCH; ++u) {
            const bool in = zb[u] == zb[u];
            add_bin(in, f32_ordered(zb[u]) >> 21);
            mine += in ? 1u : 0u;
        }
    }
    mine = wave_reduce_add(mine);
    if (l == 0) atomicAdd(&total_sh, mine);
    __syncthreads();
    const uint32_t total = total_sh;
    if (threadIdx.x == 0) reinterpret_cast<uint32_t*>(tcand)[1] = 0u;           // overflow word of the slots
    if (total == 0) { if (threadIdx.x == 0) *tcand = -INFINITY; return; }       // no estimate: every row is a candidate
    uint32_t rank = (uint32_t)((double)(total - 1) * (pct / 100.0));
    scan_pick(rank);
    const uint32_t top = pick_bin;
    rank -= pick_below;
    __syncthreads();
    for (int j = threadIdx.x; j < 2048; j += 1024) hist[j] = 0;
    if (threadIdx.x == 0) { pick_bin = 2047; pick_below = 0; }
    __syncthreads();
    for (int64_t i0 = 0; i0 < ne; i0 += 1024 * EST_BATCH) {
        float zb[EST_BATCH];
#pragma unroll
        for (int u = 0; u < EST_BATCH; ++u) {
            const int64_t i = i0 + u * 1024 + threadIdx.x;
            zb[u] = i < ne ? zs[i * EST_STRIDE] : NAN;
        }
#pragma unroll
        for (int u = 0; u < EST_BATCH; ++u) {
            const uint32_t k = f32_ordered(zb[u]);
            add_bin(zb[u] == zb[u] && (k >> 21) == top, (k >> 10) & 2047u);
        }
    }
    scan_pick(rank);
    if (threadIdx.x == 0) {
        const uint32_t key = (top << 21) | (pick_bin << 10);       // lower edge of the bin
        *tcand = f32_unordered(key) + add;
    }
}

__global__ __launch_bounds__(1024) void ms_prefix_k(MsPred* __restrict__ pred, int64_t nb2,
                                                    const float* __restrict__ sum_in, const float* __restrict__ zs,
                                                    int64_t ns, double pct, float add, float* __restrict__ tcand) {
    __shared__ double wsum[16];
    __shared__ double carry_sh;
    if (blockIdx.x == 3) { ms_zestimate(zs, ns, pct, add, tcand); return; }
    const int c = blockIdx.x;
    const int l = lane_id(), w = wave_id();
    if (threadIdx.x == 0) carry_sh = sum_in ? (double)sum_in[c] : 0.0;
    __syncthreads();
    for (int64_t r0 = 0; r0 < nb2; r0 += 1024) {
        const int64_t r = r0 + threadIdx.x;
        const double v = r < nb2 ? pred[(int64_t)c * nb2 + r].rowsum : 0.0;
        double incl = v;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) { const double t = __shfl_up(incl, o, 64); if (l >= o) incl += t; }
        if (l == 63) wsum[w] = incl;
        __syncthreads();
        double woff = 0.0, total = 0.0;
#pragma unroll
        for (int k = 0; k < 16; ++k) { const double t = wsum[k]; woff += k < w ? t : 0.0; total += t; }
        const double carry = carry_sh;
        if (r < nb2) pred[(int64_t)c * nb2 + r].pre = carry + woff + incl - v;
        __syncthreads();
        if (threadIdx.x == 0) carry_sh = carry + total;
        __syncthreads();
    }
}

__device__ __forceinline__ int ms_exp2_floor(double x) {        // floor(log2 |x|) of a finite double, -1023 for 0
    return (int)((((unsigned long long)__double_as_longlong(x)) >> 52) & 0x7FFull) - 1023;
}

__global__ __launch_bounds__(64 * MS_WAVES) void ms_summary_k(const float* __restrict__ xyz, int64_t n,
                                                             int64_t nb, MsRec* __restrict__ recs,
                                                             float* __restrict__ zcol,
                                                             const MsPred* __restrict__ pred, int64_t nb2,
                                                             float* __restrict__ cslots, uint32_t* __restrict__ ccounts,
                                                             const float* __restrict__ tcand) {
    __shared__ __attribute__((aligned(16))) float lds[MS_WAVES][MSB * 3];
    __shared__ MsRec stage[MS_WAVES];                    // a record is assembled here, stored as whole lines
    const int64_t blk = (int64_t)blockIdx.x * MS_WAVES + wave_id();
    if (blk >= nb) return;
    const int l = lane_id();
    const int64_t p0 = blk * MSB;
    const int cnt = (int)((n - p0) < MSB ? (n - p0) : MSB);
    float* tile = lds[wave_id()];
    const float* src = xyz + 3 * p0;
    if (cnt == MSB) {                                    // 12 KiB, 16-byte loads, every byte once
        const float4* s4 = reinterpret_cast<const float4*>(src);
        float4* t4 = reinterpret_cast<float4*>(tile);
#pragma unroll
        for (int i = 0; i < 12; ++i) t4[i * 64 + l] = s4[i * 64 + l];
    } else {
        for (int e = l; e < 3 * MSB; e += 64) tile[e] = e < 3 * cnt ? src[e] : 0.0f;   // ragged tail: zero padded
    }
    __builtin_amdgcn_wave_barrier();
    // the block's z values, once: for the z column copy and for the candidate test below
    float zr[MS_PER];
    if (zcol || cslots) {
#pragma unroll
        for (int i = 0; i < MS_PER; ++i) zr[i] = tile[3 * (i * 64 + l) + 2];
    }
    if (zcol) {
#pragma unroll
        for (int i = 0; i < MS_PER; ++i) {
            const int p = i * 64 + l;
            if (p < cnt) zcol[p0 + p] = zr[i];
        }
    }
    if (cslots) {
        // candidate rows of the height filter (MsCand): raw z above the low threshold estimate, kept in file order.
        // Candidates come in runs (rows next to a tower); most blocks hold none, and a block whose largest z does not
        // exceed the estimate skips the sixteen rounds of ballots, ranks and plane stores (a quarter of this kernel's
        // vector instructions) for one wave maximum.
        const float tc = *tcand;
        float zmax = -INFINITY;
#pragma unroll
        for (int i = 0; i < MS_PER; ++i) {
            const int p = i * 64 + l;
            if (p < cnt) zmax = fmaxf(zmax, zr[i]);            // (NaN never exceeds anything: not a candidate either)
        }
        const bool some = __ballot(zmax > tc) != 0;
        uint32_t at = 0;
        if (some) {
            float* slot = cslots + blk * (4 * MS_CAND_SLOT);    // planes x | y | z | row (MsCand)
#pragma unroll
            for (int i = 0; i < MS_PER; ++i) {
                const int p = i * 64 + l;
                const float z = zr[i];
                const bool take = p < cnt && z > tc;
                const unsigned long long m = __ballot(take);
                if (m == 0) continue;                           // wave-uniform
                const uint32_t pos = at + (uint32_t)__popcll(m & lanemask_lt());
                if (take && pos < (uint32_t)MS_CAND_SLOT) {
                    slot[pos] = tile[3 * p];
                    slot[MS_CAND_SLOT + pos] = tile[3 * p + 1];
                    slot[2 * MS_CAND_SLOT + pos] = z;
                    slot[3 * MS_CAND_SLOT + pos] = __uint_as_float((uint32_t)p);
                }
                at += (uint32_t)__popcll(m);
            }
        }
        if (l == 0) {
            ccounts[blk] = at < (uint32_t)MS_CAND_SLOT ? at : (uint32_t)MS_CAND_SLOT;
            if (at > (uint32_t)MS_CAND_SLOT)               // more candidates than the slot holds: the sweep reads the tile
                atomicOr(reinterpret_cast<uint32_t*>(const_cast<float*>(tcand)) + 1, 1u);
        }
    }
    const bool b5 = (l & 32) != 0, b4 = (l & 16) != 0, b3 = (l & 8) != 0;
#pragma unroll 1
    for (int c = 0; c < 3; ++c) {
        float a[MS_PER];
#pragma unroll
        for (int i = 0; i < MS_PER; ++i) {
            a[i] = tile[3 * (i * 64 + l) + c];
        }
        uint32_t mx = 0;
#pragma unroll
        for (int i = 0; i < MS_PER; ++i) {
            const uint32_t u = __float_as_uint(a[i]) & 0x7FFFFFFFu;
            mx = u > mx ? u : mx;
        }
        mx = (uint32_t)__builtin_amdgcn_readfirstlane(
            (int)ms_wave_all(mx, [](uint32_t a, uint32_t b) { return a > b ? a : b; }));
        const bool nonfinite = mx >= 0x7F800000u;             // |bits| order: inf / NaN are the largest
        const int ef = (int)(mx >> 23);
        const int emax = (ef > 0 ? ef : 1) - 127;
        // S[j] = sum_i rne(x_i / 2^j).  For j >= 1, fl(x + 1.5*2^(23+j)) rounds x to a multiple of 2^j
        // (nearest-even) and the integer difference of the two bit patterns is exactly that
        // multiple, so one float add + one integer add per (element, candidate); |x| < 2^23 keeps
        // the sum inside the constant's binade.  The 32-bit accumulators wrap; the true sums
        // (|S| <= 2^27 per lane) are what is left once the 16 constants are taken off again.
        uint32_t tie = 0;
        int A0 = 0;
        long long pfx_up = 0, pfx_dn = 0;                      // bounds of the running prefix (24-candidate path)
        bool have_pfx = false;
        const bool live = !nonfinite && mx != 0;               // wave-uniform
        int S[MS_CAND];
#pragma unroll
        for (int j = 0; j < MS_CAND; ++j) S[j] = 0;
        int W[MS_WIN] = {0, 0, 0, 0};                          // the sums of the windowed path (see below)
        // ---- candidate window (wave-uniform): see the comment above ms_sample_k
        uint32_t cmask = MS_ALLCAND;
        int j0 = 1;                                            // first candidate of the window (>= 1)
        int fix_from = MS_FIX_FROM;                            // sparse ties are settled for candidates >= this
        if (live && pred) {
            // both signs present?  (-0.0 counts as negative: such a block merely keeps all candidates)
            uint32_t orb = 0;
            float top = 0.0f;
#pragma unroll
            for (int i = 0; i < MS_PER; i += 2) {
                orb = orb | __float_as_uint(a[i]) | __float_as_uint(a[i + 1]);         // v_or3_b32
                top = fmaxf(top, fmaxf(a[i], a[i + 1]));                               // v_max3_f32 (finite: live)
            }
            const bool mixed = __ballot((orb >> 31) != 0u) != 0 && __ballot(top > 0.0f) != 0;
            const MsPred pr = pred[(int64_t)c * nb2 + (blk >> 6)];
            const double start = pr.pre + pr.rowsum * ((double)(blk & 63) * (1.0 / 64.0));
            const double mass = fabs(pr.pre) + fabs(pr.rowsum);
            // a prefix that is mostly cancellation (|sum| far below the mass that went into it) is not predictable
            const bool cancel = !(fabs(start) * 8.0 >= pr.meanabs * (double)(blk * MSB)) && blk > 0;
            // the low candidates (a running sum only 2^7 .. 2^11 times the block's largest element) are met where the
            // prefix is mostly cancellation - a column that wanders about zero; elsewhere their ties (2-16 per block and
            // candidate, a sweep of the block each) are settled for nothing
            // (tables built ahead of their walk - no prediction - keep the plain form: config 4's phase 1)
            if (cancel) fix_from = MS_FIX_LOW;
            if (!mixed && !cancel && mass > 0.0 && pr.meanabs > 0.0) {
                int E = ms_exp2_floor(start);
                const int Estag = ms_exp2_floor(pr.meanabs) + 25;       // increments round to zero up there
                E = E < Estag ? E : Estag;
                int jp = E - emax - 1;                                  // candidate of binade E for THIS block
                jp = jp < 2 ? 2 : jp;                                   // window = jp-1 .. jp+2, inside 1 .. 23
                jp = jp > MS_CAND - 3 ? MS_CAND - 3 : jp;
                j0 = jp - 1;
                // candidate 0 (a running sum within a factor two of the block's largest element) is not part of the
                // window: a same-sign block behind the first one cannot meet it
                cmask = ((1u << MS_WIN) - 1u) << j0;
            }
        }
        if (live && cmask != MS_ALLCAND) {
            float magic[MS_WIN];
            uint32_t acc[MS_WIN];
#pragma unroll
            for (int k = 0; k < MS_WIN; ++k) {
                magic[k] = ldexpf(1.5f, 23 + j0 + k);                   // 1.5 * 2^(23+j), exact
                acc[k] = 0u - (uint32_t)MS_PER * __float_as_uint(magic[k]);
            }
            // An element ties at window candidate j iff its lowest set bit is 2^(j-1) >= 2^(j0-1): only a multiple of
            // 2^(j0-1) can.  That test is one add, one subtract and one compare (the exact bit costs eight
            // instructions); the exact bits are computed only when some element passes it - for the deep windows of
            // a long column (j0 ~ 20) that is one block in a thousand.
            const float mtest = ldexpf(1.5f, 23 + j0 - 1);
            bool maybe = false;
            float absum = 0.0f;
#pragma unroll
            for (int i = 0; i < MS_PER; ++i) {
                const float x = ldexpf(a[i], 22 - emax);       // a / ulp(2^(emax+1)), |x| < 2^23, exact
                absum += fabsf(x);
                maybe |= ((x + mtest) - mtest) == x;
#pragma unroll
                for (int k = 0; k < MS_WIN; ++k) acc[k] += __float_as_uint(x + magic[k]);
            }
            if (__ballot(maybe)) {
#pragma unroll
                for (int i = 0; i < MS_PER; ++i) tie |= ms_tie_bit(ldexpf(a[i], 22 - emax));
            }
#pragma unroll
            for (int k = 0; k < MS_WIN; ++k) W[k] = (int)acc[k];       // the window j0 .. j0+3
            A0 = (int)ceilf(absum * 1.00001f) + 1;
            tie &= cmask;                                      // ties of candidates that were not computed do not matter
        } else if (live) {
            uint32_t acc[MS_CAND];                             // start at minus the 16 constants that get added
            acc[0] = 0;
#pragma unroll
            for (int j = 1; j < MS_CAND; ++j)
                acc[j] = 0u - (uint32_t)MS_PER * __float_as_uint((float)(3ull << (22 + j)));
            float absum = 0.0f;
            // Bounds of the RUNNING prefix sum_{e < k} x_e over the block's elements in file order, k = 0 .. 1024: the
            // elements i * 64 + l of one i are 64 consecutive ones, so with Cu_i = sum_l ceil(x) and P_i = the positive
            // part of it, the prefix never exceeds max_i (sum_{i' < i} Cu_i' + P_i) - and never falls below the mirror
            // image built from the floors.  A0 bounds the prefix by the sum of ALL positive steps whatever their order;
            // for a column that wanders about zero (both signs in every block) that is ~16 times what the prefix really
            // reaches, and the walk gives up on blocks it could certify (section 4 of DESIGN.md).
            long long pre_hi = 0, pre_lo = 0;
            pfx_up = 0; pfx_dn = 0;
#pragma unroll
            for (int i = 0; i < MS_PER; ++i) {
                const float x = ldexpf(a[i], 22 - emax);       // a / ulp(2^(emax+1)), |x| < 2^23, exact
                absum += fabsf(x);
                tie |= ms_tie_bit(x);
                acc[0] += (uint32_t)(int)rintf(x);
#pragma unroll
                for (int j = 1; j < MS_CAND; ++j) {
                    const float magic = (float)(3ull << (22 + j));      // 1.5 * 2^(23+j), exact
                    acc[j] += __float_as_uint(x + magic);
                }
                if (fix_from == MS_FIX_LOW) {                  // (wave-uniform: where the low candidates matter, see above)
                    const int cu = (int)ceilf(x);
                    const auto add = [](uint32_t p, uint32_t q) { return p + q; };
                    const int Cu = (int)ms_wave_all((uint32_t)cu, add);              // |.| < 2^29
                    const int P = (int)ms_wave_all((uint32_t)(cu > 0 ? cu : 0), add);
                    const long long up = pre_hi + P, dn = (long long)(P - Cu + 64) - pre_lo;   // (floor >= ceil - 1)
                    pfx_up = up > pfx_up ? up : pfx_up;
                    pfx_dn = dn > pfx_dn ? dn : pfx_dn;
                    pre_hi += Cu;
                    pre_lo += Cu - 64;
                }
            }
            have_pfx = fix_from == MS_FIX_LOW;
#pragma unroll
            for (int j = 0; j < MS_CAND; ++j) S[j] = (int)acc[j];
            // A0 only has to be an upper bound of sum |x|: the float32 sum of 16 terms is low by at most
            // 16 roundings (relative 2^-20), which the factor and the + 1 cover (|absum| < 2^27)
            A0 = (int)ceilf(absum * 1.00001f) + 1;
        }
        if (__ballot(tie != 0u)) tie = ms_wave_all(tie & ((1u << MS_CAND) - 1u), [](uint32_t a, uint32_t b) { return a | b; });
        // ---- sparse ties (candidates >= MS_FIX_FROM with at most MS_FIX_MAX tie elements): every tie
        // leaves the running mantissa EVEN, and before it the parity is (incoming parity) ^ (parity of
        // the increments in front of it), which a few ballots give.  So the block's net increment is
        // known for both incoming parities: S_j + adj0 / S_j + adj1 (S_j uses the nearest-even value
        // v of every tie element; a tie met at odd parity takes the other neighbour, v +- 1).
        uint32_t myfix = 0;                                    // lane j keeps (adj0 & 0xFFFF) | adj1 << 16
        {
            uint32_t todo = tie & ~((1u << fix_from) - 1u);
            while (todo) {
                const int j = __ffs((int)todo) - 1;
                todo &= todo - 1;
                int nt = 0;
#pragma unroll
                for (int i = 0; i < MS_PER; ++i)
                    nt += (int)__popcll(__ballot(ms_tie_bit(ldexpf(a[i], 22 - emax)) == (1u << j)));
                if (nt > MS_FIX_MAX) continue;
                const float magic = ldexpf(1.5f, 23 + j);
                int par = 0, adj0 = 0, adj1 = 0, pa0 = 0, pa1 = 0;
#pragma unroll
                for (int i = 0; i < MS_PER; ++i) {
                    const float x = ldexpf(a[i], 22 - emax);
                    const float t = x + magic;                  // nearest-even multiple of 2^j
                    const float v = t - magic;                  // exact
                    const unsigned long long low = __ballot((__float_as_uint(t) & 1u) != 0u);
                    unsigned long long tb = __ballot(ms_tie_bit(x) == (1u << j));
                    const int dirl = (v < x) ? 1 : -1;          // the other neighbour of a tie: v+2^j or v-2^j
                    while (tb) {
                        const int lt = (int)__builtin_ctzll(tb);
                        tb &= tb - 1;
                        const int pre = par ^ (int)(__popcll(low & ((1ull << lt) - 1ull)) & 1);
                        const int dir = __builtin_amdgcn_readlane(dirl, lt);
                        if ((0 ^ pre ^ pa0) & 1) { adj0 += dir; pa0 ^= 1; }
                        if ((1 ^ pre ^ pa1) & 1) { adj1 += dir; pa1 ^= 1; }
                    }
                    par ^= (int)(__popcll(low) & 1);
                }
                tie &= ~(1u << j);                             // resolved: no longer "unknown"
                if (l == j) myfix = ((uint32_t)adj0 & 0xFFFFu) | ((uint32_t)adj1 << 16);
            }
        }
        long long A0w;
        {
            uint32_t r = (uint32_t)A0;                         // per lane <= 2^27: a row of 16 fits 32 bits
            r += ms_dpp<MS_ROR8>(r); r += ms_dpp<MS_ROR4>(r); r += ms_dpp<MS_ROR2>(r); r += ms_dpp<MS_ROR1>(r);
            const auto p = __builtin_amdgcn_permlane16_swap(r, r, false, false);
            const unsigned long long two = (unsigned long long)p[0] + p[1];
            const uint32_t lo = (uint32_t)two, hi = (uint32_t)(two >> 32);
            const auto ql = __builtin_amdgcn_permlane32_swap(lo, lo, false, false);
            const auto qh = __builtin_amdgcn_permlane32_swap(hi, hi, false, false);
            A0w = (long long)((((unsigned long long)qh[0] << 32) | ql[0]) + (((unsigned long long)qh[1] << 32) | ql[1]));
        }
        MsRec* rec = &stage[wave_id()];
        if (live && cmask != MS_ALLCAND) {
            // four sums only, reduced together: two lane-swap steps (see ms_wave_all) fold the halves and the row
            // pairs and leave candidate j0 + c in row c of the wave (16 lanes, |partial| <= 2^29), one 32-bit and
            // three 64-bit rotations inside the row finish it - ~20 instructions where one butterfly per sum and
            // 16-bit half took ~80
            static_assert(MS_WIN == 4, "one window candidate per row of 16 lanes");
            if (l < MS_CAND) rec->S[l] = 0;
            __builtin_amdgcn_wave_barrier();
            const auto r02 = __builtin_amdgcn_permlane32_swap((uint32_t)W[0], (uint32_t)W[2], false, false);
            const auto r13 = __builtin_amdgcn_permlane32_swap((uint32_t)W[1], (uint32_t)W[3], false, false);
            const uint32_t v02 = r02[0] + r02[1];              // lower half: window candidate 0, upper half: 2
            const uint32_t v13 = r13[0] + r13[1];              // lower half: 1, upper half: 3
            const auto rq = __builtin_amdgcn_permlane16_swap(v02, v13, false, false);
            uint32_t v = rq[0] + rq[1];                        // rows 0..3 of the wave: candidates 0..3
            v += ms_dpp<MS_ROR8>(v);                           // |.| <= 2^30
            long long t = (int)v;
            t = ms_add_ror64<MS_ROR4>(t);
            t = ms_add_ror64<MS_ROR2>(t);
            t = ms_add_ror64<MS_ROR1>(t);                      // every lane of row c: the wave's sum of candidate c
            if ((l & 15) == 0) rec->S[j0 + (l >> 4)] = t;
        } else {
        // transposed reduction: 24 -> 12 -> 6 -> 3 values per lane while summing over the lane
        // bits 5,4,3 (|partial| <= 2^30 stays in 32 bit), then three 64-bit steps inside the 8-lane
        // groups.  Bits 5 and 4 are lane-swap steps (see ms_wave_all), bit 3 a select + DPP rotation.
        int v12[12], v6[6], v3[3];
#pragma unroll
        for (int i = 0; i < 12; ++i) {
            const auto r = __builtin_amdgcn_permlane32_swap((uint32_t)S[i], (uint32_t)S[12 + i], false, false);
            v12[i] = (int)(r[0] + r[1]);                       // lower half: candidate i, upper half: 12 + i
        }
#pragma unroll
        for (int i = 0; i < 6; ++i) {
            const auto r = __builtin_amdgcn_permlane16_swap((uint32_t)v12[i], (uint32_t)v12[6 + i], false, false);
            v6[i] = (int)(r[0] + r[1]);                        // even rows: candidate i, odd rows: 6 + i
        }
#pragma unroll
        for (int i = 0; i < 3; ++i)
            v3[i] = (b3 ? v6[3 + i] : v6[i]) + (int)ms_dpp<MS_ROR8>((uint32_t)(b3 ? v6[i] : v6[3 + i]));
        long long t[3];
#pragma unroll
        for (int i = 0; i < 3; ++i) {
            t[i] = v3[i];
            t[i] = ms_add_ror64<MS_HALF_MIRROR>(t[i]);         // lanes (i, 7-i), then (l, l^1), (l, l^2): every lane
            t[i] = ms_add_ror64<MS_QUAD_X1>(t[i]);             // of an 8-lane group ends with the group's total
            t[i] = ms_add_ror64<MS_QUAD_X2>(t[i]);
        }
        if ((l & 7) == 0) {
            const int jb = (b5 ? 12 : 0) + (b4 ? 6 : 0) + (b3 ? 3 : 0);
#pragma unroll
            for (int i = 0; i < 3; ++i) rec->S[jb + i] = t[i];
        }
        }
        if (l < MS_CAND) rec->fix[l] = myfix;
        if (l >= 4 && l < 18) rec->pad[l] = 0;
        if (l == 0) {
            rec->pad[0] = 0; rec->pad[1] = 0; rec->pad[2] = 0; rec->pad[3] = 0;
            rec->A0 = A0w;
            rec->h.emax = emax;
            rec->h.tie = tie;
            rec->h.flags = (nonfinite ? MS_NONFINITE : 0u) | (mx == 0 ? MS_ALLZERO : 0u) | (have_pfx ? MS_TIGHT : 0u);
            rec->h.mask = cmask;
            if (have_pfx) {                                    // (up to 2^33 each: two words)
                rec->pad[0] = (uint32_t)pfx_up; rec->pad[1] = (uint32_t)((unsigned long long)pfx_up >> 32);
                rec->pad[2] = (uint32_t)pfx_dn; rec->pad[3] = (uint32_t)((unsigned long long)pfx_dn >> 32);
            }
        }
        __builtin_amdgcn_wave_barrier();
        if (l < 24)                                            // 24 x 16 bytes: three full lines
            reinterpret_cast<uint4*>(recs + (int64_t)c * nb + blk)[l] = reinterpret_cast<const uint4*>(rec)[l];
        __builtin_amdgcn_wave_barrier();
    }
}

__device__ __forceinline__ long long ms_readlane64(long long v, int lane) {
    const int lo = __builtin_amdgcn_readlane((int)(v & 0xFFFFFFFFll), lane);
    const int hi = __builtin_amdgcn_readlane((int)(v >> 32), lane);
    return ((long long)hi << 32) | (unsigned int)lo;
}

// parity -> increment maps compose associatively: (f then g)(p) = f(p) + g((p + f(p)) & 1)
__device__ __forceinline__ void ms_scan_pairs(long long& c0, long long& c1) {
    const int l = lane_id();
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        const long long p0 = __shfl_up(c0, o, 64), p1 = __shfl_up(c1, o, 64);   // earlier segment f
        if (l >= o) {
            const long long n0 = p0 + ((p0 & 1) ? c1 : c0);
            const long long n1 = p1 + (((1 + p1) & 1) ? c1 : c0);
            c0 = n0; c1 = n1;
        }
    }
}

#ifdef PCH_MS_STAMPS                              // tuning build: shader-clock time per phase of ms_blocks_exact (z column) into stats[27..31]
__device__ unsigned long long ms_stamp_acc[8];
#define MS_STAMP(i) do { const unsigned long long t_ = __builtin_readcyclecounter(); \
                         if (threadIdx.x == 0 && c == 2) ms_stamp_acc[(i)] += t_ - ms_t_; ms_t_ = t_; } while (0)
#define MS_STAMP_DECL unsigned long long ms_t_ = __builtin_readcyclecounter()
#else
#define MS_STAMP(i) do { } while (0)
#define MS_STAMP_DECL do { } while (0)
#endif
constexpr int MS_SEG = 16;                       // consecutive elements per lane in the serial chain of ms_blocks_exact
constexpr int MS_XW = 4;                         // waves of the walk's workgroup (they all run the same walk and share the
                                                 // work only inside ms_blocks_exact)
constexpr int MS_XPART = MSB / MS_XW;            // elements of a block per wave there
__device__ __forceinline__ int ms_pad(int i) { return i + (i >> 4); }   // LDS bank spreading

struct alignas(16) MsExact {                     // LDS of the exact path
    float    plain[MSB];                         // the block in file order (uniform reads of the candidate chains)
    float    padded[MSB + MSB / 16];             // the same, bank-spread: lane l reads its 16 consecutive elements
    double   psum[MS_XW];                        // real sum of every wave's part (float64)
    uint32_t tab[MS_XW][64];                     // the part's chain run from 64 neighbouring start values
};

// floats as integers in their own order (-0.0 -> -1, +0.0 -> 0): the neighbours of a float are key +- 1
__device__ __forceinline__ int ms_key(uint32_t b) { return (int)(b ^ (uint32_t)(((int)b >> 31) & 0x7FFFFFFF)); }
__device__ __forceinline__ uint32_t ms_unkey(int k) { return (uint32_t)k ^ (uint32_t)((k >> 31) & 0x7FFFFFFF); }

__device__ __forceinline__ double ms_wave_sum_f64(double v) {           // every lane gets the wave's sum (fixed order)
    auto halves = [](double d, uint32_t& lo, uint32_t& hi) {
        const unsigned long long u = (unsigned long long)__double_as_longlong(d);
        lo = (uint32_t)u; hi = (uint32_t)(u >> 32);
    };
    auto whole = [](uint32_t lo, uint32_t hi) { return __longlong_as_double((long long)(((unsigned long long)hi << 32) | lo)); };
    uint32_t lo, hi;
#define MS_F64_ROR(CTRL) halves(v, lo, hi); v = v + whole(ms_dpp<CTRL>(lo), ms_dpp<CTRL>(hi))
    MS_F64_ROR(MS_ROR8); MS_F64_ROR(MS_ROR4); MS_F64_ROR(MS_ROR2); MS_F64_ROR(MS_ROR1);
#undef MS_F64_ROR
    halves(v, lo, hi);
    const auto l16 = __builtin_amdgcn_permlane16_swap(lo, lo, false, false);
    const auto h16 = __builtin_amdgcn_permlane16_swap(hi, hi, false, false);
    v = whole(l16[0], h16[0]) + whole(l16[1], h16[1]);
    halves(v, lo, hi);
    const auto l32 = __builtin_amdgcn_permlane32_swap(lo, lo, false, false);
    const auto h32 = __builtin_amdgcn_permlane32_swap(hi, hi, false, false);
    return whole(l32[0], h32[0]) + whole(l32[1], h32[1]);
}

// Adds the level-1 blocks [blk, blk + nblk) to the running sum exactly.
// The reference is the sequential float32 chain itself: lane l holds the 16 consecutive elements [16 l, 16 l + 16) of a
// block, in round r every lane chains its 16 elements onto the (wave-uniform) running sum and lane r's result becomes the
// running sum of round r + 1 (v_readlane) - 64 rounds x 17 dependent instructions ~ 3.4 us per block whatever the data:
// a lone wave issues a dependent add every ~8 cycles, and that IS the floor of a sequential float32 sum on one wave.
// Round 4: FOUR waves, each with a quarter of the block, and 64 start values per wave.  A chain depends on its start
// value only; wave w does not know the sum that will enter its quarter, but it knows it to a few ulps - the incoming sum
// plus the REAL sum of the quarters in front (float64) is off by the roundings of at most 768 additions, ~8 ulps rms - so
// lane l of wave w runs the quarter's 256-step chain from the float (l - 32) neighbours away from that estimate: 64
// literal chains at the price of one (every lane of a wave computes anyway).  Wave 0's estimate IS the incoming sum;
// its result is looked up among wave 1's start values, that result among wave 2's, and so on: three table reads
// instead of 768 dependent additions.  Nothing is approximated: a result is taken from a chain that started at exactly
// the value that arrived (same bits), or not at all - a start value outside its window (a sum that dropped by binades
// inside the block: its ulps shrank under the estimate's error) sends the rest of the block to the serial chain, from
// the last quarter that was resolved.  Non-finite sums or elements take the serial chain outright (NaN and inf
// propagate as in numpy's own loop).  Rows beyond the array's end count as -0.0: s + (-0.0) = s for every s, -0.0
// included.  All four waves run the same walk on the same data (the tables are read-only), so they arrive here
// together.
// first_decides: stop behind the first block if the sum is still in the binade (and has the sign) it came with - the
// caller's tables are still valid then; `added` returns the number of blocks that were added.
__device__ __forceinline__ uint32_t ms_blocks_exact(const float* __restrict__ xyz, int64_t n, int c,
                                                    int64_t blk, int nblk, uint32_t sb, MsExact* X, int* dbg,
                                                    bool first_decides, int& added) {
    const int l = lane_id(), w = wave_id();
    constexpr int PER = MS_XPART / 64;                   // elements per lane of a wave's part (coalesced)
    float v0[PER], v1[PER], v2[PER];                     // this wave's part of the next three blocks
    auto fetch = [&](int64_t b, float (&v)[PER]) {
        const int64_t p0 = b * MSB + w * MS_XPART;
#pragma unroll
        for (int k = 0; k < PER; ++k) {
            const int64_t i = p0 + l + 64 * k;
            v[k] = i < n ? xyz[3 * i + c] : -0.0f;
        }
    };
    fetch(blk, v0);
    if (nblk > 1) fetch(blk + 1, v1);
    if (nblk > 2) fetch(blk + 2, v2);
    float s = __uint_as_float(sb);
    added = 0;
    MS_STAMP_DECL;
    for (int q = 0; q < nblk; ++q) {
        MS_STAMP(0);
        double mine = 0.0;
#pragma unroll
        for (int k = 0; k < PER; ++k) {
            const int e = w * MS_XPART + l + 64 * k;
            X->plain[e] = v0[k];
            X->padded[ms_pad(e)] = v0[k];
            mine += (double)v0[k];
        }
        mine = ms_wave_sum_f64(mine);
        if (l == 0) X->psum[w] = mine;
        __syncthreads();
#pragma unroll
        for (int k = 0; k < PER; ++k) { v0[k] = v1[k]; v1[k] = v2[k]; }
        if (q + 3 < nblk) fetch(blk + q + 3, v2);        // in flight during this block and the next two
        const int64_t left = n - (blk + q) * MSB;
        const int cnt = (int)(left < MSB ? left : MSB);
        const int rounds = (cnt + MS_SEG - 1) / MS_SEG;
        int r0 = 0;                                      // first round the serial chain still has to do
        double tot = 0.0;
#pragma unroll
        for (int k = 0; k < MS_XW; ++k) tot += X->psum[k];
        const uint32_t sbits = __float_as_uint(s);
        MS_STAMP(1);                                     // 0 -> 1: rows arrived, staged, summed
        if (((sbits >> 23) & 0xFFu) != 255u && fabs(tot) < INFINITY) {     // finite sum, finite elements
            double pre = 0.0;
            for (int k = 0; k < w; ++k) pre += X->psum[k];
            const int kc = ms_key(__float_as_uint((float)((double)s + pre)));
            float v = __uint_as_float(ms_unkey(kc + (l - 32)));
            // the part's elements in file order: uniform 16-byte LDS reads, 64 elements requested ahead of the adds that
            // use them (a read in front of every dependent add would cost more than the add)
            const float4* p4 = reinterpret_cast<const float4*>(X->plain + w * MS_XPART);
            constexpr int G = 16;                            // float4 per stage
            float4 ea[G], eb[G];
#pragma unroll
            for (int i = 0; i < G; ++i) ea[i] = p4[i];
#pragma unroll
            for (int g = 0; g < MS_XPART / 4 / G; g += 2) {
#pragma unroll
                for (int i = 0; i < G; ++i) eb[i] = p4[(g + 1) * G + i];
#pragma unroll
                for (int i = 0; i < G; ++i) { v = v + ea[i].x; v = v + ea[i].y; v = v + ea[i].z; v = v + ea[i].w; }
                if (g + 2 < MS_XPART / 4 / G) {
#pragma unroll
                    for (int i = 0; i < G; ++i) ea[i] = p4[(g + 2) * G + i];
                }
#pragma unroll
                for (int i = 0; i < G; ++i) { v = v + eb[i].x; v = v + eb[i].y; v = v + eb[i].z; v = v + eb[i].w; }
            }
            X->tab[w][l] = __float_as_uint(v);
            MS_STAMP(2);                                 // 1 -> 2: the 64 chains
            __syncthreads();
            uint32_t rb = X->tab[0][32];                 // wave 0 started lane 32 at the incoming sum itself
            int done = 1;                                // parts resolved
            pre = X->psum[0];
            for (int k = 1; k < MS_XW; ++k) {
                const int kk = ms_key(__float_as_uint((float)((double)s + pre)));
                const long long idx = (long long)ms_key(rb) - kk + 32;
                if (idx < 0 || idx >= 64) break;
                rb = X->tab[k][(int)idx];
                pre += X->psum[k];
                ++done;
            }
            s = __uint_as_float(rb);
            r0 = done * (MS_XPART / MS_SEG);
            if (dbg) dbg[0] += done;                     // quarters taken from the tables
            MS_STAMP(3);                                 // 2 -> 3: look-ups
        }
        if (r0 < rounds) {                               // the serial chain for what is left (all of it: non-finite input)
            float a[MS_SEG];
#pragma unroll
            for (int k = 0; k < MS_SEG; ++k) a[k] = X->padded[ms_pad(MS_SEG * l + k)];
            for (int r = r0; r < rounds; ++r) {
                float t = s;
#pragma unroll
                for (int k = 0; k < MS_SEG; ++k) t = t + a[k];
                s = __uint_as_float((uint32_t)__builtin_amdgcn_readlane((int)__float_as_uint(t), r));
            }
            if (dbg) dbg[1] += (rounds - r0) * MS_SEG;   // elements added one by one
        }
        if (dbg) ++dbg[2];                               // blocks
        __syncthreads();                                 // the stage is overwritten by the next block
        MS_STAMP(4);                                     // 3 -> 4: serial rest + barrier
        ++added;
        if (q == 0 && first_decides) {
            const uint32_t ef = (sb >> 23) & 0xFFu;
            if (((__float_as_uint(s) ^ sb) & 0xFF800000u) == 0 && ef >= 1u && ef <= 254u) break;
        }
    }
    return __float_as_uint(s);
}

// ---- level 2: one row per 64 level-1 blocks (65 536 points).  Bounds are additive; the net
// increments are summed when no child depends on the incoming parity and composed IN ORDER
// ((f then g)(p) = f(p) + g((p + f(p)) & 1)) when one does.  A child contributes nothing to
// candidates 24 or more binades above its own largest element.
constexpr int MS_L2_PITCH = sizeof(MsRec) / 4 + 1;       // words per child in LDS (+1: bank spreading)

__global__ __launch_bounds__(256) void ms_level2_k(const MsRec* __restrict__ recs, int64_t nb,
                                                   int64_t nb2, MsHdr* __restrict__ hdr2,
                                                   long long* __restrict__ rows2) {
    // one workgroup per (column, row): the 64 child records are contiguous (24 KiB), copied to LDS
    // with coalesced 16-byte loads; lane l of every wave then reads the record of child l, and
    // each of the four waves composes six of the 24 candidates
    __shared__ uint32_t sh[64 * MS_L2_PITCH];
    __shared__ uint32_t tie_sh, miss_sh;
    const int64_t w = blockIdx.x;
    const int c = (int)(w / nb2);
    const int64_t g = w % nb2;
    const int l = lane_id();
    const int64_t bb = g * 64 + l;
    const bool valid = bb < nb;
    {
        const int nchild = (int)((nb - g * 64) < 64 ? (nb - g * 64) : 64);
        const uint4* src = reinterpret_cast<const uint4*>(recs + (int64_t)c * nb + g * 64);
        constexpr int VPR = sizeof(MsRec) / 16;            // 16-byte pieces per record
        uint4 q[VPR / 4];
#pragma unroll
        for (int it = 0; it < VPR / 4; ++it) {
            const int v = it * 256 + (int)threadIdx.x;
            if (v < nchild * VPR) q[it] = src[v];
        }
#pragma unroll
        for (int it = 0; it < VPR / 4; ++it) {
            const int v = it * 256 + (int)threadIdx.x;
            if (v < nchild * VPR) {
                uint32_t* d = sh + (v / VPR) * MS_L2_PITCH + (v % VPR) * 4;
                d[0] = q[it].x; d[1] = q[it].y; d[2] = q[it].z; d[3] = q[it].w;
            }
        }
        if (threadIdx.x == 0) { tie_sh = 0; miss_sh = 0; }
        __syncthreads();
    }
    const uint32_t* rec = sh + l * MS_L2_PITCH;           // words: hdr 0..3, A0 4..5, S[j] 6+2j, fix[j] 54+j
    auto rec64 = [&](int word) { return (long long)(((unsigned long long)rec[word + 1] << 32) | rec[word]); };
    MsHdr h;
    h.emax = -200; h.tie = 0; h.flags = MS_ALLZERO; h.mask = MS_ALLCAND;
    if (valid) { h.emax = (int)rec[0]; h.tie = rec[1]; h.flags = rec[2]; h.mask = rec[3]; }
    const bool zero = (h.flags & MS_ALLZERO) != 0;
    const int emax2 = wave_reduce_max(zero ? -200 : h.emax);
    const uint32_t nonfinite = __ballot((h.flags & MS_NONFINITE) != 0) ? MS_NONFINITE : 0u;
    const bool allzero = __ballot(!zero) == 0;
    uint32_t tie2 = 0, miss2 = 0;                          // miss2: candidates some child did not compute
    const int shift = emax2 - h.emax;                      // >= 0 for non-zero children
    const bool live = valid && !zero && !(h.flags & MS_NONFINITE);
    const long long A0 = live ? rec64(4) : 0;
    for (int j2 = wave_id() * (MS_CAND / 4); j2 < (wave_id() + 1) * (MS_CAND / 4); ++j2) {
        const int j = j2 + shift;
        long long n0 = 0, n1 = 0, A = 0;
        bool tie = false;
        if (__ballot(live && j < MS_CAND && !((h.mask >> j) & 1u))) {
            // a child did not compute this candidate (prediction windows): the row's entry is "unknown" to the walk
            // (ms_classify checks the mask before anything is read), so there is nothing to reduce or to store -
            // with windowed children that is ~20 of the 24 candidates of a row
            miss2 |= 1u << j2;
            continue;
        }
        if (live && j < MS_CAND) {
            const long long S = rec64(6 + 2 * j);
            const uint32_t f = rec[6 + 2 * MS_CAND + j];
            n0 = S + (long long)(short)(f & 0xFFFFu);
            n1 = S + (long long)(short)(f >> 16);
            A = (A0 >> j) + MSB;                           // >= sum |d_i| of the child at this binade
            tie = (h.tie >> j) & 1u;
        }
        if (__ballot(n0 != n1)) {
            ms_scan_pairs(n0, n1);                          // lane 63 holds the composition of all children
            n0 = ms_readlane64(n0, 63);
            n1 = ms_readlane64(n1, 63);
        } else {
            n0 = wave_reduce_add(n0);
            n1 = n0;
        }
        A = wave_reduce_add(A);
        if (__ballot(tie)) tie2 |= 1u << j2;
        if (l == 0) {
            rows2[ms_at(c, j2, g, nb2, MS_ROW2)] = n0;
            rows2[ms_at(c, MS_CAND + j2, g, nb2, MS_ROW2)] = n1;
            rows2[ms_at(c, 2 * MS_CAND + j2, g, nb2, MS_ROW2)] = A;
        }
    }
    if (l == 0 && tie2) atomicOr(&tie_sh, tie2);
    if (l == 0 && miss2) atomicOr(&miss_sh, miss2);
    __syncthreads();
    if (threadIdx.x == 0) {
        MsHdr o;
        o.emax = allzero ? 0 : emax2;
        o.tie = tie_sh;
        o.flags = nonfinite | (allzero ? MS_ALLZERO : 0u);
        o.mask = MS_ALLCAND & ~miss_sh;
        hdr2[(int64_t)c * nb2 + g] = o;
    }
}

struct MsTables {
    const MsRec* rec; int64_t nb;                                               // level 1
    const MsHdr* hdr2; const long long* rows2; int64_t nb2;                     // level 2
};

// 0: s cannot change, 1: integer increments with table bounds, 2: unknown at this binade
__device__ __forceinline__ int ms_classify(const MsHdr& h, bool valid, bool s_inf, bool s_norm, int E, int& j) {
    j = E - h.emax - 1;
    if (!valid || (h.flags & MS_ALLZERO)) return 0;
    if (s_inf) return (h.flags & MS_NONFINITE) ? 2 : 0;
    if (!s_norm || (h.flags & MS_NONFINITE) || j < 0) return 2;
    if (j >= MS_CAND) return 0;
    if (!((h.mask >> j) & 1u)) return 2;                 // candidate not computed (prediction window missed it)
    return ((h.tie >> j) & 1u) ? 2 : 1;
}

// what a lane knows about its table row at the current binade: net increment of |s| for either
// parity of the incoming mantissa, and an interval that contains every prefix
struct MsEntry { long long n0, n1, lo, hi; };
__device__ __forceinline__ MsEntry ms_entry(long long p0, long long p1, long long A, bool s_neg) {
    // p0/p1: net increments in the s > 0 frame; for s < 0 both flip sign (ties included)
    const long long smax = p0 > p1 ? p0 : p1, smin = p0 < p1 ? p0 : p1;
    const long long up = ((A + smax + 1) >> 1) + MS_FIX_MAX, dn = ((A - smin + 1) >> 1) + MS_FIX_MAX;
    MsEntry e;
    e.n0 = s_neg ? -p0 : p0;
    e.n1 = s_neg ? -p1 : p1;
    e.hi = s_neg ? dn : up;
    e.lo = -(s_neg ? up : dn);
    return e;
}

// one step of the certified walk over up to 64 rows held one per lane: scans the increments of
// the lanes >= start, returns the first lane whose certificate fails (or `count`) and advances
// m_cur / sb over the certified lanes in front of it
__device__ __forceinline__ int ms_certify(int cls, const MsEntry& e, bool valid, int start, int count,
                                          bool s_norm, long long& m_cur, uint32_t& sb) {
    const int l = lane_id();
    long long c0 = l >= start ? e.n0 : 0ll, c1 = l >= start ? e.n1 : 0ll;
    ms_scan_pairs(c0, c1);                              // inclusive: increment through lane l
    const int pc = (int)(m_cur & 1);
    long long e0 = __shfl_up(c0, 1, 64), e1 = __shfl_up(c1, 1, 64);
    if (l == 0) { e0 = 0; e1 = 0; }
    const long long m_in = m_cur + (pc ? e1 : e0);
    const bool ok = cls == 0 || (cls == 1 && m_in + e.hi + 1 < (1ll << 24) && m_in + e.lo - 1 >= (1ll << 23));
    const unsigned long long fail = __ballot(valid && l >= start && !ok);
    const int f = fail ? (int)__builtin_ctzll(fail) : count;
    if (f > start && s_norm) {                          // advance over the certified lanes
        m_cur += ms_readlane64(pc ? c1 : c0, f - 1);
        sb = (sb & 0xFF800000u) | ((uint32_t)m_cur & 0x7FFFFFu);
    }
    return fail ? f : -1;
}

// Adds the level-1 blocks [first, first+count), count <= 64, to the running sum `sb` (bits).
__device__ __forceinline__ uint32_t ms_walk_children(const float* __restrict__ xyz, int64_t n, int c,
                                                     const MsTables& T, int64_t first, int count,
                                                     uint32_t sb, MsExact* stage, int& n_exact, int& streak,
                                                     int& n_miss, int* dbg) {
    const int l = lane_id();
    int done = 0;                                       // children already added
    while (done < count) {
        const uint32_t ef = (sb >> 23) & 0xFFu;
        if (ef == 255u && (sb & 0x7FFFFFu)) return sb;  // NaN is absorbing
        const bool s_inf = ef == 255u;
        const bool s_norm = ef >= 1u && ef <= 254u;
        const bool s_neg = (sb >> 31) != 0;
        const int E = (int)ef - 127;
        const int64_t bb = first + l;
        const bool valid = l >= done && l < count;
        MsHdr h;
        h.emax = 0; h.tie = 0; h.flags = MS_ALLZERO; h.mask = MS_ALLCAND;
        const MsRec* rec = T.rec + (int64_t)c * T.nb + (valid ? bb : first);
        if (valid) h = rec->h;
        int j;
        const int cls = ms_classify(h, valid, s_inf, s_norm, E, j);
        MsEntry en;
        en.n0 = en.n1 = en.lo = en.hi = 0;
        if (cls == 1) {
            const long long S = rec->S[j];
            const uint32_t f = rec->fix[j];
            en = ms_entry(S + (long long)(short)(f & 0xFFFFu), S + (long long)(short)(f >> 16),
                          (rec->A0 >> j) + MSB, s_neg);
            if (h.flags & MS_TIGHT) {
                // the record also bounds the running prefix in file order (ms_summary_k): a step differs from x / 2^j by
                // at most 1/2 (a tie taken the other way included), so k steps stay within k / 2 <= 512 of the prefix of x
                const long long pu = (long long)(((unsigned long long)rec->pad[1] << 32) | rec->pad[0]);
                const long long pd = (long long)(((unsigned long long)rec->pad[3] << 32) | rec->pad[2]);
                const long long up = ((pu + ((1ll << j) - 1)) >> j) + MSB / 2 + MS_FIX_MAX;
                const long long dn = ((pd + ((1ll << j) - 1)) >> j) + MSB / 2 + MS_FIX_MAX;
                const long long hi = s_neg ? dn : up, lo = -(s_neg ? up : dn);
                en.hi = hi < en.hi ? hi : en.hi;
                en.lo = lo > en.lo ? lo : en.lo;
            }
        }
        int start = done;                               // first unresolved lane
        long long m_cur = (long long)((sb & 0x7FFFFFu) | 0x800000u);   // mantissa entering `start`
        for (;;) {
            const int f = ms_certify(cls, en, valid, start, count, s_norm, m_cur, sb);
            if (f < 0) { if (start < count) streak = 0; done = count; break; }
            // child f cannot be certified from the table at this binade: add it exactly
            if (f > start) streak = 0;                  // the tables carried the walk over some children: no hard stretch
            {
                const uint32_t mk = (uint32_t)__builtin_amdgcn_readlane((int)h.mask, f);
                const int jf = __builtin_amdgcn_readlane(j, f);
                if (jf >= 0 && jf < MS_CAND && !((mk >> jf) & 1u)) ++n_miss;
            }
            // In a stretch where block after block fails (a sum that keeps changing sign or binade: a zero-mean
            // column) the table read + certificate in front of every block is pure overhead - the chain is exact
            // whatever the tables say.  So behind a failing block that left the binade, `streak` more blocks are added
            // without asking; the run doubles while that keeps happening (at most 32) and is forgotten as soon as a
            // certificate advances.  An ordinary binade crossing never sees it (the run starts at 0).
            const int extra = streak < count - (f + 1) ? streak : count - (f + 1);
            int added = 0;
            const uint32_t nsb = ms_blocks_exact(xyz, n, c, first + f, 1 + extra, sb, stage, dbg, true, added);
            // (one block that stayed in its binade: the entries the lanes hold are still good for the children behind it)
            const bool same = added == 1 && ((nsb ^ sb) & 0xFF800000u) == 0 && s_norm;   // same sign and binade
            n_exact += added;
            sb = nsb;
            start = f + added;
            done = start;
            if (same) m_cur = (long long)((sb & 0x7FFFFFu) | 0x800000u);
            else {
                streak = streak ? (streak < 16 ? 2 * streak : 32) : 1;
                break;                                  // re-read the remaining children at the new binade
            }
        }
    }
    return sb;
}

// one wave per column: walks the level-2 rows, descends into the children of a row only when
// its certificate fails
// sum_in (optional): the running sum this shard continues (a file-order shard of a larger array, see
// pch_mean_seq_partial_f32); divide_n: 0 = store the running sum itself, else divide by float32(divide_n)
__global__ __launch_bounds__(64 * MS_XW) void ms_walk_k(const float* __restrict__ xyz, int64_t n, MsTables T,
                                                const float* __restrict__ sum_in, int64_t divide_n, int divide,
                                                float* __restrict__ out, int* __restrict__ stats) {
    __shared__ MsExact stage_x;
    MsExact* stage = &stage_x;
    const int c = blockIdx.x;
    const int l = lane_id();
    uint32_t sb = sum_in ? __float_as_uint(sum_in[c]) : 0u;      // bits of the running sum (+0.0 at the start)
    int64_t b = 0;                                     // next level-2 row
    int n_exact = 0, n_batches = 0, n_desc = 0, n_miss = 0;
    int dbg[3] = {0, 0, 0};                            // passes / serially added elements / calls of the exact path
    int streak = 0;                                    // blocks added without asking the tables (ms_walk_children)
    while (b < T.nb2) {
        ++n_batches;
        const uint32_t ef = (sb >> 23) & 0xFFu;
        if (ef == 255u && (sb & 0x7FFFFFu)) break;     // NaN is absorbing
        const bool s_inf = ef == 255u;
        const bool s_norm = ef >= 1u && ef <= 254u;
        const bool s_neg = (sb >> 31) != 0;
        const int E = (int)ef - 127;
        const int64_t bb = b + l;
        const bool valid = bb < T.nb2;
        MsHdr h;
        h.emax = 0; h.tie = 0; h.flags = MS_ALLZERO; h.mask = MS_ALLCAND;
        if (valid) h = T.hdr2[(int64_t)c * T.nb2 + bb];
        int j;
        const int cls = ms_classify(h, valid, s_inf, s_norm, E, j);
        MsEntry en;
        en.n0 = en.n1 = en.lo = en.hi = 0;
        if (cls == 1) {
            en = ms_entry(T.rows2[ms_at(c, j, bb, T.nb2, MS_ROW2)], T.rows2[ms_at(c, MS_CAND + j, bb, T.nb2, MS_ROW2)],
                          T.rows2[ms_at(c, 2 * MS_CAND + j, bb, T.nb2, MS_ROW2)], s_neg);
        }
        int start = 0;
        long long m_cur = (long long)((sb & 0x7FFFFFu) | 0x800000u);
        bool reload = false;
        while (!reload) {
            const int f = ms_certify(cls, en, valid, start, 64, s_norm, m_cur, sb);
            if (f < 0) break;
            // descend into the 64 children of row b+f
            ++n_desc;
            const int64_t first = (b + f) * 64;
            const int count = (int)((T.nb - first) < 64 ? (T.nb - first) : 64);
            const uint32_t nsb = ms_walk_children(xyz, n, c, T, first, count, sb, stage, n_exact, streak, n_miss, dbg);
            const bool same = ((nsb ^ sb) & 0xFF800000u) == 0 && s_norm;
            sb = nsb;
            start = f + 1;
            if (same) m_cur = (long long)((sb & 0x7FFFFFu) | 0x800000u);
            else reload = true;
        }
        b += reload ? start : 64;
    }
    if (threadIdx.x == 0) {
        // n == 0 -> 0/0 = NaN like numpy
        out[c] = divide ? __uint_as_float(sb) / (float)divide_n : __uint_as_float(sb);
        if (stats) {
            stats[4 * c + 0] = n_batches; stats[4 * c + 1] = n_miss;     // n_miss: exact blocks the window caused
            stats[4 * c + 2] = n_exact; stats[4 * c + 3] = n_desc;
            stats[16 + 4 * c + 0] = dbg[0]; stats[16 + 4 * c + 1] = dbg[1]; stats[16 + 4 * c + 2] = dbg[2];
#ifdef PCH_MS_STAMPS
            if (c == 2) for (int i = 0; i < 5; ++i) { stats[27 + i] = (int)(ms_stamp_acc[i] >> 6); ms_stamp_acc[i] = 0; }
#endif
        }
    }
}

// single-workgroup reference variant (kept for cross-checking the parallel algorithm)
constexpr int MS_TILE = 4096;   // points per LDS tile (48 KiB)

__global__ __launch_bounds__(256) void mean_seq_k(const float* __restrict__ xyz, int64_t n,
                                                  float* __restrict__ out) {
    __shared__ float tile[MS_TILE * 3];
    float s = 0.0f;
    for (int64_t base = 0; base < n; base += MS_TILE) {
        const int cnt = (int)((n - base) < MS_TILE ? (n - base) : MS_TILE);
        const float* src = xyz + 3 * base;
        for (int e = threadIdx.x; e < 3 * cnt; e += 256) tile[e] = src[e];
        __syncthreads();
        if (threadIdx.x < 3) {
            const float* col = tile + threadIdx.x;
            int i = 0;
            for (; i + 8 <= cnt; i += 8) {
                const float a0 = col[3 * (i + 0)], a1 = col[3 * (i + 1)], a2 = col[3 * (i + 2)],
                            a3 = col[3 * (i + 3)], a4 = col[3 * (i + 4)], a5 = col[3 * (i + 5)],
                            a6 = col[3 * (i + 6)], a7 = col[3 * (i + 7)];
                s = s + a0; s = s + a1; s = s + a2; s = s + a3;
                s = s + a4; s = s + a5; s = s + a6; s = s + a7;
            }
            for (; i < cnt; ++i) s = s + col[3 * i];
        }
        __syncthreads();
    }
    if (threadIdx.x < 3) out[threadIdx.x] = s / (float)n;
}

void ms_plan(Arena& a, int64_t n, MsWs& w) {
    const int64_t nb = ceil_div(n > 0 ? n : 1, MSB);
    const int64_t nb2 = ceil_div(nb, 64);
    w.stats = a.take<int>(32);               // [3][4] walk statistics, [16 + 4 c ..] exact-path counters
    w.pred = a.take<MsPred>(3 * nb2);
    w.rec = a.take<MsRec>(3 * nb);
    w.hdr2 = a.take<MsHdr>(3 * nb2);
    w.rows2 = a.take<long long>(3 * nb2 * MS_ROW2);
}

int mean_seq_launch(const float* xyz, int64_t n, float* out, MsWs& w, float* zcol, hipStream_t s,
                    hipEvent_t ev_zcol, const float* sum_in, int64_t divide_n, int phase, const MsCand* cand,
                    bool* cand_made, hipStream_t walk_stream) {
    if (cand_made) *cand_made = false;
    hipStream_t ws = s;                                  // stream of level 2 and the walk
    const int64_t nb = n > 0 ? ceil_div(n, MSB) : 0;
    const int64_t nb2 = ceil_div(nb, 64);
    if (n > 0 && phase != MS_PHASE_WALK) {
        // candidate prediction (see ms_sample_k): only when the running sum the rows continue is known now -
        // the whole sum (phase both), not for tables built ahead of their walk - and the array spans several rows
        static const bool no_predict = getenv("PCH_MEAN_NO_PREDICT") != nullptr;      // tuning toggle
        const MsPred* pred = nullptr;
        if (!no_predict && phase == MS_PHASE_BOTH && nb2 >= 2) {
            static const bool no_cand = getenv("PCH_GF_NO_CAND") != nullptr;              // tuning toggle
            const bool mk = cand && !no_cand;
            PCH_LAUNCH("mean_sample", ms_sample_k, dim3((unsigned)nb2), dim3(64), 0, s, xyz, n, nb, nb2, w.pred,
                       mk ? cand->zsample : (float*)nullptr);
            PCH_LAUNCH("mean_prefix", ms_prefix_k, dim3(mk ? 4 : 3), dim3(1024), 0, s, w.pred, nb2, sum_in,
                       mk ? (const float*)cand->zsample : (const float*)nullptr, nb, mk ? cand->pct : 0.0,
                       mk ? cand->add : 0.0f, mk ? cand->tcand : (float*)nullptr);
            pred = w.pred;
            if (mk && cand_made) *cand_made = true;
        }
        const bool emit = cand && cand_made && *cand_made;
        PCH_LAUNCH("mean_summary", ms_summary_k, dim3((unsigned)ceil_div(nb, MS_WAVES)), dim3(64 * MS_WAVES),
                   0, s, xyz, n, nb, w.rec, zcol, pred, nb2, emit ? cand->slots : (float*)nullptr,
                   emit ? cand->counts : (uint32_t*)nullptr, emit ? (const float*)cand->tcand : (const float*)nullptr);
        if (ev_zcol) PCH_HIP_TRY(hipEventRecord(ev_zcol, s));
        if (walk_stream && ev_zcol) {
            PCH_HIP_TRY(hipStreamWaitEvent(walk_stream, ev_zcol, 0));
            ws = walk_stream;
        }
        PCH_LAUNCH("mean_level2", ms_level2_k, dim3((unsigned)(3 * nb2)), dim3(256), 0, ws,
                   (const MsRec*)w.rec, nb, nb2, w.hdr2, w.rows2);
    }
    if (phase == MS_PHASE_TABLES) return PCH_OK;
    MsTables T;
    T.rec = w.rec; T.nb = nb;
    T.hdr2 = w.hdr2; T.rows2 = w.rows2; T.nb2 = nb2;
    const int divide = divide_n != MS_NO_DIVIDE;
    PCH_LAUNCH("mean_walk", ms_walk_k, dim3(3), dim3(64 * MS_XW), 0, ws, xyz, n, T, sum_in,
               divide_n == MS_DIVIDE_BY_N ? n : divide_n, divide, out, w.stats);
    return PCH_OK;
}

int mean_seq_serial_launch(const float* xyz, int64_t n, float* out, hipStream_t s) {
    PCH_LAUNCH("mean_seq_serial", mean_seq_k, dim3(1), dim3(256), 0, s, xyz, n, out);
    return PCH_OK;
}

}  // namespace pch
