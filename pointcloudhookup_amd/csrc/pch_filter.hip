// Stage B: float32 centroid (numpy sequential-sum semantics), centring, numpy 'linear'
// percentile by radix select, and the order-preserving height-filter compaction.
// Reference: utils/tower_extraction.py:62-64 (centroid, centring), :82-89 (percentile filter).
#include "pch_prims.h"

namespace pch {

// =====================================================================================
// B1: np.mean(raw, axis=0) on a C-order (n,3) float32 array == SEQUENTIAL float32 running
// sum per column, then one float32 division by float32(n)  (utils/tower_extraction.py:63).
//
// The sequential sum is reproduced bit for bit, in parallel:
// while the running sum s = sigma*m*u keeps its sign and stays inside one binade
// (m in [2^23,2^24), u = ulp(s)), fl(s + a) = sigma*(m + rne(sigma*a/u))*u, i.e. the sum is
// INTEGER addition of per-element increments d_i = rne(a_i/u) that do not depend on m (unless
// a_i/u has fraction exactly 1/2: a "tie").  So for a block of points and a candidate binade
// E the whole block collapses to two integers (sum of positive / negative increments):
//   ms_summary_k : one wave per 1024-point block, all 24 useful candidates E = emax+1..emax+24
//                  (E <= emax: an element as large as s -> handled by the slow path;
//                   E >  emax+24: every increment is 0, s cannot move)
//   ms_walk_k    : one wave per column walks the blocks 64 at a time: prefix-sums the
//                  increments for the current E, certifies per block that no prefix can leave
//                  the binade (m - neg - 1 >= 2^23, m + pos + 1 < 2^24) and that the block has
//                  no tie; the first block that fails is added element by element (exact by
//                  construction) and the walk resumes with the new s.
// Order inside a certified block is irrelevant, so the result equals the sequential sum.
// =====================================================================================
constexpr int MSB       = 1024;    // points per summary block (one wave, 16 per lane)
constexpr int MS_PER    = 16;
constexpr int MS_CAND   = 24;
constexpr int MS_WAVES  = 4;
constexpr uint32_t MS_NONFINITE = 1u, MS_ALLZERO = 2u;

struct MsHdr { int emax; uint32_t tie; uint32_t flags; uint32_t pad; };

__global__ __launch_bounds__(64 * MS_WAVES) void ms_summary_k(const float* __restrict__ xyz, int64_t n,
                                                             int64_t nb, MsHdr* __restrict__ hdr,
                                                             long long* __restrict__ apos,
                                                             long long* __restrict__ aneg) {
    const int64_t blk = (int64_t)blockIdx.x * MS_WAVES + wave_id();
    if (blk >= nb) return;
    const int l = lane_id();
    const int64_t p0 = blk * MSB;
#pragma unroll 1
    for (int c = 0; c < 3; ++c) {
        float a[MS_PER];
#pragma unroll
        for (int i = 0; i < MS_PER; ++i) {
            const int64_t p = p0 + i * 64 + l;
            a[i] = (p < n) ? xyz[3 * p + c] : 0.0f;
        }
        uint32_t mx = 0;
        bool bad = false;
#pragma unroll
        for (int i = 0; i < MS_PER; ++i) {
            const uint32_t u = __float_as_uint(a[i]) & 0x7FFFFFFFu;
            mx = u > mx ? u : mx;
            bad |= (u >= 0x7F800000u);
        }
        mx = wave_reduce_max(mx);
        const bool nonfinite = __ballot(bad) != 0;
        const int ef = (int)(mx >> 23);
        const int emax = (ef > 0 ? ef : 1) - 127;
        int pos[MS_CAND], neg[MS_CAND];
#pragma unroll
        for (int j = 0; j < MS_CAND; ++j) { pos[j] = 0; neg[j] = 0; }
        uint32_t tie = 0;
        if (!nonfinite && mx != 0) {
#pragma unroll
            for (int i = 0; i < MS_PER; ++i) {
                float x = ldexpf(a[i], 22 - emax);             // a / ulp(2^(emax+1)), |x| < 2^23
#pragma unroll
                for (int j = 0; j < MS_CAND; ++j) {
                    const float r = rintf(x);                   // round half to even
                    tie |= (fabsf(x - r) == 0.5f) ? (1u << j) : 0u;
                    const int ri = (int)r;
                    pos[j] += ri > 0 ? ri : 0;
                    neg[j] += ri < 0 ? -ri : 0;
                    x *= 0.5f;
                }
            }
        }
        uint32_t tie_all = tie;
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) tie_all |= __shfl_xor(tie_all, o, 64);
        const int64_t row = ((int64_t)c * nb + blk);
        if (l == 0) {
            MsHdr h;
            h.emax = emax;
            h.tie = tie_all;
            h.flags = (nonfinite ? MS_NONFINITE : 0u) | (mx == 0 ? MS_ALLZERO : 0u);
            h.pad = 0;
            hdr[row] = h;
        }
#pragma unroll
        for (int j = 0; j < MS_CAND; ++j) {
            const long long tp = wave_reduce_add((long long)pos[j]);
            const long long tn = wave_reduce_add((long long)neg[j]);
            if (l == j) { apos[row * MS_CAND + j] = tp; aneg[row * MS_CAND + j] = tn; }
        }
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

constexpr int MS_SEG = 16;                       // elements per lane in ms_block_exact
__device__ __forceinline__ int ms_pad(int i) { return i + (i >> 4); }   // LDS bank spreading

// Adds ONE level-1 block to the running sum exactly, using all 64 lanes: lane l owns the 16
// consecutive elements [pos+16l, pos+16l+16) and summarises them IN ORDER for the current binade
// and for both parities of its incoming mantissa (a tie a/u = q+1/2 rounds to the even mantissa,
// so its increment depends on that parity; after one tie both chains are even and coincide).
// A parity-pair scan gives every lane its incoming mantissa; lanes whose segment provably stays
// inside the binade are applied at once, the first segment that does not (an element as large as
// the sum, a binade or sign change, a non-finite value) is added element by element, and the
// rest of the block is redone at the new binade.
__device__ __forceinline__ uint32_t ms_block_exact(const float* __restrict__ xyz, int64_t n, int c,
                                                   int64_t blk, uint32_t sb, float* stage) {
    const int l = lane_id();
    const int64_t p0 = blk * MSB;
    const int cnt = (int)((n - p0) < MSB ? (n - p0) : MSB);
    for (int i = l; i < cnt; i += 64) stage[ms_pad(i)] = xyz[3 * (p0 + i) + c];
    __syncthreads();
    int pos = 0;
    while (pos < cnt) {
        const uint32_t ef = (sb >> 23) & 0xFFu;
        if (ef == 255u && (sb & 0x7FFFFFu)) break;                 // NaN is absorbing
        const bool s_norm = ef >= 1u && ef <= 254u;
        int f = 0;                                                 // first segment to add serially
        if (s_norm) {
            const bool s_neg = (sb >> 31) != 0;
            const int E = (int)ef - 127;
            const int base = pos + MS_SEG * l;
            int run0 = 0, run1 = 0, mn0 = 0, mn1 = 0, mx0 = 0, mx1 = 0, par0 = 0, par1 = 1;
            bool bad = false;
#pragma unroll
            for (int k = 0; k < MS_SEG; ++k) {
                const int i = base + k;
                const float a = (i < cnt) ? stage[ms_pad(i)] : 0.0f;
                const float x = ldexpf(s_neg ? -a : a, 23 - E);     // real increment of the mantissa
                const float r = rintf(x);
                bad |= !(fabsf(x) < 8388608.0f);                   // element >= 2^E, inf or NaN
                int d0 = (int)r, d1 = d0;
                if (fabsf(x - r) == 0.5f) {                        // tie: pick the even mantissa
                    const int fl = (int)floorf(x);
                    d0 = ((par0 + fl) & 1) ? fl + 1 : fl;
                    d1 = ((par1 + fl) & 1) ? fl + 1 : fl;
                }
                run0 += d0; par0 = (par0 + d0) & 1;
                run1 += d1; par1 = (par1 + d1) & 1;
                mn0 = run0 < mn0 ? run0 : mn0; mx0 = run0 > mx0 ? run0 : mx0;
                mn1 = run1 < mn1 ? run1 : mn1; mx1 = run1 > mx1 ? run1 : mx1;
            }
            long long c0 = run0, c1 = run1;
            ms_scan_pairs(c0, c1);
            const long long m_cur = (long long)((sb & 0x7FFFFFu) | 0x800000u);
            const int pc = (int)(m_cur & 1);
            long long e0 = __shfl_up(c0, 1, 64), e1 = __shfl_up(c1, 1, 64);
            if (l == 0) { e0 = 0; e1 = 0; }
            const long long m_in = m_cur + (pc ? e1 : e0);
            const int pl = (int)(m_in & 1);
            const long long hi = pl ? mx1 : mx0, lo = pl ? mn1 : mn0;
            const bool ok = !bad && m_in + hi + 1 < (1ll << 24) && m_in + lo - 1 >= (1ll << 23);
            const unsigned long long fail = __ballot(base < cnt && !ok);
            f = fail ? (int)__builtin_ctzll(fail) : 64;
            if (f > 0) {
                const long long m1 = m_cur + ms_readlane64(pc ? c1 : c0, f - 1);
                sb = (sb & 0xFF800000u) | ((uint32_t)m1 & 0x7FFFFFu);
                pos += MS_SEG * f;
            }
            if (!fail) break;                                      // whole remainder applied
        }
        // add segment [pos, pos+16) one element at a time (all lanes redundantly, LDS broadcast)
        float s = __uint_as_float(sb);
        const int end = pos + MS_SEG < cnt ? pos + MS_SEG : cnt;
        for (int i = pos; i < end; ++i) s = s + stage[ms_pad(i)];
        sb = __builtin_amdgcn_readfirstlane(__float_as_uint(s));
        pos = end;
    }
    __syncthreads();
    return sb;
}

// ---- level 2: one table row per 64 level-1 blocks (65 536 points), same candidate layout.
// Order-free bounds are additive, so a parent row is the sum of its children's rows taken at
// the same absolute binade; a child contributes nothing to candidates more than 24 binades
// above its own largest element.
__global__ __launch_bounds__(256) void ms_level2_k(const MsHdr* __restrict__ hdr, const long long* __restrict__ apos,
                                                   const long long* __restrict__ aneg, int64_t nb, int64_t nb2,
                                                   MsHdr* __restrict__ hdr2, long long* __restrict__ apos2,
                                                   long long* __restrict__ aneg2) {
    const int64_t w = (int64_t)blockIdx.x * 4 + wave_id();
    if (w >= 3 * nb2) return;
    const int c = (int)(w / nb2);
    const int64_t g = w % nb2;
    const int l = lane_id();
    const int64_t bb = g * 64 + l;
    const bool valid = bb < nb;
    MsHdr h;
    h.emax = -200; h.tie = 0; h.flags = MS_ALLZERO; h.pad = 0;
    if (valid) h = hdr[(int64_t)c * nb + bb];
    const bool zero = (h.flags & MS_ALLZERO) != 0;
    const int emax2 = wave_reduce_max(zero ? -200 : h.emax);
    const uint32_t nonfinite = __ballot((h.flags & MS_NONFINITE) != 0) ? MS_NONFINITE : 0u;
    const bool allzero = __ballot(!zero) == 0;
    uint32_t tie2 = 0;
    const int shift = emax2 - h.emax;                      // >= 0 for non-zero children
    for (int j2 = 0; j2 < MS_CAND; ++j2) {
        const int j = j2 + shift;
        long long p = 0, q = 0;
        bool tie = false;
        if (valid && !zero && !(h.flags & MS_NONFINITE) && j < MS_CAND) {
            const int64_t at = ((int64_t)c * nb + bb) * MS_CAND + j;
            p = apos[at]; q = aneg[at];
            tie = (h.tie >> j) & 1u;
        }
        p = wave_reduce_add(p);
        q = wave_reduce_add(q);
        if (__ballot(tie)) tie2 |= 1u << j2;
        if (l == 0) {
            const int64_t at2 = ((int64_t)c * nb2 + g) * MS_CAND + j2;
            apos2[at2] = p; aneg2[at2] = q;
        }
    }
    if (l == 0) {
        MsHdr o;
        o.emax = allzero ? 0 : emax2;
        o.tie = tie2;
        o.flags = nonfinite | (allzero ? MS_ALLZERO : 0u);
        o.pad = 0;
        hdr2[(int64_t)c * nb2 + g] = o;
    }
}

// what one lane knows about its level-1 block at the current binade
struct MsLane {
    int       cls;      // 0: s cannot change, 1: order-free table bounds, 2: unknown at this E
    long long net0, net1, lo0, lo1, hi0, hi1;
    uint32_t  flags;
    int       j;
    bool      valid;
    int64_t   bb;
};

struct MsTables {
    const MsHdr* hdr; const long long* apos; const long long* aneg; int64_t nb;      // level 1
    const MsHdr* hdr2; const long long* apos2; const long long* aneg2; int64_t nb2;  // level 2
};

__device__ __forceinline__ int ms_classify(const MsHdr& h, bool valid, bool s_inf, bool s_norm, int E, int& j) {
    j = E - h.emax - 1;
    if (!valid || (h.flags & MS_ALLZERO)) return 0;
    if (s_inf) return (h.flags & MS_NONFINITE) ? 2 : 0;
    if (!s_norm || (h.flags & MS_NONFINITE) || j < 0) return 2;
    if (j >= MS_CAND) return 0;
    return ((h.tie >> j) & 1u) ? 2 : 1;
}

// Adds the level-1 blocks [first, first+count), count <= 64, to the running sum `sb` (bits).
__device__ __forceinline__ uint32_t ms_walk_children(const float* __restrict__ xyz, int64_t n, int c,
                                                     const MsTables& T, int64_t first, int count,
                                                     uint32_t sb, float* stage, int& n_serial) {
    const int l = lane_id();
    int done = 0;                                       // children already added
    while (done < count) {
        const uint32_t ef = (sb >> 23) & 0xFFu;
        if (ef == 255u && (sb & 0x7FFFFFu)) return sb;  // NaN is absorbing
        const bool s_inf = ef == 255u;
        const bool s_norm = ef >= 1u && ef <= 254u;
        const bool s_neg = (sb >> 31) != 0;
        const int E = (int)ef - 127;
        MsLane me;
        me.bb = first + l;
        me.valid = l >= done && l < count;
        MsHdr h;
        h.emax = 0; h.tie = 0; h.flags = MS_ALLZERO; h.pad = 0;
        if (me.valid) h = T.hdr[(int64_t)c * T.nb + me.bb];
        me.cls = ms_classify(h, me.valid, s_inf, s_norm, E, me.j);
        me.flags = h.flags;
        me.net0 = me.net1 = 0; me.lo0 = me.lo1 = 0; me.hi0 = me.hi1 = 0;
        if (me.cls == 1) {
            const int64_t at = ((int64_t)c * T.nb + me.bb) * MS_CAND + me.j;
            const long long p = T.apos[at], q = T.aneg[at];
            const long long up = s_neg ? q : p, dn = s_neg ? p : q;
            me.net0 = me.net1 = up - dn; me.lo0 = me.lo1 = -dn; me.hi0 = me.hi1 = up;
        }
        int start = done;                               // first unresolved lane
        long long m_cur = (long long)((sb & 0x7FFFFFu) | 0x800000u);   // mantissa entering `start`
        for (;;) {
            long long c0 = l >= start ? me.net0 : 0ll, c1 = l >= start ? me.net1 : 0ll;
            ms_scan_pairs(c0, c1);                      // inclusive: increment through lane l
            const int pc = (int)(m_cur & 1);
            const long long incl = pc ? c1 : c0;
            long long e0 = __shfl_up(c0, 1, 64), e1 = __shfl_up(c1, 1, 64);
            if (l == 0) { e0 = 0; e1 = 0; }
            const long long m_in = m_cur + (pc ? e1 : e0);
            const int pl = (int)(m_in & 1);
            const long long hi = pl ? me.hi1 : me.hi0, lo = pl ? me.lo1 : me.lo0;
            const bool ok = me.cls == 0 || (me.cls == 1 && m_in + hi + 1 < (1ll << 24) &&
                                            m_in + lo - 1 >= (1ll << 23));
            const unsigned long long fail = __ballot(me.valid && l >= start && !ok);
            const int f = fail ? (int)__builtin_ctzll(fail) : count;
            if (f > start && s_norm) {                  // advance over the certified lanes
                m_cur += ms_readlane64(incl, f - 1);
                sb = (sb & 0xFF800000u) | ((uint32_t)m_cur & 0x7FFFFFu);
            }
            start = f;
            if (!fail) { done = count; break; }
            // child f cannot be certified from the table at this E: add it exactly
            ++n_serial;
            const uint32_t nsb = ms_block_exact(xyz, n, c, first + f, sb, stage);
            const bool same = ((nsb ^ sb) & 0xFF800000u) == 0 && s_norm;   // same sign and binade
            sb = nsb;
            start = f + 1;
            done = start;
            if (same) m_cur = (long long)((sb & 0x7FFFFFu) | 0x800000u);
            else break;                                 // re-read the remaining children at the new E
        }
    }
    return sb;
}

// one wave per column: walks the level-2 rows, descends into the children of a row only when
// its certificate fails
__global__ __launch_bounds__(64) void ms_walk_k(const float* __restrict__ xyz, int64_t n, MsTables T,
                                                float* __restrict__ out, int* __restrict__ stats) {
    __shared__ float stage[MSB + MSB / 16];
    const int c = blockIdx.x;
    const int l = lane_id();
    uint32_t sb = 0;                                   // bits of the running sum (+0.0)
    int64_t b = 0;                                     // next level-2 row
    int n_serial = 0, n_batches = 0, n_desc = 0;
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
        h.emax = 0; h.tie = 0; h.flags = MS_ALLZERO; h.pad = 0;
        if (valid) h = T.hdr2[(int64_t)c * T.nb2 + bb];
        int j;
        const int cls = ms_classify(h, valid, s_inf, s_norm, E, j);
        long long net = 0, lo = 0, hi = 0;
        if (cls == 1) {
            const int64_t at = ((int64_t)c * T.nb2 + bb) * MS_CAND + j;
            const long long p = T.apos2[at], q = T.aneg2[at];
            const long long up = s_neg ? q : p, dn = s_neg ? p : q;
            net = up - dn; lo = -dn; hi = up;
        }
        int start = 0;
        long long m_cur = (long long)((sb & 0x7FFFFFu) | 0x800000u);
        bool reload = false;
        while (!reload) {
            const long long incl = wave_scan_incl(l >= start ? net : 0ll);
            const long long m_in = m_cur + incl - (l >= start ? net : 0ll);
            const bool ok = cls == 0 || (cls == 1 && m_in + hi + 1 < (1ll << 24) && m_in + lo - 1 >= (1ll << 23));
            const unsigned long long fail = __ballot(valid && l >= start && !ok);
            const int f = fail ? (int)__builtin_ctzll(fail) : 64;
            if (f > start && s_norm) {
                m_cur += ms_readlane64(incl, f - 1);
                sb = (sb & 0xFF800000u) | ((uint32_t)m_cur & 0x7FFFFFu);
            }
            start = f;
            if (!fail) break;
            // descend into the 64 children of row b+f
            ++n_desc;
            const int64_t first = (b + f) * 64;
            const int count = (int)((T.nb - first) < 64 ? (T.nb - first) : 64);
            const uint32_t nsb = ms_walk_children(xyz, n, c, T, first, count, sb, stage, n_serial);
            const bool same = ((nsb ^ sb) & 0xFF800000u) == 0 && s_norm;
            sb = nsb;
            start = f + 1;
            if (same) m_cur = (long long)((sb & 0x7FFFFFu) | 0x800000u);
            else reload = true;
        }
        b += reload ? start : 64;
    }
    if (l == 0) {
        out[c] = __uint_as_float(sb) / (float)n;       // n == 0 -> 0/0 = NaN like numpy
        if (stats) {
            stats[4 * c + 0] = n_batches; stats[4 * c + 1] = 0;
            stats[4 * c + 2] = n_serial; stats[4 * c + 3] = n_desc;
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

struct MsWs {
    int*       stats;            // [3][4]: level-2 batches, -, exactly added blocks, descents
    MsHdr     *hdr, *hdr2;
    long long *apos, *aneg, *apos2, *aneg2;
};
static void ms_plan(Arena& a, int64_t n, MsWs& w) {
    const int64_t nb = ceil_div(n > 0 ? n : 1, MSB);
    w.stats = a.take<int>(16);
    w.hdr = a.take<MsHdr>(3 * nb);
    w.apos = a.take<long long>(3 * nb * MS_CAND);
    w.aneg = a.take<long long>(3 * nb * MS_CAND);
    const int64_t nb2 = ceil_div(nb, 64);
    w.hdr2 = a.take<MsHdr>(3 * nb2);
    w.apos2 = a.take<long long>(3 * nb2 * MS_CAND);
    w.aneg2 = a.take<long long>(3 * nb2 * MS_CAND);
}
static int mean_seq_launch(const float* xyz, int64_t n, float* out, MsWs& w, hipStream_t s) {
    const int64_t nb = n > 0 ? ceil_div(n, MSB) : 0;
    const int64_t nb2 = ceil_div(nb, 64);
    if (n > 0) {
        PCH_LAUNCH("mean_summary", ms_summary_k, dim3((unsigned)ceil_div(nb, MS_WAVES)), dim3(64 * MS_WAVES),
                   0, s, xyz, n, nb, w.hdr, w.apos, w.aneg);
        PCH_LAUNCH("mean_level2", ms_level2_k, dim3((unsigned)ceil_div(3 * nb2, 4)), dim3(256), 0, s,
                   (const MsHdr*)w.hdr, (const long long*)w.apos, (const long long*)w.aneg, nb, nb2,
                   w.hdr2, w.apos2, w.aneg2);
    }
    MsTables T;
    T.hdr = w.hdr; T.apos = w.apos; T.aneg = w.aneg; T.nb = nb;
    T.hdr2 = w.hdr2; T.apos2 = w.apos2; T.aneg2 = w.aneg2; T.nb2 = nb2;
    PCH_LAUNCH("mean_walk", ms_walk_k, dim3(3), dim3(64), 0, s, xyz, n, T, out, w.stats);
    return PCH_OK;
}

// =====================================================================================
// B2: k-th order statistics of v[i] = base[i*stride] - sub by 12/12/8-bit radix select on the
// order-preserving uint32 image of the float.  NaN sorts last (as numpy's partition).
// =====================================================================================
constexpr int SEL_BINS = 4096;

struct SelState {
    unsigned long long rank;       // remaining rank inside the current prefix
    unsigned long long less;       // elements strictly below the current prefix
    uint32_t prefix;               // resolved high bits
    uint32_t need_next;            // 1: k1 is not inside the final bin of k0
    uint32_t v0key, v1key;
    unsigned long long nan_count;
    uint32_t next_min;             // min key > v0key (pass 4)
    uint32_t pad;
};

__device__ __forceinline__ uint32_t sel_key(float v) {
    return (v != v) ? 0xFFFFFFFFu : f32_ordered(v);
}

template <int PASS>
__global__ __launch_bounds__(256) void sel_hist_k(const float* __restrict__ base, int64_t n,
                                                  int64_t stride, const float* __restrict__ sub,
                                                  SelState* __restrict__ st,
                                                  uint32_t* __restrict__ hist) {
    __shared__ uint32_t h[SEL_BINS];
    for (int j = threadIdx.x; j < SEL_BINS; j += 256) h[j] = 0;
    __syncthreads();
    const float c = sub ? *sub : 0.0f;
    const uint32_t prefix = st->prefix;
    unsigned long long nans = 0;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
        const float v = base[i * stride] - c;
        const uint32_t k = sel_key(v);
        if (PASS == 0) {
            atomicAdd(&h[k >> 20], 1u);
            nans += (v != v);
        } else if (PASS == 1) {
            if ((k >> 20) == prefix) atomicAdd(&h[(k >> 8) & 0xFFFu], 1u);
        } else {
            if ((k >> 8) == prefix) atomicAdd(&h[k & 0xFFu], 1u);
        }
    }
    __syncthreads();
    const int nb = (PASS == 2) ? 256 : SEL_BINS;
    for (int j = threadIdx.x; j < nb; j += 256)
        if (h[j]) atomicAdd(&hist[j], h[j]);
    if (PASS == 0) {
        nans = wave_reduce_add(nans);
        if (lane_id() == 0 && nans) atomicAdd(&st->nan_count, nans);
    }
}

// single workgroup: locate the bin holding the wanted rank, extend the prefix, clear hist
template <int PASS>
__global__ __launch_bounds__(256) void sel_pick_k(SelState* __restrict__ st, uint32_t* __restrict__ hist) {
    __shared__ unsigned long long wsum[4];
    __shared__ int found_bin;
    __shared__ unsigned long long found_below;
    const int nb = (PASS == 2) ? 256 : SEL_BINS;
    const int per = nb / 256;                       // bins per thread (16 or 1)
    const unsigned long long rank = st->rank;
    unsigned long long loc[16];
    unsigned long long tsum = 0;
    for (int j = 0; j < per; ++j) { loc[j] = hist[threadIdx.x * per + j]; tsum += loc[j]; }
    const unsigned long long incl = wave_scan_incl(tsum);
    if (lane_id() == 63) wsum[wave_id()] = incl;
    if (threadIdx.x == 0) found_bin = -1;
    __syncthreads();
    unsigned long long before = incl - tsum;
    for (int w = 0; w < wave_id(); ++w) before += wsum[w];
    // the thread whose bin range [before, before+tsum) contains rank owns the answer
    if (rank >= before && rank < before + tsum) {
        unsigned long long b = before;
        for (int j = 0; j < per; ++j) {
            if (rank < b + loc[j]) { found_bin = threadIdx.x * per + j; found_below = b; break; }
            b += loc[j];
        }
    }
    __syncthreads();
    for (int j = threadIdx.x; j < nb; j += 256) {
        if (PASS == 2 && j == found_bin) {
            // does rank+1 (the 'next' order statistic) still fall into this final bin?
            const unsigned long long cnt = hist[j];
            st->need_next = (rank + 1 < found_below + cnt) ? 0u : 1u;
        }
    }
    __syncthreads();
    for (int j = threadIdx.x; j < nb; j += 256) hist[j] = 0;
    if (threadIdx.x == 0) {
        const int bin = found_bin < 0 ? 0 : found_bin;   // n == 0 never reaches here
        st->less += found_below;
        st->rank = rank - found_below;
        if (PASS == 0) st->prefix = (uint32_t)bin;
        else if (PASS == 1) st->prefix = (st->prefix << 12) | (uint32_t)bin;
        else { st->v0key = (st->prefix << 8) | (uint32_t)bin; st->next_min = 0xFFFFFFFFu; }
    }
}

// pass 4 (only when needed): smallest key strictly above v0key
__global__ __launch_bounds__(256) void sel_next_k(const float* __restrict__ base, int64_t n,
                                                  int64_t stride, const float* __restrict__ sub,
                                                  SelState* __restrict__ st) {
    if (st->need_next == 0) return;
    const float c = sub ? *sub : 0.0f;
    const uint32_t v0 = st->v0key;
    uint32_t best = 0xFFFFFFFFu;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
        const uint32_t k = sel_key(base[i * stride] - c);
        if (k > v0 && k < best) best = k;
    }
    best = wave_reduce_min(best);
    if (lane_id() == 0 && best != 0xFFFFFFFFu) atomicMin(&st->next_min, best);
}

__device__ __forceinline__ float sel_key_to_float(uint32_t k) {
    return (k == 0xFFFFFFFFu) ? __uint_as_float(0x7FC00000u) : f32_unordered(k);
}

// numpy _lerp in float32: a + (b-a)*t, and b - (b-a)*(1-t) where t >= 0.5
// (numpy/lib/_function_base_impl.py:4639-4660); NaN anywhere -> NaN.
// scal: [0] = percentile, [1] = percentile + add1, [2] = percentile + add2
__global__ void sel_lerp_k(const SelState* __restrict__ st, int same_index, float gamma,
                           float add1, float add2, float* __restrict__ scal) {
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    const float a = sel_key_to_float(st->v0key);
    float b = a;
    if (!same_index && st->need_next) b = sel_key_to_float(st->next_min);
    const float diff = b - a;
    float r = a + diff * gamma;
    if (gamma >= 0.5f) r = b - diff * (1.0f - gamma);
    if (st->nan_count) r = __uint_as_float(0x7FC00000u);
    scal[0] = r;
    scal[1] = r + add1;
    scal[2] = r + add2;
}

__global__ void sel_init_k(SelState* st, unsigned long long rank) { st->rank = rank; }

struct SelWs {
    SelState* st;
    uint32_t* hist;
    float*    scal;     // 4 floats
};
static void sel_plan(Arena& a, SelWs& w) {
    w.st = a.take<SelState>(1);
    w.hist = a.take<uint32_t>(SEL_BINS);
    w.scal = a.take<float>(4);
}

// host side of np.percentile's index arithmetic (float32 under NEP 50)
struct PctIndex { int64_t k0; int same; float gamma; };
static PctIndex pct_index(int64_t n, double q_percent) {
    PctIndex r;
    const float q = (float)q_percent / 100.0f;          // np.true_divide(q, float32(100))
    const float vi = (float)(n - 1) * q;                // (n - 1) * quantiles -> float32
    float prev = floorf(vi);
    r.gamma = vi - prev;
    r.same = 0;
    if (vi >= (float)(n - 1)) { prev = (float)(n - 1); r.same = 1; }   // indexes -> -1 (last)
    if (vi < 0.0f) { prev = 0.0f; r.same = 1; }
    int64_t k0 = (int64_t)prev;
    if (k0 > n - 1) k0 = n - 1;                          // float32(n-1) may round up
    if (k0 < 0) k0 = 0;
    if (k0 == n - 1) r.same = 1;
    r.k0 = k0;
    return r;
}

static int select_percentile(const float* base, int64_t n, int64_t stride, const float* sub,
                             double q_percent, float add1, float add2, SelWs& w, hipStream_t s) {
    const PctIndex pi = pct_index(n, q_percent);
    SelState init;
    memset(&init, 0, sizeof(init));
    init.rank = (unsigned long long)pi.k0;
    PCH_HIP_TRY(hipMemsetAsync(w.hist, 0, sizeof(uint32_t) * SEL_BINS, s));
    PCH_HIP_TRY(hipMemsetAsync(w.st, 0, sizeof(SelState), s));
    PCH_LAUNCH("sel_init", sel_init_k, dim3(1), dim3(1), 0, s, w.st, init.rank);
    int64_t gb = ceil_div(n, 256 * 16);
    if (gb > 4096) gb = 4096;
    if (gb < 1) gb = 1;
    const dim3 grid((unsigned)gb), blk(256);
    PCH_LAUNCH("sel_hist0", sel_hist_k<0>, grid, blk, 0, s, base, n, stride, sub, w.st, w.hist);
    PCH_LAUNCH("sel_pick0", sel_pick_k<0>, dim3(1), blk, 0, s, w.st, w.hist);
    PCH_LAUNCH("sel_hist1", sel_hist_k<1>, grid, blk, 0, s, base, n, stride, sub, w.st, w.hist);
    PCH_LAUNCH("sel_pick1", sel_pick_k<1>, dim3(1), blk, 0, s, w.st, w.hist);
    PCH_LAUNCH("sel_hist2", sel_hist_k<2>, grid, blk, 0, s, base, n, stride, sub, w.st, w.hist);
    PCH_LAUNCH("sel_pick2", sel_pick_k<2>, dim3(1), blk, 0, s, w.st, w.hist);
    if (!pi.same)
        PCH_LAUNCH("sel_next", sel_next_k, grid, blk, 0, s, base, n, stride, sub, w.st);
    PCH_LAUNCH("sel_lerp", sel_lerp_k, dim3(1), dim3(64), 0, s, (const SelState*)w.st, pi.same, pi.gamma,
               add1, add2, w.scal);
    return PCH_OK;
}

// =====================================================================================
// B2/B3: keep = (z - cz) > thr, order preserving.  Two launches over the raw points:
// count (both thresholds at once, so the <min_keep fallback needs no host round trip),
// then scatter of the centred coordinates.
// =====================================================================================
constexpr int GF_THREADS = 256;
constexpr int GF_ROUNDS  = 8;
constexpr int GF_TILE    = GF_THREADS * GF_ROUNDS;   // 2048 points per workgroup

struct GfState {
    unsigned long long total_a, total_b;   // kept with threshold A (offset) / B (fallback)
    uint32_t use_b;
    uint32_t aabb[6];                      // ordered-uint32 min xyz / max xyz
    uint32_t pad;
};

__global__ __launch_bounds__(GF_THREADS) void gf_count_k(const float* __restrict__ raw, int64_t n,
                                                         const float* __restrict__ centroid,
                                                         const float* __restrict__ scal,
                                                         uint32_t* __restrict__ cnt_a,
                                                         uint32_t* __restrict__ cnt_b,
                                                         GfState* __restrict__ st) {
    __shared__ uint32_t sa[GF_THREADS / 64], sb[GF_THREADS / 64];
    const float cz = centroid[2];
    const float thr_a = scal[1], thr_b = scal[2];
    const int64_t base = (int64_t)blockIdx.x * GF_TILE;
    uint32_t a = 0, b = 0;
#pragma unroll
    for (int r = 0; r < GF_ROUNDS; ++r) {
        const int64_t i = base + r * GF_THREADS + threadIdx.x;
        if (i < n) {
            const float z = raw[3 * i + 2] - cz;
            a += (z > thr_a);
            b += (z > thr_b);
        }
    }
    a = wave_reduce_add(a);
    b = wave_reduce_add(b);
    if (lane_id() == 0) { sa[wave_id()] = a; sb[wave_id()] = b; }
    __syncthreads();
    if (threadIdx.x == 0) {
        const uint32_t ta = sa[0] + sa[1] + sa[2] + sa[3];
        const uint32_t tb = sb[0] + sb[1] + sb[2] + sb[3];
        cnt_a[blockIdx.x] = ta;
        cnt_b[blockIdx.x] = tb;
        if (ta) atomicAdd(&st->total_a, (unsigned long long)ta);
        if (tb) atomicAdd(&st->total_b, (unsigned long long)tb);
    }
}

// picks the threshold (utils/tower_extraction.py:87-89) and publishes the scalars
__global__ void gf_decide_k(GfState* __restrict__ st, long long min_keep,
                            const float* __restrict__ centroid, float* __restrict__ scal,
                            float* __restrict__ out_scalars, int64_t* __restrict__ out_count) {
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    const bool use_b = (long long)st->total_a < min_keep;
    st->use_b = use_b ? 1u : 0u;
    const float thr = use_b ? scal[2] : scal[1];
    scal[3] = thr;
    out_scalars[0] = centroid[0];
    out_scalars[1] = centroid[1];
    out_scalars[2] = centroid[2];
    out_scalars[3] = scal[0];
    out_scalars[4] = thr;
    out_scalars[5] = use_b ? 1.0f : 0.0f;
    out_scalars[6] = (float)st->total_a;      // kept at the first threshold (exact below 2^24)
    out_scalars[7] = 0.0f;
    *out_count = (int64_t)(use_b ? st->total_b : st->total_a);
    for (int a = 0; a < 3; ++a) { st->aabb[a] = 0xFFFFFFFFu; st->aabb[3 + a] = 0u; }
}

__global__ __launch_bounds__(GF_THREADS) void gf_scatter_k(
    const float* __restrict__ raw, int64_t n, const float* __restrict__ centroid,
    const float* __restrict__ scal, const uint32_t* __restrict__ off_a,
    const uint32_t* __restrict__ off_b, GfState* __restrict__ st,
    float* __restrict__ out_points, int32_t* __restrict__ out_index) {
    __shared__ uint32_t wtot[GF_THREADS / 64];
    __shared__ uint32_t smm[GF_THREADS / 64][6];
    const float cx = centroid[0], cy = centroid[1], cz = centroid[2];
    const float thr = scal[3];
    const uint32_t block_off = st->use_b ? off_b[blockIdx.x] : off_a[blockIdx.x];
    const int w = wave_id(), l = lane_id();
    const int64_t seg = (int64_t)blockIdx.x * GF_TILE + (int64_t)w * (64 * GF_ROUNDS);
    float px[GF_ROUNDS], py[GF_ROUNDS], pz[GF_ROUNDS];
    uint32_t pos[GF_ROUNDS];
    uint32_t run = 0;
    uint32_t mn[3] = {0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu}, mx[3] = {0u, 0u, 0u};
    const uint64_t lt = lanemask_lt();
#pragma unroll
    for (int r = 0; r < GF_ROUNDS; ++r) {
        const int64_t i = seg + r * 64 + l;
        bool keep = false;
        if (i < n) {
            px[r] = raw[3 * i + 0] - cx;          // points = raw_points - centroid (float32)
            py[r] = raw[3 * i + 1] - cy;
            pz[r] = raw[3 * i + 2] - cz;
            keep = pz[r] > thr;
        }
        const uint64_t m = __ballot(keep);
        pos[r] = keep ? run + (uint32_t)__popcll(m & lt) : 0xFFFFFFFFu;
        run += (uint32_t)__popcll(m);
        if (keep) {
            const uint32_t kx = f32_ordered(px[r]), ky = f32_ordered(py[r]), kz = f32_ordered(pz[r]);
            mn[0] = kx < mn[0] ? kx : mn[0]; mx[0] = kx > mx[0] ? kx : mx[0];
            mn[1] = ky < mn[1] ? ky : mn[1]; mx[1] = ky > mx[1] ? ky : mx[1];
            mn[2] = kz < mn[2] ? kz : mn[2]; mx[2] = kz > mx[2] ? kz : mx[2];
        }
    }
    if (l == 0) wtot[w] = run;
#pragma unroll
    for (int a = 0; a < 3; ++a) { mn[a] = wave_reduce_min(mn[a]); mx[a] = wave_reduce_max(mx[a]); }
    if (l == 0) {
#pragma unroll
        for (int a = 0; a < 3; ++a) { smm[w][a] = mn[a]; smm[w][3 + a] = mx[a]; }
    }
    __syncthreads();
    uint32_t woff = block_off;
    for (int w2 = 0; w2 < w; ++w2) woff += wtot[w2];
#pragma unroll
    for (int r = 0; r < GF_ROUNDS; ++r) {
        if (pos[r] != 0xFFFFFFFFu) {
            const int64_t o = (int64_t)woff + pos[r];
            out_points[3 * o + 0] = px[r];
            out_points[3 * o + 1] = py[r];
            out_points[3 * o + 2] = pz[r];
            if (out_index) out_index[o] = (int32_t)(seg + r * 64 + l);
        }
    }
    if (threadIdx.x < 6) {
        const int a = threadIdx.x;
        uint32_t v = smm[0][a];
        for (int w2 = 1; w2 < GF_THREADS / 64; ++w2) {
            const uint32_t o = smm[w2][a];
            v = (a < 3) ? (o < v ? o : v) : (o > v ? o : v);
        }
        if (a < 3) { if (v != 0xFFFFFFFFu) atomicMin(&st->aabb[a], v); }
        else       { if (v != 0u) atomicMax(&st->aabb[a], v); }
    }
}

__global__ void gf_finalize_k(const GfState* __restrict__ st, float* __restrict__ out_aabb) {
    if (threadIdx.x < 6 && blockIdx.x == 0 && out_aabb) {
        const uint32_t k = st->aabb[threadIdx.x];
        const bool empty = (threadIdx.x < 3) ? (k == 0xFFFFFFFFu) : (k == 0u);
        out_aabb[threadIdx.x] = empty ? 0.0f : f32_unordered(k);
    }
}

struct GfWs {
    float*    centroid;
    MsWs      ms;
    SelWs     sel;
    GfState*  st;
    uint32_t *cnt_a, *cnt_b, *scan_ws;
};
static void gf_plan(Arena& a, int64_t n, GfWs& w) {
    const int64_t nb = ceil_div(n > 0 ? n : 1, GF_TILE);
    w.centroid = a.take<float>(4);
    ms_plan(a, n, w.ms);
    sel_plan(a, w.sel);
    w.st = a.take<GfState>(1);
    w.cnt_a = a.take<uint32_t>(nb + 8);
    w.cnt_b = a.take<uint32_t>(nb + 8);
    w.scan_ws = a.take<uint32_t>(scan_ws_u32(nb));
}

}  // namespace pch

using namespace pch;

extern "C" size_t pch_mean_seq_f32_ws_bytes(int64_t n) {
    if (n < 0) return 0;
    Arena a;
    MsWs w;
    ms_plan(a, n, w);
    return a.off;
}

extern "C" int pch_mean_seq_f32(const float* xyz, int64_t n, float* out_centroid, void* ws,
                                size_t ws_bytes, void* stream) {
    prof_begin_call();
    PCH_REQUIRE(n >= 0 && out_centroid && ws, "bad argument");
    PCH_REQUIRE(n == 0 || xyz, "null input");
    Arena a(ws, ws_bytes);
    MsWs w;
    ms_plan(a, n, w);
    if (a.overflow) { set_error("workspace too small: need %zu bytes", a.off); return PCH_ERR_WORKSPACE; }
    return mean_seq_launch(xyz, n, out_centroid, w, (hipStream_t)stream);
}

extern "C" int pch_mean_seq_serial_f32(const float* xyz, int64_t n, float* out_centroid, void* stream) {
    prof_begin_call();
    PCH_REQUIRE(n >= 0 && out_centroid, "bad argument");
    PCH_REQUIRE(n == 0 || xyz, "null input");
    PCH_LAUNCH("mean_seq_serial", mean_seq_k, dim3(1), dim3(256), 0, (hipStream_t)stream, xyz, n, out_centroid);
    return PCH_OK;
}

extern "C" size_t pch_percentile_f32_ws_bytes(int64_t) {
    Arena a;
    SelWs w;
    sel_plan(a, w);
    return a.off;
}

extern "C" int pch_percentile_f32(const float* base, int64_t n, int64_t stride, const float* sub,
                                  double q_percent, float* out, void* ws, size_t ws_bytes,
                                  void* stream) {
    prof_begin_call();
    hipStream_t s = (hipStream_t)stream;
    PCH_REQUIRE(n >= 1, "percentile of an empty array (numpy raises IndexError)");
    PCH_REQUIRE(base && out && ws && stride >= 1, "bad argument");
    PCH_REQUIRE(q_percent >= 0.0 && q_percent <= 100.0, "Percentiles must be in the range [0, 100]");
    Arena a(ws, ws_bytes);
    SelWs w;
    sel_plan(a, w);
    if (a.overflow) { set_error("workspace too small: need %zu bytes", a.off); return PCH_ERR_WORKSPACE; }
    PCH_TRY(select_percentile(base, n, stride, sub, q_percent, 0.0f, 0.0f, w, s));
    PCH_HIP_TRY(hipMemcpyAsync(out, w.scal, sizeof(float), hipMemcpyDeviceToDevice, s));
    return PCH_OK;
}

extern "C" size_t pch_ground_filter_ws_bytes(int64_t n) {
    if (n < 0) return 0;
    Arena a;
    GfWs w;
    gf_plan(a, n, w);
    return a.off;
}

extern "C" int pch_ground_filter_f32(const float* raw, int64_t n, double pct, float offset,
                                     float fallback_offset, int64_t min_keep, float* out_points,
                                     int32_t* out_index, float* out_scalars, int64_t* out_count,
                                     float* out_aabb, void* ws, size_t ws_bytes, void* stream) {
    prof_begin_call();
    hipStream_t s = (hipStream_t)stream;
    PCH_REQUIRE(n >= 1 && n < (int64_t(1) << 31), "n out of range [1, 2^31) (numpy raises on empty input)");
    PCH_REQUIRE(raw && out_points && out_scalars && out_count && ws, "null buffer");
    PCH_REQUIRE(pct >= 0.0 && pct <= 100.0, "Percentiles must be in the range [0, 100]");
    Arena a(ws, ws_bytes);
    GfWs w;
    gf_plan(a, n, w);
    if (a.overflow) { set_error("workspace too small: need %zu bytes", a.off); return PCH_ERR_WORKSPACE; }
    const int64_t nb = ceil_div(n, GF_TILE);

    PCH_TRY(mean_seq_launch(raw, n, w.centroid, w.ms, s));
    PCH_TRY(select_percentile(raw + 2, n, 3, w.centroid + 2, pct, offset, fallback_offset, w.sel, s));
    PCH_HIP_TRY(hipMemsetAsync(w.st, 0, sizeof(GfState), s));
    PCH_LAUNCH("gf_count", gf_count_k, dim3((unsigned)nb), dim3(GF_THREADS), 0, s, raw, n,
               (const float*)w.centroid, (const float*)w.sel.scal, w.cnt_a, w.cnt_b, w.st);
    PCH_LAUNCH("gf_decide", gf_decide_k, dim3(1), dim3(64), 0, s, w.st, (long long)min_keep,
               (const float*)w.centroid, w.sel.scal, out_scalars, out_count);
    PCH_TRY(scan_exclusive_u32(w.cnt_a, w.cnt_a, nb, w.scan_ws, nullptr, s));
    PCH_TRY(scan_exclusive_u32(w.cnt_b, w.cnt_b, nb, w.scan_ws, nullptr, s));
    PCH_LAUNCH("gf_scatter", gf_scatter_k, dim3((unsigned)nb), dim3(GF_THREADS), 0, s, raw, n,
               (const float*)w.centroid, (const float*)w.sel.scal, (const uint32_t*)w.cnt_a,
               (const uint32_t*)w.cnt_b, w.st, out_points, out_index);
    PCH_LAUNCH("gf_finalize", gf_finalize_k, dim3(1), dim3(64), 0, s, (const GfState*)w.st, out_aabb);
    return PCH_OK;
}
