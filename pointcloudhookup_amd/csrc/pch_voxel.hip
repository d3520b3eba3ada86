// Stage A: per-chunk voxel-grid downsample (Open3D VoxelDownSample semantics) as
// key -> radix sort -> segmented in-order float64 mean.
// Reference call site: ui/import_PC.py:8-13 inside the chunk loop ui/import_PC.py:45-58.
#include "pch_prims.h"

namespace pch {

constexpr int VX_THREADS = 256;

struct VoxelGrid {
    int64_t chunk_size;
    double  voxel;
    int     bx, by, bz;      // bits per axis of the packed key
};

// ---- per-chunk float64 min / max (ordered-uint64 atomics) ---------------------------
__global__ __launch_bounds__(VX_THREADS) void vx_minmax_k(const double* __restrict__ xyz,
                                                          int64_t n, int64_t chunk_size,
                                                          int64_t blocks_per_chunk,
                                                          unsigned long long* __restrict__ mm) {
    __shared__ unsigned long long sm[VX_THREADS / 64][6];
    const int64_t chunk = blockIdx.x / blocks_per_chunk;
    const int64_t tile  = blockIdx.x % blocks_per_chunk;
    const int64_t cbeg = chunk * chunk_size;
    const int64_t cend = (cbeg + chunk_size < n) ? cbeg + chunk_size : n;
    unsigned long long lo[3] = {~0ull, ~0ull, ~0ull}, hi[3] = {0ull, 0ull, 0ull};
    for (int r = 0; r < 4; ++r) {
        const int64_t i = cbeg + (tile * 4 + r) * VX_THREADS + threadIdx.x;
        if (i < cend) {
#pragma unroll
            for (int a = 0; a < 3; ++a) {
                const unsigned long long k = f64_ordered(xyz[3 * i + a]);
                lo[a] = k < lo[a] ? k : lo[a];
                hi[a] = k > hi[a] ? k : hi[a];
            }
        }
    }
#pragma unroll
    for (int a = 0; a < 3; ++a) {
        lo[a] = wave_reduce_min(lo[a]);
        hi[a] = wave_reduce_max(hi[a]);
    }
    if (lane_id() == 0) {
#pragma unroll
        for (int a = 0; a < 3; ++a) { sm[wave_id()][a] = lo[a]; sm[wave_id()][3 + a] = hi[a]; }
    }
    __syncthreads();
    if (threadIdx.x < 6) {
        const int a = threadIdx.x;
        unsigned long long v = sm[0][a];
        for (int w = 1; w < VX_THREADS / 64; ++w) {
            const unsigned long long o = sm[w][a];
            v = (a < 3) ? (o < v ? o : v) : (o > v ? o : v);
        }
        if (a < 3) atomicMin(&mm[chunk * 6 + a], v);
        else       atomicMax(&mm[chunk * 6 + a], v);
    }
}

// one thread per chunk: grid origin, largest voxel index per axis, Open3D's range check
__global__ void vx_bounds_k(const unsigned long long* __restrict__ mm, int64_t nchunks,
                            double voxel, double* __restrict__ minb, int* __restrict__ gmeta) {
    const int64_t c = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= nchunks) return;
    double ext = 0.0;
#pragma unroll
    for (int a = 0; a < 3; ++a) {
        const double lo = f64_unordered(mm[c * 6 + a]);
        const double hi = f64_unordered(mm[c * 6 + 3 + a]);
        const double mb = lo - voxel * 0.5;           // voxel_min_bound
        const double xb = hi + voxel * 0.5;           // voxel_max_bound
        minb[c * 3 + a] = mb;
        const double e = xb - mb;
        ext = e > ext ? e : ext;
        const double q = floor((hi - mb) / voxel);
        int qi = (q >= 0.0 && q < 2147483647.0) ? (int)q : 2147483647;
        if (!(q == q)) qi = 2147483647;               // NaN coordinates
        atomicMax(&gmeta[a], qi);
    }
    if (voxel * 2147483647.0 < ext || !(ext == ext)) atomicMax(&gmeta[3], 1);
}

__global__ __launch_bounds__(VX_THREADS) void vx_keys_k(const double* __restrict__ xyz, int64_t n,
                                                        VoxelGrid g,
                                                        const double* __restrict__ minb,
                                                        uint64_t* __restrict__ keys,
                                                        uint32_t* __restrict__ vals) {
    const int64_t i = (int64_t)blockIdx.x * VX_THREADS + threadIdx.x;
    if (i >= n) return;
    const int64_t c = i / g.chunk_size;
    // ref_coord = (p - voxel_min_bound) / voxel_size ; index = floor(ref_coord)
    const uint64_t ix = (uint64_t)(int64_t)floor((xyz[3 * i + 0] - minb[3 * c + 0]) / g.voxel);
    const uint64_t iy = (uint64_t)(int64_t)floor((xyz[3 * i + 1] - minb[3 * c + 1]) / g.voxel);
    const uint64_t iz = (uint64_t)(int64_t)floor((xyz[3 * i + 2] - minb[3 * c + 2]) / g.voxel);
    keys[i] = ((((uint64_t)c << g.bx | ix) << g.by | iy) << g.bz) | iz;
    vals[i] = (uint32_t)i;
}

__global__ __launch_bounds__(VX_THREADS) void vx_heads_k(const uint64_t* __restrict__ keys,
                                                         int64_t n, uint32_t* __restrict__ flags) {
    const int64_t i = (int64_t)blockIdx.x * VX_THREADS + threadIdx.x;
    if (i >= n) return;
    flags[i] = (i == 0 || keys[i] != keys[i - 1]) ? 1u : 0u;
}

__global__ __launch_bounds__(VX_THREADS) void vx_starts_k(const uint64_t* __restrict__ keys,
                                                          const uint32_t* __restrict__ vid,
                                                          int64_t n, uint32_t* __restrict__ seg_start) {
    const int64_t i = (int64_t)blockIdx.x * VX_THREADS + threadIdx.x;
    if (i >= n) return;
    const bool head = (i == 0 || keys[i] != keys[i - 1]);
    if (head) seg_start[vid[i]] = (uint32_t)i;      // vid = exclusive scan of the head flags
    if (i == n - 1) seg_start[vid[i] + (head ? 1u : 0u)] = (uint32_t)n;   // == seg_start[m]
}

// one thread per voxel: sequential float64 sum in point order (AccumulatedPoint::AddPoint),
// mean = sum / count (GetAveragePoint)
__global__ __launch_bounds__(VX_THREADS) void vx_reduce_k(
    const double* __restrict__ xyz, const uint64_t* __restrict__ keys,
    const uint32_t* __restrict__ vals, const uint32_t* __restrict__ seg_start,
    const uint32_t* __restrict__ m_ptr, VoxelGrid g, int64_t nchunks,
    int32_t* __restrict__ out_idx, double* __restrict__ out_mean, int32_t* __restrict__ out_count,
    int64_t* __restrict__ out_chunk_offsets, int64_t* __restrict__ out_m) {
    const int64_t j = (int64_t)blockIdx.x * VX_THREADS + threadIdx.x;
    const int64_t m = *m_ptr;
    if (j == 0) {
        *out_m = m;
        if (out_chunk_offsets) out_chunk_offsets[nchunks] = m;
    }
    if (j >= m) return;
    const uint32_t s = seg_start[j], e = seg_start[j + 1];
    double ax = 0.0, ay = 0.0, az = 0.0;
    for (uint32_t i = s; i < e; ++i) {
        const int64_t p = vals[i];
        ax += xyz[3 * p + 0];
        ay += xyz[3 * p + 1];
        az += xyz[3 * p + 2];
    }
    const double cnt = (double)(e - s);
    out_mean[3 * j + 0] = ax / cnt;
    out_mean[3 * j + 1] = ay / cnt;
    out_mean[3 * j + 2] = az / cnt;
    out_count[j] = (int32_t)(e - s);
    const uint64_t k = keys[s];
    out_idx[3 * j + 2] = (int32_t)(k & ((1ull << g.bz) - 1));
    out_idx[3 * j + 1] = (int32_t)((k >> g.bz) & ((1ull << g.by) - 1));
    out_idx[3 * j + 0] = (int32_t)((k >> (g.bz + g.by)) & ((1ull << g.bx) - 1));
    if (out_chunk_offsets) {
        const int sh = g.bx + g.by + g.bz;
        const uint64_t c = sh < 64 ? (k >> sh) : 0;
        const uint64_t cp = (j == 0) ? ~0ull : (sh < 64 ? (keys[seg_start[j - 1]] >> sh) : 0);
        if (j == 0 || cp != c) out_chunk_offsets[c] = j;
    }
}

struct VoxelWs {
    unsigned long long* mm;
    double*   minb;
    int*      gmeta;
    uint64_t *k0, *k1;
    uint32_t *v0, *v1, *flags, *seg_start, *radix_ws, *scan_ws, *total;
};

static void voxel_plan(Arena& a, int64_t n, int64_t nchunks, VoxelWs& w) {
    const int64_t nn = n > 0 ? n : 1;
    w.mm = a.take<unsigned long long>(nchunks * 6);
    w.minb = a.take<double>(nchunks * 3);
    w.gmeta = a.take<int>(4);
    w.total = a.take<uint32_t>(4);
    w.k0 = a.take<uint64_t>(nn);
    w.k1 = a.take<uint64_t>(nn);
    w.v0 = a.take<uint32_t>(nn);
    w.v1 = a.take<uint32_t>(nn);
    w.flags = a.take<uint32_t>(nn + 8);
    w.seg_start = a.take<uint32_t>(nn + 8);
    w.radix_ws = a.take<uint32_t>(radix_ws_u32(nn));
    w.scan_ws = a.take<uint32_t>(scan_ws_u32(nn));
}

static int64_t voxel_nchunks(int64_t n, int64_t& chunk_size) {
    if (chunk_size <= 0 || chunk_size > n) chunk_size = n > 0 ? n : 1;
    return n > 0 ? ceil_div(n, chunk_size) : 1;
}

}  // namespace pch

using namespace pch;

extern "C" size_t pch_voxel_downsample_ws_bytes(int64_t n, int64_t chunk_size) {
    if (n < 0) return 0;
    const int64_t nchunks = voxel_nchunks(n, chunk_size);
    Arena a;
    VoxelWs w;
    voxel_plan(a, n, nchunks, w);
    return a.off;
}

extern "C" int pch_voxel_downsample_f64(const double* xyz, int64_t n, double voxel_size,
                                        int64_t chunk_size, int32_t* out_idx, double* out_mean,
                                        int32_t* out_count, int64_t* out_chunk_offsets,
                                        int64_t* out_m, void* ws, size_t ws_bytes, void* stream) {
    prof_begin_call();
    hipStream_t s = (hipStream_t)stream;
    PCH_REQUIRE(n >= 0 && n < (int64_t(1) << 31), "n out of range [0, 2^31)");
    PCH_REQUIRE(voxel_size > 0.0, "voxel_size must be > 0");
    PCH_REQUIRE(out_m != nullptr, "out_m is null");
    const int64_t nchunks = voxel_nchunks(n, chunk_size);
    if (n == 0) {
        PCH_HIP_TRY(hipMemsetAsync(out_m, 0, sizeof(int64_t), s));
        if (out_chunk_offsets) PCH_HIP_TRY(hipMemsetAsync(out_chunk_offsets, 0, 2 * sizeof(int64_t), s));
        return PCH_OK;
    }
    PCH_REQUIRE(xyz && out_idx && out_mean && out_count && ws, "null buffer");
    Arena a(ws, ws_bytes);
    VoxelWs w;
    voxel_plan(a, n, nchunks, w);
    if (a.overflow) { set_error("workspace too small: need %zu bytes", a.off); return PCH_ERR_WORKSPACE; }

    // per-chunk bounds
    PCH_HIP_TRY(hipMemsetAsync(w.mm, 0, sizeof(unsigned long long) * nchunks * 6, s));
    {   // min slots start at all-ones: strided memset via a tiny fill of 0xFF on the whole
        // table then zero the max slots would need a kernel; use two memsets per layout
        // [chunk][min3,max3] -> do it with hipMemset2DAsync (pitch 48 B, width 24 B)
        PCH_HIP_TRY(hipMemset2DAsync(w.mm, 48, 0xFF, 24, nchunks, s));
    }
    PCH_HIP_TRY(hipMemsetAsync(w.gmeta, 0, sizeof(int) * 4, s));
    const int64_t bpc = ceil_div(chunk_size, VX_THREADS * 4);
    PCH_LAUNCH("voxel_minmax", vx_minmax_k, dim3((unsigned)(bpc * nchunks)), dim3(VX_THREADS), 0, s,
               xyz, n, chunk_size, bpc, w.mm);
    PCH_LAUNCH("voxel_bounds", vx_bounds_k, dim3((unsigned)ceil_div(nchunks, 256)), dim3(256), 0, s,
               (const unsigned long long*)w.mm, nchunks, voxel_size, w.minb, w.gmeta);
    int gmeta[4];
    PCH_HIP_TRY(hipMemcpyAsync(gmeta, w.gmeta, sizeof(gmeta), hipMemcpyDeviceToHost, s));
    PCH_HIP_TRY(hipStreamSynchronize(s));
    if (gmeta[3] != 0) {   // Open3D: "[VoxelDownSample] voxel_size is too small."
        set_error("voxel_size is too small (or non-finite coordinates)");
        return PCH_ERR_RANGE;
    }
    VoxelGrid g;
    g.chunk_size = chunk_size;
    g.voxel = voxel_size;
    g.bx = bits_for((uint64_t)gmeta[0] + 1);
    g.by = bits_for((uint64_t)gmeta[1] + 1);
    g.bz = bits_for((uint64_t)gmeta[2] + 1);
    const int bc = bits_for((uint64_t)nchunks);
    const int nbits = g.bx + g.by + g.bz + bc;
    if (nbits > 64) {
        set_error("voxel grid needs %d key bits (> 64): reduce chunk extent or enlarge voxel", nbits);
        return PCH_ERR_RANGE;
    }
    const unsigned gb = (unsigned)ceil_div(n, VX_THREADS);
    PCH_LAUNCH("voxel_keys", vx_keys_k, dim3(gb), dim3(VX_THREADS), 0, s, xyz, n, g,
               (const double*)w.minb, w.k0, w.v0);
    PCH_TRY(radix_sort_pairs(w.k0, w.v0, w.k1, w.v1, n, nbits, w.radix_ws, s));
    const bool in1 = radix_sort_result_buffer(nbits) == 1;
    const uint64_t* ks = in1 ? w.k1 : w.k0;
    const uint32_t* vs = in1 ? w.v1 : w.v0;
    PCH_LAUNCH("voxel_heads", vx_heads_k, dim3(gb), dim3(VX_THREADS), 0, s, ks, n, w.flags);
    PCH_TRY(scan_exclusive_u32(w.flags, w.flags, n, w.scan_ws, w.total, s));
    PCH_LAUNCH("voxel_starts", vx_starts_k, dim3(gb), dim3(VX_THREADS), 0, s, ks,
               (const uint32_t*)w.flags, n, w.seg_start);
    PCH_LAUNCH("voxel_reduce", vx_reduce_k, dim3(gb), dim3(VX_THREADS), 0, s, xyz, ks, vs,
               (const uint32_t*)w.seg_start, (const uint32_t*)w.total, g, nchunks, out_idx, out_mean,
               out_count, out_chunk_offsets, out_m);
    return PCH_OK;
}

// ---- LAS integer <-> scaled float64 (laspy ScaledArrayView semantics) ----------------
namespace pch {
struct D3 { double v[3]; };

__global__ void las_scale_k(const int32_t* __restrict__ X, int64_t count, D3 sc, D3 of,
                            double* __restrict__ out) {
    const int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= count) return;
    const int a = (int)(e % 3);
    out[e] = (double)X[e] * sc.v[a] + of.v[a];          // separate mul, add (-ffp-contract=off)
}
// one thread per coordinate: three 4-byte loads at the head of each record (records need not be
// 4-byte aligned: formats 2, 7, 8 ... have odd lengths), assembled from bytes when misaligned
__global__ void las_records_k(const uint8_t* __restrict__ rec, int64_t count, int record_len,
                              int32_t* __restrict__ out) {
    const int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= count) return;
    const int64_t i = e / 3;
    const int a = (int)(e - 3 * i);
    const uint8_t* p = rec + i * record_len + 4 * a;
    uint32_t v;
    if ((reinterpret_cast<uintptr_t>(p) & 3) == 0) v = *reinterpret_cast<const uint32_t*>(p);
    else v = (uint32_t)p[0] | ((uint32_t)p[1] << 8) | ((uint32_t)p[2] << 16) | ((uint32_t)p[3] << 24);
    out[e] = (int32_t)v;
}
__global__ void las_unscale_k(const double* __restrict__ v, int64_t count, D3 sc, D3 of,
                              int32_t* __restrict__ out) {
    const int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= count) return;
    const int a = (int)(e % 3);
    out[e] = (int32_t)rint((v[e] - of.v[a]) / sc.v[a]); // np.round = half-to-even = rint
}
__global__ void cast_f64_f32_k(const double* __restrict__ in, int64_t count, float* __restrict__ out) {
    const int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (e < count) out[e] = (float)in[e];
}
}  // namespace pch

extern "C" int pch_las_records_xyz_i32(const uint8_t* records, int64_t n, int32_t record_len,
                                       int32_t* out_XYZ, void* stream) {
    prof_begin_call();
    PCH_REQUIRE(n >= 0 && record_len >= 12, "bad argument");
    if (n == 0) return PCH_OK;
    PCH_REQUIRE(records && out_XYZ, "null buffer");
    const int64_t count = 3 * n;
    PCH_LAUNCH("las_records", las_records_k, dim3((unsigned)ceil_div(count, 256)), dim3(256), 0,
               (hipStream_t)stream, records, count, (int)record_len, out_XYZ);
    return PCH_OK;
}

extern "C" int pch_las_scale_i32_f64(const int32_t* XYZ, int64_t n, const double* scale3_host,
                                     const double* offset3_host, double* out_xyz, void* stream) {
    prof_begin_call();
    PCH_REQUIRE(n >= 0 && scale3_host && offset3_host, "bad argument");
    if (n == 0) return PCH_OK;
    PCH_REQUIRE(XYZ && out_xyz, "null buffer");
    D3 sc, of;
    for (int a = 0; a < 3; ++a) { sc.v[a] = scale3_host[a]; of.v[a] = offset3_host[a]; }
    const int64_t count = 3 * n;
    PCH_LAUNCH("las_scale", las_scale_k, dim3((unsigned)ceil_div(count, 256)), dim3(256), 0,
               (hipStream_t)stream, XYZ, count, sc, of, out_xyz);
    return PCH_OK;
}

extern "C" int pch_las_unscale_f64_i32(const double* xyz, int64_t n, const double* scale3_host,
                                       const double* offset3_host, int32_t* out_XYZ, void* stream) {
    prof_begin_call();
    PCH_REQUIRE(n >= 0 && scale3_host && offset3_host, "bad argument");
    if (n == 0) return PCH_OK;
    PCH_REQUIRE(xyz && out_XYZ, "null buffer");
    D3 sc, of;
    for (int a = 0; a < 3; ++a) { sc.v[a] = scale3_host[a]; of.v[a] = offset3_host[a]; }
    const int64_t count = 3 * n;
    PCH_LAUNCH("las_unscale", las_unscale_k, dim3((unsigned)ceil_div(count, 256)), dim3(256), 0,
               (hipStream_t)stream, xyz, count, sc, of, out_XYZ);
    return PCH_OK;
}

extern "C" int pch_cast_f64_f32(const double* in, int64_t count, float* out, void* stream) {
    prof_begin_call();
    PCH_REQUIRE(count >= 0, "bad count");
    if (count == 0) return PCH_OK;
    PCH_REQUIRE(in && out, "null buffer");
    PCH_LAUNCH("cast_f64_f32", cast_f64_f32_k, dim3((unsigned)ceil_div(count, 256)), dim3(256), 0,
               (hipStream_t)stream, in, count, out);
    return PCH_OK;
}
