// Native LAS 1.0-1.4 point I/O for the hot path: header parse, X/Y/Z int32 of every point record
// straight into a device buffer (pread by several threads -> ring of pinned buffers -> H2D copy overlapped with the
// decode kernel), and the writer the drop-ins use for their output clouds (records laid out on the device, pwrite by
// several threads).
// Replaces: laspy.read / laspy.open(...).read() (ui/import_PC.py:28, utils/tower_extraction.py:60-61,
//           ui/extract.py:114-115) and LasData.write (ui/import_PC.py:35-42,64-65,
//           utils/tower_extraction.py:243-257).  Only what those call sites touch is implemented: the
//           public header block, scales / offsets, X, Y, Z.  VLRs, extra bytes and any padding in front of
//           the point data are honoured through header_size / offset_to_points / record_length.
#include "pch_common.h"

#include <errno.h>
#include <fcntl.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>

#include <stdlib.h>

#include <atomic>
#include <string>
#include <thread>
#include <vector>

namespace pch {

// point data record length per point format id (LAS 1.4 R15, table 7 ff.)
static int las_min_record_len(int fmt) {
    static const int len[11] = {20, 28, 26, 34, 57, 63, 30, 36, 38, 59, 67};
    return (fmt >= 0 && fmt <= 10) ? len[fmt] : -1;
}

template <typename T>
static T rd(const unsigned char* p) {
    T v;
    memcpy(&v, p, sizeof(T));          // LAS is little-endian, and so is every host we build for
    return v;
}

struct LasFile {
    int fd = -1;
    const unsigned char* map = nullptr;
    size_t size = 0;
    ~LasFile() {
        if (map) munmap(const_cast<unsigned char*>(map), size);
        if (fd >= 0) close(fd);
    }
    int open_ro(const char* path) {
        fd = ::open(path, O_RDONLY);
        if (fd < 0) { set_error("%s: %s", path, strerror(errno)); return PCH_ERR_ARG; }
        struct stat st;
        if (fstat(fd, &st) != 0) { set_error("%s: fstat failed", path); return PCH_ERR_ARG; }
        size = (size_t)st.st_size;
        if (size < 227) { set_error("%s: not a LAS file (shorter than a header)", path); return PCH_ERR_ARG; }
        void* m = mmap(nullptr, size, PROT_READ, MAP_PRIVATE, fd, 0);
        if (m == MAP_FAILED) { set_error("%s: mmap failed: %s", path, strerror(errno)); return PCH_ERR_ARG; }
        map = static_cast<const unsigned char*>(m);
        (void)madvise(m, size, MADV_SEQUENTIAL);
        return PCH_OK;
    }
};

static int las_parse(const char* path, const unsigned char* h, size_t size, PchLasHeader* out) {
    if (memcmp(h, "LASF", 4) != 0) { set_error("%s: not a LAS file (signature)", path); return PCH_ERR_ARG; }
    memset(out, 0, sizeof(*out));
    out->version_major = h[24];
    out->version_minor = h[25];
    out->header_size = rd<uint16_t>(h + 94);
    out->offset_to_points = rd<uint32_t>(h + 96);
    out->num_vlrs = rd<uint32_t>(h + 100);
    const uint8_t fmt_raw = h[104];
    if (fmt_raw & 0xC0) { set_error("%s: compressed (LAZ) point records are not supported", path); return PCH_ERR_ARG; }
    out->point_format = fmt_raw & 0x3F;
    out->record_length = rd<uint16_t>(h + 105);
    uint64_t count = rd<uint32_t>(h + 107);                          // legacy 32-bit count
    for (int a = 0; a < 3; ++a) {
        out->scales[a] = rd<double>(h + 131 + 8 * a);
        out->offsets[a] = rd<double>(h + 155 + 8 * a);
        out->maxs[a] = rd<double>(h + 179 + 16 * a);
        out->mins[a] = rd<double>(h + 187 + 16 * a);
    }
    if ((out->version_major > 1 || out->version_minor >= 4) && out->header_size >= 375 && size >= 375) {
        const uint64_t c64 = rd<uint64_t>(h + 247);                  // LAS 1.4: 64-bit count wins when set
        if (c64) count = c64;
    }
    out->point_count = count;
    const int need = las_min_record_len(out->point_format);
    if (need < 0) { set_error("%s: unsupported point format %d", path, (int)out->point_format); return PCH_ERR_ARG; }
    if ((int)out->record_length < need) {
        set_error("%s: record length %d is shorter than point format %d needs (%d)", path,
                  (int)out->record_length, (int)out->point_format, need);
        return PCH_ERR_ARG;
    }
    if (out->offset_to_points < out->header_size) {
        set_error("%s: offset to point data (%u) lies inside the %d-byte header", path, out->offset_to_points,
                  (int)out->header_size);
        return PCH_ERR_ARG;
    }
    // no multiplication: a forged 64-bit count must not wrap the product past the check
    if ((uint64_t)out->offset_to_points > (uint64_t)size ||
        count > ((uint64_t)size - (uint64_t)out->offset_to_points) / (uint64_t)out->record_length) {
        set_error("%s: truncated (%llu records of %d bytes from offset %u do not fit %zu bytes)", path,
                  (unsigned long long)count, (int)out->record_length, out->offset_to_points, size);
        return PCH_ERR_ARG;
    }
    return PCH_OK;
}

// X, Y, Z of `count` records: three 4-byte loads at the head of each record (records need not be
// 4-byte aligned: formats 2, 4, 5, 9, 10 have odd lengths), assembled from bytes when misaligned
__global__ void las_decode_k(const uint8_t* __restrict__ rec, int64_t count3, int record_len,
                             int32_t* __restrict__ out) {
    const int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= count3) return;
    const int64_t i = e / 3;
    const int a = (int)(e - 3 * i);
    const uint8_t* p = rec + i * record_len + 4 * a;
    uint32_t v;
    if ((reinterpret_cast<uintptr_t>(p) & 3) == 0) v = *reinterpret_cast<const uint32_t*>(p);
    else v = (uint32_t)p[0] | ((uint32_t)p[1] << 8) | ((uint32_t)p[2] << 16) | ((uint32_t)p[3] << 24);
    out[e] = (int32_t)v;
}

constexpr size_t LAS_HOP = size_t(64) << 20;        // bytes of point records per pinned buffer
constexpr int    LAS_RING = 3;                      // pinned buffers in flight (file -> pinned -> device)

// Pinned staging buffers, kept per thread and device across calls (pinning 192 MiB costs tens of ms - as much
// as reading a 10 M-point file); never released from the main thread's destructors (may_release_hip_objects).
struct PinnedRing {
    void* buf[LAS_RING] = {nullptr, nullptr, nullptr};
    hipEvent_t ev[LAS_RING] = {nullptr, nullptr, nullptr};
    bool ok = false;
    int init() {
        if (ok) return PCH_OK;
        for (int k = 0; k < LAS_RING; ++k) {
            PCH_HIP_TRY(hipHostMalloc(&buf[k], LAS_HOP, hipHostMallocDefault));
            PCH_HIP_TRY(hipEventCreateWithFlags(&ev[k], hipEventDisableTiming));
        }
        ok = true;
        return PCH_OK;
    }
    ~PinnedRing() {
        if (!may_release_hip_objects()) return;
        for (int k = 0; k < LAS_RING; ++k) {
            if (ev[k]) (void)hipEventDestroy(ev[k]);
            if (buf[k]) (void)hipHostFree(buf[k]);
        }
    }
};
struct PinnedRings { PinnedRing r[PCH_MAX_DEVICES]; };
static PinnedRing& pinned_ring() {
    static thread_local PinnedRings all;
    return all.r[current_device_slot()];
}

// host threads that move file bytes (page cache <-> pinned memory): one thread tops out near 2-4 GB/s of
// pread / pwrite, a PCIe Gen5 link takes ~50 GB/s.  PCH_LAS_THREADS overrides (1 = the calling thread only).
static int las_io_threads() {
    static int n = -1;
    if (n < 0) {
        const char* e = getenv("PCH_LAS_THREADS");
        int v = e ? atoi(e) : 0;
        if (v <= 0) {
            const unsigned hw = std::thread::hardware_concurrency();
            v = hw >= 16 ? 8 : (hw >= 4 ? (int)hw / 2 : 1);
        }
        n = v > 32 ? 32 : v;
    }
    return n;
}

// pread / pwrite of [off, off+bytes) split into slices, one per thread; returns false on a short transfer
template <bool WRITE>
static bool las_file_io(int fd, unsigned char* mem, size_t bytes, off_t off) {
    auto one = [&](size_t a, size_t b) -> bool {
        while (a < b) {
            const ssize_t r = WRITE ? ::pwrite(fd, mem + a, b - a, off + (off_t)a) : ::pread(fd, mem + a, b - a, off + (off_t)a);
            if (r <= 0) { if (r < 0 && errno == EINTR) continue; return false; }
            a += (size_t)r;
        }
        return true;
    };
    int nt = las_io_threads();
    if (bytes < (size_t(4) << 20)) nt = 1;
    if (nt <= 1) return one(0, bytes);
    std::atomic<bool> good{true};
    std::vector<std::thread> th;
    const size_t slice = ((bytes + nt - 1) / nt + 4095) & ~size_t(4095);
    for (int t = 1; t < nt; ++t) {
        const size_t a = (size_t)t * slice, b = a + slice < bytes ? a + slice : bytes;
        if (a >= bytes) break;
        th.emplace_back([&, a, b] { if (!one(a, b)) good.store(false); });
    }
    if (!one(0, slice < bytes ? slice : bytes)) good.store(false);
    for (auto& t : th) t.join();
    return good.load();
}

// records of `count` points laid out for the file: X,Y,Z at the head of every record_len-byte record, every other
// byte zero (what laspy writes for a LasData whose only assigned dimensions are x, y, z); one thread per 4 bytes
__global__ void las_encode_k(const int32_t* __restrict__ XYZ, int64_t count, int record_len,
                             uint8_t* __restrict__ out) {
    const int64_t total = count * record_len;
    const int64_t b = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) * 4;
    if (b >= total) return;
    uint32_t w = 0;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const int64_t at = b + k;
        if (at >= total) break;
        const int64_t i = at / record_len;
        const int o = (int)(at - i * record_len);
        if (o < 12) w |= (uint32_t)((((uint32_t)XYZ[3 * i + (o >> 2)]) >> (8 * (o & 3))) & 0xFFu) << (8 * k);
    }
    if (b + 4 <= total) *reinterpret_cast<uint32_t*>(out + b) = w;
    else for (int k = 0; b + k < total; ++k) out[b + k] = (uint8_t)(w >> (8 * k));
}

// bounding box of the integer coordinates: [0..2] minima, [3..5] maxima (int32, pre-set to +max / -max)
__global__ void las_minmax_k(const int32_t* __restrict__ XYZ, int64_t n, int32_t* __restrict__ box) {
    int32_t lo[3] = {INT32_MAX, INT32_MAX, INT32_MAX}, hi[3] = {INT32_MIN, INT32_MIN, INT32_MIN};
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
#pragma unroll
        for (int a = 0; a < 3; ++a) {
            const int32_t v = XYZ[3 * i + a];
            lo[a] = v < lo[a] ? v : lo[a];
            hi[a] = v > hi[a] ? v : hi[a];
        }
    }
#pragma unroll
    for (int a = 0; a < 3; ++a) {
        lo[a] = wave_reduce_min(lo[a]);
        hi[a] = wave_reduce_max(hi[a]);
    }
    if (lane_id() == 0) {
#pragma unroll
        for (int a = 0; a < 3; ++a) { atomicMin(&box[a], lo[a]); atomicMax(&box[3 + a], hi[a]); }
    }
}

}  // namespace pch

using namespace pch;

extern "C" int pch_las_read_header(const char* path, PchLasHeader* out) {
    PCH_REQUIRE(path && out, "null argument");
    LasFile f;
    PCH_TRY(f.open_ro(path));
    return las_parse(path, f.map, f.size, out);
}

extern "C" size_t pch_las_read_ws_bytes(void) { return 2 * LAS_HOP + 512; }

extern "C" int pch_las_read_xyz_i32(const char* path, int64_t first, int64_t count, int32_t* out_XYZ,
                                    void* ws, size_t ws_bytes, void* stream) {
    PCH_DEVICE_GUARD(out_XYZ ? (const void*)out_XYZ : (const void*)ws);
    hipStream_t s = (hipStream_t)stream;
    PCH_REQUIRE(path && first >= 0 && count >= 0, "bad argument");
    LasFile f;
    PCH_TRY(f.open_ro(path));
    PchLasHeader h;
    PCH_TRY(las_parse(path, f.map, f.size, &h));
    PCH_REQUIRE((uint64_t)first + (uint64_t)count <= h.point_count, "record range beyond the file's point count");
    if (count == 0) return PCH_OK;
    PCH_REQUIRE(out_XYZ && ws && ws_bytes >= pch_las_read_ws_bytes(), "null buffer or workspace too small");
    const size_t rl = h.record_length;
    const int64_t per_hop = (int64_t)(LAS_HOP / rl);
    uint8_t* dev[2] = {static_cast<uint8_t*>(ws), static_cast<uint8_t*>(ws) + LAS_HOP + 256};
    PinnedRing& pin = pinned_ring();
    PCH_TRY(pin.init());
    // file -> pinned buffer (pread, several threads) -> device buffer (H2D) -> decode kernel.  Three pinned buffers
    // and two device buffers: while the threads fill hop k+1, hop k crosses PCIe and hop k-1 is decoded.
    // (the two device buffers need no events: copy and decode of a hop are ordered on the one stream)
    const off_t base = (off_t)h.offset_to_points + (off_t)first * (off_t)rl;
    int64_t hop = 0;
    for (int64_t done = 0; done < count; done += per_hop, ++hop) {
        const int64_t m = (count - done) < per_hop ? (count - done) : per_hop;
        const size_t bytes = (size_t)m * rl;
        const int p = (int)(hop % LAS_RING), d = (int)(hop & 1);
        PCH_HIP_TRY(hipEventSynchronize(pin.ev[p]));        // the copy that last read this pinned buffer is done
        if (!las_file_io<false>(f.fd, static_cast<unsigned char*>(pin.buf[p]), bytes, base + (off_t)done * (off_t)rl)) {
            set_error("%s: short read of the point records: %s", path, strerror(errno));
            (void)hipStreamSynchronize(s);
            return PCH_ERR_ARG;
        }
        PCH_HIP_TRY(hipMemcpyAsync(dev[d], pin.buf[p], bytes, hipMemcpyHostToDevice, s));
        PCH_HIP_TRY(hipEventRecord(pin.ev[p], s));
        PCH_LAUNCH("las_decode", las_decode_k, dim3((unsigned)ceil_div(3 * m, 256)), dim3(256), 0, s,
                   (const uint8_t*)dev[d], 3 * m, (int)rl, out_XYZ + 3 * done);
    }
    PCH_HIP_TRY(hipStreamSynchronize(s));               // the mapping goes away on return
    return PCH_OK;
}

extern "C" int pch_las_write_xyz_i32(const char* path, const PchLasHeader* hdr, const int32_t* XYZ, int64_t n,
                                     void* stream) {
    PCH_DEVICE_GUARD(XYZ);
    hipStream_t s = (hipStream_t)stream;
    PCH_REQUIRE(path && hdr && n >= 0 && (n == 0 || XYZ), "bad argument");
    const int fmt = hdr->point_format;
    const int rl = las_min_record_len(fmt);
    PCH_REQUIRE(rl > 0, "unsupported point format");
    const int vmaj = hdr->version_major ? hdr->version_major : 1, vmin = hdr->version_major ? hdr->version_minor : 2;
    const int hs = (vmaj == 1 && vmin >= 4) ? 375 : (vmaj == 1 && vmin == 3) ? 235 : 227;
    // points first (their bounding box goes into the header): records with X,Y,Z set, every other field zero -
    // what laspy writes for a LasData whose only assigned dimensions are x, y, z
    int fd = ::open(path, O_WRONLY | O_CREAT | O_TRUNC, 0644);
    if (fd < 0) { set_error("%s: %s", path, strerror(errno)); return PCH_ERR_ARG; }
    struct Closer { int fd; ~Closer() { if (fd >= 0) close(fd); } } closer{fd};
    // the final size is known: blocks reserved up front, so the writer threads below never extend the file
    // (buffered pwrite into one file: 10.2 -> 12.1 GB/s on the bench host; failure is harmless, e.g. on tmpfs)
    if (n > 0) (void)posix_fallocate(fd, 0, (off_t)hs + (off_t)n * (off_t)rl);
    std::vector<unsigned char> head((size_t)hs, 0);
    if (::write(fd, head.data(), head.size()) != (ssize_t)head.size()) { set_error("%s: write failed", path); return PCH_ERR_ARG; }
    int32_t lo[3] = {INT32_MAX, INT32_MAX, INT32_MAX}, hi[3] = {INT32_MIN, INT32_MIN, INT32_MIN};
    if (n > 0) {
        // The DEVICE lays the records out (las_encode_k) and reduces the bounding box (las_minmax_k); the host only
        // moves finished bytes: D2H into pinned buffers, pwrite from several threads.  Hop k+1 is encoded and copied
        // while hop k is written.
        const int64_t per_hop = (int64_t)(LAS_HOP / rl);
        PinnedRing& pin = pinned_ring();
        PCH_TRY(pin.init());
        uint8_t* enc[2] = {nullptr, nullptr};
        int32_t* box = nullptr;
        struct DevGuard { uint8_t** e; int32_t** b; ~DevGuard() { for (int k = 0; k < 2; ++k) if (e[k]) (void)hipFree(e[k]); if (*b) (void)hipFree(*b); } } dg{enc, &box};
        for (int k = 0; k < 2; ++k) PCH_HIP_TRY(hipMalloc(reinterpret_cast<void**>(&enc[k]), (size_t)per_hop * rl + 16));
        PCH_HIP_TRY(hipMalloc(reinterpret_cast<void**>(&box), 6 * sizeof(int32_t)));
        const int32_t init[6] = {INT32_MAX, INT32_MAX, INT32_MAX, INT32_MIN, INT32_MIN, INT32_MIN};
        PCH_HIP_TRY(hipStreamSynchronize(s));                // XYZ is final; the blocking copy below is then ordered
        PCH_HIP_TRY(hipMemcpy(box, init, sizeof(init), hipMemcpyHostToDevice));
        {
            int64_t gb = ceil_div(n, 256 * 8);
            if (gb > 4096) gb = 4096;
            PCH_LAUNCH("las_minmax", las_minmax_k, dim3((unsigned)gb), dim3(256), 0, s, XYZ, n, box);
        }
        int32_t host_box[6];
        PCH_TRY(peek_enqueue(box, sizeof(host_box), s));     // lands in the pinned peek buffer, fetched after the hops
        const int64_t hops = ceil_div(n, per_hop);
        auto start = [&](int64_t hop) -> int {
            const int64_t at = hop * per_hop, m = (n - at) < per_hop ? (n - at) : per_hop;
            const int p = (int)(hop % 2);
            PCH_LAUNCH("las_encode", las_encode_k, dim3((unsigned)ceil_div(ceil_div(m * rl, 4), 256)), dim3(256), 0, s,
                       XYZ + 3 * at, m, rl, enc[p]);
            PCH_HIP_TRY(hipMemcpyAsync(pin.buf[p], enc[p], (size_t)m * rl, hipMemcpyDeviceToHost, s));
            PCH_HIP_TRY(hipEventRecord(pin.ev[p], s));
            return PCH_OK;
        };
        PCH_TRY(start(0));
        for (int64_t hop = 0; hop < hops; ++hop) {
            if (hop + 1 < hops) PCH_TRY(start(hop + 1));       // buffer (hop+1)%2 was written out in the round before
            PCH_HIP_TRY(hipEventSynchronize(pin.ev[hop % 2]));
            const int64_t at = hop * per_hop, m = (n - at) < per_hop ? (n - at) : per_hop;
            if (!las_file_io<true>(fd, static_cast<unsigned char*>(pin.buf[hop % 2]), (size_t)m * rl,
                                   (off_t)hs + (off_t)at * (off_t)rl)) {
                set_error("%s: write failed: %s", path, strerror(errno));
                (void)hipStreamSynchronize(s);
                return PCH_ERR_ARG;
            }
        }
        PCH_HIP_TRY(hipStreamSynchronize(s));
        PCH_TRY(peek_wait(host_box, sizeof(host_box)));
        for (int a = 0; a < 3; ++a) { lo[a] = host_box[a]; hi[a] = host_box[3 + a]; }
    }
    // ---- public header block
    unsigned char* h = head.data();
    memcpy(h, "LASF", 4);
    h[24] = (unsigned char)vmaj;
    h[25] = (unsigned char)vmin;
    memcpy(h + 26, "pointcloudhookup_amd", 20);          // system identifier (32 bytes, zero padded)
    memcpy(h + 58, "pch-hip", 7);                        // generating software
    const uint16_t doy = 1, year = 2025;
    memcpy(h + 90, &doy, 2);
    memcpy(h + 92, &year, 2);
    const uint16_t hs16 = (uint16_t)hs;
    const uint32_t otp = (uint32_t)hs, nvlr = 0;
    memcpy(h + 94, &hs16, 2);
    memcpy(h + 96, &otp, 4);
    memcpy(h + 100, &nvlr, 4);
    h[104] = (unsigned char)fmt;
    const uint16_t rl16 = (uint16_t)rl;
    memcpy(h + 105, &rl16, 2);
    const uint32_t legacy = (n < (int64_t(1) << 32) && fmt < 6) ? (uint32_t)n : 0u;
    memcpy(h + 107, &legacy, 4);
    memcpy(h + 111, &legacy, 4);                         // points by return [0]; the other four stay 0
    for (int a = 0; a < 3; ++a) {
        const double sc = hdr->scales[a], of = hdr->offsets[a];
        const double mn = n ? (double)lo[a] * sc + of : 0.0, mx = n ? (double)hi[a] * sc + of : 0.0;
        memcpy(h + 131 + 8 * a, &sc, 8);
        memcpy(h + 155 + 8 * a, &of, 8);
        memcpy(h + 179 + 16 * a, &mx, 8);
        memcpy(h + 187 + 16 * a, &mn, 8);
    }
    if (hs >= 375) {
        const uint64_t n64 = (uint64_t)n;
        memcpy(h + 247, &n64, 8);
        memcpy(h + 255, &n64, 8);
    }
    if (lseek(fd, 0, SEEK_SET) != 0 || ::write(fd, head.data(), head.size()) != (ssize_t)head.size()) {
        set_error("%s: header write failed", path);
        return PCH_ERR_ARG;
    }
    return PCH_OK;
}
