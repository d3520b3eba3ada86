// Native LAS 1.0-1.4 point I/O for the hot path: header parse, X/Y/Z int32 of every point record
// straight into a device buffer (mmap -> pinned double buffer -> H2D copy overlapped with the decode
// kernel), and the writer the drop-ins use for their output clouds.
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

#include <string>
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

constexpr size_t LAS_HOP = size_t(32) << 20;        // bytes of point records per pinned buffer

struct Pinned2 {
    void* buf[2] = {nullptr, nullptr};
    hipEvent_t ev[2] = {nullptr, nullptr};
    bool ok = false;
    int init(size_t bytes) {
        for (int k = 0; k < 2; ++k) {
            PCH_HIP_TRY(hipHostMalloc(&buf[k], bytes, hipHostMallocDefault));
            PCH_HIP_TRY(hipEventCreateWithFlags(&ev[k], hipEventDisableTiming));
        }
        ok = true;
        return PCH_OK;
    }
    ~Pinned2() {
        for (int k = 0; k < 2; ++k) {
            if (ev[k]) (void)hipEventDestroy(ev[k]);
            if (buf[k]) (void)hipHostFree(buf[k]);
        }
    }
};

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
    Pinned2 pin;
    PCH_TRY(pin.init(LAS_HOP));
    const unsigned char* recs = f.map + h.offset_to_points + (size_t)first * rl;
    int k = 0;
    for (int64_t done = 0; done < count; done += per_hop, k ^= 1) {
        const int64_t m = (count - done) < per_hop ? (count - done) : per_hop;
        const size_t bytes = (size_t)m * rl;
        // the pinned buffer is free again once the copy that last read it has finished; meanwhile the
        // GPU is busy with the previous hop's copy + decode
        PCH_HIP_TRY(hipEventSynchronize(pin.ev[k]));
        memcpy(pin.buf[k], recs + (size_t)done * rl, bytes);           // page cache / disk -> pinned
        PCH_HIP_TRY(hipMemcpyAsync(dev[k], pin.buf[k], bytes, hipMemcpyHostToDevice, s));
        PCH_HIP_TRY(hipEventRecord(pin.ev[k], s));
        PCH_LAUNCH("las_decode", las_decode_k, dim3((unsigned)ceil_div(3 * m, 256)), dim3(256), 0, s,
                   (const uint8_t*)dev[k], 3 * m, (int)rl, out_XYZ + 3 * done);
    }
    PCH_HIP_TRY(hipStreamSynchronize(s));               // the pinned buffers and the mapping go away on return
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
    std::vector<unsigned char> head((size_t)hs, 0);
    if (::write(fd, head.data(), head.size()) != (ssize_t)head.size()) { set_error("%s: write failed", path); return PCH_ERR_ARG; }
    int32_t lo[3] = {INT32_MAX, INT32_MAX, INT32_MAX}, hi[3] = {INT32_MIN, INT32_MIN, INT32_MIN};
    if (n > 0) {
        const int64_t per_hop = (int64_t)(LAS_HOP / 12);
        Pinned2 pin;
        PCH_TRY(pin.init((size_t)per_hop * 12));
        std::vector<unsigned char> out((size_t)per_hop * rl);
        // D2H of hop k+1 runs while hop k is laid out and written
        const int64_t hops = ceil_div(n, per_hop);
        auto start = [&](int64_t hop) -> int {
            const int64_t at = hop * per_hop, m = (n - at) < per_hop ? (n - at) : per_hop;
            PCH_HIP_TRY(hipMemcpyAsync(pin.buf[hop & 1], XYZ + 3 * at, (size_t)m * 12, hipMemcpyDeviceToHost, s));
            PCH_HIP_TRY(hipEventRecord(pin.ev[hop & 1], s));
            return PCH_OK;
        };
        PCH_TRY(start(0));
        for (int64_t hop = 0; hop < hops; ++hop) {
            if (hop + 1 < hops) PCH_TRY(start(hop + 1));
            PCH_HIP_TRY(hipEventSynchronize(pin.ev[hop & 1]));
            const int64_t at = hop * per_hop, m = (n - at) < per_hop ? (n - at) : per_hop;
            const int32_t* src = static_cast<const int32_t*>(pin.buf[hop & 1]);
            memset(out.data(), 0, (size_t)m * rl);
            for (int64_t i = 0; i < m; ++i) {
                memcpy(out.data() + (size_t)i * rl, src + 3 * i, 12);
                for (int a = 0; a < 3; ++a) {
                    const int32_t v = src[3 * i + a];
                    lo[a] = v < lo[a] ? v : lo[a];
                    hi[a] = v > hi[a] ? v : hi[a];
                }
            }
            size_t off = 0;
            const size_t total = (size_t)m * rl;
            while (off < total) {
                const ssize_t w = ::write(fd, out.data() + off, total - off);
                if (w <= 0) { set_error("%s: write failed: %s", path, strerror(errno)); return PCH_ERR_ARG; }
                off += (size_t)w;
            }
        }
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
