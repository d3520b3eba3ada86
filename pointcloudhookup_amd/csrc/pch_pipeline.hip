// Stages B + C + D0 behind one entry point: what utils/tower_extraction.py:62-125 does between
// "points are loaded" and "loop over the cluster labels".  Nothing new is computed here - the
// three stage entry points are called back to back on the caller's stream, sharing one
// workspace (a stage's scratch is dead once the next stage starts) and reading the two host-side
// sizes the next launch needs (points kept, clusters found) without a trip through the host
// language in between.
#include "pch_common.h"

#include <cstring>

namespace pch {

// device-side mirror of the filter's host results: scalars[8], aabb[6], pad[2], count (int64)
struct DevInfo {
    float   scalars[8];
    float   aabb[6];
    float   pad[2];
    int64_t count;
    int32_t nclusters;
    int32_t pad2;
};

static size_t max3(size_t a, size_t b, size_t c) { return a > b ? (a > c ? a : c) : (b > c ? b : c); }

}  // namespace pch

using namespace pch;

extern "C" size_t pch_tower_clusters_ws_bytes(int64_t n, int64_t nf_cap, int32_t k_cap) {
    if (n < 0 || nf_cap < 0 || k_cap < 0) return 0;
    Arena a;
    a.take<DevInfo>(1);
    return a.off + max3(pch_ground_filter_ws_bytes(n), pch_dbscan_ws_bytes(nf_cap),
                        pch_segment_by_label_ws_bytes(nf_cap, k_cap));
}

extern "C" int pch_tower_clusters_f32(const float* raw, int64_t n, double pct, float offset,
                                      float fallback_offset, int64_t min_keep, double eps,
                                      int32_t min_samples, int64_t chunk_size, float* out_points,
                                      int32_t* out_index, int32_t* out_labels, int32_t* out_perm,
                                      int64_t* out_offsets, float* out_stats, int64_t nf_cap,
                                      int32_t k_cap, PchTowerClusters* info_host, void* ws,
                                      size_t ws_bytes, void* stream) {
    hipStream_t s = (hipStream_t)stream;
    PCH_DEVICE_GUARD(raw);
    PCH_REQUIRE(info_host != nullptr, "info_host is null");
    memset(info_host, 0, sizeof(*info_host));
    PCH_REQUIRE(ws != nullptr && nf_cap >= 0 && k_cap >= 0, "bad argument");
    Arena a(ws, ws_bytes);
    DevInfo* dev = a.take<DevInfo>(1);
    if (a.overflow || ws_bytes < pch_tower_clusters_ws_bytes(n, nf_cap, k_cap)) {
        set_error("workspace too small: need %zu bytes", pch_tower_clusters_ws_bytes(n, nf_cap, k_cap));
        return PCH_ERR_WORKSPACE;
    }
    void* sub = static_cast<char*>(ws) + a.off;
    const size_t sub_bytes = ws_bytes - a.off;

    // ---- stage B
    // the filter queues the copy of its results for the host directly behind its first sweep; the launches of the
    // fallback threshold behind it return at once unless the fallback applies - the host is preparing stage C by then
    static_assert(sizeof(DevInfo) <= 256, "fits the pinned peek buffer");
    const GfEarly early = {dev, sizeof(DevInfo)};
    PCH_TRY(ground_filter_run(raw, n, pct, offset, fallback_offset, min_keep, out_points, out_index,
                              dev->scalars, &dev->count, dev->aabb, sub, sub_bytes, s, &early));
    DevInfo h;
    PCH_TRY(peek_wait(&h, sizeof(DevInfo)));
    if (h.scalars[5] != 0.0f && h.count >= 0) {          // the fallback threshold applies: the final values lie behind it
        PCH_TRY(peek_enqueue(dev, sizeof(DevInfo), s));
        PCH_TRY(peek_wait(&h, sizeof(DevInfo)));
    }
    memcpy(info_host->centroid, h.scalars, 3 * sizeof(float));
    info_host->base = h.scalars[3];
    info_host->threshold = h.scalars[4];
    info_host->used_fallback = h.scalars[5] != 0.0f;
    {
        uint32_t c0;
        memcpy(&c0, &h.scalars[6], sizeof(c0));          // integer count in the float slot's bits
        info_host->count_at_offset = (int64_t)c0;
    }
    memcpy(info_host->aabb, h.aabb, sizeof(h.aabb));
    info_host->count = h.count;
    const int64_t nf = h.count;
    if (nf < 0) {                                        // gf_compact's look-back gave up (pch_lookback.h)
        set_error("stage B: a device-side look-back wait ran out of its budget; outputs are undefined");
        return PCH_ERR_TIMEOUT;
    }
    if (nf > nf_cap) {
        set_error("the filter kept %lld points, more than nf_cap = %lld", (long long)nf, (long long)nf_cap);
        return PCH_ERR_WORKSPACE;
    }
    if (nf == 0) {                                       // nothing above the ground: no clusters
        if (out_offsets) PCH_HIP_TRY(hipMemsetAsync(out_offsets, 0, sizeof(int64_t), s));
        return PCH_OK;
    }
    // ---- stage C (the filter's scratch is dead; its outputs are the caller's buffers)
    PCH_REQUIRE(out_labels != nullptr, "out_labels is null");
    int32_t k = 0;                                       // read back while the labels are still being written
    // the cluster boxes of stage D0 are gathered where the labels are made (no pass of their own): out_stats holds
    // the encoded table until pch_segment's offsets kernel decodes it in place
    DbBoxOut box = {reinterpret_cast<uint32_t*>(out_stats), (out_perm && out_offsets && out_stats) ? k_cap : 0, false};
    PCH_TRY(dbscan_run(out_points, nf, eps, min_samples, chunk_size, h.aabb, out_labels, nullptr,
                       &dev->nclusters, sub, sub_bytes, s, &k, &box));
    info_host->nclusters = k;
    // ---- stage D0
    if (out_perm && out_offsets) {
        if (k > k_cap) {
            set_error("%d clusters found, more than k_cap = %d (labels are valid; group them with "
                      "pch_segment_by_label)", (int)k, (int)k_cap);
            return PCH_ERR_RANGE;
        }
        PCH_TRY(segment_run(out_labels, out_points, nf, k, out_perm, out_offsets, out_stats, sub, sub_bytes, s,
                            box.done));
    }
    return PCH_OK;
}
