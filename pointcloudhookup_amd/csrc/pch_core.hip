// Library plumbing: version, thread-local error text, per-kernel hipEvent profiling.
#include "pch_common.h"
#include <chrono>

#include <stdarg.h>
#include <stdlib.h>
#include <sys/syscall.h>
#include <unistd.h>
#include <atomic>
#include <string>
#include <vector>

namespace pch {

static thread_local char g_err[512] = "";

void set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

static std::atomic<bool> g_exiting{false};
static std::atomic<bool> g_exit_hook{false};
static void note_exit() { g_exiting.store(true, std::memory_order_relaxed); }
static void arm_exit_hook() {                       // first creation of a per-thread HIP object registers it
    if (!g_exit_hook.exchange(true)) atexit(note_exit);
}
bool may_release_hip_objects() {
    if (g_exiting.load(std::memory_order_relaxed)) return false;
    return (long)syscall(SYS_gettid) != (long)getpid();     // the main thread's destructors run at process exit
}

struct ProfRec {
    const char* name;
    hipEvent_t  e0, e1;
    int         slot;          // device the events belong to
};
static thread_local bool                 g_prof_on = false;
static thread_local std::vector<ProfRec> g_recs;
// HIP objects belong to the device that was current when they were made: the per-thread caches
// below are kept per device (a thread may serve several GPUs) and are destroyed when a WORKER thread
// exits (may_release_hip_objects: never from the main thread's destructors, never during process exit).
struct EventPools {
    std::vector<hipEvent_t> pool[PCH_MAX_DEVICES];
    ~EventPools() {
        if (!may_release_hip_objects()) return;
        for (auto& p : pool)
            for (hipEvent_t e : p) (void)hipEventDestroy(e);
    }
};
static thread_local EventPools g_pools;

int current_device_slot() {
    int d = 0;
    if (hipGetDevice(&d) != hipSuccess || d < 0) d = 0;
    return d < PCH_MAX_DEVICES ? d : PCH_MAX_DEVICES - 1;
}

DeviceGuard::DeviceGuard(const void* device_ptr) : rc(PCH_OK), prev(-1), dev(-1) {
    if (hipGetDevice(&prev) != hipSuccess) { prev = -1; }
    if (!device_ptr) { dev = prev; return; }            // nothing to look up (argument checks report the null)
    hipPointerAttribute_t at;
    if (hipPointerGetAttributes(&at, device_ptr) != hipSuccess) {
        (void)hipGetLastError();
        set_error("not a device pointer (hipPointerGetAttributes failed)");
        rc = PCH_ERR_ARG;
        return;
    }
    if (at.type != hipMemoryTypeDevice && at.type != hipMemoryTypeManaged) {
        set_error("expected a device pointer, got host memory");
        rc = PCH_ERR_ARG;
        return;
    }
    dev = at.device;
    if (dev != prev && hipSetDevice(dev) != hipSuccess) {
        set_error("hipSetDevice(%d) failed", dev);
        rc = PCH_ERR_HIP;
    }
}
DeviceGuard::~DeviceGuard() {
    if (prev >= 0 && dev >= 0 && dev != prev) (void)hipSetDevice(prev);
}

static hipEvent_t take_event(int slot) {
    auto& pool = g_pools.pool[slot];
    if (!pool.empty()) {
        hipEvent_t e = pool.back();
        pool.pop_back();
        return e;
    }
    hipEvent_t e = nullptr;
    arm_exit_hook();
    (void)hipEventCreate(&e);
    return e;
}


// Records accumulate over pch_* calls until pch_get_profile() collects them (bench.py reads
// one whole step at once); a cap keeps a caller that never collects from growing forever.
constexpr size_t PROF_MAX_RECORDS = 16384;

static thread_local std::vector<std::string> g_filter;   // empty: every kernel

bool prof_enabled() { return g_prof_on && g_recs.size() < PROF_MAX_RECORDS; }

bool prof_wanted(const char* name) {
    if (g_filter.empty()) return true;
    for (auto& f : g_filter)
        if (f == name) return true;
    return false;
}

static void prof_clear() {
    for (auto& r : g_recs) {
        g_pools.pool[r.slot].push_back(r.e0);
        g_pools.pool[r.slot].push_back(r.e1);
    }
    g_recs.clear();
}

void prof_pre(const char* name, hipStream_t s) {
    ProfRec r;
    r.name = name;
    r.slot = current_device_slot();
    r.e0 = take_event(r.slot);
    r.e1 = take_event(r.slot);
    (void)hipEventRecord(r.e0, s);
    g_recs.push_back(r);
}

void prof_post(hipStream_t s) { (void)hipEventRecord(g_recs.back().e1, s); }

struct HostPeeks {
    HostPeek hp[PCH_MAX_DEVICES];
    HostPeeks() { for (auto& h : hp) h = {nullptr, nullptr, false}; }
    ~HostPeeks() {
        if (!may_release_hip_objects()) return;
        for (auto& h : hp)
            if (h.ok) { (void)hipEventDestroy(h.ev); (void)hipHostFree(h.pinned); }
    }
};
HostPeek& host_peek() {
    static thread_local HostPeeks all;
    HostPeek& hp = all.hp[current_device_slot()];
    if (!hp.ok) {
        arm_exit_hook();
        if (hipHostMalloc(&hp.pinned, 256, hipHostMallocDefault) == hipSuccess &&
            hipEventCreateWithFlags(&hp.ev, hipEventDisableTiming) == hipSuccess)
            hp.ok = true;
    }
    return hp;
}
int peek_enqueue(const void* dev, size_t bytes, hipStream_t s) {
    HostPeek& hp = host_peek();
    if (!hp.ok || bytes > 256) { set_error("pinned host buffer unavailable"); return PCH_ERR_HIP; }
    PCH_HIP_TRY(hipMemcpyAsync(hp.pinned, dev, bytes, hipMemcpyDeviceToHost, s));
    PCH_HIP_TRY(hipEventRecord(hp.ev, s));
    return PCH_OK;
}
int peek_wait(void* dst, size_t bytes) {
    HostPeek& hp = host_peek();
    // The GPU idles until the host has read these words and queued the next launches, so the host polls the event
    // instead of sleeping on it (a blocking wait is woken tens of microseconds late) - for at most ~2 ms, the length
    // of a whole step; behind that it blocks like anybody else (PCH_PEEK_BLOCK=1: block at once).
    static const bool block = getenv("PCH_PEEK_BLOCK") != nullptr;
    if (!block) {
        const auto t0 = std::chrono::steady_clock::now();
        for (int spin = 0;; ++spin) {
            const hipError_t q = hipEventQuery(hp.ev);
            if (q == hipSuccess) {
                // "not ready" is an answer, not a failure: it must not be what the next launch's hipGetLastError()
                // check finds (whether the runtime latches it has differed between releases)
                if (spin) (void)hipGetLastError();
                memcpy(dst, hp.pinned, bytes);
                return PCH_OK;
            }
            if (q != hipErrorNotReady) { PCH_HIP_TRY(q); }
            if ((spin & 63) == 63 &&
                std::chrono::steady_clock::now() - t0 > std::chrono::microseconds(2000)) break;
        }
    }
    if (!block) (void)hipGetLastError();                  // (see above: the polls before this answered "not ready")
    PCH_HIP_TRY(hipEventSynchronize(hp.ev));
    memcpy(dst, hp.pinned, bytes);
    return PCH_OK;
}

}  // namespace pch

extern "C" {

int pch_version(void) { return PCH_VERSION; }

const char* pch_last_error(void) { return pch::g_err; }

int pch_device_count(void) {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    return n;
}

void pch_set_profiling(int enable) {
    pch::g_prof_on = enable != 0;
    pch::prof_clear();
}

void pch_set_profiling_filter(const char* names) {
    pch::g_filter.clear();
    if (!names) return;
    std::string cur;
    for (const char* p = names;; ++p) {
        if (*p == ',' || *p == 0) {
            if (!cur.empty()) pch::g_filter.push_back(cur);
            cur.clear();
            if (*p == 0) break;
        } else {
            cur.push_back(*p);
        }
    }
}

int pch_get_profile(int cap, char names[][48], float* ms, int* launches) {
    using namespace pch;
    // aggregate by kernel name, keeping first-seen order
    std::vector<std::string> order;
    std::vector<float> total;
    std::vector<int> count;
    for (auto& r : g_recs) {
        if (hipEventSynchronize(r.e1) != hipSuccess) continue;
        float t = 0.f;
        if (hipEventElapsedTime(&t, r.e0, r.e1) != hipSuccess) continue;
        size_t k = 0;
        for (; k < order.size(); ++k)
            if (order[k] == r.name) break;
        if (k == order.size()) {
            order.push_back(r.name);
            total.push_back(0.f);
            count.push_back(0);
        }
        total[k] += t;
        count[k] += 1;
    }
    int n = 0;
    for (size_t k = 0; k < order.size() && n < cap; ++k, ++n) {
        strncpy(names[n], order[k].c_str(), 47);
        names[n][47] = 0;
        ms[n] = total[k];
        if (launches) launches[n] = count[k];
    }
    prof_clear();
    return n;
}

}  // extern "C"
