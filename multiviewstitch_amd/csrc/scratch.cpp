// Scratch pool of the host-driven entries (Alignment, SRT, depth, files' device halves): a call of mvs_align on a 2 M-vertex scan
// makes ~40 device allocations, and hipFree alone was 4.3 of its 10 ms (rocprofv3 --hip-trace, profiles/r04/align_dev_hip_stats.csv:
// 115 us per hipFree).  Blocks handed back are kept per device and given to the next request of a similar size.
//
// Ordering rule (why a cached block can be handed out at once): a block is handed back either after its work has completed (the
// entries that take a caller's stream wait for it first) or behind its last use in the order of the LEGACY default stream of the
// device (Alignment's stages free scratch whose kernels are still queued there).  The next user on the legacy default stream is
// ordered behind that by the stream itself; a user that names another stream (`user`) gets that stream ordered behind everything
// the legacy stream has been given so far (one event) before it receives a cached block.  The deformation handles have their own arenas.
// MVS_SCRATCH_CACHE_MB (default 4096; 0 = no caching) bounds what is kept; mvs_trim() releases it.
#include <hip/hip_runtime.h>

#include <cstdlib>
#include <map>
#include <mutex>
#include <unordered_map>

#include "../../include/mvs.h"
#include "engine.h"

namespace {
struct Pool {
    std::mutex m;
    std::multimap<size_t, void*> idle[MVS_MAX_DEVICES];
    struct Info { size_t bytes; int dev; };
    std::unordered_map<void*, Info> out;               // blocks in use
    size_t kept = 0, cap = (size_t)4096 << 20;
    bool cap_read = false;
    hipEvent_t fence[MVS_MAX_DEVICES] = {};             // "everything the legacy stream has been given" (created on first use)
};
Pool& pool() { static Pool* p = new Pool; return *p; }   // (never destroyed: entries may run during process teardown)
size_t round_up(size_t b) {
    if (b < 4096) return 4096;
    size_t g = 4096;                                     // granule = 1/8 of the size's power of two: at most 12.5 % over
    while ((g << 4) <= b) g <<= 1;
    return (b + g - 1) / g * g;
}
}  // namespace

int mvs_scratch_alloc(void** p, size_t bytes, hipStream_t user) {
    Pool& P = pool();
    int dev = 0;
    HIPCHK(hipGetDevice(&dev));
    const size_t need = round_up(bytes);
    {
        std::lock_guard<std::mutex> g(P.m);
        if (!P.cap_read) {
            if (const char* e = std::getenv("MVS_SCRATCH_CACHE_MB")) P.cap = (size_t)std::strtoull(e, nullptr, 10) << 20;
            P.cap_read = true;
        }
        if (dev >= 0 && dev < MVS_MAX_DEVICES) {
            auto it = P.idle[dev].lower_bound(need);
            if (it != P.idle[dev].end() && it->first <= need + need / 4 + ((size_t)1 << 20)) {
                if (user) {                                      // (under the lock: one event per device, recorded and waited for in one go)
                    if (!P.fence[dev]) HIPCHK(hipEventCreateWithFlags(&P.fence[dev], hipEventDisableTiming));
                    HIPCHK(hipEventRecord(P.fence[dev], nullptr));
                    HIPCHK(hipStreamWaitEvent(user, P.fence[dev], 0));
                }
                *p = it->second;
                P.out[*p] = {it->first, dev};
                P.kept -= it->first;
                P.idle[dev].erase(it);
                return MVS_OK;
            }
        }
    }
    hipError_t e = hipMalloc(p, need);
    if (e != hipSuccess) {                               // out of memory with blocks kept: give them back and try once more
        (void)hipGetLastError();
        mvs_trim();
        e = hipMalloc(p, need);
    }
    int rc = mvs_check_hip(e, "hipMalloc");
    if (rc) { *p = nullptr; return rc; }
    std::lock_guard<std::mutex> g(P.m);
    P.out[*p] = {need, dev};
    return MVS_OK;
}

void mvs_scratch_free(void* p) {
    if (!p) return;
    Pool& P = pool();
    {
        std::lock_guard<std::mutex> g(P.m);
        auto it = P.out.find(p);
        if (it != P.out.end()) {
            const Pool::Info info = it->second;
            P.out.erase(it);
            if (info.dev >= 0 && info.dev < MVS_MAX_DEVICES && P.kept + info.bytes <= P.cap) {
                P.idle[info.dev].emplace(info.bytes, p);
                P.kept += info.bytes;
                return;
            }
        }
    }
    (void)hipFree(p);
}

extern "C" int mvs_trim(void) {
    Pool& P = pool();
    std::multimap<size_t, void*> take[MVS_MAX_DEVICES];
    {
        std::lock_guard<std::mutex> g(P.m);
        for (int d = 0; d < MVS_MAX_DEVICES; ++d) take[d].swap(P.idle[d]);
        P.kept = 0;
    }
    int cur = 0, rc = MVS_OK;
    bool have = false;
    for (int d = 0; d < MVS_MAX_DEVICES; ++d) {
        if (take[d].empty()) continue;
        if (!have) { (void)hipGetDevice(&cur); have = true; }
        (void)hipSetDevice(d);
        for (auto& kv : take[d]) if (hipFree(kv.second) != hipSuccess) rc = MVS_E_HIP;
    }
    if (have) (void)hipSetDevice(cur);
    return rc;
}
