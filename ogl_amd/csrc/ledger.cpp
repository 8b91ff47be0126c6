#include "ledger.hpp"

#include "common.hpp"

#include <execinfo.h>

#include <atomic>
#include <cstdio>
#include <cstdlib>
#include <mutex>
#include <unordered_map>

namespace ogl {
namespace ledger {

namespace {
struct Book {
    std::mutex mu;
    std::unordered_map<void *, size_t> dev, pin;
    int64_t dev_bytes = 0, pin_bytes = 0, dev_peak = 0, pin_peak = 0;
    int64_t dev_calls = 0, pin_calls = 0, dev_frees = 0, pin_frees = 0, unknown_frees = 0;
    std::atomic<int64_t> live[N_KINDS] = {}, made[N_KINDS] = {};
};
Book &book()
{
    static Book *b = new Book;  // never destroyed: DevBufs in static objects may be released after main()
    return *b;
}
// OGL_LEDGER_TRACE=1: every device / pinned allocation with its size and the three callers above the ledger on stderr
// (who still allocates in a time-step loop that should not)
void trace(const char *what, size_t bytes)
{
    static const bool on = [] {
        const char *e = std::getenv("OGL_LEDGER_TRACE");
        return e && *e == '1';
    }();
    if (!on) return;
    void *frames[6];
    const int n = backtrace(frames, 6);
    std::fprintf(stderr, "ogl-ledger %s %zu", what, bytes);
    for (int i = 2; i < n; ++i) std::fprintf(stderr, " %p", frames[i]);
    std::fprintf(stderr, "\n");
}
}  // namespace

hipError_t dev_malloc(void **p, size_t bytes, bool fine_grained)
{
    *p = nullptr;
    const hipError_t e = fine_grained ? hipExtMallocWithFlags(p, bytes, hipDeviceMallocFinegrained) : hipMalloc(p, bytes);
    if (e != hipSuccess) {
        *p = nullptr;
        return e;
    }
    trace("device", bytes);
    Book &b = book();
    std::lock_guard<std::mutex> g(b.mu);
    b.dev[*p] = bytes;
    b.dev_bytes += (int64_t)bytes;
    if (b.dev_bytes > b.dev_peak) b.dev_peak = b.dev_bytes;
    ++b.dev_calls;
    return e;
}

void dev_free(void *p)
{
    if (!p) return;
    {
        Book &b = book();
        std::lock_guard<std::mutex> g(b.mu);
        auto it = b.dev.find(p);
        if (it == b.dev.end()) {
            ++b.unknown_frees;
        } else {
            b.dev_bytes -= (int64_t)it->second;
            b.dev.erase(it);
            ++b.dev_frees;
        }
    }
    (void)hipFree(p);
}

hipError_t pinned_malloc(void **p, size_t bytes)
{
    *p = nullptr;
    const hipError_t e = hipHostMalloc(p, bytes, 0);
    if (e != hipSuccess) {
        *p = nullptr;
        return e;
    }
    trace("pinned", bytes);
    Book &b = book();
    std::lock_guard<std::mutex> g(b.mu);
    b.pin[*p] = bytes;
    b.pin_bytes += (int64_t)bytes;
    if (b.pin_bytes > b.pin_peak) b.pin_peak = b.pin_bytes;
    ++b.pin_calls;
    return e;
}

void pinned_free(void *p)
{
    if (!p) return;
    {
        Book &b = book();
        std::lock_guard<std::mutex> g(b.mu);
        auto it = b.pin.find(p);
        if (it == b.pin.end()) {
            ++b.unknown_frees;
        } else {
            b.pin_bytes -= (int64_t)it->second;
            b.pin.erase(it);
            ++b.pin_frees;
        }
    }
    (void)hipHostFree(p);
}

void created(Kind k)
{
    Book &b = book();
    b.live[k].fetch_add(1, std::memory_order_relaxed);
    b.made[k].fetch_add(1, std::memory_order_relaxed);
}
void destroyed(Kind k) { book().live[k].fetch_sub(1, std::memory_order_relaxed); }

void snapshot(ogl_memory_ledger *out)
{
    Book &b = book();
    std::lock_guard<std::mutex> g(b.mu);
    out->device_bytes = b.dev_bytes;
    out->device_blocks = (int64_t)b.dev.size();
    out->device_peak_bytes = b.dev_peak;
    out->device_alloc_calls = b.dev_calls;
    out->pinned_bytes = b.pin_bytes;
    out->pinned_blocks = (int64_t)b.pin.size();
    out->pinned_peak_bytes = b.pin_peak;
    out->pinned_alloc_calls = b.pin_calls;
    out->streams = b.live[STREAM].load();
    out->events = b.live[EVENT].load();
    out->graph_execs = b.live[GRAPH_EXEC].load();
    out->events_created = b.made[EVENT].load();
    out->graph_execs_created = b.made[GRAPH_EXEC].load();
    out->unknown_frees = b.unknown_frees;
}

}  // namespace ledger
}  // namespace ogl

extern "C" int ogl_memory_ledger_read(ogl_memory_ledger *out)
{
    if (!out) return ogl::fail(OGL_ERR_INVALID, "ogl_memory_ledger_read: out is NULL");
    ogl::ledger::snapshot(out);
    return OGL_OK;
}
