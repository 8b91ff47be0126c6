// ledger.hpp -- every device / pinned-host allocation of the library goes through here, so that "what does this
// process hold on the device on account of libogl_amd" is a number, not an inference from hipMemGetInfo (which moves
// with the runtime's own pools: signal pools, kernarg segments, code objects).  The reference keeps its device state in
// objects owned by the objectRegistry (DevicePersistent/Base/Base.H:53-137) and has no such counter; Ginkgo's
// executors count nothing either.  ogl_memory_ledger_read() (include/ogl_amd.h) exposes the totals; the time-step tests
// assert on them exactly.
#pragma once
#include <cstddef>
#include <cstdint>

#include <hip/hip_runtime.h>

#include "ogl_amd.h"

namespace ogl {
namespace ledger {

// hipMalloc / hipExtMallocWithFlags(hipDeviceMallocFinegrained) / hipFree, counted.  `p` is set to nullptr on failure.
hipError_t dev_malloc(void **p, size_t bytes, bool fine_grained = false);
void dev_free(void *p);  // nullptr is fine
// hipHostMalloc(flags 0) / hipHostFree, counted.
hipError_t pinned_malloc(void **p, size_t bytes);
void pinned_free(void *p);
// Runtime objects that own device or host resources of their own: counted by kind, not by bytes.
enum Kind { STREAM = 0, EVENT = 1, GRAPH_EXEC = 2, N_KINDS = 3 };
void created(Kind k);
void destroyed(Kind k);

void snapshot(ogl_memory_ledger *out);

}  // namespace ledger

inline hipError_t ev_create(hipEvent_t *e, unsigned flags = hipEventDefault)
{
    const hipError_t r = hipEventCreateWithFlags(e, flags);
    if (r == hipSuccess) ledger::created(ledger::EVENT);
    return r;
}
inline void ev_destroy(hipEvent_t e)
{
    (void)hipEventDestroy(e);
    ledger::destroyed(ledger::EVENT);
}
inline hipError_t stream_create(hipStream_t *s)
{
    const hipError_t r = hipStreamCreateWithFlags(s, hipStreamNonBlocking);
    if (r == hipSuccess) ledger::created(ledger::STREAM);
    return r;
}
inline void stream_destroy(hipStream_t s)
{
    (void)hipStreamDestroy(s);
    ledger::destroyed(ledger::STREAM);
}
}  // namespace ogl
