// comm.cpp -- see comm.hpp.
#include "comm.hpp"

#include <dlfcn.h>
#include <rccl/rccl.h>

#include <algorithm>
#include <cstdlib>
#include <cstring>
#include <mutex>

namespace ogl {

#define OGL_HIP_TRY(expr)                                                                  \
    do {                                                                                   \
        hipError_t e_ = (expr);                                                            \
        if (e_ != hipSuccess)                                                              \
            return fail(OGL_ERR_HIP, "%s failed: %s (%s:%d)", #expr, hipGetErrorString(e_), \
                        __FILE__, __LINE__);                                               \
    } while (0)

// ------------------------------------------------------------------------------------------
// HostComm: stage through pinned host memory, the application moves the bytes
// ------------------------------------------------------------------------------------------
HostComm::HostComm(int r, int n, ogl_allreduce_sum_fn ar, ogl_neighbour_exchange_fn ex, void *user)
    : ar_(ar), ex_(ex), user_(user)
{
    rank = r;
    n_ranks = n;
}

HostComm::~HostComm()
{
    ledger::pinned_free(pin_send_);
    ledger::pinned_free(pin_recv_);
}

int HostComm::reserve(size_t doubles)
{
    if (doubles <= cap_) return OGL_OK;
    ledger::pinned_free(pin_send_);
    ledger::pinned_free(pin_recv_);
    pin_send_ = pin_recv_ = nullptr;
    cap_ = 0;
    const size_t want = doubles + doubles / 2 + 64;
    OGL_HIP_TRY(ledger::pinned_malloc(reinterpret_cast<void **>(&pin_send_), want * sizeof(double)));
    OGL_HIP_TRY(ledger::pinned_malloc(reinterpret_cast<void **>(&pin_recv_), want * sizeof(double)));
    cap_ = want;
    return OGL_OK;
}

int HostComm::allreduce(double *dev, int n, hipStream_t st)
{
    if (!ar_) return fail(OGL_ERR_COMM, "host communicator has no allreduce callback");
    if (int rc = reserve((size_t)n)) return rc;
    OGL_HIP_TRY(hipMemcpyAsync(pin_send_, dev, n * sizeof(double), hipMemcpyDeviceToHost, st));
    OGL_HIP_TRY(hipStreamSynchronize(st));
    ar_(user_, pin_send_, n);
    OGL_HIP_TRY(hipMemcpyAsync(dev, pin_send_, n * sizeof(double), hipMemcpyHostToDevice, st));
    OGL_HIP_TRY(hipStreamSynchronize(st));  // pin_send_ is reused by the next call
    return OGL_OK;
}

int HostComm::exchange(const double *send, double *recv, const std::vector<int> &neighbours,
                       const std::vector<int> &counts, hipStream_t st)
{
    if (!ex_) return fail(OGL_ERR_COMM, "host communicator has no exchange callback");
    size_t total = 0;
    for (int c : counts) total += (size_t)c;
    if (total == 0) return OGL_OK;
    if (int rc = reserve(total)) return rc;
    OGL_HIP_TRY(hipMemcpyAsync(pin_send_, send, total * sizeof(double), hipMemcpyDeviceToHost, st));
    OGL_HIP_TRY(hipStreamSynchronize(st));
    ex_(user_, (int32_t)neighbours.size(), neighbours.data(), counts.data(), pin_send_, pin_recv_);
    OGL_HIP_TRY(hipMemcpyAsync(recv, pin_recv_, total * sizeof(double), hipMemcpyHostToDevice, st));
    OGL_HIP_TRY(hipStreamSynchronize(st));
    return OGL_OK;
}

// ------------------------------------------------------------------------------------------
// RcclComm: librccl is bound at run time (dlopen) so that single-rank users and the CPU-only
// symbol tests never need it.
// ------------------------------------------------------------------------------------------
namespace {

struct RcclApi {
    void *handle = nullptr;
    decltype(&ncclGetUniqueId) GetUniqueId = nullptr;
    decltype(&ncclCommInitRank) CommInitRank = nullptr;
    decltype(&ncclCommDestroy) CommDestroy = nullptr;
    decltype(&ncclAllReduce) AllReduce = nullptr;
    decltype(&ncclSend) Send = nullptr;
    decltype(&ncclRecv) Recv = nullptr;
    decltype(&ncclGroupStart) GroupStart = nullptr;
    decltype(&ncclGroupEnd) GroupEnd = nullptr;
    decltype(&ncclGetErrorString) GetErrorString = nullptr;
    decltype(&ncclCommCount) CommCount = nullptr;
    std::string error;
};

RcclApi &rccl()
{
    static RcclApi api;
    static std::once_flag once;
    std::call_once(once, [] {
        // OGL_RCCL_LIBRARY: a site's own build of RCCL -- or the test suite's stand-in (tests/cpp/rccl_standin.cpp), with
        // which this transport runs with several ranks on a box that has one GPU.  Named explicitly it is the only candidate.
        const char *site = std::getenv("OGL_RCCL_LIBRARY");
        const char *names[] = {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
        if (site && *site) {
            api.handle = dlopen(site, RTLD_NOW | RTLD_LOCAL);
        } else {
            for (const char *n : names) {
                api.handle = dlopen(n, RTLD_NOW | RTLD_GLOBAL);
                if (api.handle) break;
            }
        }
        if (!api.handle) {
            api.error = std::string("cannot load librccl: ") + dlerror();
            return;
        }
#define OGL_SYM(field, sym)                                                  \
    api.field = reinterpret_cast<decltype(api.field)>(dlsym(api.handle, sym)); \
    if (!api.field) api.error = std::string("librccl lacks ") + sym;
        OGL_SYM(GetUniqueId, "ncclGetUniqueId")
        OGL_SYM(CommInitRank, "ncclCommInitRank")
        OGL_SYM(CommDestroy, "ncclCommDestroy")
        OGL_SYM(AllReduce, "ncclAllReduce")
        OGL_SYM(Send, "ncclSend")
        OGL_SYM(Recv, "ncclRecv")
        OGL_SYM(GroupStart, "ncclGroupStart")
        OGL_SYM(GroupEnd, "ncclGroupEnd")
        OGL_SYM(GetErrorString, "ncclGetErrorString")
        OGL_SYM(CommCount, "ncclCommCount")
#undef OGL_SYM
    });
    return api;
}

#define OGL_NCCL_TRY(expr)                                                                     \
    do {                                                                                       \
        ncclResult_t r_ = (expr);                                                              \
        if (r_ != ncclSuccess)                                                                 \
            return fail(OGL_ERR_COMM, "%s failed: %s", #expr, rccl().GetErrorString(r_));      \
    } while (0)

}  // namespace

int RcclComm::library_ready()
{
    RcclApi &api = rccl();
    if (!api.error.empty()) return fail(OGL_ERR_COMM, "%s", api.error.c_str());
    return OGL_OK;
}

int RcclComm::unique_id(void *id_out)
{
    RcclApi &api = rccl();
    if (!api.error.empty()) return fail(OGL_ERR_COMM, "%s", api.error.c_str());
    static_assert(sizeof(ncclUniqueId) == OGL_RCCL_ID_BYTES, "unique id size");
    ncclUniqueId id;
    OGL_NCCL_TRY(api.GetUniqueId(&id));
    std::memcpy(id_out, &id, sizeof(id));
    return OGL_OK;
}

int RcclComm::init(int r, int n, const void *id_bytes)
{
    RcclApi &api = rccl();
    if (!api.error.empty()) return fail(OGL_ERR_COMM, "%s", api.error.c_str());
    ncclUniqueId id;
    std::memcpy(&id, id_bytes, sizeof(id));
    ncclComm_t c = nullptr;
    OGL_NCCL_TRY(api.CommInitRank(&c, n, id, r));
    comm_ = c;
    rank = r;
    n_ranks = n;
    return OGL_OK;
}

int RcclComm::self_test(hipStream_t st)
{
    if (n_ranks < 2) return OGL_OK;
    double *d = nullptr;
    OGL_HIP_TRY(ledger::dev_malloc(reinterpret_cast<void **>(&d), 8 * sizeof(double)));
    struct Free {
        double *p;
        ~Free() { ledger::dev_free(p); }
    } guard{d};
    // Both stages run on EVERY rank whatever the first one gave: a rank that left after a wrong all-reduce would leave
    // the others waiting in the ring (ncclSend / ncclRecv have no time-out).  The verdict is agreed afterwards by a
    // third collective, so all ranks return the same status and step down (or carry on) together.
    // stage 1, all-reduce: sum of (rank + 1) and of (rank + 1) / 2 over the ranks
    const double mine[2] = {rank + 1.0, 0.5 * (rank + 1.0)};
    OGL_HIP_TRY(hipMemcpyAsync(d, mine, sizeof(mine), hipMemcpyHostToDevice, st));
    if (int rc = allreduce(d, 2, st)) return rc;
    double got[2] = {0, 0};
    OGL_HIP_TRY(hipMemcpyAsync(got, d, sizeof(got), hipMemcpyDeviceToHost, st));
    OGL_HIP_TRY(hipStreamSynchronize(st));
    const double tri = 0.5 * n_ranks * (n_ranks + 1.0);
    const bool bad_sum = got[0] != tri || got[1] != 0.5 * tri;
    // stage 2, ring: every rank sends 1000 * rank + destination to its two ring neighbours and expects theirs
    const int prev = (rank + n_ranks - 1) % n_ranks, next = (rank + 1) % n_ranks;
    std::vector<int> nb, cnt;
    if (prev == next) {
        nb = {next};
    } else {
        nb = {std::min(prev, next), std::max(prev, next)};
    }
    cnt.assign(nb.size(), 1);
    double send[2] = {0, 0}, want[2] = {0, 0}, recv[2] = {-1, -1};
    for (size_t i = 0; i < nb.size(); ++i) {
        send[i] = 1000.0 * rank + nb[i];
        want[i] = 1000.0 * nb[i] + rank;
    }
    OGL_HIP_TRY(hipMemcpyAsync(d + 2, send, sizeof(send), hipMemcpyHostToDevice, st));
    if (int rc = exchange(d + 2, d + 4, nb, cnt, st)) return rc;
    OGL_HIP_TRY(hipMemcpyAsync(recv, d + 4, sizeof(recv), hipMemcpyDeviceToHost, st));
    OGL_HIP_TRY(hipStreamSynchronize(st));
    int bad_ring = -1;
    for (size_t i = 0; i < nb.size(); ++i)
        if (recv[i] != want[i] && bad_ring < 0) bad_ring = (int)i;
    // stage 3, the verdict: number of ranks that saw a wrong all-reduce / a wrong ring value.  (A broken all-reduce may
    // garble this sum too: anything but exactly 0, 0 counts as a failure, and a rank that saw a failure itself fails
    // whatever the sum says.)
    const double verdict[2] = {bad_sum ? 1.0 : 0.0, bad_ring >= 0 ? 1.0 : 0.0};
    double all[2] = {-1, -1};
    OGL_HIP_TRY(hipMemcpyAsync(d + 6, verdict, sizeof(verdict), hipMemcpyHostToDevice, st));
    if (int rc = allreduce(d + 6, 2, st)) return rc;
    OGL_HIP_TRY(hipMemcpyAsync(all, d + 6, sizeof(all), hipMemcpyDeviceToHost, st));
    OGL_HIP_TRY(hipStreamSynchronize(st));
    if (bad_sum)
        return fail(OGL_ERR_COMM_SELFTEST, "RCCL self-test: all-reduce over %d ranks gave %g, %g (expected %g, %g)", n_ranks,
                    got[0], got[1], tri, 0.5 * tri);
    if (bad_ring >= 0)
        return fail(OGL_ERR_COMM_SELFTEST, "RCCL self-test: send/recv with rank %d gave %g (expected %g)", nb[bad_ring],
                    recv[bad_ring], want[bad_ring]);
    if (all[0] != 0.0 || all[1] != 0.0)
        return fail(OGL_ERR_COMM_SELFTEST, "RCCL self-test: %g rank(s) saw a wrong all-reduce, %g a wrong send/recv", all[0],
                    all[1]);
    return OGL_OK;
}

RcclComm::~RcclComm()
{
    if (comm_) (void)rccl().CommDestroy(static_cast<ncclComm_t>(comm_));
}

int RcclComm::ranks_seen() const
{
    int count = 0;
    if (!comm_ || rccl().CommCount(static_cast<ncclComm_t>(comm_), &count) != ncclSuccess) return -1;
    return count;
}

int RcclComm::allreduce(double *dev, int n, hipStream_t st)
{
    OGL_NCCL_TRY(rccl().AllReduce(dev, dev, (size_t)n, ncclDouble, ncclSum,
                                  static_cast<ncclComm_t>(comm_), st));
    return OGL_OK;
}

int RcclComm::exchange(const double *send, double *recv, const std::vector<int> &neighbours,
                       const std::vector<int> &counts, hipStream_t st)
{
    if (neighbours.empty()) return OGL_OK;
    RcclApi &api = rccl();
    ncclComm_t c = static_cast<ncclComm_t>(comm_);
    OGL_NCCL_TRY(api.GroupStart());
    size_t off = 0;
    for (size_t i = 0; i < neighbours.size(); ++i) {
        const size_t cnt = (size_t)counts[i];
        // point-to-point over xGMI; the blocked buffers make every message contiguous
        OGL_NCCL_TRY(api.Send(send + off, cnt, ncclDouble, neighbours[i], c, st));
        OGL_NCCL_TRY(api.Recv(recv + off, cnt, ncclDouble, neighbours[i], c, st));
        off += cnt;
    }
    OGL_NCCL_TRY(api.GroupEnd());
    return OGL_OK;
}

}  // namespace ogl
