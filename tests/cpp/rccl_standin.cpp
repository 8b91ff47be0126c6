// rccl_standin.cpp -- TEST INFRASTRUCTURE: a stand-in for librccl with the ten entry points libogl_amd binds at run time
// (csrc/comm.cpp RcclApi), so that the RCCL rung of the transport ladder -- ncclCommInitRank, the self-test, the grouped
// ncclSend / ncclRecv of a halo exchange on the communication stream, the ncclAllReduce between the two finalisers --
// executes with MORE THAN ONE RANK on a box with one GPU (real RCCL refuses two ranks on one device).  The ranks are
// processes on one host; the "wire" is a POSIX shared-memory segment named after the unique id.  Stream semantics are
// honoured the blunt way: an operation synchronises the stream it was enqueued on, moves the data through the host, and
// returns -- what the library's event choreography (pack | event | comm stream | event | non-local kernel) is built to
// tolerate.  ncclAllReduce adds the ranks' contributions in RANK ORDER starting from 0.0, so a solve through this
// stand-in is bit-comparable with the distributed oracle (tests/dist_worker.py, allreduce_rank_order).
// Loaded through OGL_RCCL_LIBRARY=<this .so>.  Reference for what is being stood in for:
// DevicePersistent/ExecutorHandler/ExecutorHandler.H:29-32,140-144,167-172 (the GPU-aware communicator pair).
#include <fcntl.h>
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <time.h>
#include <unistd.h>

#include <atomic>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

namespace {

constexpr int MAX_RANKS = 16, AR_DOUBLES = 512, RING = 2;
constexpr size_t MAX_MSG = size_t(4) << 20;  // bytes per message
double timeout_s()
{
    static const double t = [] {
        const char *e = getenv("OGL_STANDIN_TIMEOUT_S");
        return e ? atof(e) : 120.0;
    }();
    return t;
}

struct Slot {
    std::atomic<uint32_t> full;
    uint32_t pad;
    uint64_t bytes;
    unsigned char data[MAX_MSG];
};
struct Channel {  // src -> dst
    Slot slot[RING];
};
struct Board {
    std::atomic<uint32_t> attached, bar_count, bar_gen;
    uint32_t n_ranks;
    double ar[MAX_RANKS][AR_DOUBLES];
    Channel ch[MAX_RANKS][MAX_RANKS];
};

struct Comm {
    Board *b = nullptr;
    int rank = 0, n = 0;
    uint32_t send_seq[MAX_RANKS] = {}, recv_seq[MAX_RANKS] = {};
    uint32_t n_allreduce = 0, n_groups = 0;
};

struct Op {
    bool send;
    const void *src;
    void *dst;
    size_t bytes;
    int peer;
    Comm *c;
    hipStream_t st;
};
thread_local int group_depth = 0;
thread_local std::vector<Op> queued;

double now()
{
    timespec t;
    clock_gettime(CLOCK_MONOTONIC, &t);
    return t.tv_sec + 1e-9 * t.tv_nsec;
}
thread_local int my_rank = -1;
template <class F>
bool wait_until(F f, const char *what = "", int peer = -1)
{
    const double t0 = now();
    double told = t0;
    // (polite: the ranks of a test share a few cores with each other and with their HIP runtimes' helper threads -- seven
    //  ranks spinning flat out starved the eighth's runtime of the CPU it needed to finish the copy they were waiting for)
    for (unsigned spin = 0; !f(); ++spin) {
        if (spin < 200) continue;
        usleep(spin < 2000 ? 20 : 200);
        if ((spin & 63) == 63) {
            const double t = now();
            if (t - t0 > timeout_s()) return false;
            if (t - told > 5.0) {  // (a rank that is killed before its own time-out has said what it was waiting for)
                std::fprintf(stderr, "rccl-standin rank %d: %.0f s in %s (peer %d)\n", my_rank, t - t0, what, peer);
                told = t;
            }
        }
    }
    return true;
}
bool barrier(Comm *c)
{
    Board *b = c->b;
    const uint32_t gen = b->bar_gen.load();
    if (b->bar_count.fetch_add(1) + 1 == (uint32_t)c->n) {
        b->bar_count.store(0);
        b->bar_gen.fetch_add(1);
        return true;
    }
    return wait_until([&] { return b->bar_gen.load() != gen; }, "barrier");
}
size_t size_of(ncclDataType_t t)
{
    switch (t) {
    case ncclDouble: case ncclInt64: case ncclUint64: return 8;
    case ncclFloat: case ncclInt32: case ncclUint32: return 4;
    default: return 1;
    }
}
ncclResult_t run(const Op &o)
{
    Comm *c = o.c;
    if (o.bytes > MAX_MSG || o.peer < 0 || o.peer >= c->n) return ncclInvalidArgument;
    if (o.send) {
        Slot &s = c->b->ch[c->rank][o.peer].slot[c->send_seq[o.peer]++ % RING];
        if (!wait_until([&] { return s.full.load(std::memory_order_acquire) == 0; }, "send: slot free", o.peer)) {
            std::fprintf(stderr, "rccl-standin rank %d: send #%u to rank %d: the slot never became free\n", c->rank,
                         c->send_seq[o.peer] - 1, o.peer);
            return ncclSystemError;
        }
        if (hipMemcpy(s.data, o.src, o.bytes, hipMemcpyDeviceToHost) != hipSuccess) return ncclUnhandledCudaError;
        s.bytes = o.bytes;
        s.full.store(1, std::memory_order_release);
    } else {
        Slot &s = c->b->ch[o.peer][c->rank].slot[c->recv_seq[o.peer]++ % RING];
        if (!wait_until([&] { return s.full.load(std::memory_order_acquire) == 1; }, "recv", o.peer)) {
            std::fprintf(stderr, "rccl-standin rank %d: recv #%u from rank %d (%zu bytes) never arrived; all-reduces so far %u\n",
                         c->rank, c->recv_seq[o.peer] - 1, o.peer, o.bytes, c->n_allreduce);
            return ncclSystemError;
        }
        if (s.bytes != o.bytes) return ncclInvalidUsage;  // (count mismatch between the two ends)
        if (hipMemcpy(o.dst, s.data, o.bytes, hipMemcpyHostToDevice) != hipSuccess) return ncclUnhandledCudaError;
        s.full.store(0, std::memory_order_release);
    }
    return ncclSuccess;
}
ncclResult_t flush()
{
    if (!queued.empty()) ++queued[0].c->n_groups;
    // everything enqueued on the operations' streams before this point has to be done: the data the sends read
    for (const Op &o : queued)
        if (hipStreamSynchronize(o.st) != hipSuccess) return ncclUnhandledCudaError;
    // all sends of the group before any receive: a ring of ranks that each send, then receive, cannot deadlock
    ncclResult_t rc = ncclSuccess;
    for (const Op &o : queued)
        if (o.send && rc == ncclSuccess) rc = run(o);
    for (const Op &o : queued)
        if (!o.send && rc == ncclSuccess) rc = run(o);
    queued.clear();
    return rc;
}
ncclResult_t enqueue(const Op &o)
{
    queued.push_back(o);
    return group_depth > 0 ? ncclSuccess : flush();
}

}  // namespace

extern "C" {

ncclResult_t ncclGetUniqueId(ncclUniqueId *id)
{
    std::memset(id, 0, sizeof(*id));
    int fd = open("/dev/urandom", O_RDONLY);
    unsigned char r[12] = {};
    if (fd >= 0) {
        (void)!read(fd, r, sizeof(r));
        close(fd);
    }
    char *p = id->internal;
    p += std::snprintf(p, 64, "/ogl_rccl_standin_%d_", (int)getpid());
    for (unsigned char c : r) p += std::snprintf(p, 3, "%02x", c);
    return ncclSuccess;
}

ncclResult_t ncclCommInitRank(ncclComm_t *comm, int nranks, ncclUniqueId id, int rank)
{
    if (nranks < 1 || nranks > MAX_RANKS || rank < 0 || rank >= nranks) return ncclInvalidArgument;
    char name[128];
    std::memcpy(name, id.internal, sizeof(name));
    name[127] = 0;
    const int fd = shm_open(name, O_CREAT | O_RDWR, 0600);
    if (fd < 0) return ncclSystemError;
    if (ftruncate(fd, sizeof(Board)) != 0) {  // (sparse: only the pages that carry messages are ever backed)
        close(fd);
        return ncclSystemError;
    }
    void *m = mmap(nullptr, sizeof(Board), PROT_READ | PROT_WRITE, MAP_SHARED, fd, 0);
    close(fd);
    if (m == MAP_FAILED) return ncclSystemError;
    Comm *c = new Comm;
    c->b = static_cast<Board *>(m);
    c->rank = rank;
    c->n = nranks;
    c->b->n_ranks = (uint32_t)nranks;
    c->b->attached.fetch_add(1);
    my_rank = rank;
    const bool all = wait_until([&] { return c->b->attached.load() >= (uint32_t)nranks; }, "init: ranks attaching");
    if (all && !barrier(c)) return ncclSystemError;
    if (rank == 0) shm_unlink(name);  // everybody has it mapped: nothing is left behind in /dev/shm
    if (!all) return ncclSystemError;
    *comm = reinterpret_cast<ncclComm_t>(c);
    return ncclSuccess;
}

ncclResult_t ncclCommDestroy(ncclComm_t comm)
{
    Comm *c = reinterpret_cast<Comm *>(comm);
    if (c) {
        munmap(c->b, sizeof(Board));
        delete c;
    }
    return ncclSuccess;
}

ncclResult_t ncclCommCount(const ncclComm_t comm, int *count)
{
    *count = reinterpret_cast<const Comm *>(comm)->n;
    return ncclSuccess;
}

const char *ncclGetErrorString(ncclResult_t r)
{
    switch (r) {
    case ncclSuccess: return "no error";
    case ncclSystemError: return "stand-in: a peer did not show up in time (or shared memory failed)";
    case ncclInvalidUsage: return "stand-in: the two ends of a message disagree on its size";
    case ncclInvalidArgument: return "stand-in: invalid argument";
    case ncclUnhandledCudaError: return "stand-in: a HIP call failed";
    default: return "stand-in: error";
    }
}

ncclResult_t ncclGroupStart()
{
    ++group_depth;
    return ncclSuccess;
}

ncclResult_t ncclGroupEnd()
{
    if (group_depth <= 0) return ncclInvalidUsage;
    return --group_depth == 0 ? flush() : ncclSuccess;
}

ncclResult_t ncclSend(const void *sendbuff, size_t count, ncclDataType_t datatype, int peer, ncclComm_t comm,
                      hipStream_t stream)
{
    return enqueue(Op{true, sendbuff, nullptr, count * size_of(datatype), peer, reinterpret_cast<Comm *>(comm), stream});
}

ncclResult_t ncclRecv(void *recvbuff, size_t count, ncclDataType_t datatype, int peer, ncclComm_t comm, hipStream_t stream)
{
    return enqueue(Op{false, nullptr, recvbuff, count * size_of(datatype), peer, reinterpret_cast<Comm *>(comm), stream});
}

ncclResult_t ncclAllReduce(const void *sendbuff, void *recvbuff, size_t count, ncclDataType_t datatype, ncclRedOp_t op,
                           ncclComm_t comm, hipStream_t stream)
{
    Comm *c = reinterpret_cast<Comm *>(comm);
    if (datatype != ncclDouble || op != ncclSum || count > (size_t)AR_DOUBLES || group_depth > 0) return ncclInvalidArgument;
    if (hipStreamSynchronize(stream) != hipSuccess) return ncclUnhandledCudaError;
    if (hipMemcpy(c->b->ar[c->rank], sendbuff, count * sizeof(double), hipMemcpyDeviceToHost) != hipSuccess)
        return ncclUnhandledCudaError;
    ++c->n_allreduce;
    if (!barrier(c)) {
        std::fprintf(stderr, "rccl-standin rank %d: all-reduce #%u (%zu doubles): not every rank arrived; exchanges so far %u\n",
                     c->rank, c->n_allreduce, count, c->n_groups);
        return ncclSystemError;
    }
    double out[AR_DOUBLES];
    for (size_t i = 0; i < count; ++i) {
        double s = 0.0;
        for (int r = 0; r < c->n; ++r) s = s + c->b->ar[r][i];  // rank order, from 0.0: what the distributed oracle adds
        out[i] = s;
    }
    if (!barrier(c)) return ncclSystemError;  // (nobody overwrites its row before everybody has read it)
    if (hipMemcpy(recvbuff, out, count * sizeof(double), hipMemcpyHostToDevice) != hipSuccess) return ncclUnhandledCudaError;
    return ncclSuccess;
}

}  // extern "C"
