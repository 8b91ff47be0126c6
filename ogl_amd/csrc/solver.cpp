// solver.cpp -- see solver.hpp.  Citations: reference tree (hpsim/OGL @ 2024-10-16).
#include "solver.hpp"

#include "launch_key.hpp"

#include "setup_kernels.hpp"

#include <algorithm>
#include <atomic>
#include <chrono>
#include <condition_variable>
#include <cmath>
#include <cstddef>
#include <cstring>
#include <future>
#if defined(__x86_64__)
#include <immintrin.h>
#endif
#include <sched.h>
#include <mutex>
#include <thread>

using namespace ogl;

namespace {

double now_ms()
{
    using clk = std::chrono::steady_clock;
    return std::chrono::duration<double, std::milli>(clk::now().time_since_epoch()).count();
}

double *sums_ptr(DevScalars *s)
{
    return reinterpret_cast<double *>(reinterpret_cast<char *>(s) + offsetof(DevScalars, sums));
}

struct EventPair {  // destroyed on every return path
    hipEvent_t e[2] = {nullptr, nullptr};
    ~EventPair()
    {
        for (auto &x : e)
            if (x) ev_destroy(x);
    }
    hipEvent_t &operator[](int i) { return e[i]; }
};

}  // namespace

// ------------------------------------------------------------------------------------------
// Stager
// ------------------------------------------------------------------------------------------
// ---- the host side of a transfer: [caller's pageable array] <-> [pinned buffer], split over helper threads ----
namespace {
#if defined(__x86_64__)
// 32-byte non-temporal stores (AVX2, checked at run time: the library travels as generic x86-64 code); head and
// tail, or the whole range without AVX2 (and on any other host architecture), by memcpy
__attribute__((target("avx2"))) void stream_copy_avx2(char *d, const char *s, size_t len)
{
    const size_t head = std::min(len, (size_t)(-(uintptr_t)d & 31));
    if (head) std::memcpy(d, s, head);
    d += head;
    s += head;
    len -= head;
    size_t i = 0;
    for (; i + 128 <= len; i += 128) {
        const __m256i a = _mm256_loadu_si256(reinterpret_cast<const __m256i *>(s + i));
        const __m256i b = _mm256_loadu_si256(reinterpret_cast<const __m256i *>(s + i + 32));
        const __m256i c = _mm256_loadu_si256(reinterpret_cast<const __m256i *>(s + i + 64));
        const __m256i e = _mm256_loadu_si256(reinterpret_cast<const __m256i *>(s + i + 96));
        _mm256_stream_si256(reinterpret_cast<__m256i *>(d + i), a);
        _mm256_stream_si256(reinterpret_cast<__m256i *>(d + i + 32), b);
        _mm256_stream_si256(reinterpret_cast<__m256i *>(d + i + 64), c);
        _mm256_stream_si256(reinterpret_cast<__m256i *>(d + i + 96), e);
    }
    _mm_sfence();
    if (i < len) std::memcpy(d + i, s + i, len - i);
}
#endif
void stream_copy(void *d, const void *s, size_t len)
{
#if defined(__x86_64__)
    static const bool avx2 = __builtin_cpu_supports("avx2") && std::getenv("OGL_STAGE_PLAIN_STORES") == nullptr;
    if (avx2 && len >= 4096) {
        stream_copy_avx2(static_cast<char *>(d), static_cast<const char *>(s), len);
        return;
    }
#endif
    std::memcpy(d, s, len);
}
}  // namespace

// Persistent helpers: a copy is cut into one part per thread (the caller takes part 0); the helpers sleep on a
// condition variable between copies, so a refresh does not pay a thread start per buffer.
class ogl::CopyPool {
public:
    explicit CopyPool(int n_threads) : n_(std::max(1, n_threads))
    {
        // (binding the helpers to different L3 domains of the caller's socket was measured and changes nothing:
        //  46 GB/s either way on the 2 x EPYC 9575F hosts -- the copy is not what limits a transfer any more)
        for (int t = 1; t < n_; ++t) helpers_.emplace_back([this, t] { run(t); });
    }
    ~CopyPool()
    {
        {
            std::lock_guard<std::mutex> g(m_);
            quit_ = true;
        }
        cv_.notify_all();
        for (auto &h : helpers_) h.join();
    }
    void copy(void *dst, const void *src, size_t len)
    {
        if (n_ == 1 || len < (size_t(1) << 20)) {
            stream_copy(dst, src, len);
            return;
        }
        {
            std::lock_guard<std::mutex> g(m_);
            dst_ = static_cast<char *>(dst);
            src_ = static_cast<const char *>(src);
            len_ = len;
            part_ = ((len + n_ - 1) / n_ + 4095) / 4096 * 4096;
            pending_ = n_ - 1;
            ++gen_;
        }
        cv_.notify_all();
        stream_copy(dst_, src_, std::min(part_, len_));
        std::unique_lock<std::mutex> g(m_);
        done_.wait(g, [this] { return pending_ == 0; });
    }

private:
    void run(int t)
    {
        uint64_t seen = 0;
        for (;;) {
            std::unique_lock<std::mutex> g(m_);
            cv_.wait(g, [&] { return quit_ || gen_ != seen; });
            if (quit_) return;
            seen = gen_;
            char *d = dst_;
            const char *s = src_;
            const size_t len = len_, part = part_;
            g.unlock();
            const size_t off = (size_t)t * part;
            if (off < len) stream_copy(d + off, s + off, std::min(part, len - off));
            g.lock();
            if (--pending_ == 0) done_.notify_one();
        }
    }
    int n_;
    std::vector<std::thread> helpers_;
    std::mutex m_;
    std::condition_variable cv_, done_;
    char *dst_ = nullptr;
    const char *src_ = nullptr;
    size_t len_ = 0, part_ = 0;
    int pending_ = 0;
    uint64_t gen_ = 0;
    bool quit_ = false;
};

Stager::~Stager()
{
    delete pool_;
    for (int i = 0; i < NBUF; ++i) {
        ledger::pinned_free(pin_[i]);
        if (ev_[i]) ev_destroy(ev_[i]);
    }
}

int Stager::init(size_t chunk_bytes)
{
    if (chunk_) return OGL_OK;
    for (int i = 0; i < NBUF; ++i) {
        OGL_HIP_CHECK(ledger::pinned_malloc(&pin_[i], chunk_bytes));
        OGL_HIP_CHECK(ev_create(&ev_[i], hipEventDisableTiming));
    }
    chunk_ = chunk_bytes;
    const char *e = std::getenv("OGL_STAGE_THREADS");
    int n_threads = std::max(1, std::min(32, e ? atoi(e) : 8));
    // ... but no more than the CPUs this thread may run on: an MPI rank bound to one core (mpirun --bind-to core) hands
    // its one-CPU mask on to the helpers, and eight of them time-slicing that core copy no faster than the caller alone
    int allowed = (int)std::thread::hardware_concurrency();
    cpu_set_t mask;
    if (sched_getaffinity(0, sizeof(mask), &mask) == 0) allowed = CPU_COUNT(&mask);
    n_threads = std::min(n_threads, std::max(1, allowed));
    pool_ = new CopyPool(n_threads);
    return OGL_OK;
}

int Stager::h2d(void *dst, const void *src, size_t bytes, hipStream_t st)
{
    const char *s = static_cast<const char *>(src);
    char *d = static_cast<char *>(dst);
    for (size_t off = 0; off < bytes; off += chunk_) {
        const size_t len = std::min(chunk_, bytes - off);
        const int k = next_;
        next_ = (next_ + 1) % NBUF;
        if (busy_[k]) OGL_HIP_CHECK(hipEventSynchronize(ev_[k]));
        pool_->copy(pin_[k], s + off, len);  // the borrowed host array is free again after this
        OGL_HIP_CHECK(hipMemcpyAsync(d + off, pin_[k], len, hipMemcpyHostToDevice, st));
        OGL_HIP_CHECK(hipEventRecord(ev_[k], st));
        busy_[k] = true;
    }
    return OGL_OK;
}

int Stager::d2h(void *dst, const void *src, size_t bytes, hipStream_t st)
{
    const char *s = static_cast<const char *>(src);
    char *d = static_cast<char *>(dst);
    // up to NBUF - 1 device-to-pinned copies in flight while the oldest buffer is copied out to the caller
    struct Flight {
        int k;
        size_t off, len;
    };
    std::vector<Flight> fl;
    size_t head = 0, off = 0;
    while (off < bytes || head < fl.size()) {
        while (off < bytes && fl.size() - head < (size_t)NBUF - 1) {
            const size_t len = std::min(chunk_, bytes - off);
            const int k = next_;
            next_ = (next_ + 1) % NBUF;
            if (busy_[k]) OGL_HIP_CHECK(hipEventSynchronize(ev_[k]));
            OGL_HIP_CHECK(hipMemcpyAsync(pin_[k], s + off, len, hipMemcpyDeviceToHost, st));
            OGL_HIP_CHECK(hipEventRecord(ev_[k], st));
            busy_[k] = true;
            fl.push_back({k, off, len});
            off += len;
        }
        const Flight f = fl[head++];
        OGL_HIP_CHECK(hipEventSynchronize(ev_[f.k]));
        pool_->copy(d + f.off, pin_[f.k], f.len);
        busy_[f.k] = false;
    }
    return OGL_OK;
}

// ------------------------------------------------------------------------------------------
// registry / solver lifetime
// ------------------------------------------------------------------------------------------
ogl_registry::~ogl_registry()
{
    if (device >= 0) (void)hipSetDevice(device);
    if (stream) (void)hipStreamSynchronize(stream);
    solvers.clear();
    comm.reset();
    peer_close();
    cached_precond.values.release();
    cached_precond.block_ptrs.release();
    cached_precond.row_block.release();
    if (comm_stream) {
        (void)hipStreamSynchronize(comm_stream);
        stream_destroy(comm_stream);
        ev_destroy(ev_packed);
        ev_destroy(ev_received);
    }
    if (own_stream && stream) stream_destroy(stream);
}

// ------------------------------------------------------------------------------------------
// peer-write all-reduce mesh (PeerArgs, kernels.hpp)
// ------------------------------------------------------------------------------------------
int ogl_registry::peer_export(void *handle_out)
{
    OGL_HIP_CHECK(hipSetDevice(device));
    if (!peer_local) {
        void *p = nullptr;
        // [mailbox | control slots | halo arena]; OGL_PEER_ARENA_MB (default 64) bounds the halo
        // blocks of all fields of this rank (2 x 8 bytes per halo entry and field)
        size_t arena_mb = 64;
        if (const char *e = std::getenv("OGL_PEER_ARENA_MB")) arena_mb = (size_t)std::max(1, atoi(e));
        arena_words = arena_mb * (1u << 20) / sizeof(unsigned long long);
        arena_used = 0;
        const size_t bytes = (PEER_ARENA_OFF + arena_words) * sizeof(unsigned long long);
        // fine-grained: stores from other GPUs become visible to a kernel that is already running
        OGL_HIP_CHECK(ledger::dev_malloc(&p, bytes, /*fine_grained=*/true));
        OGL_HIP_CHECK(hipMemset(p, 0, PEER_ARENA_OFF * sizeof(unsigned long long)));
        peer_local = static_cast<unsigned long long *>(p);
        OGL_HIP_CHECK(ledger::dev_malloc(reinterpret_cast<void **>(&peer_error), sizeof(int32_t)));
        OGL_HIP_CHECK(hipMemset(peer_error, 0, sizeof(int32_t)));
        OGL_HIP_CHECK(hipDeviceSynchronize());
    }
    hipIpcMemHandle_t h;
    static_assert(sizeof(h) == OGL_PEER_HANDLE_BYTES, "ipc handle size");
    OGL_HIP_CHECK(hipIpcGetMemHandle(&h, peer_local));
    std::memcpy(handle_out, &h, sizeof(h));
    return OGL_OK;
}

int ogl_registry::peer_connect(int rank, int n_ranks, const void *handles)
{
    OGL_HIP_CHECK(hipSetDevice(device));
    if (!peer_local) return fail(OGL_ERR_STATE, "peer_connect before peer_handle");
    if (n_ranks > PEER_MAX_RANKS)
        return fail(OGL_ERR_UNSUPPORTED, "peer all-reduce: at most %d ranks", PEER_MAX_RANKS);
    peer_ready = false;
    peer = PeerArgs{};
    for (int q = 0; q < n_ranks; ++q) {
        if (q == rank) {
            peer.box[q] = peer_local;
            continue;
        }
        if (!peer_mapped[q]) {
            hipIpcMemHandle_t h;
            std::memcpy(&h, static_cast<const char *>(handles) + (size_t)q * sizeof(h), sizeof(h));
            OGL_HIP_CHECK(hipIpcOpenMemHandle(&peer_mapped[q], h, hipIpcMemLazyEnablePeerAccess));
        }
        peer.box[q] = static_cast<unsigned long long *>(peer_mapped[q]);
    }
    peer.world = n_ranks;
    peer.rank = rank;
    if (const char *e = std::getenv("OGL_PEER_TIMEOUT_S"))
        peer.timeout_ticks = (long long)(std::max(0.001, atof(e)) * 1e8);
    // collective self-test (every rank is inside peer_connect now): two all-reduces of known values
    DevBuf<double> d;
    OGL_TRY(d.alloc(2, stream));
    for (int round = 0; round < 2; ++round) {
        const double mine[2] = {rank + 1.0 + round, 0.5 * (rank + 1.0)};
        OGL_HIP_CHECK(hipMemcpyAsync(d.p, mine, sizeof(mine), hipMemcpyHostToDevice, stream));
        launch_peer_allreduce(stream, peer_next(), d.p, 2, peer_error);
        double got[2] = {0, 0};
        int32_t err = 0;
        OGL_HIP_CHECK(hipMemcpyAsync(got, d.p, sizeof(got), hipMemcpyDeviceToHost, stream));
        OGL_HIP_CHECK(hipMemcpyAsync(&err, peer_error, sizeof(err), hipMemcpyDeviceToHost, stream));
        OGL_HIP_CHECK(hipStreamSynchronize(stream));
        const double tri = 0.5 * n_ranks * (n_ranks + 1.0);
        if (err || got[0] != tri + (double)round * n_ranks || got[1] != 0.5 * tri)
            return fail(OGL_ERR_COMM, "peer all-reduce self-test failed (round %d: %g %g, timeout %d)",
                        round, got[0], got[1], (int)err);
    }
    if (n_ranks > 1) {
        // ... and one put / wait round over the ring, the way the halo exchange moves data: a tagged record stored
        // into the NEXT rank's control slot by a kernel, the PREVIOUS rank's record awaited in this rank's memory
        const int next = (rank + 1) % n_ranks, prev = (rank + n_ranks - 1) % n_ranks;
        const unsigned long long tag = 0xFFFFFFF0ull;  // (no pattern handshake ever uses this epoch)
        launch_peer_post(stream, peer.box[next] + PEER_BOX_WORDS + (size_t)rank * 4, tag, 1000ull + rank,
                         2000ull + next, 3000ull);
        OGL_HIP_CHECK(hipStreamSynchronize(stream));
        const unsigned long long *src = peer_local + PEER_BOX_WORDS + (size_t)prev * 4;
        unsigned long long w[4] = {0, 0, 0, 0};
        const double t0 = now_ms();
        for (;;) {
            OGL_HIP_CHECK(hipMemcpy(w, src, sizeof(w), hipMemcpyDeviceToHost));
            if (w[0] == tag) {
                OGL_HIP_CHECK(hipMemcpy(w, src, sizeof(w), hipMemcpyDeviceToHost));  // (payload stored before the tag)
                break;
            }
            if (now_ms() - t0 > (double)peer.timeout_ticks / 1e5)
                return fail(OGL_ERR_COMM, "peer put self-test: nothing arrived from rank %d", prev);
        }
        if (w[1] != 1000ull + prev || w[2] != 2000ull + rank || w[3] != 3000ull)
            return fail(OGL_ERR_COMM, "peer put self-test: record from rank %d is %llu %llu %llu", prev, w[1], w[2], w[3]);
        // closing all-reduce: nobody leaves (and reuses the control slots) before everybody has read its record
        const double one = 1.0;
        OGL_HIP_CHECK(hipMemcpyAsync(d.p, &one, sizeof(one), hipMemcpyHostToDevice, stream));
        launch_peer_allreduce(stream, peer_next(), d.p, 1, peer_error);
        double sum = 0;
        OGL_HIP_CHECK(hipMemcpyAsync(&sum, d.p, sizeof(sum), hipMemcpyDeviceToHost, stream));
        OGL_HIP_CHECK(hipStreamSynchronize(stream));
        if (sum != (double)n_ranks) return fail(OGL_ERR_COMM, "peer put self-test: closing all-reduce gave %g", sum);
        // (the records stay where they are: a pattern handshake compares epochs, which count up from 1)
    }
    // Do two ranks sit on ONE device?  (ranksPerGPU > 1 is not a supported deployment, but it is what every multi-rank run
    // on a 1-GPU box does.)  Waiting workgroups of several ranks' SpMVs can then hold every slot the producers' put
    // kernels need: such a mesh runs with a single waiting workgroup per rank (peerSafeWait) without
    // being asked.  The ranks' PCI bus ids are gathered through the all-reduce that was just tested.
    peer_shared_device = false;
    if (n_ranks > 1) {
        char bus[64] = {0};
        OGL_HIP_CHECK(hipDeviceGetPCIBusId(bus, (int)sizeof(bus), device));
        unsigned long long id = 0;  // domain:bus:device.function -> its hex digits, at most 9 of them (exact in a double)
        for (const char *c = bus; *c; ++c) {
            const int v = (*c >= '0' && *c <= '9') ? *c - '0' : (*c >= 'a' && *c <= 'f') ? *c - 'a' + 10
                          : (*c >= 'A' && *c <= 'F') ? *c - 'A' + 10 : -1;
            if (v >= 0) id = id * 16 + (unsigned long long)v;
        }
        id = (id & 0xFFFFFFFFFFFull) + 1;
        DevBuf<double> ids;
        OGL_TRY(ids.alloc((size_t)n_ranks + 1, stream));
        std::vector<double> mine((size_t)n_ranks + 1, 0.0), all((size_t)n_ranks + 1, 0.0);
        mine[(size_t)rank] = (double)id;
        OGL_HIP_CHECK(hipMemcpyAsync(ids.p, mine.data(), mine.size() * sizeof(double), hipMemcpyHostToDevice, stream));
        for (int i = 0; i < n_ranks; i += 2)
            launch_peer_allreduce(stream, peer_next(), ids.p + i, std::min(2, n_ranks - i), peer_error);
        OGL_HIP_CHECK(hipMemcpyAsync(all.data(), ids.p, all.size() * sizeof(double), hipMemcpyDeviceToHost, stream));
        int32_t gather_err = 0;
        OGL_HIP_CHECK(hipMemcpyAsync(&gather_err, peer_error, sizeof(int32_t), hipMemcpyDeviceToHost, stream));
        OGL_HIP_CHECK(hipStreamSynchronize(stream));
        for (int a = 0; a < n_ranks; ++a)
            for (int b = a + 1; b < n_ranks; ++b) peer_shared_device = peer_shared_device || all[(size_t)a] == all[(size_t)b];
        // a timed-out or partial gather (an id of 0 = missing: the ids are offset by 1) must not let ranks decide
        // differently: such a mesh runs the conservative single-waiter halo path on every rank that saw the gap
        for (int a = 0; a < n_ranks; ++a) peer_shared_device = peer_shared_device || all[(size_t)a] == 0.0;
        if (gather_err != 0) peer_shared_device = true;
    }
    peer_ready = true;
    return OGL_OK;
}

void ogl_registry::peer_close()
{
    peer_ready = false;
    for (auto &m : peer_mapped)
        if (m) {
            (void)hipIpcCloseMemHandle(m);
            m = nullptr;
        }
    ledger::dev_free(peer_local);
    ledger::dev_free(peer_error);
    peer_local = nullptr;
    peer_error = nullptr;
    peer = PeerArgs{};
}

// ------------------------------------------------------------------------------------------
// Peer-put halo: per sparsity pattern, every rank takes a block of its arena, tells each neighbour
// where that neighbour's values go (control slot [this rank] of the neighbour's allocation, written
// by a one-thread kernel), reads what the neighbours said, and all ranks agree (all-reduce, which is
// also the barrier that frees the control slots) whether this field uses the peer-put exchange.
// ------------------------------------------------------------------------------------------
int ogl_solver::setup_peer_halo()
{
    peer_halo = false;
    peer_nb.clear();
    halo_seq = 0;
    props["peerHalo"] = 0.0;
    ogl_registry &R = *reg;
    if (!R.peer_ready) return OGL_OK;
    hipStream_t st = R.stream;
    const int nn = (int)neighbours.size();
    const size_t nh = (size_t)pat.non_local_nnz;
    const uint32_t epoch = ++R.halo_epoch;
    double cannot = 0.0;
    const size_t words = (2 * (size_t)nn + 2 * nh + 15) / 16 * 16;
    // a pattern rebuild reuses the field's block when it is large enough (the arena only grows)
    const bool reuse = peer_block_words >= words && words > 0;
    if (nn > PEER_MAX_NEIGH || (!reuse && R.arena_used + words > R.arena_words)) cannot = 1.0;
    if (cannot == 0.0) {
        if (!reuse) {
            peer_block = R.arena_used;
            peer_block_words = words;
            R.arena_used += words;
        }
        if (nn)
            OGL_HIP_CHECK(hipMemsetAsync(R.peer_local + PEER_ARENA_OFF + peer_block, 0,
                                         2 * (size_t)nn * sizeof(unsigned long long), st));
    }
    int32_t seg = 0;
    for (int i = 0; i < nn; ++i) {
        unsigned long long *dst = R.peer.box[neighbours[i]] + PEER_BOX_WORDS + (size_t)R.peer.rank * 4;
        launch_peer_post(st, dst, epoch, cannot == 0.0 ? (unsigned long long)peer_block : ~0ull,
                         ((unsigned long long)nn << 32) | (unsigned)i,
                         ((unsigned long long)nh << 32) | (unsigned)seg);
        seg += counts[i];
    }
    OGL_HIP_CHECK(hipStreamSynchronize(st));
    peer_nb.resize(nn);
    for (int i = 0; i < nn; ++i) {
        const unsigned long long *src = R.peer_local + PEER_BOX_WORDS + (size_t)neighbours[i] * 4;
        unsigned long long w[4] = {0, 0, 0, 0};
        const double t0 = now_ms();
        for (;;) {
            OGL_HIP_CHECK(hipMemcpy(w, src, sizeof(w), hipMemcpyDeviceToHost));
            if ((uint32_t)w[0] == epoch) {
                // the epoch word is stored last: re-read once so that the payload is the final one
                OGL_HIP_CHECK(hipMemcpy(w, src, sizeof(w), hipMemcpyDeviceToHost));
                break;
            }
            if (now_ms() - t0 > (double)R.peer.timeout_ticks / 1e5)
                return fail(OGL_ERR_COMM, "peer halo handshake: rank %d did not answer", neighbours[i]);
        }
        if (w[1] == ~0ull) cannot = 1.0;
        peer_nb[i].block = (size_t)w[1];
        peer_nb[i].n_neigh = (int32_t)(w[2] >> 32);
        peer_nb[i].my_index = (int32_t)(w[2] & 0xffffffffu);
        peer_nb[i].n_halo = (int32_t)(w[3] >> 32);
        peer_nb[i].my_seg = (int32_t)(w[3] & 0xffffffffu);
    }
    DevBuf<double> agree;
    OGL_TRY(agree.alloc(2, st));
    OGL_HIP_CHECK(hipMemcpyAsync(agree.p, &cannot, sizeof(double), hipMemcpyHostToDevice, st));
    OGL_TRY(R.allreduce(agree.p, 1));
    OGL_HIP_CHECK(hipMemcpyAsync(&cannot, agree.p, sizeof(double), hipMemcpyDeviceToHost, st));
    OGL_HIP_CHECK(hipStreamSynchronize(st));
    peer_halo = cannot == 0.0;
    props["peerHalo"] = peer_halo ? 1.0 : 0.0;
    return OGL_OK;
}

PeerHalo ogl_solver::peer_halo_args(uint32_t seq) const
{
    const ogl_registry &R = *reg;
    const int nn = (int)neighbours.size();
    const unsigned par = seq & 1u;
    PeerHalo P;
    P.n_neigh = nn;
    P.seq = seq;
    P.timeout_ticks = R.peer.timeout_ticks;
    int32_t off = 0;
    for (int i = 0; i < nn; ++i) {
        P.send_off[i] = off;
        off += counts[i];
        const PeerNeighbour &nb = peer_nb[i];
        unsigned long long *base = R.peer.box[neighbours[i]] + PEER_ARENA_OFF + nb.block;
        P.remote_flag[i] = base + (size_t)par * nb.n_neigh + nb.my_index;
        P.remote_recv[i] = reinterpret_cast<double *>(base + 2 * (size_t)nb.n_neigh +
                                                      (size_t)par * nb.n_halo + nb.my_seg);
    }
    P.send_off[nn] = off;
    P.local_flag = R.peer_local + PEER_ARENA_OFF + peer_block + (size_t)par * nn;
    return P;
}

double *ogl_solver::peer_recv(uint32_t seq) const
{
    const size_t nn = neighbours.size();
    return reinterpret_cast<double *>(reg->peer_local + PEER_ARENA_OFF + peer_block + 2 * nn +
                                      (size_t)(seq & 1u) * (size_t)pat.non_local_nnz);
}

// one waiting workgroup per rank instead of every boundary workgroup of the SpMV: asked for (property), or because
// peer_connect found two ranks on one device
bool ogl_solver::peer_safe_wait() const
{
    return prop("peerSafeWait", reg->peer_shared_device ? 1.0 : 0.0) != 0.0;
}

int ogl_registry::allreduce(double *dev, int n)
{
    if (!comm->multi()) return OGL_OK;
    if (!peer_ready) return comm->allreduce(dev, n, stream);
    for (int i = 0; i < n; i += 2)
        launch_peer_allreduce(stream, peer_next(), dev + i, std::min(2, n - i), peer_error);
    return OGL_OK;
}

void ogl_solver::drop_cg_graph()
{
    if (cg_graph) {
        (void)hipGraphExecDestroy(cg_graph);
        ledger::destroyed(ledger::GRAPH_EXEC);
    }
    cg_graph = nullptr;
    cg_graph_key = 0;
}

ogl_solver::~ogl_solver()
{
    drop_cg_graph();
    ledger::pinned_free(h_scal);
    ledger::dev_free(lead_box);
    for (auto &e : poll_ev)
        if (e) ev_destroy(e);
    for (auto &e : prof_ev)
        if (e) ev_destroy(e);
    for (auto &e : chk_ev)
        if (e) ev_destroy(e);
}

double ogl_solver::prop(const std::string &key, double dflt) const
{
    auto it = props.find(key);
    return it == props.end() ? dflt : it->second;
}

// Size above which matrix data is streamed past the caches (STREAM instantiations, common.hpp).  The
// property `streamAboveBytes` / the environment variable OGL_STREAM_ABOVE_BYTES override the built-in
// threshold: 0 forces the STREAM kernels onto small systems, which is how the parity tests reach them.
double ogl_solver::stream_above_bytes() const
{
    static const double env_default = [] {
        const char *e = std::getenv("OGL_STREAM_ABOVE_BYTES");
        return e ? atof(e) : STREAM_MATRIX_ABOVE_BYTES;
    }();
    return prop("streamAboveBytes", env_default);
}

// What a turn of this solver touches BESIDES the system matrix and GKOCG's five vectors (which the built-in threshold
// was measured with): the preconditioner's own matrices (ISAI: W and W^T, GISAI: W; block Jacobi: its blocks) and the
// further vectors of GKOBiCGStab / the Krylov basis of GKOGMRES.  All of it passes through the Infinity Cache once per
// turn, so it counts when the question is "does the turn's working set still live there" (property streamTurnSet 0:
// the matrix alone decides, as before round 5).
double ogl_solver::turn_extra_bytes() const
{
    if (prop("streamTurnSet", 1.0) == 0.0) return 0.0;
    const double N = (double)pat.n_rows, nnz = (double)pat.local_nnz;
    double extra = 0.0;
    if (cfg.preconditioner == OGL_PRECOND_ISAI) extra += 10.0 * (nnz + N);      // tril(A) twice, ~10 bytes per entry
    if (cfg.preconditioner == OGL_PRECOND_GISAI) extra += 10.0 * nnz;
    if (cfg.preconditioner == OGL_PRECOND_BJ && cfg.max_block_size > 1) extra += 8.0 * cfg.max_block_size * N;
    if (cfg.preconditioner != OGL_PRECOND_NONE && !(cfg.preconditioner == OGL_PRECOND_BJ && cfg.max_block_size == 1))
        extra += 16.0 * N;                                                      // materialised z (and the ISAI temporary)
    if (cfg.solver == OGL_SOLVER_BICGSTAB) extra += 32.0 * N;
    if (cfg.solver == OGL_SOLVER_GMRES) extra += 8.0 * N * ((cfg.krylov_dim > 0 ? cfg.krylov_dim : 100) + 1 - 3);
    return extra;
}

// chunks per XCD group of the CSR-stream / compressed SpMV (DevCsr::xcd_group): property `xcdGroup`, else the
// environment's OGL_XCD_GROUP, else what the pattern's set-up chose (0 = the kernels' built-in 4)
int32_t ogl_solver::xcd_group() const
{
    static const int env_default = [] {
        const char *e = std::getenv("OGL_XCD_GROUP");
        return e ? atoi(e) : -1;
    }();
    const int v = (int)prop("xcdGroup", (double)env_default);
    return v >= 0 ? v : pat_xcd_group;
}

DevCsr ogl_solver::csr() const
{
    DevCsr A;
    A.n_rows = pat.n_rows;
    A.nnz = pat.local_nnz;
    A.row_ptrs = d_row_ptrs.p;
    A.cols = d_cols.p;
    A.vals = d_vals.p;
    A.stream = 12.0 * (double)pat.local_nnz + 44.0 * (double)pat.n_rows + turn_extra_bytes() > stream_above_bytes();
    A.xcd_group = xcd_group();
    A.lds_rounds = prop("spmvLdsRounds", 1.0) == 2.0 ? 2 : 1;
    if (d_band_order.n) {
        A.block_order = d_band_order.p;
        A.n_blocks = (int32_t)d_band_order.n;
    }
    if (s21_use && s21_state == 1) {
        A.chunks21 = d_s21_chunks.p;
        A.codes21 = d_s21_codes.p;
        A.far_idx21 = d_s21_far_idx.p;
        A.far_col21 = d_s21_far_col.p;
    }
    return A;
}

// Packed columns for the CSR-stream kernel, from the device pattern (setup_kernels.hip).  The values stay the
// CSR array: nothing to refresh per coefficient update.
int ogl_solver::build_stream21()
{
    hipStream_t st = reg->stream;
    s21_state = -1;
    s21_use = false;
    const int32_t N = pat.n_rows;
    const size_t nc = (size_t)n_chunks(N);
    if (N == 0) return OGL_OK;
    DevBuf<int32_t> words, tmp, flags, far;
    OGL_TRY(d_s21_chunks.alloc(nc, st));
    OGL_TRY(words.alloc(nc + 1, st));
    OGL_TRY(far.alloc(nc + 1, st));
    OGL_TRY(tmp.alloc(scan_tmp_len((int64_t)nc), st));
    OGL_TRY(flags.alloc(1, st));
    Stream21Build b;
    b.n_rows = N;
    b.row_ptrs = d_row_ptrs.p;
    b.cols = d_cols.p;
    b.chunks = d_s21_chunks.p;
    b.words = words.p;
    b.scan_tmp = tmp.p;
    b.flags = flags.p;
    b.far = far.p;
    launch_stream21_plan(st, b);
    int32_t total = 0, total_far = 0;
    OGL_HIP_CHECK(hipMemcpyAsync(&total, words.p + nc, sizeof(int32_t), hipMemcpyDeviceToHost, st));
    OGL_HIP_CHECK(hipMemcpyAsync(&total_far, far.p + nc, sizeof(int32_t), hipMemcpyDeviceToHost, st));
    OGL_HIP_CHECK(hipStreamSynchronize(st));
    OGL_HIP_CHECK(hipGetLastError());
    // (a pattern whose chunks reach far beyond their 2^21-column windows all over the place -- a random numbering of a
    //  large mesh -- is left to the plain CSR-stream kernel)
    if (total < 0 || total_far < 0 || (double)total_far > STREAM21_MAX_FAR * (double)pat.local_nnz) {
        d_s21_chunks.release();
        return OGL_OK;
    }
    OGL_TRY(d_s21_codes.alloc((size_t)total + 1, st));
    OGL_TRY(d_s21_far_idx.alloc((size_t)total_far + 1, st));
    OGL_TRY(d_s21_far_col.alloc((size_t)total_far + 1, st));
    launch_stream21_fill(st, b, d_s21_codes.p, d_s21_far_idx.p, d_s21_far_col.p);
    OGL_HIP_CHECK(hipStreamSynchronize(st));
    OGL_HIP_CHECK(hipGetLastError());
    s21_state = 1;
    props["csr21FarEntries"] = (double)total_far;
    // bytes one SpMV reads of this layout: values + code words + row pointers + chunk headers (+ the far lists and the
    // values and x their entries read a second time)
    props["csr21MatrixBytes"] = 8.0 * (double)pat.local_nnz + 16.0 * (double)total + 4.0 * ((double)N + 1.0) +
                                16.0 * (double)nc + 24.0 * (double)total_far;
    return OGL_OK;
}

DevEll ogl_solver::ell() const
{
    DevEll E;
    E.n_rows = pat.n_rows;
    E.width = ell_width;
    E.stride = ell_stride;
    E.cols = d_ell_cols.p;
    E.vals = d_ell_vals.p;
    E.stream = 12.0 * (double)ell_width * (double)ell_stride + 40.0 * (double)pat.n_rows + turn_extra_bytes() > stream_above_bytes();
    return E;
}

// matrixFormat Ell (CsrMatrixWrapper.H:146-149): `width` = longest row; slot i of row r lives at
// i * stride + r.  ell_map holds the CSR position of each slot (-1 = padding), so the values are
// refreshed from the freshly permuted CSR values whatever path produced them.
int ogl_solver::build_ell()
{
    hipStream_t st = reg->stream;
    OGL_TRY(download_local_pattern(pat));
    const int32_t N = pat.n_rows;
    int32_t width = 0;
    for (int32_t r = 0; r < N; ++r) width = std::max(width, pat.row_ptrs[r + 1] - pat.row_ptrs[r]);
    const int64_t stride = ((int64_t)N + 1) / 2 * 2 + 2;  // even, and the pair load of the last row fits
    const size_t len = (size_t)width * (size_t)stride;
    std::vector<int32_t> cols(len, -1), map(len, -1);
    for (int32_t r = 0; r < N; ++r)
        for (int32_t k = pat.row_ptrs[r], i = 0; k < pat.row_ptrs[r + 1]; ++k, ++i) {
            cols[(size_t)i * stride + r] = pat.cols[k];
            map[(size_t)i * stride + r] = k;
        }
    OGL_TRY(d_ell_cols.alloc(len + 2, st));
    OGL_TRY(d_ell_map.alloc(len + 2, st));
    OGL_TRY(d_ell_vals.alloc(len + 2, st));
    OGL_TRY(reg->stager.h2d(d_ell_cols.p, cols.data(), len * sizeof(int32_t), st));
    OGL_TRY(reg->stager.h2d(d_ell_map.p, map.data(), len * sizeof(int32_t), st));
    ell_width = width;
    ell_stride = stride;
    ell_ready = true;
    return OGL_OK;
}

DevSell ogl_solver::sell() const
{
    DevSell S;
    S.n_rows = pat.n_rows;
    S.chunks = d_sell_chunks.p;
    S.dict = d_sell_dict.p;
    S.codes = d_sell_codes.p;
    S.vals = d_sell_vals.p;
    S.stream = sell_bytes + 40.0 * (double)pat.n_rows + turn_extra_bytes() > stream_above_bytes();
    S.xcd_group = xcd_group();
    if (d_band_order.n) {
        S.block_order = d_band_order.p;
        S.n_blocks = (int32_t)d_band_order.n;
    }
    if (n_spill) {
        S.spill_chunk_ptr = d_spill_chunks.p;
        S.spill_rows = d_spill_rows.p;
        S.spill_ptrs = d_spill_ptrs.p;
        S.spill_cols = d_spill_cols.p;
        S.spill_vals = d_spill_vals.p;
    }
    return S;
}

DevSym ogl_solver::sym() const
{
    DevSym S;
    S.n_rows = pat.n_rows;
    S.nd = sym_nd;
    for (int j = 0; j < 4; ++j) S.d[j] = sym_d[j];
    S.mask = d_sym_mask.p;
    S.planes = d_sym_planes.p;
    S.stream = 8.0 * (double)d_sym_planes.n + 41.0 * (double)pat.n_rows + turn_extra_bytes() > stream_above_bytes();
    if (d_sym_order.n && !band_order_off) {
        S.block_order = d_sym_order.p;
        S.n_blocks = (int32_t)d_sym_order.n;
    }
    return S;
}

DevSymx ogl_solver::symx() const
{
    DevSymx S;
    S.n_rows = pat.n_rows;
    S.chunks = d_symx_chunks.p;
    S.mask = d_symx_mask.p;
    S.planes = d_symx_planes.p;
    S.ex_rowptr = d_symx_ex_rowptr.p;
    S.ex_cols = d_symx_ex_cols.p;
    S.ex_vals = d_symx_ex_vals.p;
    S.stream = symx_bytes + 41.0 * (double)pat.n_rows + turn_extra_bytes() > stream_above_bytes();
    S.fast = symx_fast;
    S.n_blocks = (int32_t)d_symx_chunks.n;
    S.chunks_general = d_symx_chunks_general.p;
    S.n_blocks_general = (int32_t)d_symx_chunks_general.n;
    S.ex_lrow = d_symx_ex_lrow.p;
    S.xcd_group = xcd_group();
    return S;
}

// Per-chunk half storage against the compressed full-storage copy, once per pattern (same bits either way): the
// former moves about a third fewer bytes, but rows with explicit entries cost it a merge; the faster one stays.
int ogl_solver::tune_symx()
{
    hipStream_t st = reg->stream;
    EventPair ev;
    OGL_HIP_CHECK(ev_create(&ev[0]));
    OGL_HIP_CHECK(ev_create(&ev[1]));
    OGL_HIP_CHECK(hipMemsetAsync(d_p.p, 0, ((size_t)pat.n_rows + 2) * sizeof(double), st));
    SpmvDots dots;
    dots.with = d_p.p;
    dots.part = d_part0.p;
    constexpr int WARM = 2, TIMED = 5;
    float best[2] = {1e30f, 1e30f};  // [0] compressed full storage, [1] per-chunk half storage
    for (int round = 0; round < WARM + TIMED; ++round)
        for (int which = 0; which < 2; ++which) {
            OGL_HIP_CHECK(hipEventRecord(ev[0], st));
            if (which)
                launch_spmv_symx(st, symx(), SPMV_PLAIN, d_p.p, nullptr, d_q.p, dots, nullptr);
            else
                launch_spmv_sell(st, sell(), SPMV_PLAIN, d_p.p, nullptr, d_q.p, dots, nullptr);
            OGL_HIP_CHECK(hipEventRecord(ev[1], st));
            OGL_HIP_CHECK(hipEventSynchronize(ev[1]));
            float ms = 0;
            OGL_HIP_CHECK(hipEventElapsedTime(&ms, ev[0], ev[1]));
            if (round >= WARM) best[which] = std::min(best[which], ms);
        }
    OGL_HIP_CHECK(hipGetLastError());
    props["spmvTunedSellUs"] = 1e3 * best[0];
    props["spmvTunedSymxUs"] = 1e3 * best[1];
    OGL_HIP_CHECK(hipStreamSynchronize(st));
    if (best[1] <= best[0]) {  // half storage stays
        for (auto *b : {&d_sell_dict, &d_sell_map, &d_spill_rows, &d_spill_ptrs, &d_spill_cols, &d_spill_map,
                        &d_spill_chunks})
            b->release();
        d_sell_chunks.release();
        d_sell_codes.release();
        d_sell_vals.release();
        d_spill_vals.release();
        n_spill = n_spill_rows = 0;
        sell_state = -1;
        props["sellMatrixBytes"] = symx_bytes;
    } else {                   // full storage stays
        symx_state = -1;
        for (auto *b : {&d_symx_map, &d_symx_ex_rowptr, &d_symx_ex_cols, &d_symx_ex_map, &d_symx_ex_lrow}) b->release();
        d_symx_chunks.release();
        d_symx_chunks_general.release();
        d_symx_mask.release();
        d_symx_planes.release();
        d_symx_ex_vals.release();
        props["symmetricHalf"] = 0.0;
        props["symmetricHalfPerChunk"] = 0.0;
        props["sellMatrixBytes"] = sell_bytes;
    }
    return OGL_OK;
}

// Half storage with per-chunk distances (build_symx_layout, host side: it needs the whole pattern); the planes
// and the explicit entries are refreshed from the CSR values through their maps.
int ogl_solver::build_symx()
{
    hipStream_t st = reg->stream;
    symx_state = -1;
    if (pat.n_rows == 0) return OGL_OK;
    OGL_TRY(download_local_pattern(pat));
    SymxLayout L;
    if (!build_symx_layout(pat.n_rows, pat.row_ptrs.data(), pat.cols.data(), L)) return OGL_OK;
    const size_t nex = L.ex_cols.size();
    // headers in dispatch order, each naming its chunk (symx_block_order): the lean kernel's list, the general one's
    std::vector<SymxChunk> hdr_ord[2];
    int64_t general_chunks = 0;
    for (int g = 0; g < 2; ++g) {
        std::vector<int32_t> order;
        symx_block_order(L, g == 1, order);
        hdr_ord[g].resize(order.size());
        for (size_t b = 0; b < order.size(); ++b) {
            if (order[b] >= 0) hdr_ord[g][b] = L.chunks[(size_t)order[b]];
            else hdr_ord[g][b] = SymxChunk{};
            hdr_ord[g][b].chunk = order[b];
            if (g == 1 && order[b] >= 0) ++general_chunks;
        }
    }
    OGL_TRY(d_symx_chunks.alloc(hdr_ord[0].size(), st));
    OGL_TRY(d_symx_chunks_general.alloc(hdr_ord[1].size(), st));
    OGL_TRY(d_symx_ex_lrow.alloc(nex + NNZ_PAD, st));
    OGL_TRY(d_symx_mask.alloc(L.mask.size(), st));
    OGL_TRY(d_symx_map.alloc(L.map.size(), st));
    OGL_TRY(d_symx_planes.alloc(L.map.size(), st));
    OGL_TRY(d_symx_ex_rowptr.alloc(std::max<size_t>(1, L.ex_rowptr.size()), st));
    OGL_TRY(d_symx_ex_cols.alloc(nex + NNZ_PAD, st));
    OGL_TRY(d_symx_ex_map.alloc(nex + NNZ_PAD, st));
    OGL_TRY(d_symx_ex_vals.alloc(nex + NNZ_PAD, st));
    if (!hdr_ord[0].empty())
        OGL_TRY(reg->stager.h2d(d_symx_chunks.p, hdr_ord[0].data(), hdr_ord[0].size() * sizeof(SymxChunk), st));
    if (!hdr_ord[1].empty())
        OGL_TRY(reg->stager.h2d(d_symx_chunks_general.p, hdr_ord[1].data(), hdr_ord[1].size() * sizeof(SymxChunk), st));
    OGL_TRY(reg->stager.h2d(d_symx_mask.p, L.mask.data(), L.mask.size(), st));
    OGL_TRY(reg->stager.h2d(d_symx_map.p, L.map.data(), L.map.size() * sizeof(int32_t), st));
    if (!L.ex_rowptr.empty())
        OGL_TRY(reg->stager.h2d(d_symx_ex_rowptr.p, L.ex_rowptr.data(), L.ex_rowptr.size() * sizeof(int32_t), st));
    if (nex) {
        OGL_TRY(reg->stager.h2d(d_symx_ex_cols.p, L.ex_cols.data(), nex * sizeof(int32_t), st));
        OGL_TRY(reg->stager.h2d(d_symx_ex_map.p, L.ex_map.data(), nex * sizeof(int32_t), st));
        OGL_TRY(reg->stager.h2d(d_symx_ex_lrow.p, L.ex_lrow.data(), nex * sizeof(int32_t), st));
    }
    props["symxGeneralChunks"] = (double)general_chunks;
    symx_state = 1;
    symx_fast = L.all_fast;
    symx_values_stale = true;
    // bytes one SpMV reads of this layout: planes, masks, headers, explicit entries (value + column + row) and their
    // row pointers
    symx_bytes = 8.0 * (double)(L.map.size() - 2) + (double)(L.mask.size() - 16) + 96.0 * (double)L.chunks.size() +
                 16.0 * (double)nex + 4.0 * (double)L.ex_rowptr.size();
    props["sellMatrixBytes"] = symx_bytes;
    props["sellReadSlots"] = (double)(L.map.size() - 2);
    props["sellAllocatedSlots"] = (double)(L.map.size() - 2);
    props["sellChunksDelta16"] = 0.0;
    props["sellChunksCol32"] = 0.0;
    props["sellSpilledEntries"] = 0.0;
    props["symxPlanarEntries"] = (double)L.planar;
    props["symxExplicitEntries"] = (double)nex;
    return OGL_OK;
}

// Once per sparsity pattern; d_sym_map refreshes the planes from the permuted CSR values on the device.
int ogl_solver::build_sym(const SymLayout &L)
{
    hipStream_t st = reg->stream;
    OGL_TRY(d_sym_mask.alloc(L.mask.size(), st));
    OGL_TRY(d_sym_map.alloc(L.map.size(), st));
    OGL_TRY(d_sym_planes.alloc(L.map.size(), st));
    OGL_TRY(reg->stager.h2d(d_sym_mask.p, L.mask.data(), L.mask.size(), st));
    OGL_TRY(reg->stager.h2d(d_sym_map.p, L.map.data(), L.map.size() * sizeof(int32_t), st));
    return finish_sym(L.nd, L.d);
}

// the part of the half-storage set-up that does not depend on where mask and map were built
int ogl_solver::finish_sym(int nd, const int32_t *d)
{
    hipStream_t st = reg->stream;
    sym_nd = nd;
    for (int j = 0; j < 4; ++j) sym_d[j] = j < nd ? d[j] : 0;
    std::vector<int32_t> order;
    band_block_order(pat.n_rows, d[nd - 1], order);
    d_sym_order.release();
    if (!order.empty()) {
        OGL_TRY(d_sym_order.alloc(order.size(), st));
        OGL_TRY(reg->stager.h2d(d_sym_order.p, order.data(), order.size() * sizeof(int32_t), st));
    }
    sym_state = 1;
    sym_values_stale = true;
    // bytes one SpMV reads of this layout (bench.py's moved-bytes model): planes + masks
    props["sellMatrixBytes"] = 8.0 * (double)(d_sym_map.n - 2) + (double)(d_sym_mask.n - 16);
    props["sellReadSlots"] = (double)(d_sym_map.n - 2);
    props["sellAllocatedSlots"] = (double)(d_sym_map.n - 2);
    props["sellChunksDelta16"] = 0.0;
    props["sellChunksCol32"] = 0.0;
    props["sellSpilledEntries"] = 0.0;
    return OGL_OK;
}

// ------------------------------------------------------------------------------------------
// Device set-up (setup_kernels.hip): the pattern of a field straight from the lduMatrix addressing, without
// the host holding its 12 bytes per entry.  `np` carries the small parts (build_host_pattern_meta); the
// arrays land in d_row_ptrs / d_cols / d_ldu_mapping / d_diag_pos (allocated by the caller).  *built stays
// false when the addressing is not conforming (a face with owner >= neighbour): the host algorithm, which
// follows the reference's segment order for such input, takes over.
// ------------------------------------------------------------------------------------------
int ogl_solver::build_pattern_on_device(const ogl_ldu_view &ldu, HostPattern &np, bool *built)
{
    *built = false;
    hipStream_t st = reg->stream;
    const int32_t N = np.n_rows, F = np.upper_nnz;
    if (N == 0) return OGL_OK;
    std::vector<ogl_label> ir, ic;
    if (np.local_iface_nnz) local_interface_entries(ldu, ir, ic);
    DevBuf<int32_t> addr, counts, tmp, flags, d_ir, d_ic;
    OGL_TRY(addr.alloc(2 * (size_t)F + 2, st));
    OGL_TRY(counts.alloc((size_t)N, st));
    OGL_TRY(tmp.alloc(scan_tmp_len(N), st));
    OGL_TRY(flags.alloc(PATTERN_FLAGS, st));
    if (F) {
        OGL_TRY(reg->stager.h2d(addr.p, ldu.lower_addr, (size_t)F * sizeof(int32_t), st));
        OGL_TRY(reg->stager.h2d(addr.p + F, ldu.upper_addr, (size_t)F * sizeof(int32_t), st));
    }
    if (!ir.empty()) {
        OGL_TRY(d_ir.alloc(ir.size(), st));
        OGL_TRY(d_ic.alloc(ic.size(), st));
        OGL_TRY(reg->stager.h2d(d_ir.p, ir.data(), ir.size() * sizeof(int32_t), st));
        OGL_TRY(reg->stager.h2d(d_ic.p, ic.data(), ic.size() * sizeof(int32_t), st));
    }
    PatternBuild b;
    b.n_rows = N;
    b.n_faces = F;
    b.n_iface = (int32_t)ir.size();
    b.symmetric = np.symmetric ? 1 : 0;
    b.lower_addr = addr.p;
    b.upper_addr = addr.p + F;
    b.if_rows = d_ir.p;
    b.if_cols = d_ic.p;
    b.row_ptrs = d_row_ptrs.p;
    b.cols = d_cols.p;
    b.ldu_mapping = d_ldu_mapping.p;
    b.diag_pos = d_diag_pos.p;
    b.counts = counts.p;
    b.scan_tmp = tmp.p;
    b.flags = flags.p;
    launch_build_pattern(st, b);
    int32_t hf[PATTERN_FLAGS] = {};
    OGL_HIP_CHECK(hipMemcpyAsync(hf, flags.p, sizeof(hf), hipMemcpyDeviceToHost, st));
    OGL_HIP_CHECK(hipStreamSynchronize(st));
    OGL_HIP_CHECK(hipGetLastError());
    if (hf[PAT_FLAG_OUT_OF_RANGE]) return fail(OGL_ERR_INVALID, "a face addresses a cell outside [0,%d)", N);
    if (hf[PAT_FLAG_NONCONFORMING]) return OGL_OK;
    np.local_on_host = false;
    *built = true;
    return OGL_OK;
}

// cols / ldu_mapping / row_ptrs of a device-built pattern for the host code that wants them (numbering
// policy, compressed layout, Ell, block-Jacobi blocks, ISAI pattern, export): once, on demand
int ogl_solver::download_local_pattern(HostPattern &hp)
{
    if (hp.local_on_host) return OGL_OK;
    hipStream_t st = reg->stream;
    const size_t nnz = (size_t)hp.local_nnz;
    hp.row_ptrs.resize((size_t)hp.n_rows + 1);
    hp.cols.resize(nnz);
    hp.ldu_mapping.resize(nnz);
    hp.rows.clear();  // (row of entry k = the r with row_ptrs[r] <= k < row_ptrs[r + 1]; nobody here needs the array)
    OGL_TRY(reg->stager.d2h(hp.row_ptrs.data(), d_row_ptrs.p, hp.row_ptrs.size() * sizeof(int32_t), st));
    if (nnz) {
        OGL_TRY(reg->stager.d2h(hp.cols.data(), d_cols.p, nnz * sizeof(int32_t), st));
        OGL_TRY(reg->stager.d2h(hp.ldu_mapping.data(), d_ldu_mapping.p, nnz * sizeof(int32_t), st));
    }
    hp.local_on_host = true;
    return OGL_OK;
}

// Reverse Cuthill-McKee on the device, level by level (setup_kernels.hip): start node of every component by one
// breadth-first sweep (the node of smallest degree in its last level), then Cuthill-McKee from it -- children of a
// node by ascending (degree, index), a node belonging to the earliest parent of the level before.  The same
// order as rcm_order (tests/test_gpu_device_setup.py).  A level costs four small launches and one 4-byte
// read-back; graphs with very many components or levels (chains) are left to the host.
int ogl_solver::rcm_on_device(const HostPattern &hp, std::vector<ogl_label> &new_id)
{
    new_id.clear();
    hipStream_t st = reg->stream;
    const int32_t N = hp.n_rows;
    if (N < 2) return OGL_OK;
    constexpr int MAX_COMPONENTS = 64, MAX_LEVELS = 60000;
    DevBuf<int32_t> lvl, key, order, scratch, cnt, tmp, nid;
    DevBuf<unsigned long long> cell;
    OGL_TRY(lvl.alloc((size_t)N, st));
    OGL_TRY(key.alloc((size_t)N, st));
    OGL_TRY(order.alloc((size_t)N, st));
    OGL_TRY(scratch.alloc((size_t)N, st));
    OGL_TRY(cnt.alloc((size_t)N + 2, st));
    OGL_TRY(tmp.alloc(scan_tmp_len(N), st));
    OGL_TRY(nid.alloc((size_t)N, st));
    OGL_TRY(cell.alloc(4, st));
    RcmWork w;
    w.n_rows = N;
    w.row_ptrs = d_row_ptrs.p;
    w.cols = d_cols.p;
    w.lvl = lvl.p;
    w.key = key.p;
    w.order = order.p;
    w.scratch = scratch.p;
    w.cnt = cnt.p;
    w.scan_tmp = tmp.p;
    w.cell = cell.p;
    launch_rcm_init(st, w);
    int levels = 0;
    // breadth-first levels from list[begin] (already placed at `begin`); returns the end of the list and the
    // start of its last level
    auto run_levels = [&](int32_t *list, int32_t begin, bool by_degree, int32_t *last_begin, int32_t *end_out) -> int {
        int32_t b = begin, e = begin + 1, level = 0;
        for (;;) {
            launch_rcm_level(st, w, list, b, e, level, by_degree);
            int32_t total = 0;
            OGL_HIP_CHECK(hipMemcpyAsync(&total, cnt.p + (e - b), sizeof(int32_t), hipMemcpyDeviceToHost, st));
            OGL_HIP_CHECK(hipStreamSynchronize(st));
            if (total == 0) break;
            b = e;
            e += total;
            ++level;
            if (++levels > MAX_LEVELS) return 1;
            // thin levels for thousands of levels on end (a chain, a very long duct): a launch-bound crawl
            if (level > 2000 && (int64_t)(e - begin) < 64 * (int64_t)level) return 1;
        }
        *last_begin = b;
        *end_out = e;
        return OGL_OK;
    };
    int32_t filled = 0;
    for (int comp = 0; filled < N; ++comp) {
        if (comp >= MAX_COMPONENTS) return OGL_OK;  // (new_id stays empty: host)
        launch_rcm_find_seed(st, w);
        unsigned long long seed64 = 0;
        OGL_HIP_CHECK(hipMemcpyAsync(&seed64, cell.p, sizeof(seed64), hipMemcpyDeviceToHost, st));
        OGL_HIP_CHECK(hipStreamSynchronize(st));
        if (seed64 >= (unsigned long long)N) return fail(OGL_ERR_STATE, "device RCM lost %d nodes", N - filled);
        const int32_t seed = (int32_t)seed64;
        int32_t rp2[2] = {0, 0};
        OGL_HIP_CHECK(hipMemcpy(rp2, d_row_ptrs.p + seed, sizeof(rp2), hipMemcpyDeviceToHost));
        int32_t start = seed;
        if (rp2[1] - rp2[0] > 1) {  // (isolated cells and chain ends are peripheral already)
            int32_t last_begin = 0, end = 0;
            launch_rcm_start(st, w, scratch.p, 0, seed, 0);
            const int rc = run_levels(scratch.p, 0, /*by_degree*/ false, &last_begin, &end);
            if (rc == 1) return OGL_OK;
            if (rc != OGL_OK) return rc;
            launch_rcm_far_node(st, w, scratch.p, last_begin, end);
            unsigned long long far = 0;
            OGL_HIP_CHECK(hipMemcpyAsync(&far, cell.p + 1, sizeof(far), hipMemcpyDeviceToHost, st));
            OGL_HIP_CHECK(hipStreamSynchronize(st));
            OGL_HIP_CHECK(hipMemcpy(&start, scratch.p + last_begin + (int32_t)(far & 0xffffffffu), sizeof(int32_t),
                                    hipMemcpyDeviceToHost));
            launch_rcm_reset(st, w, scratch.p, end);
        }
        int32_t last_begin = 0, end = 0;
        launch_rcm_start(st, w, order.p, filled, start, 0);
        const int rc = run_levels(order.p, filled, /*by_degree*/ true, &last_begin, &end);
        if (rc == 1) return OGL_OK;
        if (rc != OGL_OK) return rc;
        filled = end;
    }
    launch_rcm_finish(st, w, nid.p);
    new_id.resize((size_t)N);
    OGL_TRY(reg->stager.d2h(new_id.data(), nid.p, (size_t)N * sizeof(int32_t), st));
    OGL_HIP_CHECK(hipGetLastError());
    props["rcmLevels"] = (double)levels;
    return OGL_OK;
}

int ogl_solver::renumber_on_device(HostPattern &hp, const std::vector<ogl_label> &new_id)
{
    hipStream_t st = reg->stream;
    const int32_t N = hp.n_rows;
    const size_t nnz = (size_t)hp.local_nnz;
    DevBuf<int32_t> nid, old_of, rp2, cols2, map2, dpos2, tmp;
    OGL_TRY(nid.alloc((size_t)N, st));
    OGL_TRY(old_of.alloc((size_t)N, st));
    OGL_TRY(rp2.alloc((size_t)N + 1, st));
    OGL_TRY(cols2.alloc(nnz + NNZ_PAD, st));
    OGL_TRY(map2.alloc(nnz + NNZ_PAD, st));
    OGL_TRY(dpos2.alloc(std::max<size_t>(1, (size_t)N), st));
    OGL_TRY(tmp.alloc(scan_tmp_len(N), st));
    OGL_TRY(reg->stager.h2d(nid.p, new_id.data(), (size_t)N * sizeof(int32_t), st));
    RenumberWork w;
    w.n_rows = N;
    w.new_id = nid.p;
    w.row_ptrs = d_row_ptrs.p;
    w.cols = d_cols.p;
    w.map = d_ldu_mapping.p;
    w.old_of = old_of.p;
    w.row_ptrs_out = rp2.p;
    w.cols_out = cols2.p;
    w.map_out = map2.p;
    w.diag_pos_out = dpos2.p;
    w.scan_tmp = tmp.p;
    launch_renumber_pattern(st, w);
    OGL_HIP_CHECK(hipStreamSynchronize(st));
    OGL_HIP_CHECK(hipGetLastError());
    d_row_ptrs.swap(rp2);
    d_cols.swap(cols2);
    d_ldu_mapping.swap(map2);
    d_diag_pos.swap(dpos2);
    hp.local_on_host = false;  // the host arrays are in the old numbering: fetch the new ones
    return download_local_pattern(hp);
}

// Half storage from the device pattern (build_sym_layout's rules, host_matrix.cpp): the distances that occur,
// then mask and map per row.  *done stays false when the pattern does not qualify.
int ogl_solver::build_sym_on_device(const HostPattern &np, SymDistances *sd_out, bool *done)
{
    *done = false;
    hipStream_t st = reg->stream;
    const int32_t N = np.n_rows;
    if (N <= 0) return OGL_OK;
    DevBuf<int32_t> work;  // [table | flags]
    OGL_TRY(work.alloc(SYM_TABLE + SYM_FLAGS, st));
    int32_t init[SYM_TABLE + SYM_FLAGS];
    for (int j = 0; j < SYM_TABLE; ++j) init[j] = SYM_EMPTY;
    for (int j = 0; j < SYM_FLAGS; ++j) init[SYM_TABLE + j] = 0;
    OGL_HIP_CHECK(hipMemcpyAsync(work.p, init, sizeof(init), hipMemcpyHostToDevice, st));
    launch_sym_distances(st, N, d_row_ptrs.p, d_cols.p, work.p, work.p + SYM_TABLE);
    int32_t got[SYM_TABLE + SYM_FLAGS];
    OGL_HIP_CHECK(hipMemcpyAsync(got, work.p, sizeof(got), hipMemcpyDeviceToHost, st));
    OGL_HIP_CHECK(hipStreamSynchronize(st));
    OGL_HIP_CHECK(hipGetLastError());
    if (got[SYM_TABLE + SYM_FLAG_TOO_MANY] || got[SYM_TABLE + SYM_FLAG_UNSORTED_ROW]) return OGL_OK;
    int32_t dist[SYM_TABLE];
    int nd = 0;
    for (int j = 0; j < SYM_TABLE; ++j)
        if (got[j] != SYM_EMPTY) dist[nd++] = got[j];
    std::sort(dist, dist + nd);
    if (nd < 2 || nd > SYM_MAX_OFFSETS || dist[0] != 0 || dist[nd - 1] > INT32_MAX / 2) return OGL_OK;
    const int64_t nc = n_chunks(N);
    const double upper_entries = (double)N + (double)np.upper_nnz;  // diagonal + one entry per face
    if ((double)nd * (double)nc * CHUNK_ROWS > SYM_MAX_PADDING * upper_entries + 8.0 * CHUNK_ROWS) return OGL_OK;
    const size_t mask_len = (size_t)nc * CHUNK_ROWS + 16, map_len = (size_t)nc * nd * CHUNK_ROWS + 2;
    OGL_TRY(d_sym_mask.alloc(mask_len, st));
    OGL_TRY(d_sym_map.alloc(map_len, st));
    OGL_TRY(d_sym_planes.alloc(map_len, st));
    OGL_HIP_CHECK(hipMemsetAsync(d_sym_mask.p, 0, mask_len, st));
    OGL_HIP_CHECK(hipMemsetAsync(d_sym_map.p, 0xFF, map_len * sizeof(int32_t), st));
    SymDistances sd{};
    sd.nd = nd;
    for (int j = 0; j < nd; ++j) sd.d[j] = dist[j];
    OGL_HIP_CHECK(hipMemsetAsync(work.p + SYM_TABLE, 0, SYM_FLAGS * sizeof(int32_t), st));
    launch_sym_fill(st, N, d_row_ptrs.p, d_cols.p, sd, d_sym_mask.p, d_sym_map.p, work.p + SYM_TABLE);
    OGL_HIP_CHECK(hipMemcpyAsync(got, work.p + SYM_TABLE, SYM_FLAGS * sizeof(int32_t), hipMemcpyDeviceToHost, st));
    OGL_HIP_CHECK(hipStreamSynchronize(st));
    OGL_HIP_CHECK(hipGetLastError());
    if (got[SYM_FLAG_TOO_MANY]) return OGL_OK;
    *sd_out = sd;
    *done = true;
    return OGL_OK;
}

int SellDev::build(ogl_label n_rows, const ogl_label *row_ptrs, const ogl_label *cols, Stager &stager,
                   hipStream_t st, bool sort_windows)
{
    ready = false;
    sorted = false;
    rmap.release();
    SellLayout L;
    if (n_rows == 0) return OGL_OK;
    if (!sort_windows) {
        if (!build_sell_layout(n_rows, row_ptrs, cols, L, /*allow_spill*/ false)) return OGL_OK;
    } else {
        // The rows of every wavefront's window (SELL_WAVE_ROWS rows) longest first, in a copy of the pattern that only
        // this layout sees: what choose_numbering does for the system matrix through the numbering itself is done here
        // with a slot order of the layout's own, undone by the kernel (DevSell::rmap) -- W in the CALLER's triangle on a
        // renumbered copy has rows of 1 .. 7 entries next to each other and does not qualify otherwise.
        const int64_t nc = n_chunks(n_rows);
        std::vector<ogl_label> order((size_t)nc * CHUNK_ROWS);
        for (size_t i = 0; i < order.size(); ++i) order[i] = (ogl_label)i;
        auto len = [&](ogl_label r) { return row_ptrs[r + 1] - row_ptrs[r]; };
        bool moved = false;
        for (ogl_label k0 = 0; k0 < n_rows; k0 += SELL_WAVE_ROWS) {
            const auto b = order.begin() + k0, e = order.begin() + std::min<int64_t>(n_rows, (int64_t)k0 + SELL_WAVE_ROWS);
            std::stable_sort(b, e, [&](ogl_label x, ogl_label y) { return len(x) > len(y); });
            for (auto it = b; it != e && !moved; ++it) moved = *it != k0 + (ogl_label)(it - b);
        }
        if (!moved) return OGL_OK;
        std::vector<ogl_label> prp((size_t)n_rows + 1, 0), pc((size_t)row_ptrs[n_rows]), at((size_t)row_ptrs[n_rows]);
        for (ogl_label sr = 0; sr < n_rows; ++sr) prp[(size_t)sr + 1] = prp[(size_t)sr] + len(order[(size_t)sr]);
        for (ogl_label sr = 0; sr < n_rows; ++sr) {
            const ogl_label r = order[(size_t)sr];
            for (ogl_label k = row_ptrs[r], q = prp[(size_t)sr]; k < row_ptrs[r + 1]; ++k, ++q) {
                pc[(size_t)q] = cols[k];
                at[(size_t)q] = k;
            }
        }
        if (!build_sell_layout(n_rows, prp.data(), pc.data(), L, /*allow_spill*/ false)) return OGL_OK;
        for (auto &m : L.map)
            if (m >= 0) m = at[(size_t)m];  // (values are gathered from the CSR values of the pattern itself)
        std::vector<uint16_t> rm(order.size());
        for (size_t i = 0; i < order.size(); ++i) rm[i] = (uint16_t)(order[i] - (ogl_label)(i / CHUNK_ROWS * CHUNK_ROWS));
        OGL_TRY(rmap.alloc(rm.size(), st));
        OGL_TRY(stager.h2d(rmap.p, rm.data(), rm.size() * sizeof(uint16_t), st));
        sorted = true;
    }
    OGL_TRY(chunks.alloc(L.chunks.size(), st));
    OGL_TRY(dict.alloc(L.dict.size(), st));
    OGL_TRY(codes.alloc(L.codes.size(), st));
    OGL_TRY(map.alloc(L.map.size(), st));
    OGL_TRY(vals.alloc(L.map.size(), st));
    OGL_TRY(stager.h2d(chunks.p, L.chunks.data(), L.chunks.size() * sizeof(SellChunk), st));
    OGL_TRY(stager.h2d(dict.p, L.dict.data(), L.dict.size() * sizeof(int32_t), st));
    OGL_TRY(stager.h2d(codes.p, L.codes.data(), L.codes.size(), st));
    OGL_TRY(stager.h2d(map.p, L.map.data(), L.map.size() * sizeof(int32_t), st));
    slots = L.n_slots;
    read_slots = L.read_slots;
    ready = true;
    return OGL_OK;
}

// Once per sparsity pattern: derive the compressed layout on the host; sell_map (like ell_map)
// refreshes the values from the permuted CSR values on the device.
int ogl_solver::build_sell(SellLayout *pre, bool pre_qualifies)
{
    hipStream_t st = reg->stream;
    SellLayout own;
    SellLayout &L = pre ? *pre : own;
    n_spill = n_spill_rows = 0;
    if (!pre) OGL_TRY(download_local_pattern(pat));
    const bool ok = pre ? pre_qualifies
                        : (pat.n_rows > 0 &&
                           build_sell_layout(pat.n_rows, pat.row_ptrs.data(), pat.cols.data(), own));
    if (pat.n_rows == 0 || !ok) {
        sell_state = -1;
        return OGL_OK;
    }
    OGL_TRY(d_sell_chunks.alloc(L.chunks.size(), st));
    OGL_TRY(d_sell_dict.alloc(L.dict.size(), st));
    OGL_TRY(d_sell_codes.alloc(L.codes.size(), st));
    OGL_TRY(d_sell_map.alloc(L.map.size(), st));
    OGL_TRY(d_sell_vals.alloc(L.map.size(), st));
    OGL_TRY(reg->stager.h2d(d_sell_chunks.p, L.chunks.data(), L.chunks.size() * sizeof(SellChunk), st));
    OGL_TRY(reg->stager.h2d(d_sell_dict.p, L.dict.data(), L.dict.size() * sizeof(int32_t), st));
    OGL_TRY(reg->stager.h2d(d_sell_codes.p, L.codes.data(), L.codes.size(), st));
    OGL_TRY(reg->stager.h2d(d_sell_map.p, L.map.data(), L.map.size() * sizeof(int32_t), st));
    // spill: tails of the rows longer than their chunk's cap (row-sorted), added by a second pass
    n_spill_rows = (int32_t)L.spill_rows.size();
    n_spill = (int32_t)L.spill_cols.size();
    if (n_spill) {
        OGL_TRY(d_spill_rows.alloc(L.spill_rows.size(), st));
        OGL_TRY(d_spill_ptrs.alloc(L.spill_ptrs.size(), st));
        OGL_TRY(d_spill_cols.alloc(L.spill_cols.size(), st));
        OGL_TRY(d_spill_map.alloc(L.spill_map.size() + NNZ_PAD, st));
        OGL_TRY(d_spill_vals.alloc(L.spill_cols.size() + NNZ_PAD, st));
        OGL_TRY(d_spill_chunks.alloc(L.spill_chunk_ptr.size(), st));
        OGL_TRY(reg->stager.h2d(d_spill_rows.p, L.spill_rows.data(), L.spill_rows.size() * sizeof(int32_t), st));
        OGL_TRY(reg->stager.h2d(d_spill_ptrs.p, L.spill_ptrs.data(), L.spill_ptrs.size() * sizeof(int32_t), st));
        OGL_TRY(reg->stager.h2d(d_spill_cols.p, L.spill_cols.data(), L.spill_cols.size() * sizeof(int32_t), st));
        OGL_TRY(reg->stager.h2d(d_spill_map.p, L.spill_map.data(), L.spill_map.size() * sizeof(int32_t), st));
        OGL_TRY(reg->stager.h2d(d_spill_chunks.p, L.spill_chunk_ptr.data(), L.spill_chunk_ptr.size() * sizeof(int32_t), st));
    }
    props["sellSpilledEntries"] = (double)n_spill;
    sell_slots = L.n_slots;
    sell_state = 1;
    sell_irregular = L.n_delta16 + L.n_col32 > 0;
    // a banded pattern (1-byte codes throughout: a structured mesh): the largest offset any chunk's table holds = the
    // band the workgroup order below is built for
    sell_band_rows = 0;
    if (!sell_irregular)
        for (int32_t d : L.dict)
            if (d != SELL_PAD_OFFSET) sell_band_rows = std::max<int64_t>(sell_band_rows, std::abs((int64_t)d));
    sell_tuned = 0;
    // bytes one SpMV reads of this layout (bench.py's moved-bytes model): the value planes and codes
    // up to every wavefront's own width (planes beyond it are allocated, not read), headers, tables
    const double read_frac = L.n_slots ? (double)L.read_slots / (double)L.n_slots : 1.0;
    props["sellMatrixBytes"] = 8.0 * (double)L.read_slots + read_frac * (double)(L.codes.size() - 16) +
                               (double)(L.chunks.size() * sizeof(SellChunk)) + 4.0 * (double)L.dict.size() +
                               16.0 * (double)L.spill_cols.size();  // spilled entries: value + column + their share of row data
    sell_bytes = props["sellMatrixBytes"];
    props["sellReadSlots"] = (double)L.read_slots;
    props["sellAllocatedSlots"] = (double)L.n_slots;
    props["sellChunksDelta16"] = (double)L.n_delta16;
    props["sellChunksCol32"] = (double)L.n_col32;
    return OGL_OK;
}

DevHalo ogl_solver::halo() const
{
    DevHalo H;
    H.n_boundary_rows = (int32_t)boundary_rows.size();
    H.boundary_rows = d_boundary_rows.p;
    H.entry_ptrs = d_boundary_ptrs.p;
    H.cols = d_nl_cols.p;
    H.vals = d_nl_vals.p;
    H.n_send = (int32_t)pat.send_idxs.size();
    H.send_idxs = d_send_idxs.p;
    return H;
}

int ogl_solver::upload_vec(DevBuf<double> &dst, const double *src)
{
    const size_t n = (size_t)pat.n_rows;
    if (n == 0) return OGL_OK;
    if (!src) {
        OGL_HIP_CHECK(hipMemsetAsync(dst.p, 0, n * sizeof(double), reg->stream));
        return OGL_OK;
    }
    return upload_rows(dst.p, src);
}

int ogl_solver::upload_rows(double *dst, const double *src)
{
    const size_t bytes = (size_t)pat.n_rows * sizeof(double);
    if (!pat.renumbered()) return reg->stager.h2d(dst, src, bytes, reg->stream);
    OGL_TRY(reg->stager.h2d(d_perm_tmp.p, src, bytes, reg->stream));
    launch_permute_scatter(reg->stream, pat.n_rows, d_new_id.p, d_perm_tmp.p, dst);
    return OGL_OK;
}

int ogl_solver::download_rows(double *dst, const double *src)
{
    const size_t bytes = (size_t)pat.n_rows * sizeof(double);
    if (!pat.renumbered()) return reg->stager.d2h(dst, src, bytes, reg->stream);
    launch_permute_gather(reg->stream, pat.n_rows, d_new_id.p, src, d_perm_tmp.p);
    return reg->stager.d2h(dst, d_perm_tmp.p, bytes, reg->stream);
}


// Which kernel runs the in-loop SpMV of a pattern with irregular chunks: measured, once per pattern.  Both
// read the same matrix and give the same bits (y and the fused dot partials), so this is a speed choice
// only and ranks are free to differ.  Work vectors p (input, zeroed: the time does not depend on the
// values) and q (output) are free between solves.
int ogl_solver::tune_spmv_layout()
{
    hipStream_t st = reg->stream;
    EventPair ev;
    OGL_HIP_CHECK(ev_create(&ev[0]));
    OGL_HIP_CHECK(ev_create(&ev[1]));
    OGL_HIP_CHECK(hipMemsetAsync(d_p.p, 0, ((size_t)pat.n_rows + 2) * sizeof(double), st));
    SpmvDots dots;
    dots.with = d_p.p;
    dots.part = d_part0.p;
    constexpr int WARM = 2, TIMED = 5;
    // [0] CSR-stream, [1] compressed chunked ELL, [2] CSR-stream with packed columns
    const bool have[3] = {true, sell_state == 1 && !sell_values_stale, s21_state == 1};
    float best[3] = {1e30f, 1e30f, 1e30f};
    for (int round = 0; round < WARM + TIMED; ++round)
        for (int which = 0; which < 3; ++which) {
            if (!have[which]) continue;
            OGL_HIP_CHECK(hipEventRecord(ev[0], st));
            if (which == 1) {
                launch_spmv_sell(st, sell(), SPMV_PLAIN, d_p.p, nullptr, d_q.p, dots, nullptr);
            } else {
                s21_use = which == 2;
                launch_spmv(st, csr(), SPMV_PLAIN, d_p.p, nullptr, d_q.p, dots, nullptr);
            }
            OGL_HIP_CHECK(hipEventRecord(ev[1], st));
            OGL_HIP_CHECK(hipEventSynchronize(ev[1]));
            float ms = 0;
            OGL_HIP_CHECK(hipEventElapsedTime(&ms, ev[0], ev[1]));
            if (round >= WARM) best[which] = std::min(best[which], ms);
        }
    OGL_HIP_CHECK(hipGetLastError());
    int winner = 0;
    for (int which = 1; which < 3; ++which)
        if (have[which] && best[which] <= best[winner]) winner = which;
    // (property spmvForceLayout 0 | 1 | 2: CSR-stream | compressed | packed columns whatever the timing says -- how the
    //  parity tests reach a layout on a pattern where another one wins)
    const int forced = (int)prop("spmvForceLayout", -1.0);
    if (forced >= 0 && forced < 3 && have[forced]) winner = forced;
    layout_tuned = true;
    sell_tuned = winner == 1 ? 1 : -1;
    s21_use = winner == 2;
    props["spmvTunedCsrUs"] = 1e3 * best[0];
    if (have[1]) props["spmvTunedSellUs"] = 1e3 * best[1];
    if (have[2]) props["spmvTunedCsr21Us"] = 1e3 * best[2];
    OGL_HIP_CHECK(hipStreamSynchronize(st));
    if (have[1] && sell_tuned < 0) {  // the compressed copy is of no use for this pattern: no refreshes, no memory
        for (auto *b : {&d_sell_dict, &d_sell_map, &d_spill_rows, &d_spill_ptrs, &d_spill_cols, &d_spill_map,
                        &d_spill_chunks})
            b->release();
        d_sell_chunks.release();
        d_sell_codes.release();
        d_sell_vals.release();
        d_spill_vals.release();
        n_spill = n_spill_rows = 0;
        sell_state = -1;
    }
    if (have[2] && !s21_use) {
        d_s21_chunks.release();
        d_s21_codes.release();
        d_s21_far_idx.release();
        d_s21_far_col.release();
        s21_state = -1;
    }
    return OGL_OK;
}

int ogl_solver::ensure_vectors()
{
    hipStream_t st = reg->stream;
    // +2 so that a trailing double2 access of the last (odd) row stays inside the allocation
    const size_t n = (size_t)pat.n_rows + 2;
    const size_t nc = (size_t)n_chunks(pat.n_rows) + 1;
    OGL_TRY(d_x.alloc(n, st));
    OGL_TRY(d_b.alloc(n, st));
    OGL_TRY(d_r.alloc(n, st));
    OGL_TRY(d_p.alloc(n, st));
    OGL_TRY(d_q.alloc(n, st));
    OGL_TRY(d_w.alloc(n, st));
    OGL_TRY(d_part0.alloc(nc, st));
    OGL_TRY(d_part1.alloc(nc, st));
    OGL_TRY(d_part2.alloc(nc, st));
    OGL_TRY(d_scal.alloc(2, st));  // (two slots: the fused-finaliser kernels of small systems ping-pong between them)
    if (!h_scal) {
        OGL_HIP_CHECK(ledger::pinned_malloc(reinterpret_cast<void **>(&h_scal), 2 * sizeof(DevScalars)));
        for (auto &e : poll_ev) OGL_HIP_CHECK(ev_create(&e, hipEventDisableTiming));
    }
    return OGL_OK;
}

bool ogl_solver::saw_addressing(const ogl_ldu_view &ldu) const
{
    if (seen_lower_addr != ldu.lower_addr || seen_upper_addr != ldu.upper_addr || seen_faces != ldu.n_faces ||
        (ogl_label)seen_iface_cells.size() != ldu.n_interfaces)
        return false;
    for (ogl_label i = 0; i < ldu.n_interfaces; ++i) {
        const SeenIface &a = seen_iface_cells[(size_t)i];
        const ogl_interface &b = ldu.interfaces[i];
        if (a.face_cells != b.face_cells || a.size != b.size || a.kind != b.kind || a.neighb_proc != b.neighb_proc ||
            a.neighb_patch != b.neighb_patch)
            return false;
    }
    return true;
}

// checksum over (at most) 4096 evenly spaced entries of each off-diagonal array plus their last ones (bit patterns):
// what ogl_solver_set_matrix_like compares before it trusts a donor's device copy
static uint64_t offdiag_sample(const double *upper, const double *lower, int64_t F)
{
    uint64_t h = 1469598103934665603ull;
    auto mix = [&h](const double *a, int64_t i) {
        uint64_t w;
        std::memcpy(&w, a + i, sizeof(w));
        h = (h ^ w) * 1099511628211ull;
    };
    if (F <= 0 || !upper) return h;
    const int64_t step = std::max<int64_t>(1, F / 4096);
    for (int64_t i = 0; i < F; i += step) {
        mix(upper, i);
        if (lower) mix(lower, i);
    }
    mix(upper, F - 1);
    if (lower) mix(lower, F - 1);
    return h;
}

// ------------------------------------------------------------------------------------------
// HostMatrixWrapper: pattern once, coefficients every call (HostMatrix.C:15-96)
// ------------------------------------------------------------------------------------------
int ogl_solver::set_matrix(const ogl_ldu_view &ldu)
{
    OGL_HIP_CHECK(hipSetDevice(reg->device));
    hipStream_t st = reg->stream;
    TraceRange trace("update_matrix", field);
    const double t0 = now_ms();
    // (what a later sibling may take over from THIS call is recorded at its successful end only: a call that fails part-way
    //  leaves nothing to be trusted)
    seen_lower_addr = seen_upper_addr = nullptr;
    seen_faces = -1;
    seen_iface_cells.clear();
    const bool try_sell = cfg.matrix_format != OGL_FORMAT_ELL && cfg.compress_indices;
    // ---- coefficients (update_local_matrix_data :592-705) ----
    // MatrixInitFunctor::update only overwrites the matrix values when updateSysMatrix is set
    // (CsrMatrixWrapper.H:259); the fresh coefficients have no other consumer, so the transfer
    // is skipped altogether in that case.
    auto upload_coefficients = [&]() -> int {
        const int32_t N = pat.n_rows, F = pat.upper_nnz;
        const size_t nnz = (size_t)pat.local_nnz;
        std::vector<double> iface;
        if (pat.local_iface_nnz) {
            iface.resize(pat.local_iface_nnz);
            collect_interface_coeffs(ldu, true, iface.data());
        }
        if (cfg.reorder_on_host) {  // :608-633, scaling applied by the host update functions
            offdiag_valid = false;  // (d_source is not filled on this path: nothing for a sibling to take)
            props["offDiagReused"] = 0.0;
            OGL_TRY(download_local_pattern(pat));
            std::vector<double> sorted(nnz);
            if (pat.local_iface_nnz) {
                if (pat.symmetric)
                    ogl_host_symmetric_update_w_interface(pat.local_nnz, N, F, pat.ldu_mapping.data(),
                                                          cfg.scaling, ldu.diag, ldu.upper,
                                                          iface.data(), sorted.data());
                else
                    ogl_host_non_symmetric_update_w_interface(
                        pat.local_nnz, N, F, pat.ldu_mapping.data(), cfg.scaling, ldu.diag,
                        ldu.upper, ldu.lower, iface.data(), sorted.data());
            } else if (pat.symmetric) {
                ogl_host_symmetric_update(pat.local_nnz, F, pat.ldu_mapping.data(), cfg.scaling,
                                          ldu.diag, ldu.upper, sorted.data());
            } else {
                ogl_host_non_symmetric_update(pat.local_nnz, F, pat.ldu_mapping.data(), cfg.scaling,
                                              ldu.diag, ldu.upper, ldu.lower, sorted.data());
            }
            OGL_TRY(reg->stager.h2d(d_vals.p, sorted.data(), nnz * sizeof(double), st));
        } else {  // :634-704 -- H2D into the unsorted slots, then the device permutation (K9)
            double *src = d_source.p;
            // (ogl_solver_set_matrix_like: a sibling component's device copy of the same upper / lower arrays)
            const ogl_solver *d = share_from;
            const uint64_t sum = offdiag_sample(ldu.upper, pat.symmetric ? nullptr : ldu.lower, F);
            const bool reuse = d && d->offdiag_valid && d->matrix_set && !d->cfg.reorder_on_host && d->have_pattern &&
                               d->pat.fingerprint == pat.fingerprint && d->pat.upper_nnz == F && d->pat.n_rows == N &&
                               d->pat.symmetric == pat.symmetric && d->offdiag_upper == ldu.upper &&
                               (pat.symmetric || d->offdiag_lower == ldu.lower) && d->offdiag_sum == sum && F > 0;
            if (reuse) {
                OGL_HIP_CHECK(hipMemcpyAsync(src, d->d_source.p, (size_t)F * (pat.symmetric ? 1 : 2) * sizeof(double),
                                             hipMemcpyDeviceToDevice, st));
            } else {
                OGL_TRY(reg->stager.h2d(src, ldu.upper, (size_t)F * sizeof(double), st));          // :644-650
                if (!pat.symmetric)
                    OGL_TRY(reg->stager.h2d(src + F, ldu.lower, (size_t)F * sizeof(double), st));  // :653-660
            }
            offdiag_upper = ldu.upper;
            offdiag_lower = pat.symmetric ? nullptr : ldu.lower;
            offdiag_sum = sum;
            offdiag_valid = true;
            props["offDiagReused"] = reuse ? 1.0 : 0.0;
            OGL_TRY(reg->stager.h2d(src + pat.diag_start(), ldu.diag, (size_t)N * sizeof(double),
                                    st));                                                     // :663-669
            if (pat.local_iface_nnz)
                OGL_TRY(reg->stager.h2d(src + pat.diag_start() + N, iface.data(),
                                        iface.size() * sizeof(double), st));                  // :672-682
            launch_gather_coeffs(st, pat.local_nnz, d_ldu_mapping.p, src, d_vals.p);          // :700-703
        }
        // ---- non-local coefficients (:708-732): tiny, permuted on the host ----
        if (pat.non_local_nnz) {
            std::vector<double> cc(pat.non_local_nnz);
            collect_interface_coeffs(ldu, false, cc.data());
            h_nl_vals.resize(pat.non_local_nnz);
            for (int32_t e = 0; e < pat.non_local_nnz; ++e) h_nl_vals[e] = cc[pat.nl_ldu_mapping[e]];
            OGL_TRY(reg->stager.h2d(d_nl_vals.p, h_nl_vals.data(),
                                    h_nl_vals.size() * sizeof(double), st));
        }
        matrix_set = true;
        ell_values_stale = true;
        sell_values_stale = true;
        sym_values_stale = true;
        symx_values_stale = true;
        return OGL_OK;
    };
    // Has the addressing changed?  Counts first (free); then the hash of every face and interface cell,
    // which runs on helper threads WHILE the coefficients of the (presumably unchanged) pattern are
    // staged to the device: the arrays have the right sizes either way, and if the hash disagrees the
    // pattern is rebuilt and the coefficients go up again.
    const bool try_sym = try_sell && cfg.symmetric_half;
    const bool config_same = have_pattern && pat_renumber_mode == cfg.renumber &&
                             !(cfg.renumber != 0 && pat_try_sell != try_sell) && pat_try_sym == try_sym;
    bool first = true, coefficients_done = false;
    int upload_rc = OGL_OK;  // (a failed speculative upload is reported after the ranks have agreed below)
    if (config_same && same_counts(ldu, pat)) {
        // (ogl_solver_set_matrix_like: the sibling hashed these very addressing arrays a moment ago -- same pointers, same
        //  counts -- and this field's pattern carries the fingerprint it found: the 240 MB are not read a second and
        //  third time per time step)
        const bool hashed_by_sibling = share_from && share_from->have_pattern &&
                                       share_from->pat.fingerprint == pat.fingerprint && share_from->saw_addressing(ldu);
        const uint64_t known = pat.fingerprint;
        auto fp = hashed_by_sibling ? std::async(std::launch::deferred, [known] { return known; })
                                    : std::async(std::launch::async, [&ldu] { return addressing_fingerprint(ldu); });
        int rc = OGL_OK;
        if (!matrix_set || cfg.update_sys_matrix || cfg.regenerate) {
            rc = upload_coefficients();
            coefficients_done = rc == OGL_OK;
        }
        first = fp.get() != pat.fingerprint;  // (joined before any return)
        upload_rc = rc;
    }
    if (reg->comm->multi()) {
        // a rebuild is collective once the peer mesh is up (setup_peer_halo): every rank rebuilds
        // when any rank's addressing changed
        OGL_TRY(d_flag.alloc(2, st));
        const double mine = first ? 1.0 : 0.0;
        double any = 0.0;
        OGL_HIP_CHECK(hipMemcpyAsync(d_flag.p, &mine, sizeof(double), hipMemcpyHostToDevice, st));
        OGL_TRY(reg->allreduce(d_flag.p, 1));
        OGL_HIP_CHECK(hipMemcpyAsync(&any, d_flag.p, sizeof(double), hipMemcpyDeviceToHost, st));
        OGL_HIP_CHECK(hipStreamSynchronize(st));
        first = any != 0.0;
    }
    if (upload_rc != OGL_OK) return upload_rc;
    // a rebuild -- this rank's own or one another rank asked for -- re-creates the value arrays: whatever was
    // uploaded speculatively above is gone
    if (first) coefficients_done = false;
    if (first) {  // :79-87
        // Nothing of the old pattern survives from here on (the value arrays are re-created below before the new
        // pattern is validated): a failure inside this block must not leave the old fingerprint in charge of
        // buffers of another size, and a captured turn must not be replayed on the new layouts
        have_pattern = false;
        matrix_set = false;
        drop_cg_graph();
        HostPattern np;
        // Where the pattern is built: on the device from the face addressing (setup_kernels.hip), unless the
        // environment / property says otherwise or the addressing is not conforming; host_matrix.cpp then
        static const bool host_setup_env = std::getenv("OGL_HOST_SETUP") != nullptr;
        const bool device_setup = !host_setup_env && prop("deviceSetup", 1.0) != 0.0;
        OGL_TRY(build_host_pattern_meta(ldu, np, /*check_faces*/ !device_setup));
        // (the hash of the addressing runs on helper threads next to everything below; joined before any return)
        auto fp_new = std::async(std::launch::async, [&ldu] { return addressing_fingerprint(ldu); });
        const size_t nnz = (size_t)np.local_nnz;
        OGL_TRY(d_row_ptrs.alloc((size_t)np.n_rows + 1, st));
        OGL_TRY(d_cols.alloc(nnz + NNZ_PAD, st));
        OGL_TRY(d_vals.alloc(nnz + NNZ_PAD, st));
        OGL_TRY(d_ldu_mapping.alloc(nnz + NNZ_PAD, st));
        OGL_TRY(d_source.alloc((size_t)np.source_len() + NNZ_PAD, st));
        OGL_TRY(d_diag_pos.alloc(std::max<size_t>(1, (size_t)np.n_rows), st));
        bool built_on_device = false;
        if (device_setup) OGL_TRY(build_pattern_on_device(ldu, np, &built_on_device));
        if (!built_on_device) {
            if (device_setup)  // (non-conforming addressing: the device pass has checked the face range already)
                for (ogl_label f = 0; f < np.upper_nnz; ++f)
                    if (ldu.lower_addr[f] < 0 || ldu.lower_addr[f] >= np.n_rows || ldu.upper_addr[f] < 0 ||
                        ldu.upper_addr[f] >= np.n_rows)
                        return fail(OGL_ERR_INVALID, "face %d addresses a cell outside [0,%d)", f, np.n_rows);
            build_local_pattern(ldu, np);
        }
        props["deviceSetup"] = device_setup ? 1.0 : 0.0;
        props["patternBuiltOnDevice"] = built_on_device ? 1.0 : 0.0;
        // A symmetric lduMatrix without same-rank interfaces is tried on the half storage first (banded with
        // at most SYM_MAX_OFFSETS - 1 distances = a structured mesh, whose numbering the policy below would
        // keep anyway): everything stays on the device then
        SymDistances sym_dist{};
        bool sym_on_device = false, sym_tried = false, renumbered_on_device = false;
        // (symmetricHalfWhole_enable 0: an experiment switch -- the per-chunk variant on a pattern the whole-matrix one takes)
        const bool whole_sym = prop("symmetricHalfWhole_enable", 1.0) != 0.0;
        if (built_on_device && whole_sym && try_sym && np.symmetric && np.local_iface_nnz == 0 && cfg.renumber != 1) {
            sym_tried = true;
            OGL_TRY(build_sym_on_device(np, &sym_dist, &sym_on_device));
        }
        // numbering of the device copy (config `renumber`); the compressed layout the policy may
        // have derived on the way is kept for build_sell below
        SellLayout pre_sell;
        bool pre_built = false;
        RenumberReport rep;
        if (sym_on_device) {
            rep.ratio_natural = rep.ratio_used = -1.0;  // (not measured: the pattern never came to the host)
        } else {
            OGL_TRY(download_local_pattern(np));
            // with the pattern on the device the two heavy steps of a renumbering run there (same results)
            NumberingHooks hooks;
            if (built_on_device) {
                hooks.rcm = [&](const HostPattern &hp, std::vector<ogl_label> &nid) {
                    return rcm_on_device(hp, nid) == OGL_OK && !nid.empty();
                };
                hooks.renumber_local = [&](HostPattern &hp, const std::vector<ogl_label> &nid) {
                    renumbered_on_device = renumber_on_device(hp, nid) == OGL_OK;
                    return renumbered_on_device;
                };
            }
            // (cell centres, when the caller passes them: the Hilbert-curve candidate; property renumberCurve 0 = RCM only)
            if (ldu.cell_centres && prop("renumberCurve", 1.0) != 0.0) hooks.centres = ldu.cell_centres;
            OGL_TRY(choose_numbering(np, cfg.renumber, try_sell, &pre_sell, &pre_built, rep, &hooks));
        }
        pat_renumber_mode = cfg.renumber;
        pat_try_sell = try_sell;
        pat_try_sym = try_sym;
        props["rowsSortedByLength"] = rep.sorted_by_length ? 1.0 : 0.0;
        props["gatherSlotSectorRatio"] = rep.slot_ratio;
        props["renumbered"] = rep.applied ? 1.0 : 0.0;
        props["gatherSectorRatioNatural"] = rep.ratio_natural;
        props["gatherSectorRatio"] = rep.ratio_used;
        props["gatherSectorRatioRcm"] = rep.ratio_rcm;
        props["gatherSectorRatioCurve"] = rep.ratio_curve;
        props["renumberedAlongCurve"] = rep.curve_used ? 1.0 : 0.0;
        props["curveFarEntries"] = (double)rep.curve_far_entries;
        np.fingerprint = fp_new.get();
        pat = std::move(np);
        have_pattern = true;
        static std::atomic<uint64_t> pattern_counter{0};  // registries may live on different threads
        pat_id = ++pattern_counter;
        matrix_set = false;
        ell_ready = false;
        sell_state = 0;
        sym_state = 0;
        s21_state = 0;
        s21_use = false;
        symx_state = 0;
        layout_tuned = false;
        d_s21_chunks.release();
        d_s21_codes.release();
        d_s21_far_idx.release();
        d_s21_far_col.release();
        d_band_order.release();  // (the band-aware workgroup order belongs to the pattern it was built for)
        band_order_rows = 0;
        x_resident = b_resident = false;
        props["renumberedOnDevice"] = renumbered_on_device ? 1.0 : 0.0;
        if (!built_on_device || (rep.applied && !renumbered_on_device)) {  // the device does not hold the pattern (in this numbering) yet
            OGL_TRY(reg->stager.h2d(d_row_ptrs.p, pat.row_ptrs.data(),
                                    pat.row_ptrs.size() * sizeof(int32_t), st));
            OGL_TRY(reg->stager.h2d(d_cols.p, pat.cols.data(), nnz * sizeof(int32_t), st));
            OGL_TRY(reg->stager.h2d(d_ldu_mapping.p, pat.ldu_mapping.data(), nnz * sizeof(int32_t), st));
            // Csr::extract_diagonal takes the first (i, i) entry of a row
            std::vector<int32_t> dpos((size_t)pat.n_rows, -1);
            for (int32_t r = 0; r < pat.n_rows; ++r)
                for (int32_t k = pat.row_ptrs[r]; k < pat.row_ptrs[r + 1]; ++k)
                    if (pat.cols[k] == r) {
                        dpos[(size_t)r] = k;
                        break;
                    }
            if (!dpos.empty())
                OGL_TRY(reg->stager.h2d(d_diag_pos.p, dpos.data(), dpos.size() * sizeof(int32_t), st));
        }

        // halo: rows owning non-local entries (row-sorted triplets -> one run per row)
        boundary_rows.clear();
        boundary_ptrs.clear();
        for (int32_t e = 0; e < pat.non_local_nnz; ++e) {
            if (e == 0 || pat.nl_rows[e] != pat.nl_rows[e - 1]) {
                boundary_rows.push_back(pat.nl_rows[e]);
                boundary_ptrs.push_back(e);
            }
        }
        boundary_ptrs.push_back(pat.non_local_nnz);
        const size_t hn = (size_t)pat.non_local_nnz;
        {  // chunks whose fused dot partials must be redone after the non-local part was added
            std::vector<int32_t> bc, bc_ptr;
            for (size_t i = 0; i < boundary_rows.size(); ++i) {
                const int32_t r = boundary_rows[i];
                if (bc.empty() || bc.back() != r / CHUNK_ROWS) {
                    bc.push_back(r / CHUNK_ROWS);
                    bc_ptr.push_back((int32_t)i);
                }
            }
            bc_ptr.push_back((int32_t)boundary_rows.size());
            n_boundary_chunks = (int32_t)bc.size();
            OGL_TRY(d_boundary_chunk_ptr.alloc(bc_ptr.size(), st));
            OGL_TRY(reg->stager.h2d(d_boundary_chunk_ptr.p, bc_ptr.data(), bc_ptr.size() * sizeof(int32_t), st));
            OGL_TRY(d_ticket.alloc(1, st));
            OGL_TRY(d_boundary_chunks.alloc(bc.size(), st));
            if (!bc.empty())
                OGL_TRY(reg->stager.h2d(d_boundary_chunks.p, bc.data(), bc.size() * sizeof(int32_t), st));
        }
        {   // per chunk: its boundary rows (HaloFused) and its send rows (HaloPutFused)
            const size_t nc1 = (size_t)n_chunks(pat.n_rows) + 1;
            std::vector<int32_t> bptr(nc1, 0), sptr(nc1, 0), spos(pat.send_idxs.size());
            for (int32_t r : boundary_rows) ++bptr[(size_t)(r / CHUNK_ROWS) + 1];
            for (int32_t r : pat.send_idxs) ++sptr[(size_t)(r / CHUNK_ROWS) + 1];
            n_put_chunks = 0;
            for (size_t c = 1; c < nc1; ++c) {
                if (sptr[c]) ++n_put_chunks;
                bptr[c] += bptr[c - 1];
                sptr[c] += sptr[c - 1];
            }
            std::vector<int32_t> fill(sptr.begin(), sptr.end() - 1);
            for (size_t j = 0; j < pat.send_idxs.size(); ++j)
                spos[(size_t)fill[(size_t)(pat.send_idxs[j] / CHUNK_ROWS)]++] = (int32_t)j;
            OGL_TRY(d_chunk_bptr.alloc(nc1, st));
            OGL_TRY(d_chunk_sptr.alloc(nc1, st));
            OGL_TRY(d_send_pos.alloc(std::max<size_t>(1, spos.size()), st));
            OGL_TRY(reg->stager.h2d(d_chunk_bptr.p, bptr.data(), nc1 * sizeof(int32_t), st));
            OGL_TRY(reg->stager.h2d(d_chunk_sptr.p, sptr.data(), nc1 * sizeof(int32_t), st));
            if (!spos.empty())
                OGL_TRY(reg->stager.h2d(d_send_pos.p, spos.data(), spos.size() * sizeof(int32_t), st));
        }
        OGL_TRY(d_boundary_rows.alloc(boundary_rows.size(), st));
        OGL_TRY(d_boundary_ptrs.alloc(boundary_ptrs.size(), st));
        OGL_TRY(d_nl_cols.alloc(hn, st));
        OGL_TRY(d_nl_vals.alloc(hn, st));
        OGL_TRY(d_send_idxs.alloc(pat.send_idxs.size(), st));
        OGL_TRY(d_send.alloc(pat.send_idxs.size(), st));
        OGL_TRY(d_recv.alloc(hn, st));
        if (hn) {
            OGL_TRY(reg->stager.h2d(d_boundary_rows.p, boundary_rows.data(),
                                    boundary_rows.size() * sizeof(int32_t), st));
            OGL_TRY(reg->stager.h2d(d_boundary_ptrs.p, boundary_ptrs.data(),
                                    boundary_ptrs.size() * sizeof(int32_t), st));
            OGL_TRY(reg->stager.h2d(d_nl_cols.p, pat.nl_cols.data(), hn * sizeof(int32_t), st));
            OGL_TRY(reg->stager.h2d(d_send_idxs.p, pat.send_idxs.data(),
                                    pat.send_idxs.size() * sizeof(int32_t), st));
        }
        neighbours.assign(pat.target_ids.begin(), pat.target_ids.end());
        counts.assign(pat.target_sizes.begin(), pat.target_sizes.end());
        if (!neighbours.empty() && !reg->comm->multi())
            return fail(OGL_ERR_STATE,
                        "matrix has processor interfaces but the registry has no communicator");
        // the receive side is assumed to mirror the send side (Partition.H:66-67)
        if ((size_t)pat.non_local_nnz != pat.send_idxs.size())
            return fail(OGL_ERR_INVALID, "send/receive sizes differ");
        OGL_TRY(ensure_vectors());
        if (pat.renumbered()) {
            OGL_TRY(d_new_id.alloc((size_t)pat.n_rows, st));
            OGL_TRY(d_perm_tmp.alloc((size_t)pat.n_rows + 2, st));
            OGL_TRY(reg->stager.h2d(d_new_id.p, pat.new_id.data(), (size_t)pat.n_rows * sizeof(int32_t), st));
            // (the inverse: block-Jacobi blocks stay those of the caller's numbering, generate_preconditioner)
            OGL_TRY(d_old_of.alloc((size_t)pat.n_rows, st));
            OGL_TRY(reg->stager.h2d(d_old_of.p, pat.old_of.data(), (size_t)pat.n_rows * sizeof(int32_t), st));
        }
        // a symmetric lduMatrix on a banded pattern keeps the OpenFOAM storage (diagonal + upper); the
        // compressed full-storage copy is then not built at all
        props["symmetricHalf"] = 0.0;
        bool sym_ok = false;
        if (sym_on_device) {
            OGL_TRY(finish_sym(sym_dist.nd, sym_dist.d));
            sym_ok = true;
        } else if (!sym_tried && prop("symmetricHalfWhole_enable", 1.0) != 0.0 && try_sym && pat.symmetric &&
                   pat.local_iface_nnz == 0 && !pat.renumbered()) {
            SymLayout symL;
            if (build_sym_layout(pat.n_rows, pat.row_ptrs.data(), pat.cols.data(), symL)) {
                OGL_TRY(build_sym(symL));
                sym_ok = true;
            }
        }
        props["symmetricHalfPerChunk"] = 0.0;
        if (!sym_ok && try_sym && pat.symmetric && pat.local_iface_nnz == 0 && !pat.renumbered() &&
            prop("symmetricHalfPerChunk_enable", 1.0) != 0.0) {
            // banded only locally (multi-block mesh, refinement shell): per-chunk distances + explicit exceptions
            OGL_TRY(build_symx());
            if (symx_state == 1) {
                sym_ok = true;
                props["symmetricHalfPerChunk"] = 1.0;
            }
        }
        if (symx_state != 1) {
            symx_state = -1;
            for (auto *b : {&d_symx_map, &d_symx_ex_rowptr, &d_symx_ex_cols, &d_symx_ex_map, &d_symx_ex_lrow}) b->release();
            d_symx_chunks.release();
            d_symx_chunks_general.release();
            d_symx_mask.release();
            d_symx_planes.release();
            d_symx_ex_vals.release();
        }
        symx_tune_pending = false;
        if (sym_ok) {
            props["symmetricHalf"] = 1.0;
            // (per-chunk half storage of a system that is not launch-bound: timed once against the compressed full
            //  storage the numbering policy has laid out on the way, after the first values are in)
            symx_tune_pending = symx_state == 1 && cfg.compress_indices == 1 && pat.n_rows >= SPMV_TUNE_MIN_ROWS;
            if (symx_tune_pending) {
                if (pre_built)
                    OGL_TRY(build_sell(&pre_sell, rep.sell_used));
                else
                    OGL_TRY(build_sell());
                symx_tune_pending = sell_state == 1;  // (nothing to time against when the compressed copy does not qualify)
                props["sellMatrixBytes"] = symx_bytes;
            }
            if (!symx_tune_pending) {
                sell_state = -1;
                d_sell_chunks.release();
                d_sell_codes.release();
                d_sell_vals.release();
                d_sell_map.release();
                d_sell_dict.release();
            }
        }
        if (sym_state != 1) {
            sym_state = -1;
            d_sym_mask.release();
            d_sym_map.release();
            d_sym_planes.release();
        }
        if (sym_state != 1 && symx_state != 1 && pre_built && try_sell) OGL_TRY(build_sell(&pre_sell, rep.sell_used));
        OGL_TRY(setup_peer_halo());  // collective when the peer mesh is up (every rank, every pattern)
    }

    if (!coefficients_done && (!matrix_set || cfg.update_sys_matrix || cfg.regenerate)) OGL_TRY(upload_coefficients());
    if (cfg.matrix_format == OGL_FORMAT_ELL) {
        if (!ell_ready) OGL_TRY(build_ell());
        if (ell_values_stale) {
            launch_gather_coeffs_masked(st, (int64_t)ell_width * ell_stride, d_ell_map.p, d_vals.p,
                                        d_ell_vals.p);
            ell_values_stale = false;
        }
    } else if (cfg.compress_indices && sym_state == 1) {
        if (sym_values_stale) {
            launch_gather_coeffs_masked(st, (int64_t)d_sym_map.n - 2, d_sym_map.p, d_vals.p, d_sym_planes.p);
            sym_values_stale = false;
        }
    } else if (cfg.compress_indices && symx_state == 1) {
        if (symx_values_stale) {
            launch_gather_coeffs_masked(st, (int64_t)d_symx_map.n - 2, d_symx_map.p, d_vals.p, d_symx_planes.p);
            const int32_t nex = (int32_t)(d_symx_ex_cols.n - NNZ_PAD);
            if (nex > 0) launch_gather_coeffs(st, nex, d_symx_ex_map.p, d_vals.p, d_symx_ex_vals.p);
            symx_values_stale = false;
        }
        if (symx_tune_pending) {
            symx_tune_pending = false;
            if (sell_state == 1) {
                launch_gather_sell(st, (int32_t)d_sell_chunks.n, d_sell_chunks.p, d_sell_map.p, d_vals.p, d_sell_vals.p);
                if (n_spill) launch_gather_coeffs(st, n_spill, d_spill_map.p, d_vals.p, d_spill_vals.p);
                sell_values_stale = false;
                OGL_TRY(tune_symx());
            }
        }
    } else if (cfg.compress_indices) {
        if (sell_state == 0) OGL_TRY(build_sell());
        if (sell_state == 1 && sell_values_stale) {
            launch_gather_sell(st, (int32_t)d_sell_chunks.n, d_sell_chunks.p, d_sell_map.p, d_vals.p,
                               d_sell_vals.p);
            if (n_spill) launch_gather_coeffs(st, n_spill, d_spill_map.p, d_vals.p, d_spill_vals.p);
            sell_values_stale = false;
        }
        // irregular patterns (16 / 32-bit codes in the chunked ELL, or one that does not qualify for it at all: a
        // polyhedral mesh) of a size where the SpMV is not launch-bound: the CSR-stream kernel gets its packed
        // columns, and the candidates are timed once per pattern
        const bool big = pat.n_rows >= SPMV_TUNE_MIN_ROWS;
        const bool irregular_sell = sell_state == 1 && sell_irregular;
        if (big && s21_state == 0 && (irregular_sell || sell_state == -1)) OGL_TRY(build_stream21());
        if (big && cfg.compress_indices == 1 && !layout_tuned && (irregular_sell || s21_state == 1))
            OGL_TRY(tune_spmv_layout());
        if (cfg.compress_indices == 2) s21_use = s21_state == 1 && sell_state != 1;  // force: no timing
    }
    // which layout the in-loop SpMV runs on: 0 CSR-stream, 1 ELL, 2 index-compressed chunked ELL
    // (2 also for the half storage of a symmetric matrix: property symmetricHalf tells them apart)
    // 3: CSR-stream with packed columns
    {   // Band-aware workgroup order of the compressed / CSR-stream kernels (the half-storage kernels take theirs from
        // their own distances): the chunks of rows r and r +- band run on one XCD, so a strip of x is fetched into one L2
        // instead of three.  Full storage of the 216^3 box, STREAM instantiation: 128.2 -> 124.6 us (0.718 -> 0.739 of
        // peak); the plain CSR-stream kernel measures the same either way (195.7 / 195.8 us) and is left alone.  Property
        // spmvBandRows: the band in rows, 0 = off, -1 (default) = the compressed layout's own largest offset on a
        // banded pattern.
        int64_t band = (int64_t)prop("spmvBandRows", -1.0);
        if (band < 0) band = (cfg.matrix_format != OGL_FORMAT_ELL && use_sell() && !use_sym() && !use_symx()) ? sell_band_rows : 0;
        if (band != band_order_rows) {
            d_band_order.release();
            band_order_rows = band;
            std::vector<int32_t> order;
            if (band > 0) band_block_order(pat.n_rows, band, order);
            if (!order.empty()) {
                OGL_TRY(d_band_order.alloc(order.size(), st));
                OGL_TRY(reg->stager.h2d(d_band_order.p, order.data(), order.size() * sizeof(int32_t), st));
            }
            drop_cg_graph();
        }
    }
    const bool on_csr = cfg.matrix_format != OGL_FORMAT_ELL && !use_sell() && !use_sym() && !use_symx();
    if (!cfg.compress_indices) s21_use = false;
    props["spmvLayout"] = cfg.matrix_format == OGL_FORMAT_ELL ? 1.0 : (!on_csr ? 2.0 : (s21_use && s21_state == 1 ? 3.0 : 0.0));
    {   // ... and which instantiation of its kernel (what a profiler lists; bench.py looks up exactly that one)
        bool stream = false, fast = false;
        if (cfg.matrix_format == OGL_FORMAT_ELL) {
            stream = ell().stream;
        } else if (use_sym()) {
            const DevSym S = sym();
            stream = S.stream;
            fast = S.nd >= 2 && S.d[1] == 1;
            for (int j = 2; j < S.nd; ++j) fast = fast && (S.d[j] % 2 == 0);
            props["spmvSymPlanes"] = (double)S.nd;
        } else if (use_symx()) {
            stream = symx().stream;
            fast = symx().fast;
        } else if (use_sell()) {
            stream = sell().stream;
        } else {
            stream = csr().stream;
        }
        props["spmvStream"] = stream ? 1.0 : 0.0;
        props["spmvSymFast"] = fast ? 1.0 : 0.0;
    }
    OGL_HIP_CHECK(hipStreamSynchronize(st));
    OGL_HIP_CHECK(hipGetLastError());
    // what this call's addressing arrays were (ogl_solver_set_matrix_like: a sibling built on the same arrays right after
    // need not hash them again)
    seen_lower_addr = ldu.lower_addr;
    seen_upper_addr = ldu.upper_addr;
    seen_faces = ldu.n_faces;
    seen_iface_cells.clear();
    for (ogl_label i = 0; i < ldu.n_interfaces; ++i) {
        const ogl_interface &f = ldu.interfaces[i];
        seen_iface_cells.push_back(SeenIface{f.face_cells, f.size, f.kind, f.neighb_proc, f.neighb_patch});
    }
    t_update_matrix_ms = now_ms() - t0;
    return OGL_OK;
}

// ------------------------------------------------------------------------------------------
// Preconditioner::init_preconditioner (Preconditioner.H:353-431) with its caching rules:
//   * nothing stored yet      -> generate, store, counter := caching
//   * stored and counter > 0  -> counter -= 1, use the STORED one
//   * stored and counter == 0 -> counter := caching, generate a fresh one for this solve only;
//                                the stored object is not replaced (:411-413)
// The store is registry-wide (one key for all fields, :357), as in the reference.
// ------------------------------------------------------------------------------------------
int ogl_solver::generate_preconditioner(PrecondData &P)
{
    hipStream_t st = reg->stream;
    const size_t n = (size_t)pat.n_rows;
    // structures of a renumbered device copy (block-Jacobi blocks, ISAI(spd)'s triangle) in the CALLER's numbering:
    // the reference's operator (property precondCallerNumbering 0 = the backend's numbering, for A/B)
    const bool caller_numbering = prop("precondCallerNumbering", 1.0) != 0.0;
    const bool through_perm = pat.renumbered() && caller_numbering;
    if (P.struct_caller_numbering != caller_numbering) P.struct_pat_id = 0;  // (the switch was flipped: rebuild)
    P.struct_caller_numbering = caller_numbering;
    if (cfg.preconditioner == OGL_PRECOND_ISAI || cfg.preconditioner == OGL_PRECOND_GISAI) {
        // Isai<spd|general> with sparsity_power 1 and skip_sorting (Preconditioner.H:225-258).
        // Pattern of W on the host (tril(A) for spd, A for general), its transpose + map for spd,
        // values on the device (one dense solve per row).
        const bool spd = cfg.preconditioner == OGL_PRECOND_ISAI;
        const int32_t N = pat.n_rows;
        const int kind = spd ? 3 : 4;
        if (!P.has_structure(pat_id, kind, cfg.sparsity_power)) {
            P.struct_pat_id = 0;
            std::vector<int32_t> wrp, wc;
            ogl_label wide = -1;
            OGL_TRY(download_local_pattern(pat));
            if (!isai_pattern(pat, spd, cfg.sparsity_power, MAX_ISAI_HUGE_ROW, wrp, wc, wide, caller_numbering))
                return fail(OGL_ERR_UNSUPPORTED,
                            "preconditioner %s, sparsityPower %d: row %d of the approximate inverse has more than %d "
                            "pattern entries (lower sparsityPower)", spd ? "ISAI" : "GISAI", cfg.sparsity_power, wide,
                            MAX_ISAI_HUGE_ROW);
            int32_t max_row = 0;
            // rows solved by one wavefront each (33 .. 64 entries) / by one workgroup each in global scratch
            // (65 .. 2048 = MAX_ISAI_HUGE_ROW); the others: one thread
            std::vector<int32_t> wide_rows, huge_rows;
            std::vector<int64_t> huge_off;
            P.huge_batches.assign(1, 0);
            // scratch for the dense systems of the huge rows: batches of rows within a budget (property, bytes)
            const int64_t budget = (int64_t)(prop("isaiScratchBytes", 2147483648.0) / sizeof(double));
            int64_t used = 0, most = 0;
            for (int32_t r = 0; r < N; ++r) {
                const int32_t len = wrp[(size_t)r + 1] - wrp[(size_t)r];
                max_row = std::max(max_row, len);
                if (len > MAX_ISAI_ROW) {
                    const int64_t need = (int64_t)len * len;
                    if (used + need > budget && used > 0) {
                        P.huge_batches.push_back((int32_t)huge_rows.size());
                        used = 0;
                    }
                    huge_rows.push_back(r);
                    huge_off.push_back(used);
                    used += need;
                    most = std::max(most, used);
                } else if (len > ISAI_THREAD_ROW) {
                    wide_rows.push_back(r);
                }
            }
            P.huge_batches.push_back((int32_t)huge_rows.size());
            P.n_wide_rows = (int32_t)wide_rows.size();
            P.n_huge_rows = (int32_t)huge_rows.size();
            OGL_TRY(P.wide_rows.alloc(std::max<size_t>(1, wide_rows.size()), st));
            if (!wide_rows.empty())
                OGL_TRY(reg->stager.h2d(P.wide_rows.p, wide_rows.data(), wide_rows.size() * sizeof(int32_t), st));
            OGL_TRY(P.huge_rows.alloc(std::max<size_t>(1, huge_rows.size()), st));
            OGL_TRY(P.huge_off.alloc(std::max<size_t>(1, huge_off.size()), st));
            if (!huge_rows.empty()) {
                OGL_TRY(reg->stager.h2d(P.huge_rows.p, huge_rows.data(), huge_rows.size() * sizeof(int32_t), st));
                OGL_TRY(reg->stager.h2d(P.huge_off.p, huge_off.data(), huge_off.size() * sizeof(int64_t), st));
            }
            P.huge_scratch_len = most;
            P.huge_scratch.release();
            props["isaiWideRows"] = (double)wide_rows.size();
            props["isaiHugeRows"] = (double)huge_rows.size();
            const size_t wn = wc.size();
            // (property isaiSortRows 0: W / W^T on the compressed layout only where their own row order qualifies, A/B)
            const bool sort_w = prop("isaiSortRows", 1.0) != 0.0;
            OGL_TRY(P.w_row_ptrs.alloc((size_t)N + 1, st));
            OGL_TRY(P.w_cols.alloc(wn + NNZ_PAD, st));
            OGL_TRY(P.w_vals.alloc(wn + NNZ_PAD, st));
            OGL_TRY(reg->stager.h2d(P.w_row_ptrs.p, wrp.data(), wrp.size() * sizeof(int32_t), st));
            OGL_TRY(reg->stager.h2d(P.w_cols.p, wc.data(), wn * sizeof(int32_t), st));
            if (spd) {  // W^T: counting transpose keeps every row sorted by column
                std::vector<int32_t> trp((size_t)N + 1, 0), tc(wn), tmap(wn);
                for (size_t k = 0; k < wn; ++k) ++trp[wc[k] + 1];
                for (int32_t r = 0; r < N; ++r) trp[r + 1] += trp[r];
                std::vector<int32_t> fill(trp.begin(), trp.end() - 1);
                for (int32_t r = 0; r < N; ++r)
                    for (int32_t k = wrp[r]; k < wrp[r + 1]; ++k) {
                        const int32_t e = fill[wc[k]]++;
                        tc[e] = r;
                        tmap[e] = k;
                    }
                OGL_TRY(P.wt_row_ptrs.alloc((size_t)N + 1, st));
                OGL_TRY(P.wt_cols.alloc(wn + NNZ_PAD, st));
                OGL_TRY(P.wt_map.alloc(wn + NNZ_PAD, st));
                OGL_TRY(P.wt_vals.alloc(wn + NNZ_PAD, st));
                OGL_TRY(reg->stager.h2d(P.wt_row_ptrs.p, trp.data(), trp.size() * sizeof(int32_t), st));
                OGL_TRY(reg->stager.h2d(P.wt_cols.p, tc.data(), wn * sizeof(int32_t), st));
                OGL_TRY(reg->stager.h2d(P.wt_map.p, tmap.data(), wn * sizeof(int32_t), st));
                if (cfg.compress_indices) {
                    OGL_TRY(P.wt_sell.build(N, trp.data(), tc.data(), reg->stager, st));
                    if (!P.wt_sell.ready && sort_w) OGL_TRY(P.wt_sell.build(N, trp.data(), tc.data(), reg->stager, st, true));
                }
            }
            if (!spd || !cfg.compress_indices) P.wt_sell.ready = false;
            P.w_sell.ready = false;
            if (cfg.compress_indices) {
                OGL_TRY(P.w_sell.build(N, wrp.data(), wc.data(), reg->stager, st));
                if (!P.w_sell.ready && sort_w) OGL_TRY(P.w_sell.build(N, wrp.data(), wc.data(), reg->stager, st, true));
            }
            props["isaiWSorted"] = P.w_sell.sorted ? 1.0 : 0.0;
            props["isaiWtSorted"] = P.wt_sell.sorted ? 1.0 : 0.0;
            P.w_nnz = (int32_t)wn;
            P.w_max_row = max_row;
            props["isaiWCompressed"] = P.w_sell.ready ? 1.0 : 0.0;
            props["isaiWtCompressed"] = P.wt_sell.ready ? 1.0 : 0.0;
            P.struct_pat_id = pat_id;
            P.struct_kind = kind;
            P.struct_stride = cfg.sparsity_power;
        }
        launch_isai_generate(st, csr(), spd ? 1 : 0, P.w_row_ptrs.p, P.w_cols.p, P.w_vals.p,
                             P.w_max_row, P.wide_rows.p, P.n_wide_rows);
        // the dense systems of the huge rows live in a scratch of up to isaiScratchBytes that only this generation
        // needs: allocated here, released below (a field's own preconditioner plus the registry-wide cached one would
        // otherwise sit on 2 GiB each for the whole run)
        if (P.n_huge_rows > 0) OGL_TRY(P.huge_scratch.alloc((size_t)P.huge_scratch_len, st));
        for (size_t bt = 0; bt + 1 < P.huge_batches.size(); ++bt)  // (stream order: a batch reuses the scratch)
            launch_isai_generate_huge(st, csr(), spd ? 1 : 0, P.w_row_ptrs.p, P.w_cols.p, P.w_vals.p, P.huge_rows.p,
                                      P.huge_off.p, P.huge_batches[bt], P.huge_batches[bt + 1] - P.huge_batches[bt],
                                      P.huge_scratch.p);
        if (P.n_huge_rows > 0 && prop("isaiKeepScratch", 0.0) == 0.0) {
            OGL_HIP_CHECK(hipStreamSynchronize(st));
            P.huge_scratch.release();
        }
        if (spd) launch_gather_coeffs(st, P.w_nnz, P.wt_map.p, P.w_vals.p, P.wt_vals.p);
        P.w_sell.refresh(P.w_vals.p, st);
        if (spd) P.wt_sell.refresh(P.wt_vals.p, st);
        P.kind = spd ? 3 : 4;
        P.stride = cfg.sparsity_power;
    } else if (cfg.max_block_size == 1) {  // scalar Jacobi: 1 / diag
        OGL_TRY(P.values.alloc(n + 2, st));
        launch_jacobi_generate_pos(st, csr(), d_diag_pos.p, P.values.p);
        P.kind = 1;
        P.stride = 0;
    } else {
        // Jacobi factory with max_block_size = maxBlockSize, skip_sorting (Preconditioner.H:100-104)
        const size_t k = (size_t)cfg.max_block_size;
        if (!P.has_structure(pat_id, 2, cfg.max_block_size)) {
            P.struct_pat_id = 0;
            std::vector<int32_t> ptrs, row_block;
            OGL_TRY(download_local_pattern(pat));
            find_jacobi_blocks(pat, cfg.max_block_size, ptrs, row_block, caller_numbering);
            P.n_blocks = (int32_t)ptrs.size() - 1;
            P.uniform_blocks = true;
            for (int32_t b = 0; b + 1 < P.n_blocks; ++b)
                P.uniform_blocks = P.uniform_blocks && ptrs[(size_t)b + 1] - ptrs[(size_t)b] == cfg.max_block_size;
            if (P.n_blocks > 0)
                P.uniform_blocks = P.uniform_blocks && ptrs[(size_t)P.n_blocks - 1] == (P.n_blocks - 1) * cfg.max_block_size;
            OGL_TRY(P.block_ptrs.alloc(ptrs.size(), st));
            OGL_TRY(P.row_block.alloc(std::max<size_t>(1, row_block.size()), st));
            OGL_TRY(P.values.alloc(std::max<size_t>(1, (size_t)P.n_blocks * k * k), st));
            OGL_TRY(reg->stager.h2d(P.block_ptrs.p, ptrs.data(), ptrs.size() * sizeof(int32_t), st));
            OGL_TRY(reg->stager.h2d(P.row_block.p, row_block.data(),
                                    row_block.size() * sizeof(int32_t), st));
            P.struct_pat_id = pat_id;
            P.struct_kind = 2;
            P.struct_stride = cfg.max_block_size;
        }
        DevBlockJacobi J;
        J.n_rows = pat.n_rows;
        J.n_blocks = P.n_blocks;
        J.stride = cfg.max_block_size;
        J.block_ptrs = P.block_ptrs.p;
        J.row_block = P.row_block.p;
        J.blocks = P.values.p;
        if (through_perm) {  // blocks of the caller's numbering, reached through the permutation
            J.rows = d_new_id.p;
            J.pos = d_old_of.p;
            // staged apply (default): blocks stay block-major in the caller's order, the vectors are carried there and
            // back by two gather kernels -- one scattered access per row instead of one per block member (128^3
            // shuffled, BJ(4): 316 us per turn with the direct apply on block rows stored by device row)
            J.by_device_row = prop("bjStagedApply", 1.0) != 0.0 ? 0 : 1;
        }
        P.by_device_row = J.by_device_row != 0;
        P.through_perm = through_perm;
        P.perm_pat_id = through_perm ? pat_id : 0;

        launch_bj_generate(st, csr(), J);
        P.kind = 2;
        P.stride = cfg.max_block_size;
    }
    P.n_rows = n;
    P.gen_pat_id = pat_id;
    P.gen_device_numbering =
        pat.renumbered() && !(P.kind == 2 && P.through_perm && !P.by_device_row);  // (see PrecondData::foreign_to)
    return OGL_OK;
}

void ogl_solver::apply_preconditioner(const double *in, double *out, const DevScalars *gate,
                                      double *dot_part)
{
    hipStream_t st = reg->stream;
    // the last kernel of the apply also leaves the partials of in . out
    SpmvDots last{};
    if (dot_part) {
        last.with = in;
        last.part = dot_part;
    }
    if (precond_data->kind == 3 || precond_data->kind == 4) {  // ISAI: one or two SpMVs
        DevCsr W;
        W.n_rows = pat.n_rows;
        W.nnz = precond_data->w_nnz;
        W.row_ptrs = precond_data->w_row_ptrs.p;
        W.cols = precond_data->w_cols.p;
        W.vals = precond_data->w_vals.p;
        const bool w_sell = cfg.compress_indices && precond_data->w_sell.ready;
        const bool general = precond_data->kind == 4;
        double *w_out = general ? out : d_isai_tmp.p;
        // (W / W^T are streamed past the caches exactly when the system matrix is: one working set, one policy)
        const bool stream_w = props.count("spmvStream") && props.at("spmvStream") == 1.0;
        W.stream = stream_w;
        if (w_sell)
            launch_spmv_sell(st, precond_data->w_sell.view(pat.n_rows, stream_w), SPMV_PLAIN, in, nullptr, w_out,
                             general ? last : SpmvDots{}, gate);
        else
            launch_spmv(st, W, SPMV_PLAIN, in, nullptr, w_out, general ? last : SpmvDots{}, gate);
        if (general) return;
        DevCsr WT = W;
        WT.row_ptrs = precond_data->wt_row_ptrs.p;
        WT.cols = precond_data->wt_cols.p;
        WT.vals = precond_data->wt_vals.p;
        if (cfg.compress_indices && precond_data->wt_sell.ready)
            launch_spmv_sell(st, precond_data->wt_sell.view(pat.n_rows, stream_w), SPMV_PLAIN, d_isai_tmp.p,
                             nullptr, out, last, gate);
        else
            launch_spmv(st, WT, SPMV_PLAIN, d_isai_tmp.p, nullptr, out, last, gate);
        return;
    }
    DevBlockJacobi J;
    J.n_rows = pat.n_rows;
    J.n_blocks = precond_data->n_blocks;
    J.stride = precond_data->stride;
    J.block_ptrs = precond_data->block_ptrs.p;
    J.row_block = precond_data->row_block.p;
    J.blocks = precond_data->values.p;
    J.uniform = precond_data->uniform_blocks ? 1 : 0;
    // (blocks kept in the caller's order -- also a stored object that a field WITHOUT a numbering of its own generated --
    //  are reached through this solver's permutation)
    if (pat.renumbered() && (precond_data->through_perm || precond_data->caller_order_blocks())) {
        J.rows = d_new_id.p;
        J.pos = d_old_of.p;
        J.by_device_row = precond_data->by_device_row ? 1 : 0;
        if (!precond_data->by_device_row) {
            // (property bjFusedPerm 0: the three-launch form with two staging vectors, for A/B)
            const bool fused_perm = prop("bjFusedPerm", 1.0) != 0.0 && in != out;
            launch_bj_apply_staged(st, J, in, out, dot_part, gate, fused_perm ? nullptr : d_bj_tmp0.p,
                                   fused_perm ? nullptr : d_bj_tmp1.p);
            return;
        }
    }
    launch_bj_apply(st, J, in, out, dot_part, gate);
}

int ogl_solver::init_preconditioner()
{
    precond = nullptr;
    precond_data = nullptr;
    if (cfg.preconditioner == OGL_PRECOND_NONE) return OGL_OK;  // :342
    const bool isai = cfg.preconditioner == OGL_PRECOND_ISAI || cfg.preconditioner == OGL_PRECOND_GISAI;
    if (cfg.preconditioner != OGL_PRECOND_BJ && !isai)
        return fail(OGL_ERR_UNSUPPORTED, "preconditioner kind %d is not built", cfg.preconditioner);
    if (isai && (cfg.sparsity_power < 1 || cfg.sparsity_power > 8))
        return fail(OGL_ERR_INVALID, "ISAI sparsityPower %d outside [1, 8]", cfg.sparsity_power);
    if (!isai && (cfg.max_block_size < 1 || cfg.max_block_size > MAX_JACOBI_BLOCK))
        return fail(OGL_ERR_INVALID, "BJ maxBlockSize %d outside [1, %d]", cfg.max_block_size,
                    MAX_JACOBI_BLOCK);
    if (cfg.preconditioner == OGL_PRECOND_ISAI)
        OGL_TRY(d_isai_tmp.alloc((size_t)pat.n_rows + 2, reg->stream));
    const int kind = isai ? (cfg.preconditioner == OGL_PRECOND_ISAI ? 3 : 4)
                          : (cfg.max_block_size == 1 ? 1 : 2);
    const int stride = kind == 2 ? cfg.max_block_size : (isai ? cfg.sparsity_power : 0);
    const int cache = (int)prop("preconditionerCaching", 0);
    const bool stored =
        reg->has_cached_precond && reg->cached_precond.matches(kind, (size_t)pat.n_rows, stride);
    // (the store is shared by all fields, Preconditioner.H:357: a stored object whose values live in ANOTHER pattern's
    //  device numbering -- inverse diagonal, W / W^T, block rows stored by device row, the backend's own blocks -- would be
    //  a silently permuted operator here: generate for this solve instead.  Blocks kept block-major in the caller's
    //  order are applied through this solver's permutation.)
    const bool foreign = stored && reg->cached_precond.foreign_to(pat_id, pat.renumbered());
    if (stored && cache > 0 && !foreign) {
        props["preconditionerCaching"] = cache - 1;
        precond_data = &reg->cached_precond;
    } else {
        props["preconditionerCaching"] = cfg.caching;
        PrecondData &P = stored ? own_precond : reg->cached_precond;
        OGL_TRY(generate_preconditioner(P));
        if (!stored) reg->has_cached_precond = true;
        precond_data = &P;
    }
    if (precond_data->kind == 1) precond = precond_data->values.p;
    if (pat.renumbered() && precond_data->kind == 2 && !precond_data->by_device_row &&
        (precond_data->through_perm || precond_data->caller_order_blocks())) {
        // (the staged apply's two vectors in the caller's order; also for a stored object another field generated)
        OGL_TRY(d_bj_tmp0.alloc((size_t)pat.n_rows + 2, reg->stream));
        OGL_TRY(d_bj_tmp1.alloc((size_t)pat.n_rows + 2, reg->stream));
    }
    return OGL_OK;
}

// ------------------------------------------------------------------------------------------
// distributed::Matrix::apply: y = A_local x (+ dot partials), then y += A_non_local recv
// ------------------------------------------------------------------------------------------
// arguments of the next SpMV's halo exchange for a producer kernel that puts the values itself (step_1x)
HaloPutFused ogl_solver::begin_halo_put()
{
    HaloPutFused put;
    if (!(pat.non_local_nnz > 0 && peer_halo) || prop("haloFused", 1.0) == 0.0 || peer_safe_wait())
        return put;
    if (++halo_seq == 0) ++halo_seq;
    cur_halo = peer_halo_args(halo_seq);
    put.P = cur_halo;
    put.chunk_sptr = d_chunk_sptr.p;
    put.send_pos = d_send_pos.p;
    put.send_idxs = d_send_idxs.p;
    put.ticket = d_ticket.p;
    put.n_put_chunks = n_put_chunks;
    return put;
}

// what a kernel needs to wait for exchange `ph` and to add the non-local part itself
HaloFused ogl_solver::halo_fused_args(const PeerHalo &ph) const
{
    HaloFused hf;
    hf.chunk_bptr = d_chunk_bptr.p;
    hf.boundary_rows = d_boundary_rows.p;
    hf.entry_ptrs = d_boundary_ptrs.p;
    hf.cols = d_nl_cols.p;
    hf.vals = d_nl_vals.p;
    hf.recv = peer_recv(ph.seq);
    hf.local_flag = ph.local_flag;
    hf.n_neigh = ph.n_neigh;
    hf.seq = ph.seq;
    hf.timeout_ticks = ph.timeout_ticks;
    hf.s = d_scal.p;
    return hf;
}

int ogl_solver::dist_spmv(int mode, const double *x, const double *b, double *y,
                          const SpmvDots &dots, const DevScalars *gate, bool prepacked)
{
    hipStream_t st = reg->stream;
    const bool has_halo = pat.non_local_nnz > 0;
    const double *recv = d_recv.p;
    PeerHalo ph;
    // peer-put transport: by default the non-local part is added inside the local kernel (HaloFused) -- a
    // distributed SpMV is then 2 launches (pack + put + signal | local + wait + non-local), 1 when the producer
    // of x has put the values itself; property haloFused 0 keeps the separate finish kernel (A/B)
    // peerSafeWait 1 (ranks that SHARE a device; switched on by peer_connect itself when two ranks
    // report the same PCI bus id): no workgroup of the SpMV waits -- ONE workgroup of a
    // kernel of its own does, then the non-local part is added by kernels that find the values there.  Many waiting
    // workgroups of several ranks on one device can hold every slot the producers' put kernels need.
    const bool safe = has_halo && peer_halo && peer_safe_wait();
    const bool fuse = has_halo && peer_halo && !safe && prop("haloFused", 1.0) != 0.0;
    HaloFused hf;
    if (has_halo && peer_halo) {
        // peer-put: the values go straight into the neighbours' receive blocks over xGMI, then the
        // flags; they fly while the local SpMV below runs
        if (prepacked && fuse) {
            ph = cur_halo;
        } else {
            if (++halo_seq == 0) ++halo_seq;
            ph = peer_halo_args(halo_seq);
            launch_pack_put_signal(st, halo(), ph, x, gate, d_ticket.p);
        }
        recv = peer_recv(ph.seq);
        if (fuse) hf = halo_fused_args(ph);
    } else if (has_halo) {
        // pack on the compute stream, exchange on the communication stream: the neighbour copies
        // fly while the local SpMV below runs; the non-local kernel waits for their arrival
        if (!reg->comm_stream) {
            OGL_HIP_CHECK(stream_create(&reg->comm_stream));
            OGL_HIP_CHECK(ev_create(&reg->ev_packed, hipEventDisableTiming));
            OGL_HIP_CHECK(ev_create(&reg->ev_received, hipEventDisableTiming));
        }
        launch_pack(st, halo(), x, d_send.p, gate);
        OGL_HIP_CHECK(hipEventRecord(reg->ev_packed, st));
        OGL_HIP_CHECK(hipStreamWaitEvent(reg->comm_stream, reg->ev_packed, 0));
        OGL_TRY(reg->comm->exchange(d_send.p, d_recv.p, neighbours, counts, reg->comm_stream));
        OGL_HIP_CHECK(hipEventRecord(reg->ev_received, reg->comm_stream));
    }
    // The fused dot partials of the local kernel are final for every chunk without boundary rows;
    // the few chunks that hold boundary rows are redone after "y += A_non_local recv" (same
    // per-chunk tree, so the sums are bit-identical to a dot over the finished y).
    if (cfg.matrix_format == OGL_FORMAT_ELL && ell_ready && !ell_values_stale)
        launch_spmv_ell(st, ell(), mode, x, b, y, dots, gate, hf);
    else if (use_sym())
        launch_spmv_sym(st, sym(), mode, x, b, y, dots, gate, hf);
    else if (use_symx())
        launch_spmv_symx(st, symx(), mode, x, b, y, dots, gate, hf);
    else if (use_sell())
        launch_spmv_sell(st, sell(), mode, x, b, y, dots, gate, hf);
    else
        launch_spmv(st, csr(), mode, x, b, y, dots, gate, hf);
    if (safe) {
        launch_halo_wait(st, ph, gate, d_scal.p);
        launch_spmv_non_local(st, halo(), mode, recv, y, gate);
        if (dots.part)
            launch_partials_dot_chunks(st, pat.n_rows, dots.with, y, dots.part, gate, d_boundary_chunks.p,
                                       n_boundary_chunks);
        if (dots.part_yy)
            launch_partials_dot_chunks(st, pat.n_rows, y, y, dots.part_yy, gate, d_boundary_chunks.p,
                                       n_boundary_chunks);
    } else if (has_halo && peer_halo && !fuse) {
        // wait for the neighbours' flags, add the non-local part, redo the touched chunks' partials
        launch_halo_finish(st, halo(), mode, pat.n_rows, d_boundary_chunks.p, d_boundary_chunk_ptr.p,
                           n_boundary_chunks, recv, y, dots, ph, gate, d_scal.p);
    } else if (has_halo && !peer_halo) {
        OGL_HIP_CHECK(hipStreamWaitEvent(st, reg->ev_received, 0));
        launch_spmv_non_local(st, halo(), mode, recv, y, gate);
        if (dots.part)
            launch_partials_dot_chunks(st, pat.n_rows, dots.with, y, dots.part, gate,
                                       d_boundary_chunks.p, n_boundary_chunks);
        if (dots.part_yy)
            launch_partials_dot_chunks(st, pat.n_rows, y, y, dots.part_yy, gate,
                                       d_boundary_chunks.p, n_boundary_chunks);
    }
    return OGL_OK;
}

int ogl_solver::finalize(int phase, FinArgs &a)
{
    hipStream_t st = reg->stream;
    if (a.n_sums == 0) {  // nothing to reduce: scalar logic only (identical on every rank)
        a.do_reduce = 0;
        a.do_logic = 1;
        launch_finalize(st, phase, d_scal.p, a);
        return OGL_OK;
    }
    if (!reg->comm->multi()) {
        a.do_reduce = 1;
        a.do_logic = 1;
        launch_finalize(st, phase, d_scal.p, a);
        return OGL_OK;
    }
    if (reg->peer_ready) {  // the all-reduce runs inside the finaliser (peer mailboxes over xGMI)
        a.peer = reg->peer_next();
        a.do_reduce = 1;
        a.do_logic = 1;
        launch_finalize(st, phase, d_scal.p, a);
        a.peer = PeerArgs{};
        return OGL_OK;
    }
    a.do_reduce = 1;
    a.do_logic = 0;
    launch_finalize(st, phase, d_scal.p, a);
    OGL_TRY(reg->comm->allreduce(sums_ptr(d_scal.p), a.n_sums, st));
    a.do_reduce = 0;
    a.do_logic = 1;
    launch_finalize(st, phase, d_scal.p, a);
    return OGL_OK;
}

// ------------------------------------------------------------------------------------------
// The Krylov drivers
// ------------------------------------------------------------------------------------------
// One driver for GKOCG, GKOBiCGStab and GKOGMRES: plan (which turn shape) -> prepare (criterion, buffers, norm factor,
// initial residual, the sums of turn 0) -> batches of turns with the stop flag polled one batch late -> finish (x,
// history, perf).  One member function per solver x turn shape (turn_*); what they share per solve lives in KrylovRun.
//
// GKOBiCGStab ([UPSTREAM] gko::solver::Bicgstab), per turn:
//   rho = rr.r, sum|r| -> check#1 -> p = r + (rho/prev_rho * alpha/omega)(p - omega v) -> y = M^-1 p
//   -> v = A y, beta = rr.v -> alpha = rho/beta, s = r - alpha v, sum|s| -> check#2 (x += alpha y
//   when it stops) -> z = M^-1 s -> t = A z, gamma = s.t, beta = t.t -> omega = gamma/beta,
//   x += alpha y + omega z, r = s - omega t.
// Two checks per turn: maxIter is doubled (StoppingCriterion.H:188) and the reported count halved
// (GKOBiCGStab.H:114).
struct ogl_solver::KrylovRun {
    hipStream_t st = nullptr;
    int n = 0, nc = 0;
    DevScalars *s = nullptr, *s2 = nullptr;
    DevScalars *slot_s[2] = {nullptr, nullptr};
    int cur = 0;  // the slot that holds the scalars after everything enqueued so far (folded GKOBiCGStab turn only)
    bool bicg = false, gmres = false, generic = false, multi = false;
    // turn shapes (see plan)
    bool fused = false, fused2 = false, merged = false, merged_halo = false, bicg_fold = false, gmres_fold = false;
    int m = 0;        // Krylov dimension of GKOGMRES
    int64_t ldv = 0;  // leading dimension of the Krylov bases
    double n_global = 0.0;
    size_t n_halo = 0;
    double *p0 = nullptr, *p1 = nullptr, *ph = nullptr;  // p of even / odd turns (merged turn), old p at the halo columns
    double *z_kept = nullptr;  // z = r / d, left behind by step_2r_fin for the gathers
    LeadBox lead{};  // leader finalisation of the folded turn (box == nullptr: every workgroup reduces for itself)
    DevCriterion crit{};
    bool is_final = false;
    int max_checks = 0, max_turns = 0;
    int prof_stride = 0, prof_cap = 0;
    hipEvent_t ev_chk[2] = {nullptr, nullptr};  // (the solver's own pair, created once: ogl_solver::chk_ev)
    double t_start = 0.0;
    FinArgs fg{}, chk{}, f1{}, f2{};  // GMRES finaliser arguments; the head-of-turn check; one / two partial arrays
    const double *beta_ptr = nullptr;
    double *gm = nullptr, *gm_y = nullptr;
    double *y = nullptr, *z = nullptr;  // BiCGStab: identity preconditioner -> y aliases p, z aliases s
    int enq = 0;                        // turns enqueued so far
    double *gm_h(int i, int j) const { return gm + (size_t)j * (m + 1) + i; }
    double *p_of_turn(int turn) const { return (merged && (turn & 1)) ? p1 : p0; }  // p that turn `turn` reads
    double *p_halo_of_turn(int turn) const { return ph + (size_t)(turn & 1) * n_halo; }
    // scalar Jacobi: V_it is divided by its norm at the head of turn `it`, in the pass that applies the preconditioner
    bool gmres_scale_late() const { return gmres && !generic && has_diag; }
    bool has_diag = false;
    bool folded() const { return fused || bicg_fold; }  // the check of a turn runs at the head of the next kernel
};

int ogl_solver::run_cg(ogl_perf *perf) { return run_krylov(perf); }
int ogl_solver::run_bicgstab(ogl_perf *perf) { return run_krylov(perf); }

int ogl_solver::run_krylov(ogl_perf *perf)
{
    KrylovRun k;
    OGL_TRY(krylov_plan(k));
    OGL_TRY(krylov_prepare(k));
    OGL_TRY(krylov_loop(k));
    return krylov_finish(k, perf);
}

// Which solver, which turn shape.
int ogl_solver::krylov_plan(KrylovRun &k)
{
    hipStream_t st = k.st = reg->stream;
    const int n = k.n = pat.n_rows;
    DevScalars *s = k.s = d_scal.p;
    const int nc = k.nc = (int)n_chunks(n);
    const bool bicg = k.bicg = cfg.solver == OGL_SOLVER_BICGSTAB;
    const bool gmres = k.gmres = cfg.solver == OGL_SOLVER_GMRES;
    // Ginkgo's default Krylov dimension is 100; the reference has no keyword for it
    // (GKOGMRES.H:46-63), `krylovDim` is this build's addition
    k.m = cfg.krylov_dim > 0 ? cfg.krylov_dim : 100;
    k.ldv = (int64_t)n + 2;
    // block Jacobi (maxBlockSize > 1): z = M^-1 r is materialised by its own kernel; the scalar
    // case stays fused into the step kernels
    const bool generic = k.generic = precond_data && precond_data->kind >= 2;  // block Jacobi, ISAI, GISAI
    k.has_diag = precond != nullptr;
    const bool multi = k.multi = reg->comm->multi();
    const bool small = !multi && nc >= 1 &&
                       nc <= std::min((int)prop("fusedFinMaxChunks", (double)FUSED_FIN_MAX_CHUNKS), FUSED_FIN_MAX_CHUNKS) &&
                       prop("fusedFinalizers", 1.0) != 0.0;
    // small single-rank GKOCG systems: finalisers folded into the step kernels, 3 launches per turn (kernels_krylov.hip)
    // ... and for LARGER single-rank GKOCG systems the same three launches with the LEADER finalisation (device_common.hpp):
    // workgroup 0 of the consuming kernel is the finaliser, the others poll its mailbox -- instead of two
    // single-workgroup launches (10 + 7 us at 10 M rows) and their dispatch gaps per turn (property leadFinalizers)
    const bool lead_any = !multi && !small && nc >= 3 * 16 && prop("leadFinalizers", 1.0) != 0.0;
    const bool lead_ok = lead_any && !bicg && !gmres && !generic;
    bool fused = k.fused = !bicg && !gmres && !generic && (small || lead_ok);
    k.lead = LeadBox{};
    k.s2 = s + 1;
    // ... and the same for small single-rank GKOBiCGStab systems: three finalisers folded into step_1 / step_2 / step_3
    // (k_bicg_fold1/2/3: 5 launches per turn instead of 8, plus the preconditioner's own)
    // (larger systems: the same five launches with the leader finalisation, any preconditioner -- the 2 M-row momentum
    //  systems of configs[2] spend a tenth of a turn in three single-workgroup launches and their gaps)
    k.bicg_fold = bicg && (small || lead_any) && prop("bicgFold", 1.0) != 0.0;
    // ... and for small single-rank GKOGMRES systems the finaliser between two Gram-Schmidt links is folded into the next
    // link's kernel (k_gmres_mgs_fold: one launch per link instead of two)
    k.gmres_fold = gmres && small && prop("gmresFold", 1.0) != 0.0;
    k.slot_s[0] = s;
    k.slot_s[1] = k.s2;
    props["fusedFinalizersInUse"] = (((fused || k.bicg_fold) && small) || k.gmres_fold) ? 1.0 : 0.0;
    // ... and on half storage step_1x(_fin) and the SpMV are one kernel (k_cg_turn_sym, k_cg_turn_sym_big): 2 launches
    // per turn for small systems, 4 for larger ones, p alternating between two buffers (single rank: with halos the
    // put and the wait for the neighbours' puts would sit in one kernel)
    // (larger systems, property fusedTurnBig: on while matrix and vectors live in the Infinity Cache -- 128^3 58.8 -> 54.7 us
    //  per turn, 136^3 66.3 -> 61.5 -- and off once they are streamed: 160^3 105.1 -> 104.5, 216^3 4149 -> 4190 turns/s,
    //  where the merged kernel runs 152 us for the 162 of step_1x + SpMV and step_2r pays 8 N more bytes for keeping z)
    // Several ranks (peer-put transport with the non-local part inside the local kernel): the same merge, 4 launches
    // per turn instead of 5 -- the neighbours' step_2r puts z of their send rows, this rank keeps the old p of its halo
    // columns and forms p_new there itself (kernels_spmv_sym.hip, k_cg_turn_sym_big<.., HALO>), so the merged kernel has
    // nothing to put and only waits for a put of the PREVIOUS launch.  Every rank must run the same turn (what the
    // neighbours put differs): agreed below together with the global row count.
    bool merged = !bicg && !gmres && !generic && nc >= 1 && use_sym() && cfg.matrix_format != OGL_FORMAT_ELL &&
                  ((fused && small) ? prop("fusedTurn", 1.0) != 0.0 : prop("fusedTurnBig", sym().stream ? 0.0 : 1.0) != 0.0);
    if (multi)
        merged = merged && peer_halo && prop("haloFused", 1.0) != 0.0 && prop("fusedTurnMulti", 1.0) != 0.0 &&
                 !peer_safe_wait();
    k.n_global = (double)n;
    if (multi) {
        // global row count (Partition.H:118-121) and the agreement on the turn, through the device all-reduce
        const double mine[2] = {(double)n, merged ? 0.0 : 1.0};
        double got[2] = {0.0, 0.0};
        OGL_HIP_CHECK(hipMemcpyAsync(sums_ptr(s), mine, sizeof(mine), hipMemcpyHostToDevice, st));
        OGL_HIP_CHECK(hipStreamSynchronize(st));
        OGL_TRY(reg->allreduce(sums_ptr(s), 2));
        OGL_HIP_CHECK(hipMemcpyAsync(got, sums_ptr(s), sizeof(got), hipMemcpyDeviceToHost, st));
        OGL_HIP_CHECK(hipStreamSynchronize(st));
        k.n_global = got[0];
        merged = got[1] == 0.0;
    }
    if ((lead_ok && fused) || (lead_any && k.bicg_fold)) {
        // (with the merged kernel two launches per turn, k_cg_turn_sym<.., LEAD> | k_cg_step2r_fin<LEAD>; that kernel has no
        //  streaming instantiation: the merge is on by default only where matrix and vectors live in the Infinity Cache)
        if (!lead_box) {
            void *b = nullptr;
            OGL_HIP_CHECK(ledger::dev_malloc(&b, LEAD_REPLICAS * LEAD_REPLICA_STRIDE * sizeof(unsigned long long), /*fine_grained=*/true));
            lead_box = static_cast<unsigned long long *>(b);
        }
        // (tags restart at 1 with every solve: no word of an earlier solve may survive)
        OGL_HIP_CHECK(hipMemsetAsync(lead_box, 0, LEAD_REPLICAS * LEAD_REPLICA_STRIDE * sizeof(unsigned long long), st));
        k.lead.box = lead_box;
        k.lead.timeout_ticks = (long long)(prop("leadTimeoutS", 10.0) * 1e8);
        k.lead.early_loads = prop("leadEarlyLoads", 1.0) != 0.0 ? 1 : 0;
    }
    props["leadFinalizersInUse"] = k.lead.box ? 1.0 : 0.0;
    props["fusedFinalizersInUse"] = (((fused || k.bicg_fold) && small) || k.gmres_fold) ? 1.0 : 0.0;
    k.merged = merged;
    k.fused2 = fused && merged;
    k.merged_halo = merged && multi && pat.non_local_nnz > 0;  // (a rank without neighbours: the single-rank kernel)
    props["fusedTurnInUse"] = merged ? 1.0 : 0.0;
    return OGL_OK;
}

// Buffers, the stopping criterion, the norm factor, r = b - A x, and the sums the first check needs.
int ogl_solver::krylov_prepare(KrylovRun &k)
{
    hipStream_t st = k.st;
    const int n = k.n, nc = k.nc, m = k.m;
    DevScalars *s = k.s;
    const bool bicg = k.bicg, gmres = k.gmres, generic = k.generic, merged = k.merged;
    if (merged) OGL_TRY(d_p2.alloc((size_t)n + 2, st));
    if (merged && precond) OGL_TRY(d_z.alloc((size_t)n + 2, st));
    k.n_halo = (size_t)pat.non_local_nnz;
    if (k.merged_halo) {  // old p at the halo columns, two buffers like p itself; p = 0 before the first turn
        OGL_TRY(d_p_halo.alloc(2 * k.n_halo + 2, st));
        OGL_HIP_CHECK(hipMemsetAsync(d_p_halo.p, 0, d_p_halo.n * sizeof(double), st));
    }
    k.p0 = d_p.p;
    k.p1 = d_p2.p;
    k.ph = d_p_halo.p;
    k.z_kept = merged && precond ? d_z.p : nullptr;

    // StoppingCriterion ctor + build_dist_stopping_criterion (StoppingCriterion.H:164-234)
    k.is_final = cfg.rel_tol == 0.0;  // get_is_final, :242
    const int prev_iters = (int)prop(k.is_final ? "prevSolveIters_final" : "prevSolveIters", 1);
    const double prev_cost = prop("_prev_solve", 0.0);
    DevCriterion &crit = k.crit;
    crit.tolerance = cfg.tolerance;
    crit.rel_tol = cfg.rel_tol;
    crit.max_iter = bicg ? 2 * cfg.max_iter : cfg.max_iter;  // :188
    crit.export_res = cfg.export_res;
    ogl_host_adapt_criterion(&cfg, prev_iters, prev_cost, &crit.min_iter, &crit.frequency);
    if (crit.frequency < 1) return fail(OGL_ERR_INVALID, "evalFrequency must be >= 1");
    // the criterion stops at the first evaluated check at or after max(maxIter, minIter): checks
    // below minIter are skipped without a verdict (StoppingCriterion.C:77-81), so a minIter above
    // maxIter keeps the loop going, as in the reference
    k.max_checks = std::max(crit.max_iter, crit.min_iter) + crit.frequency + 1;
    k.max_turns = bicg ? k.max_checks / 2 + 1 : k.max_checks;  // CG and GMRES: one check per turn
    // (sized by what the keywords allow, not by this solve's adaptive frequency / minIter: the same block solve after solve)
    OGL_TRY(d_history.alloc((size_t)std::max(k.max_checks, crit.max_iter + std::max(1, cfg.norm_eval_limit) + 1) + 4, st));
    if (cfg.export_res)
        OGL_HIP_CHECK(hipMemsetAsync(d_history.p, 0, d_history.n * sizeof(double), st));
    if (k.bicg_fold) {  // (a folded kernel never writes a partial array it reads: six of them per turn)
        OGL_TRY(d_part3.alloc((size_t)nc, st));
        OGL_TRY(d_part4.alloc((size_t)nc, st));
        OGL_TRY(d_part5.alloc((size_t)nc, st));
    }
    if (bicg) {
        const size_t nv = (size_t)n + 2;
        OGL_TRY(d_v.alloc(nv, st));
        OGL_TRY(d_s.alloc(nv, st));
        OGL_TRY(d_t.alloc(nv, st));
        OGL_TRY(d_rr.alloc(nv, st));
        if (precond || generic) {
            OGL_TRY(d_y.alloc(nv, st));
            OGL_TRY(d_z.alloc(nv, st));
        }
    } else if (gmres) {
        OGL_TRY(d_V.alloc((size_t)(m + 1) * (size_t)k.ldv, st));
        OGL_TRY(d_gm.alloc(gmres_state_len(m), st));
        OGL_HIP_CHECK(hipMemsetAsync(d_gm.p, 0, gmres_state_len(m) * sizeof(double), st));
        if (generic) OGL_TRY(d_z.alloc((size_t)n + 2, st));
    } else if (generic) {
        OGL_TRY(d_z.alloc((size_t)n + 2, st));
    }

    // profile_kernels = k > 0: every k-th turn's in-loop SpMV (the first of a BiCGStab turn) is bracketed by an event pair
    k.prof_stride = std::max(0, cfg.profile_kernels);
    k.prof_cap = k.prof_stride ? std::min((k.max_turns + k.prof_stride - 1) / k.prof_stride, 4096) : 0;
    while ((int)prof_ev.size() < 2 * k.prof_cap) {
        hipEvent_t e;
        OGL_HIP_CHECK(ev_create(&e));
        prof_ev.push_back(e);
    }
    for (int i = 0; i < 2; ++i) {  // (once per solver, not per solve: no runtime object comes and goes with a time step)
        if (!chk_ev[i]) OGL_HIP_CHECK(ev_create(&chk_ev[i]));
        k.ev_chk[i] = chk_ev[i];
    }

    k.t_start = now_ms();
    launch_reset_scalars(st, s, crit);

    FinArgs fa;
    // norm factor, part 1: xbar = mean(x) (StoppingCriterion.C:17-19)
    launch_partials_sum(st, n, d_x.p, d_part0.p);
    fa = FinArgs{};
    fa.part[0] = d_part0.p;
    fa.n_part = nc;
    fa.n_sums = 1;
    fa.n_local = (double)n;
    fa.n_global = k.n_global;  // (all-reduced in plan)
    OGL_TRY(finalize(FIN_MEAN, fa));
    // Axref = A * (xbar 1) (:24-29) into q
    launch_fill_xbar(st, n, d_w.p, s);
    OGL_TRY(dist_spmv(SPMV_PLAIN, d_w.p, nullptr, d_q.p, SpmvDots{}, nullptr));
    // r = b - A x  ([UPSTREAM] r = b; r = -1*A*x + 1*r)
    OGL_TRY(dist_spmv(SPMV_RESIDUAL, d_x.p, d_b.p, d_r.p, SpmvDots{}, nullptr));
    // norm factor, part 2 (:53-68)
    launch_partials_normfactor(st, n, d_b.p, d_q.p, d_r.p, d_part0.p);
    fa = FinArgs{};
    fa.part[0] = d_part0.p;
    fa.n_part = nc;
    fa.n_sums = 1;
    OGL_TRY(finalize(FIN_NORMFACTOR, fa));

    // solver initialisation + turn 0: rho, sum|r|, check (timed once as "time per residual norm
    // calculation", lduLduBase.H:287)
    OGL_HIP_CHECK(hipMemsetAsync(d_p.p, 0, (size_t)n * sizeof(double), st));
    k.fg = FinArgs{};
    k.fg.part[0] = d_part0.p;
    k.fg.part[1] = d_part1.p;
    k.fg.n_part = nc;
    k.fg.history = d_history.p;
    k.fg.gm = d_gm.p;
    k.fg.m = m;
    k.beta_ptr = reinterpret_cast<const double *>(reinterpret_cast<const char *>(s) + offsetof(DevScalars, beta));
    k.gm = d_gm.p;
    k.gm_y = d_gm.p + (size_t)(m + 1) * m + 2 * (size_t)m + (m + 1);
    if (gmres) {
        OGL_HIP_CHECK(hipEventRecord(k.ev_chk[0], st));
        OGL_TRY(gmres_restart(k, nullptr));
        OGL_HIP_CHECK(hipEventRecord(k.ev_chk[1], st));
    } else if (bicg) {
        // rr = r ; p = v = 0 ([UPSTREAM] bicgstab::initialize); rho = rr.r = r.r
        OGL_HIP_CHECK(hipMemcpyAsync(d_rr.p, d_r.p, (size_t)n * sizeof(double),
                                     hipMemcpyDeviceToDevice, st));
        OGL_HIP_CHECK(hipMemsetAsync(d_v.p, 0, (size_t)n * sizeof(double), st));
        launch_cg_rho_norm(st, n, d_r.p, nullptr, d_part0.p, d_part1.p, s);
    } else {
        // p = q = 0 ([UPSTREAM] cg::initialize); z is never materialised (z = r * inv_diag on the fly)
        launch_cg_rho_norm(st, n, d_r.p, precond, d_part0.p, d_part1.p, s);
        if (k.z_kept) launch_mul(st, n, k.z_kept, d_r.p, precond, nullptr);  // (the z of the first k_cg_turn_sym)
        if (generic) {  // rho = r . (M^-1 r) with the block preconditioner
            apply_preconditioner(d_r.p, d_z.p, s, d_part0.p);
        }
    }
    k.chk = FinArgs{};
    k.chk.part[0] = d_part0.p;
    k.chk.part[1] = d_part1.p;
    k.chk.n_part = nc;
    k.chk.n_sums = 2;
    k.chk.history = d_history.p;
    if (!gmres && !k.folded()) {  // (folded turns: this check opens the first folded kernel)
        OGL_HIP_CHECK(hipEventRecord(k.ev_chk[0], st));
        OGL_TRY(finalize(FIN_CG_CHECK, k.chk));
        OGL_HIP_CHECK(hipEventRecord(k.ev_chk[1], st));
    }
    if (k.merged_halo) {  // the z of the first merged turn (later ones: put by step_2r)
        if (++halo_seq == 0) ++halo_seq;
        cur_halo = peer_halo_args(halo_seq);
        launch_pack_put_signal(st, halo(), cur_halo, k.z_kept ? k.z_kept : d_r.p, s, d_ticket.p);
    }

    k.f1 = FinArgs{};  // one partial array
    k.f1.part[0] = d_part0.p;
    k.f1.n_part = nc;
    k.f1.n_sums = 1;
    k.f1.history = d_history.p;
    k.f2 = k.f1;  // two partial arrays
    k.f2.part[1] = d_part1.p;
    k.f2.n_sums = 2;

    k.y = (precond || generic) ? d_y.p : d_p.p;  // identity: y aliases p, z aliases s
    k.z = (precond || generic) ? d_z.p : d_s.p;
    k.enq = 0;
    return OGL_OK;
}

// gmres::restart: rn = ||r||, rnc[0] = rn, V_0 = r / rn; the criterion keeps sum|r| of this r.  gate == nullptr: the
// restart before the first turn, whose finaliser runs the first check too.  With scalar Jacobi the division waits for the
// turn that follows (k_gmres_scale_mul).
int ogl_solver::gmres_restart(KrylovRun &k, const DevScalars *gate)
{
    launch_cg_rho_norm(k.st, k.n, d_r.p, nullptr, d_part0.p, d_part1.p, gate);  // r.r and sum|r|
    k.fg.n_sums = 2;
    k.fg.check_after = gate ? 0 : 1;
    OGL_TRY(finalize(FIN_GMRES_RESTART, k.fg));
    k.fg.check_after = 0;
    if (!k.gmres_scale_late()) launch_gmres_scale(k.st, k.n, d_V.p, d_r.p, k.beta_ptr, gate);
    return OGL_OK;
}

// solve_krylov + x += M^-1 (V y) over `cols` columns of the cycle
int ogl_solver::gmres_update_x(KrylovRun &k, int cols, const DevScalars *gate)
{
    if (cols <= 0) return OGL_OK;
    k.fg.n_sums = 0;
    k.fg.turn = cols;
    OGL_TRY(finalize(FIN_GMRES_SOLVE, k.fg));
    if (k.generic) {
        launch_gmres_update_x(k.st, k.n, d_V.p, k.ldv, k.gm_y, cols, nullptr, d_x.p, d_w.p, gate);
        apply_preconditioner(d_w.p, d_z.p, gate);
        launch_add(k.st, k.n, d_x.p, d_z.p, gate);
    } else {
        launch_gmres_update_x(k.st, k.n, d_V.p, k.ldv, k.gm_y, cols, precond, d_x.p, nullptr, gate);
    }
    return OGL_OK;
}

// ---- one turn of every solver x turn shape.  enq = index of the turn; pe >= 0: the event pair that brackets the turn's
// in-loop SpMV (profile_kernels), -1: none.  Kernels enqueued after the stop are no-ops (gated on the device scalars).

// GKOGMRES ([UPSTREAM] Gmres loop): check (on the residual of the last restart), restart when the cycle is full, then one
// Arnoldi step; gmres_fold: the finaliser between two Gram-Schmidt links runs inside the next link's kernel
int ogl_solver::turn_gmres(KrylovRun &k, int enq, int pe)
{
    hipStream_t st = k.st;
    const int n = k.n, m = k.m;
    const int64_t ldv = k.ldv;
    DevScalars *s = k.s;
    FinArgs &fg = k.fg;  // (the check at the head of this turn ran in the finaliser before it: restart or the last column's)
    if (enq > 0 && enq % m == 0) {
        OGL_TRY(gmres_update_x(k, m, s));
        OGL_TRY(dist_spmv(SPMV_RESIDUAL, d_x.p, d_b.p, d_r.p, SpmvDots{}, s));
        OGL_TRY(gmres_restart(k, s));
    }
    const int it = enq % m;
    double *v_it = d_V.p + (size_t)it * ldv, *nx = d_V.p + (size_t)(it + 1) * ldv;
    const double *w = v_it;  // identity preconditioner: w aliases V_it
    if (k.generic) {
        apply_preconditioner(v_it, d_w.p, s);
        w = d_w.p;
    } else if (precond) {  // V_it = (r | the last turn's new vector) / its norm, w = M^-1 V_it
        launch_gmres_scale_mul(st, n, v_it, it == 0 ? d_r.p : v_it, k.beta_ptr, precond, d_w.p, s);
        w = d_w.p;
    }
    if (pe >= 0) OGL_HIP_CHECK(hipEventRecord(prof_ev[2 * pe], st));
    OGL_TRY(dist_spmv(SPMV_PLAIN, w, nullptr, nx, SpmvDots{}, s));
    if (pe >= 0) OGL_HIP_CHECK(hipEventRecord(prof_ev[2 * pe + 1], st));
    // finish_arnoldi (modified Gram-Schmidt): H(k,it) = nx.V_k ; nx -= H(k,it) V_k
    fg.turn = it;
    fg.n_sums = 1;
    if (k.gmres_fold) {
        double *pin = d_part1.p, *pout = d_part0.p;  // (a link reads the partials of the one before it)
        for (int j = 0; j <= it; ++j) {
            launch_gmres_mgs_fold(st, n, nx, j > 0 ? d_V.p + (size_t)(j - 1) * ldv : nullptr,
                                  j > 0 ? k.gm_h(j - 1, it) : nullptr, d_V.p + (size_t)j * ldv, pin, pout, s);
            std::swap(pin, pout);
        }
        launch_gmres_mgs_fold(st, n, nx, v_it, k.gm_h(it, it), nullptr, pin, pout, s);
        fg.part[0] = pout;
        fg.check_after = 1;
        OGL_TRY(finalize(FIN_GMRES_COL, fg));  // ||nx||, Givens, residual-norm recurrence, the next turn's check
        fg.check_after = 0;
        fg.part[0] = d_part0.p;
    } else {
        for (int j = 0; j <= it; ++j) {
            launch_gmres_mgs(st, n, nx, j > 0 ? d_V.p + (size_t)(j - 1) * ldv : nullptr,
                             j > 0 ? k.gm_h(j - 1, it) : nullptr, d_V.p + (size_t)j * ldv,
                             d_part0.p, s);
            fg.k = j;
            OGL_TRY(finalize(FIN_GMRES_H, fg));
        }
        launch_gmres_mgs(st, n, nx, v_it, k.gm_h(it, it), nullptr, d_part0.p, s);
        fg.check_after = 1;
        OGL_TRY(finalize(FIN_GMRES_COL, fg));  // ||nx||, Givens, residual-norm recurrence, the next turn's check
        fg.check_after = 0;
    }
    if (!k.gmres_scale_late()) launch_gmres_scale(st, n, nx, nx, k.beta_ptr, s);
    return OGL_OK;
}

// GKOCG with a materialised z = M^-1 r (block Jacobi, ISAI): step_1 | SpMV | beta | step_2 | M^-1 | check
int ogl_solver::turn_cg_generic(KrylovRun &k, int, int pe)
{
    hipStream_t st = k.st;
    const int n = k.n;
    DevScalars *s = k.s;
    launch_cg_step1(st, n, d_p.p, d_z.p, nullptr, s);  // p = z + (rho/prev_rho) p
    if (pe >= 0) OGL_HIP_CHECK(hipEventRecord(prof_ev[2 * pe], st));
    OGL_TRY(dist_spmv(SPMV_PLAIN, d_p.p, nullptr, d_q.p, SpmvDots{d_p.p, d_part0.p, nullptr}, s));
    if (pe >= 0) OGL_HIP_CHECK(hipEventRecord(prof_ev[2 * pe + 1], st));
    OGL_TRY(finalize(FIN_BETA, k.f1));
    launch_cg_step2(st, n, d_x.p, d_r.p, d_p.p, d_q.p, nullptr, d_part0.p, d_part1.p, s);
    apply_preconditioner(d_r.p, d_z.p, s, d_part0.p);  // z = M^-1 r and the partials of r.z
    OGL_TRY(finalize(FIN_CG_CHECK, k.chk));
    return OGL_OK;
}

// small single-rank GKOCG on half storage, 2 launches: [check of the previous turn + pending x update + step_1 + SpMV] |
// beta + step_2r
int ogl_solver::turn_cg_two_launch(KrylovRun &k, int enq, int pe)
{
    hipStream_t st = k.st;
    if (enq == 0) OGL_HIP_CHECK(hipEventRecord(k.ev_chk[0], st));
    if (pe >= 0) OGL_HIP_CHECK(hipEventRecord(prof_ev[2 * pe], st));
    launch_cg_turn_sym(st, sym(), k.p_of_turn(enq), k.p_of_turn(enq + 1), d_x.p, k.z_kept ? k.z_kept : d_r.p,
                       d_q.p, d_part2.p, k.s, k.s2, d_part0.p, d_part1.p, d_history.p, enq == 0 ? 1 : 0, k.lead);
    if (pe >= 0) OGL_HIP_CHECK(hipEventRecord(prof_ev[2 * pe + 1], st));
    if (enq == 0) OGL_HIP_CHECK(hipEventRecord(k.ev_chk[1], st));
    launch_cg_step2r_fin(st, k.n, d_r.p, d_q.p, precond, d_part0.p, d_part1.p, k.s2, k.s, d_part2.p, k.z_kept, k.lead);
    return OGL_OK;
}

// small single-rank GKOCG, 3 launches: check of the previous turn (or of the initial residual) + pending x update +
// step_1 | SpMV | beta + step_2r: the scalars go s -> s2 -> s
int ogl_solver::turn_cg_three_launch(KrylovRun &k, int enq, int pe)
{
    hipStream_t st = k.st;
    const int n = k.n;
    if (enq == 0) OGL_HIP_CHECK(hipEventRecord(k.ev_chk[0], st));
    launch_cg_step1x_fin(st, n, d_p.p, d_x.p, d_r.p, precond, k.s, k.s2, d_part0.p, d_part1.p, d_history.p,
                         enq == 0 ? 1 : 0, k.lead);
    if (enq == 0) OGL_HIP_CHECK(hipEventRecord(k.ev_chk[1], st));
    if (pe >= 0) OGL_HIP_CHECK(hipEventRecord(prof_ev[2 * pe], st));
    OGL_TRY(dist_spmv(SPMV_PLAIN, d_p.p, nullptr, d_q.p, SpmvDots{d_p.p, d_part2.p, nullptr}, k.s2));
    if (pe >= 0) OGL_HIP_CHECK(hipEventRecord(prof_ev[2 * pe + 1], st));
    launch_cg_step2r_fin(st, n, d_r.p, d_q.p, precond, d_part0.p, d_part1.p, k.s2, k.s, d_part2.p, nullptr, k.lead);
    return OGL_OK;
}

// GKOCG on half storage between the single-workgroup finalisers, 4 launches: [pending x update + step_1 + SpMV] | beta |
// step_2r (keeps z) | check; several ranks: the neighbours' step_2r has put z, p_new is formed at the halo columns here
int ogl_solver::turn_cg_merged(KrylovRun &k, int enq, int pe)
{
    hipStream_t st = k.st;
    const int n = k.n;
    DevScalars *s = k.s;
    if (pe >= 0) OGL_HIP_CHECK(hipEventRecord(prof_ev[2 * pe], st));
    if (k.merged_halo) {
        // (waits for the z the neighbours put one kernel -- or, before turn 0, one launch -- earlier)
        launch_cg_turn_sym_big(st, sym(), k.p_of_turn(enq), k.p_of_turn(enq + 1), d_x.p,
                               k.z_kept ? k.z_kept : d_r.p, d_q.p, d_part0.p, s,
                               halo_fused_args(cur_halo), k.p_halo_of_turn(enq), k.p_halo_of_turn(enq + 1));
    } else {
        launch_cg_turn_sym_big(st, sym(), k.p_of_turn(enq), k.p_of_turn(enq + 1), d_x.p,
                               k.z_kept ? k.z_kept : d_r.p, d_q.p, d_part0.p, s);
    }
    if (pe >= 0) OGL_HIP_CHECK(hipEventRecord(prof_ev[2 * pe + 1], st));
    OGL_TRY(finalize(FIN_BETA, k.f1));
    if (k.merged_halo) {
        const HaloPutFused put = begin_halo_put();  // z of the next turn
        launch_cg_step2r(st, n, d_r.p, d_q.p, precond, d_part0.p, d_part1.p, s, k.z_kept, &put);
    } else {
        launch_cg_step2r(st, n, d_r.p, d_q.p, precond, d_part0.p, d_part1.p, s, k.z_kept);
    }
    k.chk.turn = 1;  // this check leaves an x update pending for the next turn's kernel
    OGL_TRY(finalize(FIN_CG_CHECK, k.chk));
    return OGL_OK;
}

// GKOCG, 5 launches (the headline's turn): step_1x | SpMV | beta | step_2r | check.  x += t p is deferred into the next
// turn's step_1x (kernels_krylov.hip): p is read once (peer-put transport: the halo values of the SpMV are put by step_1x itself)
int ogl_solver::turn_cg_five_launch(KrylovRun &k, int, int pe)
{
    hipStream_t st = k.st;
    const int n = k.n;
    DevScalars *s = k.s;
    const HaloPutFused put = begin_halo_put();
    launch_cg_step1x(st, n, d_p.p, d_x.p, d_r.p, precond, s, &put);
    if (pe >= 0) OGL_HIP_CHECK(hipEventRecord(prof_ev[2 * pe], st));
    OGL_TRY(dist_spmv(SPMV_PLAIN, d_p.p, nullptr, d_q.p,
                      SpmvDots{d_p.p, d_part0.p, nullptr}, s, put.chunk_sptr != nullptr));
    if (pe >= 0) OGL_HIP_CHECK(hipEventRecord(prof_ev[2 * pe + 1], st));
    OGL_TRY(finalize(FIN_BETA, k.f1));
    launch_cg_step2r(st, n, d_r.p, d_q.p, precond, d_part0.p, d_part1.p, s);
    k.chk.turn = 1;  // this check leaves an x update pending for the next step_1x
    OGL_TRY(finalize(FIN_CG_CHECK, k.chk));
    return OGL_OK;
}

// small single-rank GKOBiCGStab, 5 launches + the preconditioner's own: [check + step_1] | M^-1 | SpMV | [alpha + step_2] |
// M^-1 | SpMV | [mid-turn check + omega + step_3]; partials: rho, sum|r| in part0 / part1; rr.v in part2; sum|s| in
// part3; s.t, t.t in part4 / part5; the scalars alternate between the two slots (k.cur)
int ogl_solver::turn_bicg_folded(KrylovRun &k, int enq, int pe)
{
    hipStream_t st = k.st;
    const int n = k.n;
    DevScalars **slot_s = k.slot_s;
    int &cur = k.cur;
    double *y = k.y, *z = k.z;
    if (enq == 0) OGL_HIP_CHECK(hipEventRecord(k.ev_chk[0], st));
    launch_bicg_fold1(st, n, d_p.p, d_r.p, d_v.p, precond, y, slot_s[cur], slot_s[cur ^ 1], d_part0.p,
                      d_part1.p, d_history.p, k.lead);
    cur ^= 1;
    if (enq == 0) OGL_HIP_CHECK(hipEventRecord(k.ev_chk[1], st));
    if (k.generic) apply_preconditioner(d_p.p, y, slot_s[cur]);
    if (pe >= 0) OGL_HIP_CHECK(hipEventRecord(prof_ev[2 * pe], st));
    OGL_TRY(dist_spmv(SPMV_PLAIN, y, nullptr, d_v.p, SpmvDots{d_rr.p, d_part2.p, nullptr}, slot_s[cur]));
    if (pe >= 0) OGL_HIP_CHECK(hipEventRecord(prof_ev[2 * pe + 1], st));
    launch_bicg_fold2(st, n, d_r.p, d_v.p, d_s.p, precond, z, d_part3.p, slot_s[cur], slot_s[cur ^ 1],
                      d_part2.p, k.lead);
    cur ^= 1;
    if (k.generic) apply_preconditioner(d_s.p, z, slot_s[cur]);
    OGL_TRY(dist_spmv(SPMV_PLAIN, z, nullptr, d_t.p, SpmvDots{d_s.p, d_part4.p, d_part5.p}, slot_s[cur]));
    launch_bicg_fold3(st, n, d_x.p, d_r.p, d_s.p, d_t.p, y, z, d_rr.p, d_part0.p, d_part1.p, slot_s[cur],
                      slot_s[cur ^ 1], d_part4.p, d_part5.p, d_part3.p, d_history.p, enq, k.lead);
    cur ^= 1;
    return OGL_OK;
}

// GKOBiCGStab, 8 launches + the preconditioner's own (9 with several ranks: the mid-turn check keeps its own finaliser)
int ogl_solver::turn_bicg(KrylovRun &k, int enq, int pe)
{
    hipStream_t st = k.st;
    const int n = k.n;
    DevScalars *s = k.s;
    double *y = k.y, *z = k.z;
    FinArgs &f1 = k.f1, &f2 = k.f2;
    launch_bicg_step1(st, n, d_p.p, d_r.p, d_v.p, precond, y, s);
    if (k.generic) apply_preconditioner(d_p.p, y, s);
    if (pe >= 0) OGL_HIP_CHECK(hipEventRecord(prof_ev[2 * pe], st));
    OGL_TRY(dist_spmv(SPMV_PLAIN, y, nullptr, d_v.p,
                      SpmvDots{d_rr.p, d_part0.p, nullptr}, s));
    if (pe >= 0) OGL_HIP_CHECK(hipEventRecord(prof_ev[2 * pe + 1], st));
    OGL_TRY(finalize(FIN_BICG_ALPHA, f1));
    if (!k.multi && prop("bicgMergedCheck", 1.0) != 0.0) {
        // single rank: the mid-turn check moves behind the second SpMV and shares its finaliser (8 launches
        // per turn instead of 9; when it stops the solve that SpMV ran for nothing)
        launch_bicg_step2(st, n, d_r.p, d_v.p, d_s.p, precond, z, d_part2.p, s);
        if (k.generic) apply_preconditioner(d_s.p, z, s);
        OGL_TRY(dist_spmv(SPMV_PLAIN, z, nullptr, d_t.p,
                          SpmvDots{d_s.p, d_part0.p, d_part1.p}, s));
        FinArgs f3 = f2;
        f3.part_extra = d_part2.p;
        f3.n_sums = 3;
        f3.turn = enq;
        OGL_TRY(finalize(FIN_BICG_CHECK2_OMEGA, f3));
    } else {
        launch_bicg_step2(st, n, d_r.p, d_v.p, d_s.p, precond, z, d_part0.p, s);
        if (k.generic) apply_preconditioner(d_s.p, z, s);
        f1.turn = enq;
        OGL_TRY(finalize(FIN_BICG_CHECK2, f1));
        OGL_TRY(dist_spmv(SPMV_PLAIN, z, nullptr, d_t.p,
                          SpmvDots{d_s.p, d_part0.p, d_part1.p}, s));
        OGL_TRY(finalize(FIN_BICG_OMEGA, f2));
    }
    launch_bicg_step3(st, n, d_x.p, d_r.p, d_s.p, d_t.p, y, z, d_rr.p, d_part0.p,
                      d_part1.p, s, enq);
    OGL_TRY(finalize(FIN_CG_CHECK, k.chk));
    return OGL_OK;
}

// `count` turns into the stream, each in its solver's / system's shape
int ogl_solver::krylov_enqueue(KrylovRun &k, int count)
{
    for (int i = 0; i < count; ++i, ++k.enq) {
        const int enq = k.enq;
        const bool prof = k.prof_stride && enq % k.prof_stride == 0 && enq / k.prof_stride < k.prof_cap;
        const int pe = prof ? enq / k.prof_stride : -1;  // event pair of this turn
        if (k.gmres)
            OGL_TRY(turn_gmres(k, enq, pe));
        else if (k.bicg)
            OGL_TRY(k.bicg_fold ? turn_bicg_folded(k, enq, pe) : turn_bicg(k, enq, pe));
        else if (k.generic)
            OGL_TRY(turn_cg_generic(k, enq, pe));
        else if (k.fused2)
            OGL_TRY(turn_cg_two_launch(k, enq, pe));
        else if (k.fused)
            OGL_TRY(turn_cg_three_launch(k, enq, pe));
        else if (k.merged)
            OGL_TRY(turn_cg_merged(k, enq, pe));
        else
            OGL_TRY(turn_cg_five_launch(k, enq, pe));
    }
    return OGL_OK;
}

// GKOCG: gko::solver::Cg step order ([UPSTREAM]) with the OpenFOAM criterion
// evaluated on the device.  Per turn:
//   (z = M^-1 r, rho = r.z, sum|r|)  -> check -> p = z + (rho/prev_rho) p -> q = A p, beta = p.q
//   -> x += (rho/beta) p, r -= (rho/beta) q
// The host only enqueues; it looks at the stop flag one batch late, and kernels enqueued after
// the stop are no-ops, so x, r and the counters are exactly those of the stopping turn.
int ogl_solver::krylov_loop(KrylovRun &k)
{
    hipStream_t st = k.st;
    const bool fused = k.fused;
    auto poll_record = [&](int slot) -> int {
        OGL_HIP_CHECK(hipMemcpyAsync(&h_scal[slot], k.bicg_fold ? k.slot_s[k.cur] : k.s, sizeof(DevScalars),
                                     hipMemcpyDeviceToHost, st));
        OGL_HIP_CHECK(hipEventRecord(poll_ev[slot], st));
        return OGL_OK;
    };

    // The host never waits for the turn it has just enqueued: it looks at the stop flag of batch j
    // only after batch j+1 is in the queue.  Every rank sees the same flags (the norms are
    // all-reduced), hence enqueues the same number of batches and of RCCL calls.
    const int batch = k.bicg ? 8 : 16;

    // hipGraph replay of a full batch of single-rank GKOCG turns (property "hipGraph").  Nothing in the
    // captured launches depends on the turn or on the solve (criterion and flags live in the device
    // scalars); the key lists every pointer they do bake in.  For the 5-launch turn it does not pay on
    // MI355X / ROCm 7.2 (23.7 us per turn with plain stream launches against 24.2 us replayed at 262k rows,
    // 286.1 against 285.3 us at 10M rows: the gap between two dependent kernels is the device's dispatch
    // latency, not host launch cost) and stays off by default.
    const bool graphable = !k.gmres && !k.bicg && !k.generic && !k.multi && k.prof_cap == 0 &&
                           prop("hipGraph", fused ? 1.0 : 0.0) != 0.0;
    // (on by default for the folded 2- / 3-launch turns of small systems, where the host's launch rate shows: 32^3
    //  15.1 -> 13.0 us per 3-launch turn, 64^3 17.3 -> 16.7; the 5-launch turn of larger systems measures the same either way;
    //  a batch of 16 turns leaves the two p buffers of the 2-launch turn where it found them)
    auto enqueue_turns = [&](int count) -> int {
        // (the fused-finaliser turn: its first step_1x_fin differs from the later ones -- the first batch runs direct)
        if (!graphable || count != batch || (fused && k.enq == 0)) return krylov_enqueue(k, count);
        // the key: every view a captured launcher reads, hashed field by field (launch_key.hpp), the vectors and scalar
        // slots the turn kernels take, the turn's shape, and the pattern the layouts belong to (a rebuild with the same
        // sizes usually gets the same pointers back: 32x64x32 -> 64x32x32)
        KeyHasher kh;
        kh(k.n), kh(batch), kh(cfg.matrix_format), kh(use_sell()), kh(use_sym()), kh(use_symx()), kh(symx_fast), kh(s21_use);
        kh(k.fused), kh(k.fused2), kh(k.merged), kh(k.p0), kh(k.p1), kh(k.z_kept), kh(k.s), kh(k.s2), kh(pat_id);
        for (const void *v : {(const void *)d_p.p, (const void *)d_x.p, (const void *)d_r.p, (const void *)d_q.p,
                              (const void *)precond, (const void *)d_part0.p, (const void *)d_part1.p,
                              (const void *)d_part2.p, (const void *)d_history.p, (const void *)d_z.p, (const void *)d_p2.p})
            kh(v);
        visit(kh, csr());
        visit(kh, ell());
        if (sell_state == 1) visit(kh, sell());
        if (use_sym()) visit(kh, sym());
        if (use_symx()) visit(kh, symx());
        visit(kh, k.lead);
        const uint64_t key = kh.h;
        if (!cg_graph || key != cg_graph_key) {
            if (cg_graph) {
                (void)hipGraphExecDestroy(cg_graph);
                ledger::destroyed(ledger::GRAPH_EXEC);
            }
            cg_graph = nullptr;
            OGL_HIP_CHECK(hipStreamBeginCapture(st, hipStreamCaptureModeThreadLocal));
            const int before = k.enq;
            const int rc = krylov_enqueue(k, batch);
            k.enq = before;  // captured, not run
            hipGraph_t g = nullptr;
            const hipError_t e = hipStreamEndCapture(st, &g);
            if (rc != OGL_OK || e != hipSuccess) {
                if (g) (void)hipGraphDestroy(g);
                return rc != OGL_OK ? rc : fail(OGL_ERR_HIP, "stream capture failed: %s", hipGetErrorString(e));
            }
            const hipError_t ei = hipGraphInstantiate(&cg_graph, g, nullptr, nullptr, 0);
            (void)hipGraphDestroy(g);
            if (ei != hipSuccess) {
                cg_graph = nullptr;
                return fail(OGL_ERR_HIP, "hipGraphInstantiate failed: %s", hipGetErrorString(ei));
            }
            ledger::created(ledger::GRAPH_EXEC);
            cg_graph_key = key;
            props["hipGraphCaptures"] = prop("hipGraphCaptures", 0.0) + 1.0;
        }
        OGL_HIP_CHECK(hipGraphLaunch(cg_graph, st));
        k.enq += batch;
        return OGL_OK;
    };
    OGL_TRY(enqueue_turns(std::min(batch, k.max_turns - k.enq)));
    OGL_TRY(poll_record(0));
    for (int j = 0;; ++j) {
        const bool more = k.enq < k.max_turns;
        if (more) {
            OGL_TRY(enqueue_turns(std::min(batch, k.max_turns - k.enq)));
            OGL_TRY(poll_record((j + 1) & 1));
        }
        OGL_HIP_CHECK(hipEventSynchronize(poll_ev[j & 1]));
        if (h_scal[j & 1].stop) break;
        if (!more && k.folded()) break;  // (the check of the last enqueued turn is still to come: krylov_finish)
        if (!more) return fail(OGL_ERR_STATE, "criterion did not stop within maxIter + frequency");
    }
    return OGL_OK;
}

// The closing check of the folded turns, the pending x update, GMRES' final solve_krylov; history, perf, the
// properties the adaptive criterion of the next solve reads.
int ogl_solver::krylov_finish(KrylovRun &k, ogl_perf *perf)
{
    hipStream_t st = k.st;
    const int n = k.n, m = k.m;
    DevScalars *s = k.s, *s2 = k.s2;
    const bool bicg = k.bicg, gmres = k.gmres, fused = k.fused, bicg_fold = k.bicg_fold;
    if (fused)  // the check that closes the last turn run so far (a plain copy s -> s2 when the solve has stopped)
        launch_cg_step1x_fin(st, n, k.p_of_turn(k.enq), d_x.p, d_r.p, precond, s, s2, d_part0.p, d_part1.p, d_history.p, 0,
                             k.lead);
    if (bicg_fold) {  // the check that closes the last turn run so far (a plain copy of the scalars when the solve has stopped)
        launch_bicg_fold1(st, n, d_p.p, d_r.p, d_v.p, precond, k.y, k.slot_s[k.cur], k.slot_s[k.cur ^ 1], d_part0.p,
                          d_part1.p, d_history.p, k.lead);
        k.cur ^= 1;
    }
    OGL_HIP_CHECK(hipStreamSynchronize(st));
    OGL_HIP_CHECK(hipGetLastError());
    DevScalars fin;
    OGL_HIP_CHECK(hipMemcpy(&fin, bicg_fold ? k.slot_s[k.cur] : (fused ? s2 : s), sizeof(fin), hipMemcpyDeviceToHost));
    if (k.folded() && !fin.stop) return fail(OGL_ERR_STATE, "criterion did not stop within maxIter + frequency");
    if (fin.comm_error)
        return fail(OGL_ERR_COMM, "peer all-reduce timed out: a rank did not take part (check %d)",
                    fin.iter);
    if (fin.x_pending) {
        // the stop came with the check of the last enqueued turn: no step_1x followed to apply
        // that turn's x update
        launch_cg_step1x(st, n, k.p_of_turn(k.enq), d_x.p, d_r.p, precond, s);
        OGL_HIP_CHECK(hipStreamSynchronize(st));
        OGL_HIP_CHECK(hipGetLastError());
    }
    if (gmres) {
        // final solve_krylov on the (partial) cycle: Arnoldi steps done since the last restart
        const int steps = fin.iter - 1;
        const int cols = steps <= 0 ? 0 : (steps - 1) % m + 1;
        OGL_TRY(gmres_update_x(k, cols, nullptr));
        OGL_HIP_CHECK(hipStreamSynchronize(st));
        OGL_HIP_CHECK(hipGetLastError());
    }
    const double t_solve = now_ms() - k.t_start;
    history.clear();
    if (cfg.export_res) {
        history.resize(fin.iter);
        OGL_HIP_CHECK(hipMemcpy(history.data(), d_history.p, (size_t)fin.iter * sizeof(double),
                                hipMemcpyDeviceToHost));
    }
    float chk_ms = 0.f;
    OGL_HIP_CHECK(hipEventElapsedTime(&chk_ms, k.ev_chk[0], k.ev_chk[1]));

    // where the multi-rank turns of this solve waited (DevScalars, kernels.hpp; wall_clock64 counts 10 ns)
    props["haloWaits"] = (double)fin.halo_waits;
    props["haloWaitUs"] = (double)fin.halo_wait_ticks / 100.0;
    props["allreduceWaits"] = (double)fin.reduce_waits;
    props["allreduceWaitUs"] = (double)fin.reduce_wait_ticks / 100.0;
    props["peerSafeWaitInUse"] = (pat.non_local_nnz > 0 && peer_halo && peer_safe_wait()) ? 1.0 : 0.0;
    props["peerSharedDevice"] = reg->peer_shared_device ? 1.0 : 0.0;
    perf->initial_residual = fin.init_res;                  // lduLduBase.H:283
    perf->final_residual = fin.res;                         // :284
    perf->n_iterations = bicg ? fin.iter / 2 : fin.iter;    // :285, GKOCG.H:105-108, GKOBiCGStab.H:114
    perf->n_norm_evals = fin.n_evals;
    perf->norm_factor = fin.norm_factor;
    perf->t_solve_ms = t_solve;
    const int turns_done = bicg ? fin.iter / 2 : std::max(0, fin.iter - 1);
    perf->spmv_avg_ms = 0;
    perf->spmv_launches = 0;
    if (k.prof_cap) {
        double acc = 0;
        const int cnt = std::min((turns_done + k.prof_stride - 1) / k.prof_stride, k.prof_cap);
        for (int i = 0; i < cnt; ++i) {
            float ms = 0.f;
            OGL_HIP_CHECK(hipEventElapsedTime(&ms, prof_ev[2 * i], prof_ev[2 * i + 1]));
            acc += ms;
        }
        perf->spmv_launches = cnt;
        perf->spmv_avg_ms = cnt ? acc / cnt : 0.0;
    }

    // store_number_of_iterations + relative residual-evaluation cost (lduLduBase.H:286-293);
    // both are stored as labels, i.e. truncated (common.C:75-76,117-123).  The stored count is the
    // raw number of checks for every solver (GKOBiCGStab.H:98-103).
    props[k.is_final ? "prevSolveIters_final" : "prevSolveIters"] = fin.iter;
    const double time_per_iter = t_solve * 1e3 / std::max(perf->n_iterations, 1);
    const double res_norm_time = std::max(1e-3, (double)chk_ms * 1e3);
    double rel_cost = time_per_iter / res_norm_time;
    perf->t_res_norm_us = res_norm_time;
    perf->n_global_rows = k.n_global;
    if (reg->comm->multi()) {  // broadcast from rank 0 (:291-292) so every rank adapts alike
        double v = reg->comm->rank == 0 ? rel_cost : 0.0;
        OGL_HIP_CHECK(hipMemcpy(sums_ptr(s), &v, sizeof(double), hipMemcpyHostToDevice));
        OGL_TRY(reg->allreduce(sums_ptr(s), 1));
        OGL_HIP_CHECK(hipStreamSynchronize(st));
        OGL_HIP_CHECK(hipMemcpy(&rel_cost, sums_ptr(s), sizeof(double), hipMemcpyDeviceToHost));
    }
    props["_prev_solve"] = std::floor(rel_cost);
    return OGL_OK;
}

// solver->apply(b, x) on the resident vectors (lduLduBase.H:254-276)
int ogl_solver::apply_resident(ogl_perf *perf)
{
    if (!matrix_set) return fail(OGL_ERR_STATE, "solve before set_matrix");
    if (!x_resident || !b_resident) return fail(OGL_ERR_STATE, "rhs/solution not resident");
    OGL_HIP_CHECK(hipSetDevice(reg->device));
    ogl_perf local{};
    if (!perf) perf = &local;
    TraceRange trace("solve", field);
    {
        TraceRange trace_pc("init_preconditioner", field);
        OGL_TRY(init_preconditioner());
    }
    switch (cfg.solver) {
    case OGL_SOLVER_CG:
        return run_cg(perf);
    case OGL_SOLVER_BICGSTAB:
        return run_bicgstab(perf);
    case OGL_SOLVER_GMRES:
        return run_krylov(perf);
    default:
        return fail(OGL_ERR_UNSUPPORTED, "solver kind %d is not built", cfg.solver);
    }
}

// lduLduBase::solve_multi_gpu_impl (lduLduBase.H:189-308)
int ogl_solver::solve(const double *source, double *psi, ogl_perf *perf)
{
    if (!matrix_set) return fail(OGL_ERR_STATE, "solve before set_matrix");
    if (!source || !psi) return fail(OGL_ERR_INVALID, "source/psi is NULL");
    OGL_HIP_CHECK(hipSetDevice(reg->device));
    hipStream_t st = reg->stream;
    ogl_perf local{};
    if (!perf) perf = &local;
    *perf = ogl_perf{};
    const double t0 = now_ms();
    {
        TraceRange trace("upload_rhs_and_guess", field);
        if (!b_resident || cfg.update_rhs) {  // :217-226
            OGL_TRY(upload_vec(d_b, source));
            b_resident = true;
        }
        if (!x_resident || cfg.update_init_guess) {  // :228-237
            OGL_TRY(upload_vec(d_x, psi));
            x_resident = true;
        }
        if (cfg.scaling != 1.0) launch_scale(st, pat.n_rows, d_b.p, cfg.scaling);  // :242-252
        OGL_HIP_CHECK(hipStreamSynchronize(st));
    }
    perf->t_upload_ms = now_ms() - t0;
    perf->t_update_matrix_ms = t_update_matrix_ms;
    OGL_TRY(apply_resident(perf));
    const double t1 = now_ms();
    {
        TraceRange trace("copy_back", field);
        OGL_TRY(download_rows(psi, d_x.p));  // :278-279
    }
    perf->t_copy_back_ms = now_ms() - t1;
    return OGL_OK;
}

// `repeats` in-loop SpMVs (q = A b, fused dot) timed with HIP events on the solver's stream
int ogl_solver::time_spmv(int repeats, double *avg_ms)
{
    if (!matrix_set) return fail(OGL_ERR_STATE, "time_spmv before set_matrix");
    OGL_HIP_CHECK(hipSetDevice(reg->device));
    hipStream_t st = reg->stream;
    EventPair ev;
    OGL_HIP_CHECK(ev_create(&ev[0]));
    OGL_HIP_CHECK(ev_create(&ev[1]));
    hipEvent_t e0 = ev[0], e1 = ev[1];
    OGL_TRY(dist_spmv(SPMV_PLAIN, d_b.p, nullptr, d_q.p, SpmvDots{d_b.p, d_part0.p, nullptr}, nullptr));  // warm-up
    OGL_HIP_CHECK(hipEventRecord(e0, st));
    for (int i = 0; i < repeats; ++i) {
        // alternate the input so consecutive launches do not read what the last one wrote
        const double *x = (i & 1) ? d_r.p : d_b.p;
        OGL_TRY(dist_spmv(SPMV_PLAIN, x, nullptr, d_q.p, SpmvDots{x, d_part0.p, nullptr}, nullptr));
    }
    OGL_HIP_CHECK(hipEventRecord(e1, st));
    OGL_HIP_CHECK(hipEventSynchronize(e1));
    float ms = 0.f;
    OGL_HIP_CHECK(hipEventElapsedTime(&ms, e0, e1));
    *avg_ms = repeats > 0 ? (double)ms / repeats : 0.0;
    return OGL_OK;
}
