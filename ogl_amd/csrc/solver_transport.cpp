// solver_transport.cpp -- the peer mesh over xGMI (hipIpc mailboxes, halo puts) and the registry's all-reduce
// (ExecutorHandler.H:29-32,140-144,167-172 is what they stand in for).  See solver.hpp, solver_internal.hpp.
#include "solver_internal.hpp"

#include <algorithm>
#include <cmath>
#include <cstring>
#include <vector>

using namespace ogl;

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
