// solver.cpp -- see solver.hpp.  Citations: reference tree (hpsim/OGL @ 2024-10-16).
#include "solver_internal.hpp"

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

// The cells along a Hilbert curve through their centres (hilbert_order, host_matrix.cpp) on the device: bounding box on the
// host (one pass over 3 N doubles), keys + stable radix sort + inverse permutation on the device.  Same numbering as the host's.
int ogl_solver::curve_on_device(ogl_label n, const double *centres, std::vector<ogl_label> &new_id)
{
    new_id.clear();
    if (n < 2) return OGL_OK;
    hipStream_t st = reg->stream;
    double lo[3] = {centres[0], centres[1], centres[2]}, hi[3] = {centres[0], centres[1], centres[2]};
    for (ogl_label c = 0; c < n; ++c)
        for (int d = 0; d < 3; ++d) {
            lo[d] = std::min(lo[d], centres[3 * (size_t)c + d]);
            hi[d] = std::max(hi[d], centres[3 * (size_t)c + d]);
        }
    double ext = 0.0;
    for (int d = 0; d < 3; ++d) ext = std::max(ext, hi[d] - lo[d]);
    const double scale = ext > 0.0 ? 65535.0 / ext : 0.0;
    DevBuf<double> d_c;
    DevBuf<unsigned long long> keys, keys_out;
    DevBuf<int32_t> cells, cells_out, nid;
    DevBuf<unsigned char> temp;
    const size_t temp_bytes = hilbert_sort_temp_bytes(n);
    OGL_TRY(d_c.alloc(3 * (size_t)n, st));
    OGL_TRY(keys.alloc((size_t)n, st));
    OGL_TRY(keys_out.alloc((size_t)n, st));
    OGL_TRY(cells.alloc((size_t)n, st));
    OGL_TRY(cells_out.alloc((size_t)n, st));
    OGL_TRY(nid.alloc((size_t)n, st));
    OGL_TRY(temp.alloc(std::max<size_t>(temp_bytes, 16), st));
    OGL_TRY(reg->stager.h2d(d_c.p, centres, 3 * (size_t)n * sizeof(double), st));
    OGL_TRY(hilbert_order_device(st, n, d_c.p, lo, scale, keys.p, keys_out.p, cells.p, cells_out.p, temp.p, temp_bytes, nid.p));
    new_id.resize((size_t)n);
    OGL_TRY(reg->stager.d2h(new_id.data(), nid.p, (size_t)n * sizeof(int32_t), st));
    return OGL_OK;
}

// ... and the entries that would fall outside their chunk's window of packed columns along that curve (choose_numbering)
int ogl_solver::curve_far_on_device(const HostPattern &hp, const std::vector<ogl_label> &new_id,
                                    const std::vector<ogl_label> &old_of, int64_t &far)
{
    hipStream_t st = reg->stream;
    const int32_t N = hp.n_rows;
    DevBuf<int32_t> nid, old;
    DevBuf<unsigned long long> cnt;
    OGL_TRY(nid.alloc((size_t)N, st));
    OGL_TRY(old.alloc((size_t)N, st));
    OGL_TRY(cnt.alloc(1, st));  // (zero-filled)
    OGL_TRY(reg->stager.h2d(nid.p, new_id.data(), (size_t)N * sizeof(int32_t), st));
    OGL_TRY(reg->stager.h2d(old.p, old_of.data(), (size_t)N * sizeof(int32_t), st));
    launch_curve_far_count(st, N, d_row_ptrs.p, d_cols.p, nid.p, old.p, cnt.p);
    unsigned long long got = 0;
    OGL_TRY(reg->stager.d2h(&got, cnt.p, sizeof(got), st));
    far = (int64_t)got;
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

// The band of the device CSR arrays: the largest distance column - row when the pattern holds at most SYM_TABLE distinct
// ones (a structured mesh in its own numbering), else 0.  One small kernel per pattern (k_sym_distances), cached.
int ogl_solver::csr_band(int64_t *band)
{
    if (csr_band_pat != pat_id) {
        hipStream_t st = reg->stream;
        csr_band_rows = 0;
        DevBuf<int32_t> work;  // [table | flags]
        OGL_TRY(work.alloc(SYM_TABLE + SYM_FLAGS, st));
        int32_t io[SYM_TABLE + SYM_FLAGS];
        for (int j = 0; j < SYM_TABLE; ++j) io[j] = SYM_EMPTY;
        for (int j = 0; j < SYM_FLAGS; ++j) io[SYM_TABLE + j] = 0;
        OGL_HIP_CHECK(hipMemcpyAsync(work.p, io, sizeof(io), hipMemcpyHostToDevice, st));
        launch_sym_distances(st, pat.n_rows, d_row_ptrs.p, d_cols.p, work.p, work.p + SYM_TABLE);
        OGL_HIP_CHECK(hipMemcpyAsync(io, work.p, sizeof(io), hipMemcpyDeviceToHost, st));
        OGL_HIP_CHECK(hipStreamSynchronize(st));
        OGL_HIP_CHECK(hipGetLastError());
        if (!io[SYM_TABLE + SYM_FLAG_TOO_MANY])
            for (int j = 0; j < SYM_TABLE; ++j)
                if (io[j] != SYM_EMPTY) csr_band_rows = std::max<int64_t>(csr_band_rows, io[j]);
        csr_band_pat = pat_id;
    }
    *band = csr_band_rows;
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
            source_diag_valid = false;
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
            source_diag_valid = true;  // (d_vals' diagonal = the N doubles at src + diag_start)
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
        source_diag_valid = false;
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
            if (ldu.cell_centres && prop("renumberCurve", 1.0) != 0.0) {
                hooks.centres = ldu.cell_centres;
                if (built_on_device && prop("curveOnDevice", 1.0) != 0.0) {
                    hooks.curve = [&](ogl_label n, const double *centres, std::vector<ogl_label> &nid) {
                        return curve_on_device(n, centres, nid) == OGL_OK && !nid.empty();
                    };
                    hooks.curve_far = [&](const HostPattern &hp, const std::vector<ogl_label> &nid,
                                          const std::vector<ogl_label> &old, int64_t &far) {
                        return curve_far_on_device(hp, nid, old, far) == OGL_OK;
                    };
                }
            }
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
        // peak); the plain CSR-stream kernel takes the same time either way (195.7 / 195.8 us) but fetches 14 % less over the
        // fabric (1.19 -> 1.02 x the model's bytes, profiles/r06_pmc_nocompress_summary.json), so it gets the order too.
        // Property spmvBandRows: the band in rows, 0 = off, -1 (default) = the largest offset of a banded pattern (the
        // compressed layout's tables; for the CSR arrays the distance table of the half-storage set-up, csr_band()).
        int64_t band = (int64_t)prop("spmvBandRows", -1.0);
        if (band < 0) {
            const bool by_sell = cfg.matrix_format != OGL_FORMAT_ELL && use_sell() && !use_sym() && !use_symx();
            const bool by_csr = cfg.matrix_format != OGL_FORMAT_ELL && !use_sell() && !use_sym() && !use_symx();
            band = 0;
            if (by_sell) band = sell_band_rows;
            // (single rank: the order has not been run next to the halo waits of a multi-rank SpMV on the CSR arrays)
            if (by_csr && pat.n_rows >= SPMV_TUNE_MIN_ROWS && !reg->comm->multi()) OGL_TRY(csr_band(&band));
        }
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
