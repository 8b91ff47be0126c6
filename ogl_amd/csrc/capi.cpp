// capi.cpp -- extern "C" entry points of include/ogl_amd.h.  No exception crosses this file.
#include <algorithm>
#include <cstdio>
#include <cstring>
#include <new>
#include <string>
#include <vector>

#include "solver.hpp"

using namespace ogl;

#define OGL_GUARD_BEGIN try {
#define OGL_GUARD_END                                                         \
    }                                                                         \
    catch (const std::bad_alloc &) { return fail(OGL_ERR_HIP, "out of host memory"); } \
    catch (const std::exception &e) { return fail(OGL_ERR_INVALID, "exception: %s", e.what()); } \
    catch (...) { return fail(OGL_ERR_INVALID, "unknown exception"); }

extern "C" int ogl_registry_create(ogl_registry **out, int device_id, void *hip_stream)
{
    OGL_GUARD_BEGIN
    if (!out) return fail(OGL_ERR_INVALID, "out is NULL");
    *out = nullptr;
    int n_dev = 0;
    if (hipGetDeviceCount(&n_dev) != hipSuccess || n_dev <= 0)
        return fail(OGL_ERR_NO_DEVICE,
                    "no HIP device visible: libogl_amd has no CPU path (gfx950 required)");
    // ExecutorHandler.H:90-91: device_id_ % get_num_devices() -- rank 8 of a 2 x 8 run (or any
    // rank / ranksPerGPU beyond the node's device count) lands on a local device
    device_id = device_id < 0 ? 0 : device_id % n_dev;
    hipDeviceProp_t prop;
    OGL_HIP_CHECK(hipGetDeviceProperties(&prop, device_id));
    if (std::strncmp(prop.gcnArchName, "gfx950", 6) != 0)
        return fail(OGL_ERR_NO_DEVICE, "device %d is %s; the kernels are built for gfx950 only",
                    device_id, prop.gcnArchName);
    OGL_HIP_CHECK(hipSetDevice(device_id));
    auto reg = std::make_unique<ogl_registry>();
    reg->device = device_id;
    if (hip_stream) {
        reg->stream = static_cast<hipStream_t>(hip_stream);
    } else {
        OGL_HIP_CHECK(stream_create(&reg->stream));
        reg->own_stream = true;
    }
    reg->comm = std::make_unique<SelfComm>();
    OGL_TRY(reg->stager.init(size_t(16) << 20));  // x Stager::NBUF pinned buffers
    *out = reg.release();
    return OGL_OK;
    OGL_GUARD_END
}

extern "C" void ogl_registry_destroy(ogl_registry *reg) { delete reg; }

extern "C" int ogl_registry_set_host_comm(ogl_registry *reg, int32_t rank, int32_t n_ranks,
                                          ogl_allreduce_sum_fn allreduce,
                                          ogl_neighbour_exchange_fn exchange, void *user)
{
    OGL_GUARD_BEGIN
    if (!reg) return fail(OGL_ERR_INVALID, "registry is NULL");
    if (n_ranks < 1 || rank < 0 || rank >= n_ranks) return fail(OGL_ERR_INVALID, "bad rank/n_ranks");
    if (n_ranks > 1 && (!allreduce || !exchange))
        return fail(OGL_ERR_INVALID, "host communicator needs both callbacks");
    OGL_HIP_CHECK(hipSetDevice(reg->device));
    reg->comm = std::make_unique<HostComm>(rank, n_ranks, allreduce, exchange, user);
    return OGL_OK;
    OGL_GUARD_END
}

extern "C" int ogl_rccl_unique_id(void *id_out)
{
    OGL_GUARD_BEGIN
    if (!id_out) return fail(OGL_ERR_INVALID, "id_out is NULL");
    return RcclComm::unique_id(id_out);
    OGL_GUARD_END
}

extern "C" int ogl_registry_rccl_ready(ogl_registry *reg)
{
    OGL_GUARD_BEGIN
    if (!reg) return fail(OGL_ERR_INVALID, "registry is NULL");
    OGL_HIP_CHECK(hipSetDevice(reg->device));
    return RcclComm::library_ready();
    OGL_GUARD_END
}

extern "C" int ogl_registry_init_rccl(ogl_registry *reg, int32_t rank, int32_t n_ranks,
                                      const void *id)
{
    OGL_GUARD_BEGIN
    if (!reg || !id) return fail(OGL_ERR_INVALID, "registry/id is NULL");
    if (n_ranks < 1 || rank < 0 || rank >= n_ranks) return fail(OGL_ERR_INVALID, "bad rank/n_ranks");
    OGL_HIP_CHECK(hipSetDevice(reg->device));
    auto c = std::make_unique<RcclComm>();
    OGL_TRY(c->init(rank, n_ranks, id));
    // collective self-test over the links: a transport that does not deliver known numbers is refused here
    // (the caller keeps / falls back to the host-buffer transport in-process; nothing is re-exec'ed)
    OGL_TRY(c->self_test(reg->stream));
    reg->comm = std::move(c);
    return OGL_OK;
    OGL_GUARD_END
}

extern "C" int ogl_registry_peer_handle(ogl_registry *reg, void *handle_out)
{
    OGL_GUARD_BEGIN
    if (!reg || !handle_out) return fail(OGL_ERR_INVALID, "registry/handle_out is NULL");
    return reg->peer_export(handle_out);
    OGL_GUARD_END
}

extern "C" int ogl_registry_peer_connect(ogl_registry *reg, int32_t rank, int32_t n_ranks,
                                         const void *handles)
{
    OGL_GUARD_BEGIN
    if (!reg || !handles) return fail(OGL_ERR_INVALID, "registry/handles is NULL");
    if (n_ranks < 1 || rank < 0 || rank >= n_ranks) return fail(OGL_ERR_INVALID, "bad rank/n_ranks");
    if (!reg->comm || reg->comm->n_ranks != n_ranks || reg->comm->rank != rank)
        return fail(OGL_ERR_STATE, "peer_connect: set the transport (host or RCCL) with the same rank/n_ranks first");
    return reg->peer_connect(rank, n_ranks, handles);
    OGL_GUARD_END
}

extern "C" int ogl_registry_peer_disable(ogl_registry *reg)
{
    OGL_GUARD_BEGIN
    if (!reg) return fail(OGL_ERR_INVALID, "registry is NULL");
    reg->peer_ready = false;
    return OGL_OK;
    OGL_GUARD_END
}

extern "C" int ogl_registry_comm_info(ogl_registry *reg, ogl_comm_info *info)
{
    if (!reg || !info) return fail(OGL_ERR_INVALID, "NULL argument");
    const std::string name = reg->comm ? reg->comm->name() : "self";
    info->transport = name == "rccl" ? 2 : (name == "host-buffer" ? 1 : 0);
    info->rank = reg->comm ? reg->comm->rank : 0;
    info->n_ranks = reg->comm ? reg->comm->n_ranks : 1;
    info->ranks_seen = reg->comm ? reg->comm->ranks_seen() : 1;
    info->peer_mesh = reg->peer_ready ? 1 : 0;
    info->device = reg->device;
    return OGL_OK;
}

extern "C" int ogl_registry_mem_info(ogl_registry *reg, int64_t *free_bytes, int64_t *total_bytes)
{
    OGL_GUARD_BEGIN
    if (!reg) return fail(OGL_ERR_INVALID, "registry is NULL");
    OGL_HIP_CHECK(hipSetDevice(reg->device));
    size_t f = 0, t = 0;
    OGL_HIP_CHECK(hipMemGetInfo(&f, &t));
    if (free_bytes) *free_bytes = (int64_t)f;
    if (total_bytes) *total_bytes = (int64_t)t;
    return OGL_OK;
    OGL_GUARD_END
}

template <class T>
static int fetch(ogl_solver *s, T *dst, const T *dev, size_t count)
{
    if (!dst || count == 0) return OGL_OK;
    return s->reg->stager.d2h(dst, dev, count * sizeof(T), s->reg->stream);
}

static int check_config(const ogl_config &c)
{
    if (c.solver < OGL_SOLVER_CG || c.solver > OGL_SOLVER_GMRES)
        return fail(OGL_ERR_INVALID, "unknown solver kind %d", c.solver);
    if (c.matrix_format < OGL_FORMAT_COO || c.matrix_format > OGL_FORMAT_ELL)
        return fail(OGL_ERR_INVALID, "Matrix format %d not supported", c.matrix_format);  // CsrMatrixWrapper.H:159
    if (c.ranks_per_gpu != 1)
        return fail(OGL_ERR_UNSUPPORTED, "ranksPerGPU %d: only 1 works (Vector.H:78-82)", c.ranks_per_gpu);
    if (c.max_iter < 0 || c.eval_frequency < 1 || c.norm_eval_limit < 1)
        return fail(OGL_ERR_INVALID, "maxIter/evalFrequency/normEvalLimit out of range");
    return OGL_OK;
}

extern "C" int ogl_solver_get_or_create(ogl_registry *reg, const char *field_name,
                                        const ogl_config *cfg, ogl_solver **out)
{
    OGL_GUARD_BEGIN
    if (!reg || !field_name || !cfg || !out) return fail(OGL_ERR_INVALID, "NULL argument");
    OGL_TRY(check_config(*cfg));
    auto &slot = reg->solvers[field_name];
    if (!slot) {  // "initialising <name>"  (Base.H:98-113)
        slot = std::make_unique<ogl_solver>();
        slot->reg = reg;
        slot->field = field_name;
    }
    slot->cfg = *cfg;  // the dictionary is re-read at every construction
    *out = slot.get();
    return OGL_OK;
    OGL_GUARD_END
}

extern "C" int ogl_solver_set_matrix(ogl_solver *s, const ogl_ldu_view *ldu)
{
    OGL_GUARD_BEGIN
    if (!s || !ldu) return fail(OGL_ERR_INVALID, "NULL argument");
    return s->set_matrix(*ldu);
    OGL_GUARD_END
}

extern "C" int ogl_solver_set_matrix_like(ogl_solver *s, const ogl_ldu_view *ldu, ogl_solver *donor)
{
    OGL_GUARD_BEGIN
    if (!s || !ldu) return fail(OGL_ERR_INVALID, "NULL argument");
    // (the donor is looked at during this call only: the pointer is dropped again on every way out)
    struct Lend {
        ogl_solver *s;
        ~Lend() { s->share_from = nullptr; }
    } lend{s};
    s->share_from = (donor && donor != s && donor->reg == s->reg) ? donor : nullptr;
    return s->set_matrix(*ldu);
    OGL_GUARD_END
}

extern "C" int ogl_solver_solve(ogl_solver *s, const ogl_scalar *source, ogl_scalar *psi,
                                ogl_perf *perf)
{
    OGL_GUARD_BEGIN
    if (!s) return fail(OGL_ERR_INVALID, "NULL solver");
    return s->solve(source, psi, perf);
    OGL_GUARD_END
}

extern "C" int ogl_solver_history(ogl_solver *s, double *out, int32_t capacity)
{
    if (!s || (!out && capacity > 0)) return fail(OGL_ERR_INVALID, "NULL argument");
    const int n = std::min<int>(capacity, (int)s->history.size());
    std::copy(s->history.begin(), s->history.begin() + n, out);
    return n;
}

// MatrixMarket export (common.C:31-58): what test/data_validation.py of the reference inspects
extern "C" int ogl_solver_export_system(ogl_solver *s, const char *directory)
{
    OGL_GUARD_BEGIN
    if (!s || !directory) return fail(OGL_ERR_INVALID, "NULL argument");
    if (!s->matrix_set) return fail(OGL_ERR_STATE, "no matrix yet");
    OGL_HIP_CHECK(hipSetDevice(s->reg->device));
    OGL_TRY(s->download_local_pattern(s->pat));
    const HostPattern &p = s->pat;
    const std::string base = std::string(directory) + "/" + s->field;
    std::vector<double> vals((size_t)p.local_nnz), nl((size_t)p.non_local_nnz), b((size_t)p.n_rows);
    OGL_TRY(fetch(s, vals.data(), s->d_vals.p, vals.size()));
    OGL_TRY(fetch(s, nl.data(), s->d_nl_vals.p, nl.size()));
    if (s->b_resident && p.n_rows > 0) OGL_TRY(s->download_rows(b.data(), s->d_b.p));
    auto open = [&](const std::string &fn) -> FILE * {
        FILE *f = std::fopen(fn.c_str(), "w");
        if (!f) fail(OGL_ERR_INVALID, "cannot write %s", fn.c_str());
        return f;
    };
    // The files are written in the CALLER's cell numbering, row-major sorted (what
    // test/data_validation.py expects), whatever numbering the device copy uses.
    struct Entry {
        int32_t col;
        double val;
    };
    std::vector<Entry> row;
    FILE *f = open(base + "_A_local.mtx");
    if (!f) return OGL_ERR_INVALID;
    std::fprintf(f, "%%%%MatrixMarket matrix coordinate real general\n%d %d %d\n", p.n_rows, p.n_rows,
                 p.local_nnz);
    for (int32_t r = 0; r < p.n_rows; ++r) {
        const int32_t k = p.renumbered() ? p.new_id[(size_t)r] : r;
        row.clear();
        for (int32_t e = p.row_ptrs[(size_t)k]; e < p.row_ptrs[(size_t)k + 1]; ++e)
            row.push_back({p.renumbered() ? p.old_of[(size_t)p.cols[(size_t)e]] : p.cols[(size_t)e], vals[(size_t)e]});
        if (p.renumbered())
            std::stable_sort(row.begin(), row.end(), [](const Entry &a, const Entry &b) { return a.col < b.col; });
        for (const Entry &e : row) std::fprintf(f, "%d %d %.15g\n", r + 1, e.col + 1, e.val);
    }
    std::fclose(f);
    f = open(base + "_A_non_local.mtx");
    if (!f) return OGL_ERR_INVALID;
    std::fprintf(f, "%%%%MatrixMarket matrix coordinate real general\n%d %d %d\n", p.n_rows,
                 p.non_local_nnz, p.non_local_nnz);  // N x H, CsrMatrixWrapper.H:185-188
    {
        std::vector<int32_t> ord((size_t)p.non_local_nnz);
        for (size_t e = 0; e < ord.size(); ++e) ord[e] = (int32_t)e;
        auto old_row = [&](int32_t e) { return p.renumbered() ? p.old_of[(size_t)p.nl_rows[(size_t)e]] : p.nl_rows[(size_t)e]; };
        if (p.renumbered())
            std::stable_sort(ord.begin(), ord.end(), [&](int32_t a, int32_t b) {
                return old_row(a) != old_row(b) ? old_row(a) < old_row(b) : p.nl_cols[(size_t)a] < p.nl_cols[(size_t)b];
            });
        for (int32_t e : ord)
            std::fprintf(f, "%d %d %.15g\n", old_row(e) + 1, p.nl_cols[(size_t)e] + 1, nl[(size_t)e]);
    }
    std::fclose(f);
    if (s->b_resident) {
        f = open(base + "_rhs_b_.mtx");
        if (!f) return OGL_ERR_INVALID;
        std::fprintf(f, "%%%%MatrixMarket matrix array real general\n%d 1\n", p.n_rows);
        for (double v : b) std::fprintf(f, "%.15g\n", v);
        std::fclose(f);
    }
    if (!s->history.empty()) {
        f = open(base + "_res_norms.mtx");
        if (!f) return OGL_ERR_INVALID;
        std::fprintf(f, "%%%%MatrixMarket matrix array real general\n%zu 1\n", s->history.size());
        for (double v : s->history) std::fprintf(f, "%.17g\n", v);
        std::fclose(f);
    }
    return OGL_OK;
    OGL_GUARD_END
}

extern "C" int ogl_solver_get_property(ogl_solver *s, const char *key, double *value)
{
    if (!s || !key || !value) return fail(OGL_ERR_INVALID, "NULL argument");
    if (!std::strcmp(key, "deviceBytesInUse") || !std::strcmp(key, "pinnedBytesInUse")) {  // (process-wide: csrc/ledger.hpp)
        ogl_memory_ledger l;
        ogl::ledger::snapshot(&l);
        *value = (double)(key[0] == 'd' ? l.device_bytes : l.pinned_bytes);
        return OGL_OK;
    }
    auto it = s->props.find(key);
    if (it == s->props.end()) return fail(OGL_ERR_INVALID, "no property %s", key);
    *value = it->second;
    return OGL_OK;
}

extern "C" int ogl_solver_set_property(ogl_solver *s, const char *key, double value)
{
    OGL_GUARD_BEGIN
    if (!s || !key) return fail(OGL_ERR_INVALID, "NULL argument");
    s->props[key] = value;
    return OGL_OK;
    OGL_GUARD_END
}

extern "C" int ogl_solver_apply_resident(ogl_solver *s, ogl_perf *perf)
{
    OGL_GUARD_BEGIN
    if (!s) return fail(OGL_ERR_INVALID, "NULL solver");
    if (perf) *perf = ogl_perf{};
    return s->apply_resident(perf);
    OGL_GUARD_END
}

static int upload_into(ogl_solver *s, DevBuf<double> ogl_solver::*which, bool ogl_solver::*flag,
                       const double *src)
{
    if (!s) return fail(OGL_ERR_INVALID, "NULL solver");
    if (!s->have_pattern) return fail(OGL_ERR_STATE, "upload before set_matrix");
    OGL_HIP_CHECK(hipSetDevice(s->reg->device));
    OGL_TRY(s->upload_vec(s->*which, src));
    OGL_HIP_CHECK(hipStreamSynchronize(s->reg->stream));
    s->*flag = true;
    return OGL_OK;
}

extern "C" int ogl_solver_upload_solution(ogl_solver *s, const ogl_scalar *psi)
{
    OGL_GUARD_BEGIN
    return upload_into(s, &ogl_solver::d_x, &ogl_solver::x_resident, psi);
    OGL_GUARD_END
}

extern "C" int ogl_solver_upload_rhs(ogl_solver *s, const ogl_scalar *source)
{
    OGL_GUARD_BEGIN
    return upload_into(s, &ogl_solver::d_b, &ogl_solver::b_resident, source);
    OGL_GUARD_END
}

extern "C" int ogl_solver_download_solution(ogl_solver *s, ogl_scalar *psi)
{
    OGL_GUARD_BEGIN
    if (!s || !psi) return fail(OGL_ERR_INVALID, "NULL argument");
    if (!s->x_resident) return fail(OGL_ERR_STATE, "no resident solution");
    OGL_HIP_CHECK(hipSetDevice(s->reg->device));
    return s->download_rows(psi, s->d_x.p);
    OGL_GUARD_END
}

extern "C" int ogl_solver_spmv(ogl_solver *s, const ogl_scalar *x, ogl_scalar *y)
{
    OGL_GUARD_BEGIN
    if (!s || !x || !y) return fail(OGL_ERR_INVALID, "NULL argument");
    if (!s->matrix_set) return fail(OGL_ERR_STATE, "spmv before set_matrix");
    OGL_HIP_CHECK(hipSetDevice(s->reg->device));
    hipStream_t st = s->reg->stream;
    (void)st;
    OGL_TRY(s->upload_rows(s->d_w.p, x));
    OGL_TRY(s->dist_spmv(SPMV_PLAIN, s->d_w.p, nullptr, s->d_q.p, SpmvDots{}, nullptr));
    OGL_TRY(s->download_rows(y, s->d_q.p));
    OGL_HIP_CHECK(hipGetLastError());
    // a neighbour whose halo values did not arrive within the time-out leaves the LOCAL product behind and raises
    // comm_error in the device scalars (device_common.hpp halo_fused_add / kernels_comm.hip k_halo_finish): never hand that out silently
    if (s->pat.non_local_nnz > 0 && s->peer_halo) {
        DevScalars sc;
        OGL_HIP_CHECK(hipMemcpy(&sc, s->d_scal.p, sizeof(sc), hipMemcpyDeviceToHost));
        if (sc.comm_error) {
            sc.comm_error = 0;
            sc.stop = 0;
            OGL_HIP_CHECK(hipMemcpy(s->d_scal.p, &sc, sizeof(sc), hipMemcpyHostToDevice));
            return fail(OGL_ERR_COMM, "halo exchange timed out: a neighbour's values did not arrive (ranks that share "
                                      "one device can starve each other's put kernels: every waiting workgroup holds a "
                                      "slot the producer needs)");
        }
    }
    return OGL_OK;
    OGL_GUARD_END
}

extern "C" int ogl_solver_time_spmv(ogl_solver *s, int32_t repeats, double *avg_ms)
{
    OGL_GUARD_BEGIN
    if (!s || !avg_ms) return fail(OGL_ERR_INVALID, "NULL argument");
    return s->time_spmv(repeats, avg_ms);
    OGL_GUARD_END
}

extern "C" int ogl_solver_reduce(ogl_solver *s, int32_t op, const ogl_scalar *a, const ogl_scalar *b,
                                 double *out)
{
    OGL_GUARD_BEGIN
    if (!s || !a || !out || (op == 0 && !b)) return fail(OGL_ERR_INVALID, "NULL argument");
    if (!s->have_pattern) return fail(OGL_ERR_STATE, "reduce before set_matrix");
    OGL_HIP_CHECK(hipSetDevice(s->reg->device));
    hipStream_t st = s->reg->stream;
    const int n = s->pat.n_rows;
    const size_t bytes = (size_t)n * sizeof(double);
    OGL_TRY(s->reg->stager.h2d(s->d_w.p, a, bytes, st));
    if (op == 0) {
        OGL_TRY(s->reg->stager.h2d(s->d_q.p, b, bytes, st));
        launch_partials_dot(st, n, s->d_w.p, s->d_q.p, s->d_part0.p, nullptr);
    } else if (op == 1) {
        launch_partials_norm1(st, n, s->d_w.p, s->d_part0.p);
    } else if (op == 2) {
        launch_partials_sum(st, n, s->d_w.p, s->d_part0.p);
    } else {
        return fail(OGL_ERR_INVALID, "unknown reduction op %d", op);
    }
    FinArgs fa{};
    fa.part[0] = s->d_part0.p;
    fa.n_part = (int)n_chunks(n);
    fa.n_sums = 1;
    OGL_TRY(s->finalize(FIN_RAW, fa));
    DevScalars h;
    OGL_HIP_CHECK(hipMemcpyAsync(&h, s->d_scal.p, sizeof(h), hipMemcpyDeviceToHost, st));
    OGL_HIP_CHECK(hipStreamSynchronize(st));
    OGL_HIP_CHECK(hipGetLastError());
    *out = h.sums[0];
    return OGL_OK;
    OGL_GUARD_END
}

extern "C" int ogl_reduction_chunk_rows(void) { return CHUNK_ROWS; }

extern "C" int ogl_solver_matrix_dims(ogl_solver *s, ogl_matrix_dims *d)
{
    if (!s || !d) return fail(OGL_ERR_INVALID, "NULL argument");
    if (!s->have_pattern) return fail(OGL_ERR_STATE, "no matrix yet");
    d->n_rows = s->pat.n_rows;
    d->local_nnz = s->pat.local_nnz;
    d->non_local_nnz = s->pat.non_local_nnz;
    d->n_halo = s->pat.non_local_nnz;
    d->n_neighbours = (ogl_label)s->pat.target_ids.size();
    d->n_send = (ogl_label)s->pat.send_idxs.size();
    return OGL_OK;
}

extern "C" int ogl_solver_get_local_matrix(ogl_solver *s, ogl_label *row_ptrs, ogl_label *cols,
                                           ogl_label *ldu_mapping, ogl_scalar *coeffs)
{
    OGL_GUARD_BEGIN
    if (!s) return fail(OGL_ERR_INVALID, "NULL solver");
    if (!s->matrix_set) return fail(OGL_ERR_STATE, "no matrix yet");
    OGL_HIP_CHECK(hipSetDevice(s->reg->device));
    const size_t nnz = (size_t)s->pat.local_nnz;
    OGL_TRY(fetch(s, row_ptrs, s->d_row_ptrs.p, (size_t)s->pat.n_rows + 1));
    OGL_TRY(fetch(s, cols, s->d_cols.p, nnz));
    OGL_TRY(fetch(s, ldu_mapping, s->d_ldu_mapping.p, nnz));
    OGL_TRY(fetch(s, coeffs, s->d_vals.p, nnz));
    return OGL_OK;
    OGL_GUARD_END
}

extern "C" int ogl_solver_get_non_local_matrix(ogl_solver *s, ogl_label *rows, ogl_label *cols,
                                               ogl_label *ldu_mapping, ogl_scalar *coeffs)
{
    OGL_GUARD_BEGIN
    if (!s) return fail(OGL_ERR_INVALID, "NULL solver");
    if (!s->matrix_set) return fail(OGL_ERR_STATE, "no matrix yet");
    OGL_HIP_CHECK(hipSetDevice(s->reg->device));
    const size_t nnz = (size_t)s->pat.non_local_nnz;
    if (rows) std::copy(s->pat.nl_rows.begin(), s->pat.nl_rows.end(), rows);
    if (ldu_mapping) std::copy(s->pat.nl_ldu_mapping.begin(), s->pat.nl_ldu_mapping.end(), ldu_mapping);
    OGL_TRY(fetch(s, cols, s->d_nl_cols.p, nnz));
    OGL_TRY(fetch(s, coeffs, s->d_nl_vals.p, nnz));
    return OGL_OK;
    OGL_GUARD_END
}

extern "C" int ogl_solver_get_renumbering(ogl_solver *s, ogl_label *new_id)
{
    if (!s) return fail(OGL_ERR_INVALID, "NULL solver");
    if (!s->have_pattern) return fail(OGL_ERR_STATE, "no matrix yet");
    if (!s->pat.renumbered()) return 0;
    if (new_id) std::copy(s->pat.new_id.begin(), s->pat.new_id.end(), new_id);
    return 1;
}

extern "C" int ogl_solver_get_comm_pattern(ogl_solver *s, ogl_label *target_ids,
                                           ogl_label *target_sizes, ogl_label *send_idxs)
{
    if (!s) return fail(OGL_ERR_INVALID, "NULL solver");
    if (!s->have_pattern) return fail(OGL_ERR_STATE, "no matrix yet");
    if (target_ids) std::copy(s->pat.target_ids.begin(), s->pat.target_ids.end(), target_ids);
    if (target_sizes) std::copy(s->pat.target_sizes.begin(), s->pat.target_sizes.end(), target_sizes);
    if (send_idxs) std::copy(s->pat.send_idxs.begin(), s->pat.send_idxs.end(), send_idxs);
    return OGL_OK;
}
