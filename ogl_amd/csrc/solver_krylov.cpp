// solver_krylov.cpp -- the distributed product, the finalisers and the Krylov drivers: plan, prepare, the turns of every
// solver x turn shape, the loop with its hipGraph replay, finish (lduLduBase.H:189-308; [UPSTREAM] step order: SURVEY 8 a19-a21).
// See solver.hpp, solver_internal.hpp.
#include "launch_key.hpp"
#include "solver_internal.hpp"

#include <algorithm>
#include <cmath>
#include <cstring>
#include <vector>

using namespace ogl;

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
    uint32_t lead_tag = 0;  // GKOGMRES: tag of the last leader launch (host-side sequence, from 1 per solve)
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
    PRing ring{};  // three-launch leader turn with ring.k p buffers: x touched every ring.k-th turn (k_cg_step1x_fin)
    double *p_of_turn(int turn) const  // p that turn `turn` reads
    {
        if (ring.k > 0) return ring.b[turn % ring.k];
        return (merged && (turn & 1)) ? p1 : p0;
    }
    PRing ring_of_turn(int turn) const
    {
        PRing r = ring;
        r.phase = ring.k > 0 ? turn % ring.k : 0;
        return r;
    }
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
    // (GKOCG with a materialised z -- block Jacobi, ISAI -- takes the leader finalisation too: the same two kernels with
    //  z in r's place, the preconditioner's launches between step_2r and the next turn's check)
    const bool lead_ok = lead_any && !bicg && !gmres;
    bool fused = k.fused = !bicg && !gmres && ((small && !generic) || lead_ok);
    k.lead = LeadBox{};
    k.s2 = s + 1;
    // ... and the same for small single-rank GKOBiCGStab systems: three finalisers folded into step_1 / step_2 / step_3
    // (k_bicg_fold1/2/3: 5 launches per turn instead of 8, plus the preconditioner's own)
    // (larger systems: the same five launches with the leader finalisation, any preconditioner -- the 2 M-row momentum
    //  systems of configs[2] spend a tenth of a turn in three single-workgroup launches and their gaps)
    k.bicg_fold = bicg && (small || lead_any) && prop("bicgFold", 1.0) != 0.0;
    // ... and for small single-rank GKOGMRES systems the finaliser between two Gram-Schmidt links is folded into the next
    // link's kernel (k_gmres_mgs_fold: one launch per link instead of two)
    // (larger systems: with the leader finalisation on request only, property gmresLead -- a Gram-Schmidt link moves 16 N
    //  bytes, and the wait for the leaders costs it what the finaliser launch did: 216^3 GMRES(30) 1068 against 1047 us per turn)
    k.gmres_fold = gmres && (small || (lead_any && prop("gmresLead", 0.0) != 0.0)) && prop("gmresFold", 1.0) != 0.0;
    k.slot_s[0] = s;
    k.slot_s[1] = k.s2;
    props["fusedFinalizersInUse"] = ((fused || k.bicg_fold || k.gmres_fold) && small) ? 1.0 : 0.0;
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
    if ((lead_ok && fused) || (lead_any && (k.bicg_fold || k.gmres_fold))) {
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
    props["fusedFinalizersInUse"] = ((fused || k.bicg_fold || k.gmres_fold) && small) ? 1.0 : 0.0;
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
    // (x every K-th turn: the leader turn of three launches -- not the merged kernel, which forms p_new at its gathers;
    //  property deferX = number of p buffers, 2 | 4 | 8, anything below 2 or deferX2 0: x every turn, p in place.  At 10 M
    //  rows two buffers give +1.2 % over none, four the same as two, eight lose 6 % -- the pending directions push the
    //  turn's other vectors out of the Infinity Cache: profiles/r06_defer_ab.txt)
    int ring_k = 0;
    if (k.lead.box != nullptr && k.fused && !k.fused2 && prop("deferX2", 1.0) != 0.0) {
        const double want = prop("deferX", 2.0);
        ring_k = want >= 8.0 ? 8 : want >= 4.0 ? 4 : want >= 2.0 ? 2 : 0;
    }
    props["deferXInUse"] = (double)ring_k;
    props["deferX2InUse"] = ring_k ? 1.0 : 0.0;
    if (merged || ring_k) OGL_TRY(d_p2.alloc((size_t)n + 2, st));
    for (int i = 2; i < ring_k; ++i) OGL_TRY(d_pring[i - 2].alloc((size_t)n + 2, st));
    if (merged && precond) OGL_TRY(d_z.alloc((size_t)n + 2, st));
    k.n_halo = (size_t)pat.non_local_nnz;
    if (k.merged_halo) {  // old p at the halo columns, two buffers like p itself; p = 0 before the first turn
        OGL_TRY(d_p_halo.alloc(2 * k.n_halo + 2, st));
        OGL_HIP_CHECK(hipMemsetAsync(d_p_halo.p, 0, d_p_halo.n * sizeof(double), st));
    }
    k.p0 = d_p.p;
    k.p1 = d_p2.p;
    k.ring.k = ring_k;
    for (int i = 0; i < ring_k; ++i) k.ring.b[i] = i == 0 ? d_p.p : i == 1 ? d_p2.p : d_pring[i - 2].p;
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
                                  j > 0 ? k.gm_h(j - 1, it) : nullptr, d_V.p + (size_t)j * ldv, pin, pout, s, k.lead,
                                  ++k.lead_tag);
            std::swap(pin, pout);
        }
        launch_gmres_mgs_fold(st, n, nx, v_it, k.gm_h(it, it), nullptr, pin, pout, s, k.lead, ++k.lead_tag);
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

// ... with the leader finalisation (single rank, more than 1,024 chunks): [check of the previous turn + pending x update +
// step_1 on the materialised z] | SpMV | [beta + step_2r] | M^-1 (z and the partials of r.z); scalars s -> s2 -> s
int ogl_solver::turn_cg_generic_led(KrylovRun &k, int enq, int pe)
{
    hipStream_t st = k.st;
    const int n = k.n;
    if (enq == 0) OGL_HIP_CHECK(hipEventRecord(k.ev_chk[0], st));
    double *p_new = k.p_of_turn(enq + 1);
    launch_cg_step1x_fin(st, n, k.p_of_turn(enq), d_x.p, d_z.p, nullptr, k.s, k.s2, d_part0.p, d_part1.p, d_history.p,
                         enq == 0 ? 1 : 0, k.lead, p_new, k.ring_of_turn(enq));
    if (enq == 0) OGL_HIP_CHECK(hipEventRecord(k.ev_chk[1], st));
    if (pe >= 0) OGL_HIP_CHECK(hipEventRecord(prof_ev[2 * pe], st));
    OGL_TRY(dist_spmv(SPMV_PLAIN, p_new, nullptr, d_q.p, SpmvDots{p_new, d_part2.p, nullptr}, k.s2));
    if (pe >= 0) OGL_HIP_CHECK(hipEventRecord(prof_ev[2 * pe + 1], st));
    launch_cg_step2r_fin(st, n, d_r.p, d_q.p, nullptr, d_part0.p, d_part1.p, k.s2, k.s, d_part2.p, nullptr, k.lead);
    apply_preconditioner(d_r.p, d_z.p, k.s, d_part0.p);  // z = M^-1 r and the partials of r.z (over step_2r's r.r)
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
    double *p_new = k.p_of_turn(enq + 1);
    launch_cg_step1x_fin(st, n, k.p_of_turn(enq), d_x.p, d_r.p, precond, k.s, k.s2, d_part0.p, d_part1.p, d_history.p,
                         enq == 0 ? 1 : 0, k.lead, p_new, k.ring_of_turn(enq));
    if (enq == 0) OGL_HIP_CHECK(hipEventRecord(k.ev_chk[1], st));
    if (pe >= 0) OGL_HIP_CHECK(hipEventRecord(prof_ev[2 * pe], st));
    OGL_TRY(dist_spmv(SPMV_PLAIN, p_new, nullptr, d_q.p, SpmvDots{p_new, d_part2.p, nullptr}, k.s2));
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
            OGL_TRY(k.fused ? turn_cg_generic_led(k, enq, pe) : turn_cg_generic(k, enq, pe));
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
        kh(k.fused), kh(k.fused2), kh(k.merged), kh(k.ring.k), kh(k.p0), kh(k.p1), kh(k.z_kept), kh(k.s), kh(k.s2), kh(pat_id);
        for (const void *v : {(const void *)d_p.p, (const void *)d_x.p, (const void *)d_r.p, (const void *)d_q.p,
                              (const void *)precond, (const void *)d_part0.p, (const void *)d_part1.p,
                              (const void *)d_part2.p, (const void *)d_history.p, (const void *)d_z.p, (const void *)d_p2.p})
            kh(v);
        for (int i = 0; i < k.ring.k; ++i) kh((const void *)k.ring.b[i]);
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
        launch_cg_step1x_fin(st, n, k.p_of_turn(k.enq), d_x.p, k.generic ? d_z.p : d_r.p, k.generic ? nullptr : precond, s, s2,
                             d_part0.p, d_part1.p, d_history.p, 0, k.lead, k.p_of_turn(k.enq + 1), k.ring_of_turn(k.enq));
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
    if (fin.comm_error && !k.multi && k.lead.box)
        return fail(OGL_ERR_STATE, "leader finalisation timed out: the sums of a turn were not published within "
                    "leadTimeoutS = %g s (check %d)", prop("leadTimeoutS", 10.0), fin.iter);
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
