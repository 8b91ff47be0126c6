// kernels_krylov.hip -- vector steps of GKOCG / GKOBiCGStab / GKOGMRES, per-chunk partials and the finalisers
// (geometry, reduction tree and the -ffp-contract=off rule: device_common.hpp)
#include "device_common.hpp"

namespace ogl {

namespace {

__global__ __launch_bounds__(BLOCK) void k_scale(int n, double *__restrict__ v, double f)
{
    const int i = blockIdx.x * BLOCK + threadIdx.x;
    if (i < n) v[i] = v[i] * f;
}

__global__ __launch_bounds__(BLOCK) void k_fill_xbar(int n, double *__restrict__ v,
                                                     const DevScalars *s)
{
    const int i = blockIdx.x * BLOCK + threadIdx.x;
    if (i < n) v[i] = s->xbar;
}

enum PartialOp { P_SUM = 0, P_DOT = 1, P_NORM1 = 2 };
template <int OP>
__global__ __launch_bounds__(BLOCK) void k_partials(int n, int n_chunks,
                                                    const double *__restrict__ a,
                                                    const double *__restrict__ b,
                                                    double *__restrict__ part,
                                                    const DevScalars *gate,
                                                    const int *__restrict__ chunk_list)
{
    __shared__ double slot[N_WAVES];
    if (gate && gate->stop) return;
    // chunk_list: only these chunks (the ones holding boundary rows, whose fused partials of the
    // local SpMV are stale once the non-local part has been added)
    const int chunk = chunk_list ? chunk_list[blockIdx.x] : (int)blockIdx.x;
    const RowPair r = my_rows(chunk, n);
    const double2 va = ld2(a, r);
    double d = 0.0;
    if (OP == P_SUM) {
        if (r.n > 0) d += va.x;
        if (r.n > 1) d += va.y;
    } else if (OP == P_NORM1) {
        if (r.n > 0) d += fabs(va.x);
        if (r.n > 1) d += fabs(va.y);
    } else {
        const double2 vb = ld2(b, r);
        if (r.n > 0) d += va.x * vb.x;
        if (r.n > 1) d += va.y * vb.y;
    }
    const double s = block_sum(d, slot);
    if (threadIdx.x == 0) part[chunk] = s;
    (void)n_chunks;
}

// StoppingCriterion.C:53-61: t = b - Axref ; e = |t - r| + |t|
__global__ __launch_bounds__(BLOCK) void k_partials_normfactor(int n, const double *__restrict__ b,
                                                               const double *__restrict__ w,
                                                               const double *__restrict__ r,
                                                               double *__restrict__ part)
{
    __shared__ double slot[N_WAVES];
    const int chunk = blockIdx.x;
    const RowPair rp = my_rows(chunk, n);
    const double2 vb = ld2(b, rp), vw = ld2(w, rp), vr = ld2(r, rp);
    double d = 0.0;
    if (rp.n > 0) {
        const double t = vb.x - 1.0 * vw.x;
        const double p2 = fabs(t);
        d += fabs(fabs(t - 1.0 * vr.x) + 1.0 * p2);
    }
    if (rp.n > 1) {
        const double t = vb.y - 1.0 * vw.y;
        const double p2 = fabs(t);
        d += fabs(fabs(t - 1.0 * vr.y) + 1.0 * p2);
    }
    const double s = block_sum(d, slot);
    if (threadIdx.x == 0) part[chunk] = s;
}

// z = M^-1 r (scalar Jacobi: r * inv_diag; identity: r), rho partial = sum r z, norm partial = sum |r|
__global__ __launch_bounds__(BLOCK) void k_cg_rho_norm(int n, const double *__restrict__ r,
                                                       const double *__restrict__ inv_diag,
                                                       double *__restrict__ part_rho,
                                                       double *__restrict__ part_norm,
                                                       const DevScalars *gate)
{
    __shared__ double slot[N_WAVES];
    if (gate && gate->stop) return;
    const int chunk = blockIdx.x;
    const RowPair rp = my_rows(chunk, n);
    const double2 vr = ld2(r, rp);
    double2 vz = vr;
    if (inv_diag) {
        const double2 vi = ld2(inv_diag, rp);
        vz.x = vr.x * vi.x;
        vz.y = vr.y * vi.y;
    }
    double d = 0.0, a = 0.0;
    if (rp.n > 0) {
        d += vr.x * vz.x;
        a += fabs(vr.x);
    }
    if (rp.n > 1) {
        d += vr.y * vz.y;
        a += fabs(vr.y);
    }
    const double s0 = block_sum(d, slot);
    const double s1 = block_sum(a, slot);
    if (threadIdx.x == 0) {
        part_rho[chunk] = s0;
        part_norm[chunk] = s1;
    }
}

// step_1: p = z + (rho / prev_rho) * p   (tmp = 0 when prev_rho == 0)
__global__ __launch_bounds__(BLOCK) void k_cg_step1(int n, double *__restrict__ p,
                                                    const double *__restrict__ r,
                                                    const double *__restrict__ inv_diag,
                                                    const DevScalars *s)
{
    if (s->stop) return;
    const double rho = s->rho, prev = s->prev_rho;
    const double tmp = (prev == 0.0) ? 0.0 : rho / prev;
    const RowPair rp = my_rows(blockIdx.x, n);
    double2 vz = ld2(r, rp);
    if (inv_diag) {
        const double2 vi = ld2(inv_diag, rp);
        vz.x = vz.x * vi.x;
        vz.y = vz.y * vi.y;
    }
    double2 vp = ld2(p, rp);
    vp.x = vz.x + tmp * vp.x;
    vp.y = vz.y + tmp * vp.y;
    st2(p, rp, vp);
}

// step_2: if (beta != 0) { t = rho / beta ; x += t p ; r -= t q } -- then the next turn's
// z = M^-1 r, rho = r.z and sum|r| partials, fused so r is not re-read (K5+K6+K8).
__global__ __launch_bounds__(BLOCK) void k_cg_step2(int n, double *__restrict__ x,
                                                    double *__restrict__ r,
                                                    const double *__restrict__ p,
                                                    const double *__restrict__ q,
                                                    const double *__restrict__ inv_diag,
                                                    double *__restrict__ part_rho,
                                                    double *__restrict__ part_norm,
                                                    const DevScalars *s)
{
    __shared__ double slot[N_WAVES];
    if (s->stop) return;
    const double rho = s->rho, beta = s->beta;
    const int chunk = blockIdx.x;
    const RowPair rp = my_rows(chunk, n);
    double2 vr = ld2(r, rp);
    if (beta != 0.0) {
        const double t = rho / beta;
        double2 vx = ld2_stream(x, rp);  // x: once per turn; q: last use of the turn
        const double2 vp = ld2(p, rp), vq = ld2_stream(q, rp);
        vx.x += t * vp.x;
        vx.y += t * vp.y;
        vr.x -= t * vq.x;
        vr.y -= t * vq.y;
        st2_stream(x, rp, vx);
        st2(r, rp, vr);
    }
    double2 vz = vr;
    if (inv_diag) {
        const double2 vi = ld2(inv_diag, rp);
        vz.x = vr.x * vi.x;
        vz.y = vr.y * vi.y;
    }
    double d = 0.0, a = 0.0;
    if (rp.n > 0) {
        d += vr.x * vz.x;
        a += fabs(vr.x);
    }
    if (rp.n > 1) {
        d += vr.y * vz.y;
        a += fabs(vr.y);
    }
    const double s0 = block_sum(d, slot);
    const double s1 = block_sum(a, slot);
    if (threadIdx.x == 0) {
        part_rho[chunk] = s0;
        part_norm[chunk] = s1;
    }
}

// The same two steps with the x update DEFERRED by one turn, so that p is read once per turn instead
// of twice (80 N instead of 88 N bytes per turn with scalar Jacobi):
//   step_2r (turn j)  : r -= t_j q ; partials of the next rho and sum|r|        (x, p untouched)
//   step_1x (turn j+1): x += t_j p  with the OLD p, then p = z + (rho/prev_rho) p
// t_j = rho_j / beta_j is formed from the same two scalars as in step_2 (after the check that closed
// turn j they sit in prev_rho and beta), so x receives the same bits, one kernel later.  When that
// check stops the solve, the step_1x of turn j+1 still applies the pending update (it recognises
// its turn by iter == turn + 1: no check runs after the stop) and leaves p alone; the host flushes
// with an extra step_1x when the stop came after the last enqueued turn.

template <bool PUT>
__global__ __launch_bounds__(BLOCK) void k_cg_step1x(int n, double *__restrict__ p,
                                                     double *__restrict__ x,
                                                     const double *__restrict__ r,
                                                     const double *__restrict__ inv_diag,
                                                     const DevScalars *s, HaloPutFused put)
{
    const int stop = s->stop;
    const bool pending = s->x_pending != 0;
    if (stop && !pending) return;
    const RowPair rp = my_rows(blockIdx.x, n);
    double2 vp = ld2(p, rp);
    if (pending) {
        const double beta = s->beta;
        if (beta != 0.0) {
            const double t = s->prev_rho / beta;
            double2 vx = ld2_stream(x, rp);  // x is touched once per turn
            vx.x += t * vp.x;
            vx.y += t * vp.y;
            st2_stream(x, rp, vx);
        }
    }
    if (stop) return;
    const double rho = s->rho, prev = s->prev_rho;
    const double tmp = (prev == 0.0) ? 0.0 : rho / prev;
    double2 vz = ld2_stream(r, rp);  // r and inv_diag: last use of this turn
    if (inv_diag) {
        const double2 vi = ld2_stream(inv_diag, rp);
        vz.x = vz.x * vi.x;
        vz.y = vz.y * vi.y;
    }
    vp.x = vz.x + tmp * vp.x;
    vp.y = vz.y + tmp * vp.y;
    st2(p, rp, vp);
    if (PUT) {  // the halo values of the SpMV that follows
        __shared__ double ps[CHUNK_ROWS];
        __shared__ int last;
        halo_put_chunk(put, blockIdx.x, vp.x, vp.y, ps, &last);
    }
}

// PUT (multi-rank merged turn): the z of this chunk's send rows goes to the neighbours, whose next merged kernel
// forms p_new = z + tmp p_old at its halo columns itself (halo_fused_add<.., TURN>)
template <bool PUT>
__global__ __launch_bounds__(BLOCK) void k_cg_step2r(int n, double *__restrict__ r,
                                                     const double *__restrict__ q,
                                                     const double *__restrict__ inv_diag,
                                                     double *__restrict__ part_rho,
                                                     double *__restrict__ part_norm,
                                                     const DevScalars *s, double *__restrict__ z_out,
                                                     HaloPutFused put)
{
    __shared__ double slot[N_WAVES];
    if (s->stop) return;
    const double rho = s->rho, beta = s->beta;
    const int chunk = blockIdx.x;
    const RowPair rp = my_rows(chunk, n);
    double2 vr = ld2(r, rp);
    if (beta != 0.0) {
        const double t = rho / beta;
        const double2 vq = ld2_stream(q, rp);  // q: last use of this turn
        vr.x -= t * vq.x;
        vr.y -= t * vq.y;
        st2(r, rp, vr);
    }
    double2 vz = vr;
    if (inv_diag) {
        const double2 vi = ld2(inv_diag, rp);
        vz.x = vr.x * vi.x;
        vz.y = vr.y * vi.y;
    }
    if (z_out) st2(z_out, rp, vz);  // (kept for the gathers of k_cg_turn_sym_big)
    double d = 0.0, a = 0.0;
    if (rp.n > 0) {
        d += vr.x * vz.x;
        a += fabs(vr.x);
    }
    if (rp.n > 1) {
        d += vr.y * vz.y;
        a += fabs(vr.y);
    }
    const double s0 = block_sum(d, slot);
    const double s1 = block_sum(a, slot);
    if (threadIdx.x == 0) {
        part_rho[chunk] = s0;
        part_norm[chunk] = s1;
    }
    if (PUT) {
        __shared__ double ps[CHUNK_ROWS];
        __shared__ int last;
        halo_put_chunk(put, chunk, vz.x, vz.y, ps, &last);
    }
}

// ------------------------------------------------------------------------------------------
// Small systems (a turn of a 64^3 case is 5 dependent launches of ~4.5 us for ~6 us of memory time): the two
// single-workgroup finalisers of a GKOCG turn are folded into the kernels that consume their results.  Every
// workgroup of step_1x / step_2r first reduces the (few hundred) per-chunk partials ITSELF -- with 256 threads
// walking the 1024-thread tree of k_finalize, so the sums have the same bits -- and runs the scalar logic on its
// own copy of the solver scalars; workgroup 0 stores the new scalars.  The scalars ping-pong between two slots
// (a kernel reads `sin`, writes `sout`), so that no workgroup can see them half-way.  Turn = 3 launches:
//   [check of the previous turn + pending x update + step_1]  ->  SpMV  ->  [beta + step_2r]
// ------------------------------------------------------------------------------------------

// LEAD (systems of more than FUSED_FIN_MAX_CHUNKS chunks): the first 16 workgroups are the finaliser's 16 wavefronts and
// publish their sums; every workgroup fetches them, adds them and runs the same logic (leader finalisation, device_common.hpp)
template <bool LEAD, int K>
__global__ __launch_bounds__(BLOCK) void k_cg_step1x_fin(int n, double *p, double *__restrict__ x,
                                                         const double *__restrict__ r,
                                                         const double *__restrict__ inv_diag,
                                                         const DevScalars *sin, DevScalars *sout,
                                                         const double *__restrict__ part_rho,
                                                         const double *__restrict__ part_norm, int n_part,
                                                         double *history, int first, LeadBox lead,
                                                         double *p_out, PRing ring)
{
    // K > 0 (a ring of K p buffers; p = ring.b[phase], p_out = ring.b[(phase + 1) % K]): x is touched by every K-th head
    // only.  A head at ring position phase != 0 leaves its term t_j p_j of x pending: t_j goes into the scalars'
    // t_ring[phase], p_j stays intact in its buffer, which nobody writes before the ring comes round.  The head at
    // position 0 adds the K - 1 pending terms, oldest first (the oldest p sits in ITS p_out: read before it is
    // overwritten), then its own -- ((x + t_(j-K+1) p_(j-K+1)) + ...) + t_j p_j, the bits of K single updates -- so that
    // a turn moves 162 / K MB of x instead of 162 at 10 M rows, plus one read of each pending p.  A head that stops the
    // solve adds what is pending at once.  K == 0: every turn, p in place.
    __shared__ double red[2 * FIN_WAVES];
    __shared__ double sh[4];
    __shared__ int sh_stop;
    __shared__ double lead_words[LEAD ? LEAD_BOX_WORDS / 2 : 1];
    __shared__ double lead_stage[LEAD ? LEAD_STAGE : 1];
    __shared__ int lead_timed_out;
    const uint32_t seq = LEAD ? sin->launch_seq : 0u;
    // (the leaders first of all: their partial loads go out ahead of the chip's row loads, and nothing below -- not even
    //  the stop flag -- sits between the launch and the sums everybody else will wait for; after a stop nobody polls)
    if (LEAD) lead_leaders<2>(lead, seq, part_rho, part_norm, nullptr, n_part, lead_stage);
    // everything this workgroup will need is asked for at once -- the scalars, the partials and its own rows of
    // p, x, r, 1/d: one memory round trip instead of three in a row (scalars -> partials -> vectors).  The
    // scalars are read field by field into registers (a private copy of the struct would live in scratch memory,
    // and a kernel with scratch costs more to dispatch than the finaliser launch this is meant to save).
    const int stopped = sin->stop;
    const double s_rho = sin->rho, s_beta = sin->beta, s_nf = sin->norm_factor, s_init = sin->init_res;
    const int phase = K > 0 ? ring.phase : 0;
    const bool defers = K > 0 && phase != 0;  // this head leaves its term of x pending
    const unsigned s_pending = K > 0 && !first ? (unsigned)sin->defer_valid : 0u;
    constexpr int KQ = K > 0 ? K : 1;
    double s_t[KQ];
#pragma unroll
    for (int i = 1; i < KQ; ++i) s_t[i] = sin->t_ring[i];
    const int s_iter = sin->iter, s_evals = sin->n_evals;
    const double c_tol = sin->crit.tolerance, c_rel = sin->crit.rel_tol;
    const int c_min = sin->crit.min_iter, c_max = sin->crit.max_iter, c_freq = sin->crit.frequency,
              c_exp = sin->crit.export_res;
    static_assert(sizeof(DevScalars) % 8 == 0, "copied as 8-byte words");
    if (blockIdx.x == 0 && threadIdx.x < sizeof(DevScalars) / 8)  // fields this kernel leaves alone
        reinterpret_cast<unsigned long long *>(sout)[threadIdx.x] =
            reinterpret_cast<const unsigned long long *>(sin)[threadIdx.x];
    const RowPair rp = my_rows(blockIdx.x, n);
    double2 vp, vx, vz, vi, vq[KQ];
    vi.x = vi.y = 1.0;
    // (the terms pending at ring positions 1 .. K-1: the head at position 0 asks for all of them with its other rows)
    auto load_pending = [&](int upto) {
#pragma unroll
        for (int i = 1; i < KQ; ++i)
            if (i < upto && ((s_pending >> i) & 1u)) vq[i] = ld2(ring.b[i], rp);
    };
    const bool early = !LEAD || lead.early_loads != 0;
    // this head's own term of x (t_j p_j)
    const bool own_term = !first && s_beta != 0.0;
    if (early) {
        vp = ld2(p, rp);
        if (!defers) {  // (a deferring head leaves x alone -- unless its check stops the solve)
            vx = ld2_stream(x, rp);
            load_pending(KQ);
        }
        vz = ld2_stream(r, rp);
        if (inv_diag) vi = ld2_stream(inv_diag, rp);
    }
    // x after this head (when it does not defer): the pending terms of positions 1 .. upto-1 first, then its own -- the
    // order of single updates
    auto update_x = [&](int upto) {
        bool any = own_term;
#pragma unroll
        for (int i = 1; i < KQ; ++i)
            if (i < upto && ((s_pending >> i) & 1u)) {
                vx.x += s_t[i] * vq[i].x;
                vx.y += s_t[i] * vq[i].y;
                any = true;
            }
        if (own_term) {
            const double t = s_rho / s_beta;
            vx.x += t * vp.x;
            vx.y += t * vp.y;
        }
        if (any) st2_stream(x, rp, vx);
    };
    double pv[2][FIN_VT];
    if (!LEAD) load_partials_as_finaliser<2>(part_rho, part_norm, n_part, pv);
    if (stopped) return;  // (the solve has ended: workgroup 0 has handed the scalars on, nothing else to do)
    double v[2] = {0.0, 0.0};
    if (LEAD) {
        // x += t_j p of the turn this check will close needs nothing the leaders compute (prev_rho = the incoming rho,
        // beta as it stands): it goes out while the mailbox is awaited -- same scalars, same bits, one store less behind the wait
        if (early && !defers) update_x(KQ);
        if (!lead_wait(lead, 4 * FIN_WAVES, seq, lead_words, &lead_timed_out)) {
            if (threadIdx.x == 0) sout->comm_error = sout->stop = 1;
            return;
        }
        if (!early) {
            vp = ld2(p, rp);
            if (!defers) {
                vx = ld2_stream(x, rp);
                load_pending(KQ);
            }
            vz = ld2_stream(r, rp);
            if (inv_diag) vi = ld2_stream(inv_diag, rp);
        }
        if (threadIdx.x == 0) {
            v[0] = lead_total(lead_words, 0);
            v[1] = lead_total(lead_words, 1);
        }
    } else {
        reduce_partials_as_finaliser<2>(pv, n_part, red, v);  // (its barriers order the copy above before the stores below)
    }
    if (threadIdx.x == 0) {
        // FIN_CG_CHECK: swap(prev_rho, rho) of the previous turn, then criterion_check (StoppingCriterion.C:71-151)
        const double prev_rho = s_rho, rho = v[0];
        int iter = s_iter, n_evals = s_evals, stop = 0;
        double init_res = s_init, res = 0.0;
        bool evaluated = false;
        if (iter > 0 && iter < c_min) {           // :77-81
            iter += 1;
        } else if (iter % c_freq != 0) {          // :84-87
            iter += 1;
        } else {
            evaluated = true;
            n_evals += 1;
            res = v[1];
            if (iter == 0) init_res = res / s_nf;  // :102-111
            res /= s_nf;                           // :113
            if (c_exp && history && blockIdx.x == 0) history[iter] = res;  // :115-117
            if (iter >= c_max) stop = 1;                                   // :124
            if (res < c_tol) stop = 1;                                     // :128
            if (c_rel > 0 && res < c_rel * init_res) stop = 1;             // :132-136
            iter += 1;                                                     // :143
        }
        sh[0] = s_beta;
        sh[1] = prev_rho;
        sh[2] = rho;
        sh_stop = stop;
        if (blockIdx.x == 0) {
            sout->prev_rho = prev_rho;
            sout->rho = rho;
            sout->iter = iter;
            sout->x_pending = 0;
            if (evaluated) {
                sout->n_evals = n_evals;
                sout->init_res = init_res;
                sout->res = res;
            }
            if (stop) sout->stop = 1;
            if (LEAD) sout->launch_seq = seq + 1;
            // what is pending after this head (nothing when it has updated x itself or stops)
            sout->defer_valid = (defers && !stop) ? (int)(s_pending | (own_term ? 1u << phase : 0u)) : 0;
            if (defers && own_term) sout->t_ring[phase] = s_rho / s_beta;
        }
    }
    __syncthreads();
    const double prev = sh[1], rho = sh[2];
    const int stop = sh_stop;
    // x += t_j p of the turn this check closed (same scalars, same bits as step_2) -- unless it went out before the wait
    if (!defers && !(LEAD && early)) update_x(KQ);
    if (defers && stop) {  // a deferring head that ends the solve: the terms pending so far and its own go in now
        vx = ld2_stream(x, rp);
        load_pending(phase);
        update_x(phase);
    }
    if (stop) return;
    const double tmp = (prev == 0.0) ? 0.0 : rho / prev;
    if (inv_diag) {
        vz.x = vz.x * vi.x;
        vz.y = vz.y * vi.y;
    }
    vp.x = vz.x + tmp * vp.x;
    vp.y = vz.y + tmp * vp.y;
    st2(p_out, rp, vp);
}

template <bool LEAD>
__global__ __launch_bounds__(BLOCK) void k_cg_step2r_fin(int n, double *__restrict__ r,
                                                         const double *__restrict__ q,
                                                         const double *__restrict__ inv_diag,
                                                         double *__restrict__ part_rho,
                                                         double *__restrict__ part_norm, const DevScalars *sin,
                                                         DevScalars *sout, const double *__restrict__ part_beta,
                                                         int n_part, double *__restrict__ z_out, LeadBox lead)
{
    __shared__ double red[2 * FIN_WAVES];
    __shared__ double sh[2];
    __shared__ double slot[2 * N_WAVES];
    __shared__ double lead_words[LEAD ? LEAD_BOX_WORDS / 2 : 1];
    __shared__ double lead_stage[LEAD ? LEAD_STAGE : 1];
    __shared__ int lead_timed_out;
    const uint32_t seq = LEAD ? sin->launch_seq : 0u;
    if (LEAD) lead_leaders<1>(lead, seq, part_beta, nullptr, nullptr, n_part, lead_stage);  // (as step_1x_fin)
    // (all loads up front and the scalars field by field, as in step_1x_fin)
    const int stopped = sin->stop;
    const double s_rho = sin->rho;
    if (blockIdx.x == 0 && threadIdx.x < sizeof(DevScalars) / 8)
        reinterpret_cast<unsigned long long *>(sout)[threadIdx.x] =
            reinterpret_cast<const unsigned long long *>(sin)[threadIdx.x];
    const int chunk = blockIdx.x;
    const RowPair rp = my_rows(chunk, n);
    double2 vr, vq, vi;
    vi.x = vi.y = 1.0;
    const bool early = !LEAD || lead.early_loads != 0;
    if (early) {
        vr = ld2(r, rp);
        vq = ld2_stream(q, rp);  // q: last use of this turn
        if (inv_diag) vi = ld2(inv_diag, rp);
    }
    double pv[2][FIN_VT];
    if (!LEAD) load_partials_as_finaliser<1>(part_beta, nullptr, n_part, pv);
    if (stopped) return;
    double v[2] = {0.0, 0.0};
    if (LEAD) {
        if (!lead_wait(lead, 2 * FIN_WAVES, seq, lead_words, &lead_timed_out)) {
            if (threadIdx.x == 0) sout->comm_error = sout->stop = 1;
            return;
        }
        if (!early) {
            vr = ld2(r, rp);
            vq = ld2_stream(q, rp);
            if (inv_diag) vi = ld2(inv_diag, rp);
        }
        if (threadIdx.x == 0) v[0] = lead_total(lead_words, 0);
    } else {
        reduce_partials_as_finaliser<1>(pv, n_part, red, v);
    }
    if (threadIdx.x == 0) {
        sh[0] = s_rho;
        sh[1] = v[0];
        if (blockIdx.x == 0) {
            sout->beta = v[0];  // FIN_BETA
            if (LEAD) sout->launch_seq = seq + 1;
        }
    }
    __syncthreads();
    const double rho = sh[0], beta = sh[1];
    if (beta != 0.0) {
        const double t = rho / beta;
        vr.x -= t * vq.x;
        vr.y -= t * vq.y;
        st2(r, rp, vr);
    }
    double2 vz = vr;
    if (inv_diag) {
        vz.x = vr.x * vi.x;
        vz.y = vr.y * vi.y;
    }
    if (z_out) st2(z_out, rp, vz);  // (the 2-launch turn gathers z at the columns of its rows)
    double d = 0.0, a = 0.0;
    if (rp.n > 0) {
        d += vr.x * vz.x;
        a += fabs(vr.x);
    }
    if (rp.n > 1) {
        d += vr.y * vz.y;
        a += fabs(vr.y);
    }
    block_sum2(d, a, slot);
    if (threadIdx.x == 0) {
        part_rho[chunk] = d;
        part_norm[chunk] = a;
    }
}

// ------------------------------------------------------------------------------------------
// BiCGStab steps ([UPSTREAM] bicgstab::step_1 / step_2 / step_3 / finalize)
// ------------------------------------------------------------------------------------------
// step_1: p = r + (rho/prev_rho * alpha/omega) (p - omega v)   [p = r when prev_rho*omega == 0];
// then y = M^-1 p (scalar Jacobi; with the identity y aliases p and is not written)
__global__ __launch_bounds__(BLOCK) void k_bicg_step1(int n, double *__restrict__ p,
                                                      const double *__restrict__ r,
                                                      const double *__restrict__ v,
                                                      const double *__restrict__ inv_diag,
                                                      double *__restrict__ y, const DevScalars *s)
{
    if (s->stop) return;
    const double rho = s->rho, prev = s->prev_rho, alpha = s->alpha, omega = s->omega;
    const RowPair rp = my_rows(blockIdx.x, n);
    // r, p, v, inv_diag: a whole SpMV passes before any of them is touched again -> streamed past the caches;
    // y is gathered by the SpMV that follows and stays cached
    const double2 vr = ld2_stream(r, rp);
    double2 vp = vr;
    if (prev * omega != 0.0) {
        const double tmp = rho / prev * alpha / omega;
        const double2 po = ld2_stream(p, rp), vv = ld2_stream(v, rp);
        vp.x = vr.x + tmp * (po.x - omega * vv.x);
        vp.y = vr.y + tmp * (po.y - omega * vv.y);
    }
    if (inv_diag) st2_stream(p, rp, vp); else st2(p, rp, vp);  // (without a preconditioner p itself is the SpMV's input)
    if (inv_diag) {
        const double2 vi = ld2_stream(inv_diag, rp);
        double2 vy;
        vy.x = vp.x * vi.x;
        vy.y = vp.y * vi.y;
        st2(y, rp, vy);
    }
}

// step_2: s = r - alpha v (alpha = rho/beta from the finaliser; s = r when beta == 0); z = M^-1 s;
// partial of sum|s| for the mid-turn criterion check
__global__ __launch_bounds__(BLOCK) void k_bicg_step2(int n, const double *__restrict__ r,
                                                      const double *__restrict__ v,
                                                      double *__restrict__ sv,
                                                      const double *__restrict__ inv_diag,
                                                      double *__restrict__ z,
                                                      double *__restrict__ part_norm,
                                                      const DevScalars *s)
{
    __shared__ double slot[N_WAVES];
    if (s->stop) return;
    const double alpha = s->alpha, beta = s->beta;
    const int chunk = blockIdx.x;
    const RowPair rp = my_rows(chunk, n);
    double2 vs = ld2_stream(r, rp);  // (as in step_1: r, v and inv_diag are not touched again before an SpMV has passed)
    if (beta != 0.0) {
        const double2 vv = ld2_stream(v, rp);
        vs.x = vs.x - alpha * vv.x;
        vs.y = vs.y - alpha * vv.y;
    }
    st2(sv, rp, vs);  // (s is read by the SpMV that follows, for the fused t.s: stays cached)
    if (inv_diag) {
        const double2 vi = ld2_stream(inv_diag, rp);
        double2 vz;
        vz.x = vs.x * vi.x;
        vz.y = vs.y * vi.y;
        st2(z, rp, vz);
    }
    double a = 0.0;
    if (rp.n > 0) a += fabs(vs.x);
    if (rp.n > 1) a += fabs(vs.y);
    const double s1 = block_sum(a, slot);
    if (threadIdx.x == 0) part_norm[chunk] = s1;
}

// step_3: x += alpha y + omega z ; r = s - omega t ; then the partials of the next turn's
// rho = rr.r and of sum|r|
__global__ __launch_bounds__(BLOCK) void k_bicg_step3(int n, double *__restrict__ x,
                                                      double *__restrict__ r,
                                                      const double *__restrict__ sv,
                                                      const double *__restrict__ t,
                                                      const double *__restrict__ y,
                                                      const double *__restrict__ z,
                                                      const double *__restrict__ rr,
                                                      double *__restrict__ part_rho,
                                                      double *__restrict__ part_norm,
                                                      const DevScalars *s, int turn)
{
    __shared__ double slot[N_WAVES];
    if (s->stop) {
        // bicgstab::finalize: x += alpha y, only on the turn whose mid-step check stopped the solver
        if (s->stop_phase == 1 && s->stop_turn == turn) {
            const double alpha = s->alpha;
            const RowPair rp = my_rows(blockIdx.x, n);
            double2 vx = ld2(x, rp);
            const double2 vy = ld2(y, rp);
            vx.x += alpha * vy.x;
            vx.y += alpha * vy.y;
            st2(x, rp, vx);
        }
        return;
    }
    const double alpha = s->alpha, omega = s->omega;
    const int chunk = blockIdx.x;
    const RowPair rp = my_rows(chunk, n);
    // x, y, z, s, t, rr: touched here for the last (x, rr: only) time of the turn -> streamed past the caches
    double2 vx = ld2_stream(x, rp);
    const double2 vy = ld2_stream(y, rp), vz = ld2_stream(z, rp), vs = ld2_stream(sv, rp), vt = ld2_stream(t, rp),
                  vrr = ld2_stream(rr, rp);
    vx.x += alpha * vy.x + omega * vz.x;
    vx.y += alpha * vy.y + omega * vz.y;
    double2 vr;
    vr.x = vs.x - omega * vt.x;
    vr.y = vs.y - omega * vt.y;
    st2_stream(x, rp, vx);
    st2(r, rp, vr);
    double d = 0.0, a = 0.0;
    if (rp.n > 0) {
        d += vrr.x * vr.x;
        a += fabs(vr.x);
    }
    if (rp.n > 1) {
        d += vrr.y * vr.y;
        a += fabs(vr.y);
    }
    const double s0 = block_sum(d, slot);
    const double s1 = block_sum(a, slot);
    if (threadIdx.x == 0) {
        part_rho[chunk] = s0;
        part_norm[chunk] = s1;
    }
}

// ------------------------------------------------------------------------------------------
// GKOBiCGStab on small systems (<= FUSED_FIN_MAX_CHUNKS chunks, one rank): the three single-workgroup finalisers of
// a turn folded into the step kernels that consume their results, as for GKOCG (k_cg_step1x_fin): every workgroup
// reduces the per-chunk partials itself -- 256 threads walking the 1024-thread tree of k_finalize, same bits -- and runs
// the scalar logic on its own copy of the scalars; workgroup 0 stores them.  The scalars ping-pong between two slots
// (a kernel reads `sin`, writes `sout`).  Turn:
//   [check of the previous turn + step_1] -> (M^-1) -> SpMV -> [alpha + step_2] -> (M^-1) -> SpMV
//   -> [mid-turn check + omega + step_3 (or bicgstab::finalize when that check stops the solve)]
// 5 launches instead of 8 (+ the preconditioner's own).  A kernel that reads partials never writes the arrays it
// reads -- another workgroup may still be reducing them -- so the turn uses six partial arrays.
// ------------------------------------------------------------------------------------------
// FIN_CG_CHECK + step_1.  The closing check of a solve is one more launch of this kernel (the step it then takes on
// p is harmless: the solve has stopped, or fails with "did not stop").
// LEAD (any number of chunks): leader finalisation as in k_cg_step1x_fin<true>
template <bool LEAD>
__global__ __launch_bounds__(BLOCK) void k_bicg_fold1(int n, double *__restrict__ p, const double *__restrict__ r,
                                                      const double *__restrict__ v,
                                                      const double *__restrict__ inv_diag, double *__restrict__ y,
                                                      const DevScalars *sin, DevScalars *sout,
                                                      const double *__restrict__ part_rho,
                                                      const double *__restrict__ part_norm, int n_part,
                                                      double *history, LeadBox lead)
{
    __shared__ double red[2 * FIN_WAVES];
    __shared__ double sh[2];
    __shared__ int sh_stop;
    __shared__ double lead_words[LEAD ? LEAD_BOX_WORDS / 2 : 1];
    __shared__ double lead_stage[LEAD ? LEAD_STAGE : 1];
    __shared__ int lead_timed_out;
    const uint32_t seq = LEAD ? sin->launch_seq : 0u;
    if (LEAD) lead_leaders<2>(lead, seq, part_rho, part_norm, nullptr, n_part, lead_stage);
    // (everything asked for at once, the scalars field by field: see k_cg_step1x_fin)
    const int stopped = sin->stop;
    const double s_rho = sin->rho, alpha = sin->alpha, omega = sin->omega, s_nf = sin->norm_factor,
                 s_init = sin->init_res;
    const int s_iter = sin->iter, s_evals = sin->n_evals;
    const double c_tol = sin->crit.tolerance, c_rel = sin->crit.rel_tol;
    const int c_min = sin->crit.min_iter, c_max = sin->crit.max_iter, c_freq = sin->crit.frequency,
              c_exp = sin->crit.export_res;
    if (blockIdx.x == 0 && threadIdx.x < sizeof(DevScalars) / 8)  // fields this kernel leaves alone
        reinterpret_cast<unsigned long long *>(sout)[threadIdx.x] =
            reinterpret_cast<const unsigned long long *>(sin)[threadIdx.x];
    const RowPair rp = my_rows(blockIdx.x, n);
    const double2 vr = ld2_stream(r, rp);
    const double2 po = ld2_stream(p, rp), vv = ld2_stream(v, rp);
    double2 vi;
    vi.x = vi.y = 1.0;
    if (inv_diag) vi = ld2_stream(inv_diag, rp);
    double pv[2][FIN_VT];
    if (!LEAD) load_partials_as_finaliser<2>(part_rho, part_norm, n_part, pv);
    if (stopped) return;
    double vsum[2] = {0.0, 0.0};
    if (LEAD) {
        if (!lead_wait(lead, 4 * FIN_WAVES, seq, lead_words, &lead_timed_out)) {
            if (threadIdx.x == 0) sout->comm_error = sout->stop = 1;
            return;
        }
        if (threadIdx.x == 0) {
            vsum[0] = lead_total(lead_words, 0);
            vsum[1] = lead_total(lead_words, 1);
        }
    } else {
        reduce_partials_as_finaliser<2>(pv, n_part, red, vsum);
    }
    if (threadIdx.x == 0) {
        // FIN_CG_CHECK: swap(prev_rho, rho) of the previous turn, then criterion_check (StoppingCriterion.C:71-151)
        const double prev_rho = s_rho, rho = vsum[0];
        int iter = s_iter, n_evals = s_evals, stop = 0;
        double init_res = s_init, res = 0.0;
        bool evaluated = false;
        if (iter > 0 && iter < c_min) {           // :77-81
            iter += 1;
        } else if (iter % c_freq != 0) {          // :84-87
            iter += 1;
        } else {
            evaluated = true;
            n_evals += 1;
            res = vsum[1];
            if (iter == 0) init_res = res / s_nf;  // :102-111
            res /= s_nf;                           // :113
            if (c_exp && history && blockIdx.x == 0) history[iter] = res;  // :115-117
            if (iter >= c_max) stop = 1;                                   // :124
            if (res < c_tol) stop = 1;                                     // :128
            if (c_rel > 0 && res < c_rel * init_res) stop = 1;             // :132-136
            iter += 1;                                                     // :143
        }
        sh[0] = prev_rho;
        sh[1] = rho;
        sh_stop = stop;
        if (blockIdx.x == 0) {
            sout->prev_rho = prev_rho;
            sout->rho = rho;
            sout->iter = iter;
            sout->x_pending = 0;
            if (evaluated) {
                sout->n_evals = n_evals;
                sout->init_res = init_res;
                sout->res = res;
            }
            if (stop) sout->stop = 1;
            if (LEAD) sout->launch_seq = seq + 1;
        }
    }
    __syncthreads();
    if (sh_stop) return;
    const double prev = sh[0], rho = sh[1];
    double2 vp = vr;
    if (prev * omega != 0.0) {  // step_1 (k_bicg_step1)
        const double tmp = rho / prev * alpha / omega;
        vp.x = vr.x + tmp * (po.x - omega * vv.x);
        vp.y = vr.y + tmp * (po.y - omega * vv.y);
    }
    if (inv_diag) st2_stream(p, rp, vp); else st2(p, rp, vp);
    if (inv_diag) {
        double2 vy;
        vy.x = vp.x * vi.x;
        vy.y = vp.y * vi.y;
        st2(y, rp, vy);
    }
}

// FIN_BICG_ALPHA + step_2
template <bool LEAD>
__global__ __launch_bounds__(BLOCK) void k_bicg_fold2(int n, const double *__restrict__ r,
                                                      const double *__restrict__ v, double *__restrict__ sv,
                                                      const double *__restrict__ inv_diag, double *__restrict__ z,
                                                      double *__restrict__ part_norm_out, const DevScalars *sin,
                                                      DevScalars *sout, const double *__restrict__ part_beta,
                                                      int n_part, LeadBox lead)
{
    __shared__ double red[2 * FIN_WAVES];
    __shared__ double sh[2];
    __shared__ double slot[N_WAVES];
    __shared__ double lead_words[LEAD ? LEAD_BOX_WORDS / 2 : 1];
    __shared__ double lead_stage[LEAD ? LEAD_STAGE : 1];
    __shared__ int lead_timed_out;
    const uint32_t seq = LEAD ? sin->launch_seq : 0u;
    if (LEAD) lead_leaders<1>(lead, seq, part_beta, nullptr, nullptr, n_part, lead_stage);
    const int stopped = sin->stop;
    const double s_rho = sin->rho;
    if (blockIdx.x == 0 && threadIdx.x < sizeof(DevScalars) / 8)
        reinterpret_cast<unsigned long long *>(sout)[threadIdx.x] =
            reinterpret_cast<const unsigned long long *>(sin)[threadIdx.x];
    const int chunk = blockIdx.x;
    const RowPair rp = my_rows(chunk, n);
    double2 vs = ld2_stream(r, rp);
    const double2 vv = ld2_stream(v, rp);
    double2 vi;
    vi.x = vi.y = 1.0;
    if (inv_diag) vi = ld2_stream(inv_diag, rp);
    double pv[2][FIN_VT];
    if (!LEAD) load_partials_as_finaliser<1>(part_beta, nullptr, n_part, pv);
    if (stopped) return;
    double vsum[2] = {0.0, 0.0};
    if (LEAD) {
        if (!lead_wait(lead, 2 * FIN_WAVES, seq, lead_words, &lead_timed_out)) {
            if (threadIdx.x == 0) sout->comm_error = sout->stop = 1;
            return;
        }
        if (threadIdx.x == 0) vsum[0] = lead_total(lead_words, 0);
    } else {
        reduce_partials_as_finaliser<1>(pv, n_part, red, vsum);
    }
    if (threadIdx.x == 0) {  // beta = rr.v ; alpha = rho / beta (0 when beta == 0)
        const double beta = vsum[0], alpha = (beta != 0.0) ? s_rho / beta : 0.0;
        sh[0] = alpha;
        sh[1] = beta;
        if (blockIdx.x == 0) {
            sout->beta = beta;
            sout->alpha = alpha;
            if (LEAD) sout->launch_seq = seq + 1;
        }
    }
    __syncthreads();
    const double alpha = sh[0], beta = sh[1];
    if (beta != 0.0) {  // step_2 (k_bicg_step2)
        vs.x = vs.x - alpha * vv.x;
        vs.y = vs.y - alpha * vv.y;
    }
    st2(sv, rp, vs);
    if (inv_diag) {
        double2 vz;
        vz.x = vs.x * vi.x;
        vz.y = vs.y * vi.y;
        st2(z, rp, vz);
    }
    double a = 0.0;
    if (rp.n > 0) a += fabs(vs.x);
    if (rp.n > 1) a += fabs(vs.y);
    const double s1 = block_sum(a, slot);
    if (threadIdx.x == 0) part_norm_out[chunk] = s1;
}

// FIN_BICG_CHECK2_OMEGA + step_3 (bicgstab::finalize, x += alpha y, when the mid-turn check stops the solve)
template <bool LEAD>
__global__ __launch_bounds__(BLOCK) void k_bicg_fold3(int n, double *__restrict__ x, double *__restrict__ r,
                                                      const double *__restrict__ sv, const double *__restrict__ t,
                                                      const double *__restrict__ y, const double *__restrict__ z,
                                                      const double *__restrict__ rr,
                                                      double *__restrict__ part_rho_out,
                                                      double *__restrict__ part_norm_out, const DevScalars *sin,
                                                      DevScalars *sout, const double *__restrict__ part_gamma,
                                                      const double *__restrict__ part_tt,
                                                      const double *__restrict__ part_snorm, int n_part,
                                                      double *history, int turn, LeadBox lead)
{
    __shared__ double red[2 * FIN_WAVES];
    __shared__ double sh[1];
    __shared__ int sh_stop;
    __shared__ double slot[2 * N_WAVES];
    __shared__ double lead_words[LEAD ? LEAD_BOX_WORDS / 2 : 1];
    __shared__ double lead_stage[LEAD ? LEAD_STAGE : 1];
    __shared__ int lead_timed_out;
    const uint32_t seq = LEAD ? sin->launch_seq : 0u;
    if (LEAD) lead_leaders<3>(lead, seq, part_gamma, part_tt, part_snorm, n_part, lead_stage);
    const int stopped = sin->stop;
    const double alpha = sin->alpha, s_nf = sin->norm_factor, s_init = sin->init_res;
    const int s_iter = sin->iter, s_evals = sin->n_evals;
    const double c_tol = sin->crit.tolerance, c_rel = sin->crit.rel_tol;
    const int c_min = sin->crit.min_iter, c_max = sin->crit.max_iter, c_freq = sin->crit.frequency,
              c_exp = sin->crit.export_res;
    if (blockIdx.x == 0 && threadIdx.x < sizeof(DevScalars) / 8)
        reinterpret_cast<unsigned long long *>(sout)[threadIdx.x] =
            reinterpret_cast<const unsigned long long *>(sin)[threadIdx.x];
    const int chunk = blockIdx.x;
    const RowPair rp = my_rows(chunk, n);
    double2 vx = ld2_stream(x, rp);
    const double2 vy = ld2_stream(y, rp), vz = ld2_stream(z, rp), vs = ld2_stream(sv, rp), vt = ld2_stream(t, rp),
                  vrr = ld2_stream(rr, rp);
    double pv[2][FIN_VT], pn[2][FIN_VT];
    if (!LEAD) {
        load_partials_as_finaliser<2>(part_gamma, part_tt, n_part, pv);
        load_partials_as_finaliser<1>(part_snorm, nullptr, n_part, pn);
    }
    if (stopped) return;
    double vsum[2] = {0.0, 0.0}, vnorm[2] = {0.0, 0.0};
    if (LEAD) {
        if (!lead_wait(lead, 6 * FIN_WAVES, seq, lead_words, &lead_timed_out)) {
            if (threadIdx.x == 0) sout->comm_error = sout->stop = 1;
            return;
        }
        if (threadIdx.x == 0) {
            vsum[0] = lead_total(lead_words, 0);
            vsum[1] = lead_total(lead_words, 1);
            vnorm[0] = lead_total(lead_words, 2);
        }
    } else {
        reduce_partials_as_finaliser<2>(pv, n_part, red, vsum);
        reduce_partials_as_finaliser<1>(pn, n_part, red, vnorm);
    }
    if (threadIdx.x == 0) {
        // the mid-turn check on s (criterion_check, StoppingCriterion.C:71-151), then gamma = s.t, beta = t.t,
        // omega = gamma / beta unless it stopped
        int iter = s_iter, n_evals = s_evals, stop = 0;
        double init_res = s_init, res = 0.0;
        bool evaluated = false;
        if (iter > 0 && iter < c_min) {
            iter += 1;
        } else if (iter % c_freq != 0) {
            iter += 1;
        } else {
            evaluated = true;
            n_evals += 1;
            res = vnorm[0];
            if (iter == 0) init_res = res / s_nf;
            res /= s_nf;
            if (c_exp && history && blockIdx.x == 0) history[iter] = res;
            if (iter >= c_max) stop = 1;
            if (res < c_tol) stop = 1;
            if (c_rel > 0 && res < c_rel * init_res) stop = 1;
            iter += 1;
        }
        const double omega = (vsum[1] != 0.0) ? vsum[0] / vsum[1] : 0.0;
        sh[0] = omega;
        sh_stop = stop;
        if (blockIdx.x == 0) {
            sout->iter = iter;
            if (evaluated) {
                sout->n_evals = n_evals;
                sout->init_res = init_res;
                sout->res = res;
            }
            if (stop) {
                sout->stop = 1;
                sout->stop_phase = 1;
                sout->stop_turn = turn;
            } else {
                sout->gamma = vsum[0];
                sout->beta = vsum[1];
                sout->omega = omega;
            }
            if (LEAD) sout->launch_seq = seq + 1;
        }
    }
    __syncthreads();
    if (sh_stop) {  // bicgstab::finalize: x += alpha y
        vx.x += alpha * vy.x;
        vx.y += alpha * vy.y;
        st2(x, rp, vx);
        return;
    }
    const double omega = sh[0];
    vx.x += alpha * vy.x + omega * vz.x;  // step_3 (k_bicg_step3)
    vx.y += alpha * vy.y + omega * vz.y;
    double2 vr;
    vr.x = vs.x - omega * vt.x;
    vr.y = vs.y - omega * vt.y;
    st2_stream(x, rp, vx);
    st2(r, rp, vr);
    double d = 0.0, a = 0.0;
    if (rp.n > 0) {
        d += vrr.x * vr.x;
        a += fabs(vr.x);
    }
    if (rp.n > 1) {
        d += vrr.y * vr.y;
        a += fabs(vr.y);
    }
    block_sum2(d, a, slot);
    if (threadIdx.x == 0) {
        part_rho_out[chunk] = d;
        part_norm_out[chunk] = a;
    }
}

// ------------------------------------------------------------------------------------------
// GMRES vector kernels
// ------------------------------------------------------------------------------------------
__global__ __launch_bounds__(BLOCK) void k_gmres_scale(int n, double *__restrict__ out,
                                                       const double *__restrict__ in,
                                                       const double *__restrict__ denom,
                                                       const DevScalars *gate)
{
    if (gate && gate->stop) return;
    const double d = *denom;
    const RowPair rp = my_rows(blockIdx.x, n);
    double2 v = ld2(in, rp);
    v.x = v.x / d;
    v.y = v.y / d;
    st2(out, rp, v);
}

__global__ __launch_bounds__(BLOCK) void k_gmres_scale_mul(int n, double *__restrict__ out, const double *in,
                                                           const double *__restrict__ denom,
                                                           const double *__restrict__ inv_diag,
                                                           double *__restrict__ w, const DevScalars *gate)
{
    if (gate && gate->stop) return;
    const double d = *denom;
    const RowPair rp = my_rows(blockIdx.x, n);
    double2 v = ld2(in, rp);  // (in may be out: the vector scaled in place)
    const double2 vi = ld2(inv_diag, rp);
    v.x = v.x / d;
    v.y = v.y / d;
    st2(out, rp, v);
    v.x = v.x * vi.x;
    v.y = v.y * vi.y;
    st2(w, rp, v);
}

__global__ __launch_bounds__(BLOCK) void k_gmres_mgs(int n, double *__restrict__ w,
                                                     const double *__restrict__ vprev,
                                                     const double *__restrict__ hprev,
                                                     const double *__restrict__ vdot,
                                                     double *__restrict__ part,
                                                     const DevScalars *gate)
{
    __shared__ double slot[N_WAVES];
    if (gate && gate->stop) return;
    const int chunk = blockIdx.x;
    const RowPair rp = my_rows(chunk, n);
    double2 vw = ld2(w, rp);
    if (vprev) {
        const double h = *hprev;
        const double2 vp = ld2(vprev, rp);
        vw.x -= h * vp.x;
        vw.y -= h * vp.y;
        st2(w, rp, vw);
    }
    const double2 vd = vdot ? ld2(vdot, rp) : vw;
    double d = 0.0;
    if (rp.n > 0) d += vw.x * vd.x;
    if (rp.n > 1) d += vw.y * vd.y;
    const double s0 = block_sum(d, slot);
    if (threadIdx.x == 0) part[chunk] = s0;
}

// Small single-rank systems (<= FUSED_FIN_MAX_CHUNKS chunks): the finaliser between two Gram-Schmidt links (FIN_GMRES_H:
// H(k, it) = sum of the link's partials) folded into the next link's kernel -- every workgroup reduces the partials
// itself in the finaliser's order (same bits), workgroup 0 stores H(k, it).  One launch per link instead of two; the
// link reads `part_in` and writes `part_out` (never the same array: another workgroup may still be reducing).
// LEAD (any number of chunks): leader finalisation (device_common.hpp); the tag of the launch comes from the host -- these
// turns are never captured in a graph, and there are no ping-pong scalar slots here to carry a sequence number
template <bool LEAD>
__global__ __launch_bounds__(BLOCK) void k_gmres_mgs_fold(int n, double *__restrict__ w,
                                                          const double *__restrict__ vprev,
                                                          double *__restrict__ h_out,
                                                          const double *__restrict__ vdot,
                                                          const double *__restrict__ part_in, int n_part,
                                                          double *__restrict__ part_out, const DevScalars *gate,
                                                          LeadBox lead, uint32_t tag)
{
    __shared__ double red[2 * FIN_WAVES];
    __shared__ double sh_h;
    __shared__ double slot[N_WAVES];
    __shared__ double lead_words[LEAD ? LEAD_BOX_WORDS / 2 : 1];
    __shared__ double lead_stage[LEAD ? LEAD_STAGE : 1];
    __shared__ int lead_timed_out;
    if (LEAD && vprev) lead_leaders<1>(lead, tag, part_in, nullptr, nullptr, n_part, lead_stage);
    if (gate && gate->stop) return;
    const int chunk = blockIdx.x;
    const RowPair rp = my_rows(chunk, n);
    double2 vw = ld2(w, rp);
    if (vprev) {
        const double2 vp = ld2(vprev, rp);
        double pv[2][FIN_VT], v[2] = {0.0, 0.0};
        if (LEAD) {
            if (!lead_wait(lead, 2 * FIN_WAVES, tag, lead_words, &lead_timed_out)) return;  // (the host's poll of the solve times out)
            if (threadIdx.x == 0) v[0] = lead_total(lead_words, 0);
        } else {
            load_partials_as_finaliser<1>(part_in, nullptr, n_part, pv);
            reduce_partials_as_finaliser<1>(pv, n_part, red, v);
        }
        if (threadIdx.x == 0) {
            sh_h = v[0];
            if (blockIdx.x == 0) *h_out = v[0];  // FIN_GMRES_H
        }
        __syncthreads();
        const double h = sh_h;
        vw.x -= h * vp.x;
        vw.y -= h * vp.y;
        st2(w, rp, vw);
    }
    const double2 vd = vdot ? ld2(vdot, rp) : vw;
    double d = 0.0;
    if (rp.n > 0) d += vw.x * vd.x;
    if (rp.n > 1) d += vw.y * vd.y;
    const double s0 = block_sum(d, slot);
    if (threadIdx.x == 0) part_out[chunk] = s0;
}

__global__ __launch_bounds__(BLOCK) void k_gmres_update_x(int n, const double *__restrict__ V,
                                                          long ld, const double *__restrict__ y,
                                                          int it, const double *__restrict__ inv_diag,
                                                          double *__restrict__ x,
                                                          double *__restrict__ before,
                                                          const DevScalars *gate)
{
    if (gate && gate->stop) return;
    const RowPair rp = my_rows(blockIdx.x, n);
    double2 sum;
    sum.x = 0.0;
    sum.y = 0.0;
    for (int j = 0; j < it; ++j) {
        const double yj = y[j];
        const double2 v = ld2(V + (size_t)j * ld, rp);
        sum.x += v.x * yj;
        sum.y += v.y * yj;
    }
    if (before) {
        st2(before, rp, sum);
        return;
    }
    if (inv_diag) {
        const double2 vi = ld2(inv_diag, rp);
        sum.x = sum.x * vi.x;
        sum.y = sum.y * vi.y;
    }
    double2 vx = ld2(x, rp);
    vx.x += sum.x;
    vx.y += sum.y;
    st2(x, rp, vx);
}

__global__ __launch_bounds__(BLOCK) void k_mul(int n, double *__restrict__ out,
                                               const double *__restrict__ in,
                                               const double *__restrict__ inv_diag,
                                               const DevScalars *gate)
{
    if (gate && gate->stop) return;
    const RowPair rp = my_rows(blockIdx.x, n);
    double2 v = ld2(in, rp);
    const double2 vi = ld2(inv_diag, rp);
    v.x = v.x * vi.x;
    v.y = v.y * vi.y;
    st2(out, rp, v);
}

__global__ __launch_bounds__(BLOCK) void k_add(int n, double *__restrict__ x,
                                               const double *__restrict__ a, const DevScalars *gate)
{
    if (gate && gate->stop) return;
    const RowPair rp = my_rows(blockIdx.x, n);
    double2 vx = ld2(x, rp);
    const double2 va = ld2(a, rp);
    vx.x += va.x;
    vx.y += va.y;
    st2(x, rp, vx);
}

// GMRES dense-state accessors (layout: kernels.hpp gmres_state_len)
struct GmresState {
    double *H, *gs, *gc, *rnc, *y;
    int m;
    __device__ GmresState(double *base, int m_) : m(m_)
    {
        H = base;
        gs = H + (size_t)(m + 1) * m;
        gc = gs + m;
        rnc = gc + m;
        y = rnc + (m + 1);
    }
    __device__ double &h(int i, int j) const { return H[(size_t)j * (m + 1) + i]; }
};

// ------------------------------------------------------------------------------------------
// finalisers (one workgroup): reduce the per-chunk partials, then the scalar logic
// ------------------------------------------------------------------------------------------

template <int PHASE>
__global__ __launch_bounds__(FIN_BLOCK) void k_finalize(DevScalars *s, FinArgs a)
{
    __shared__ double slot[FIN_WAVES];
    if (PHASE != FIN_MEAN && PHASE != FIN_NORMFACTOR && PHASE != FIN_RAW &&
        PHASE != FIN_GMRES_SOLVE && s->stop) {
        // after a stop the one step_1x that followed has applied the pending x update
        if (PHASE == FIN_BETA && threadIdx.x == 0 && s->x_pending) s->x_pending = 0;
        return;
    }
    // thread 0 fetches the scalar block up front (its latency hides behind the partial loads),
    // does the logic in registers and stores the block once: the criterion's dependent global
    // round trips would otherwise cost more than the reduction itself
    DevScalars L;
    if (threadIdx.x == 0 && a.do_logic) L = *s;
    double v0 = 0.0, v1 = 0.0, v2 = 0.0;
    if (a.do_reduce) {
        const double *const parts[2] = {a.part[0], a.part[1]};
        double r[2];
        if (a.n_sums > 1)
            reduce_partials<2>(parts, a.n_part, slot, r);
        else
            reduce_partials<1>(parts, a.n_part, slot, r);
        v0 = r[0];
        v1 = r[1];
        if (PHASE == FIN_BICG_CHECK2_OMEGA) {  // (single rank: nothing of this goes through an all-reduce)
            const double *const extra[2] = {a.part_extra, nullptr};
            reduce_partials<1>(extra, a.n_part, slot, r);
            v2 = r[0];
        }
        if (PHASE == FIN_MEAN) {
            // distributed compute_mean [UPSTREAM]: local mean, weighted by n_local / n_global
            v0 /= a.n_local;
            v0 *= a.n_local / a.n_global;
        }
        if (threadIdx.x == 0 && !a.do_logic) {
            s->sums[0] = v0;
            s->sums[1] = v1;
        }
    }
    bool comm_ok = true;
    unsigned reduce_waited = 0;
    const bool peer_reduce = a.peer.world > 1 && a.do_reduce && a.do_logic;
    if (peer_reduce) comm_ok = peer_allreduce2(a.peer, v0, v1, &reduce_waited);
    if (!a.do_logic || threadIdx.x != 0) return;
    if (peer_reduce) {
        L.reduce_wait_ticks += reduce_waited;
        L.reduce_waits += 1;
    }
    if (!comm_ok) {  // a rank is gone: end the solve, the host reports OGL_ERR_COMM
        L.comm_error = 1;
        L.stop = 1;
    }
    if (!a.do_reduce) {
        v0 = L.sums[0];
        v1 = L.sums[1];
    } else {
        L.sums[0] = v0;
        L.sums[1] = v1;
    }
    if (PHASE == FIN_MEAN) {
        L.xbar = v0;
    } else if (PHASE == FIN_NORMFACTOR) {
        L.norm_factor = v0 + 1.0e-15;  // + SMALL, StoppingCriterion.C:68
    } else if (PHASE == FIN_CG_CHECK) {
        L.prev_rho = L.rho;  // swap(prev_rho, rho) of the previous turn
        L.rho = v0;
        criterion_check(&L, L.crit, v1, a.history);
        L.x_pending = a.turn ? 1 : 0;  // deferred-x path: step_2r's update waits for the next step_1x
    } else if (PHASE == FIN_BETA) {
        L.beta = v0;
        L.x_pending = 0;  // the step_1x before this SpMV has applied it
    } else if (PHASE == FIN_BICG_ALPHA) {  // beta = rr.v ; alpha = rho / beta (0 when beta == 0)
        L.beta = v0;
        L.alpha = (v0 != 0.0) ? L.rho / v0 : 0.0;
    } else if (PHASE == FIN_BICG_CHECK2) {  // mid-turn check on s
        criterion_check(&L, L.crit, v0, a.history);
        if (L.stop) {
            L.stop_phase = 1;
            L.stop_turn = a.turn;
        }
    } else if (PHASE == FIN_BICG_CHECK2_OMEGA) {  // the two phases around the second SpMV in one
        criterion_check(&L, L.crit, v2, a.history);
        if (L.stop) {
            L.stop_phase = 1;
            L.stop_turn = a.turn;
        } else {
            L.gamma = v0;
            L.beta = v1;
            L.omega = (v1 != 0.0) ? v0 / v1 : 0.0;
        }
    } else if (PHASE == FIN_BICG_OMEGA) {  // gamma = s.t ; beta = t.t ; omega = gamma / beta
        L.gamma = v0;
        L.beta = v1;
        L.omega = (v1 != 0.0) ? v0 / v1 : 0.0;
    } else if (PHASE == FIN_GMRES_RESTART) {  // gmres::restart
        GmresState g(a.gm, a.m);
        const double rn = sqrt(v0);
        g.rnc[0] = rn;
        L.beta = rn;  // V_0 = r / rn
        L.stale_norm = v1;
        if (a.check_after) criterion_check(&L, L.crit, L.stale_norm, a.history);
    } else if (PHASE == FIN_GMRES_H) {  // finish_arnoldi: H(k, it)
        GmresState g(a.gm, a.m);
        g.h(a.k, a.turn) = v0;
    } else if (PHASE == FIN_GMRES_COL) {  // norm of the new basis vector, then givens_rotation
        GmresState g(a.gm, a.m);
        const int it = a.turn;
        const double hn = sqrt(v0);
        g.h(it + 1, it) = hn;
        L.beta = hn;  // V_{it+1} /= hn
        for (int j = 0; j < it; ++j) {
            const double t = g.gc[j] * g.h(j, it) + g.gs[j] * g.h(j + 1, it);
            g.h(j + 1, it) = -g.gs[j] * g.h(j, it) + g.gc[j] * g.h(j + 1, it);
            g.h(j, it) = t;
        }
        if (g.h(it, it) == 0.0) {
            g.gc[it] = 0.0;
            g.gs[it] = 1.0;
        } else {
            const double scale = fabs(g.h(it, it)) + fabs(g.h(it + 1, it));
            const double a0 = g.h(it, it) / scale, a1 = g.h(it + 1, it) / scale;
            const double hyp = scale * sqrt(a0 * a0 + a1 * a1);
            g.gc[it] = g.h(it, it) / hyp;
            g.gs[it] = g.h(it + 1, it) / hyp;
        }
        g.h(it, it) = g.gc[it] * g.h(it, it) + g.gs[it] * g.h(it + 1, it);
        g.h(it + 1, it) = 0.0;
        g.rnc[it + 1] = -g.gs[it] * g.rnc[it];
        g.rnc[it] = g.gc[it] * g.rnc[it];
        if (a.check_after) criterion_check(&L, L.crit, L.stale_norm, a.history);
    } else if (PHASE == FIN_GMRES_SOLVE) {  // solve_upper_triangular over `turn` columns
        GmresState g(a.gm, a.m);
        for (int i = a.turn - 1; i >= 0; --i) {
            double t = g.rnc[i];
            for (int j = i + 1; j < a.turn; ++j) t -= g.h(i, j) * g.y[j];
            g.y[i] = t / g.h(i, i);
        }
    }
    *s = L;
}

__global__ void k_reset_scalars(DevScalars *s, DevCriterion crit)
{
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    DevScalars z{};
    z.crit = crit;
    z.rho = 1.0;  // becomes prev_rho = 1 at the first check ([UPSTREAM] cg::initialize)
    z.prev_rho = 1.0;
    z.alpha = z.omega = z.gamma = z.beta = 1.0;
    z.norm_factor = 1.0;  // StoppingCriterion.H:136
    z.launch_seq = 1;     // (0 is what an untouched LeadBox word carries)
    *s = z;
}

}  // namespace

// ------------------------------------------------------------------------------------------
// launchers
// ------------------------------------------------------------------------------------------
void launch_scale(hipStream_t st, int32_t n, double *v, double factor)
{
    if (n == 0) return;
    hipLaunchKernelGGL(k_scale, dim3(blocks_for(n)), dim3(BLOCK), 0, st, n, v, factor);
}

void launch_fill_xbar(hipStream_t st, int32_t n, double *v, const DevScalars *s)
{
    if (n == 0) return;
    hipLaunchKernelGGL(k_fill_xbar, dim3(blocks_for(n)), dim3(BLOCK), 0, st, n, v, s);
}

void launch_partials_sum(hipStream_t st, int32_t n, const double *a, double *part)
{
    const int nc = (int)n_chunks(n);
    if (nc == 0) return;
    hipLaunchKernelGGL((k_partials<P_SUM>), dim3(nc), dim3(BLOCK), 0, st, n, nc, a, nullptr, part,
                       nullptr, nullptr);
}

void launch_partials_dot(hipStream_t st, int32_t n, const double *a, const double *b, double *part,
                         const DevScalars *gate)
{
    const int nc = (int)n_chunks(n);
    if (nc == 0) return;
    hipLaunchKernelGGL((k_partials<P_DOT>), dim3(nc), dim3(BLOCK), 0, st, n, nc, a, b, part, gate,
                       nullptr);
}

void launch_partials_dot_chunks(hipStream_t st, int32_t n, const double *a, const double *b,
                                double *part, const DevScalars *gate, const int32_t *chunk_list,
                                int32_t count)
{
    if (count == 0) return;
    hipLaunchKernelGGL((k_partials<P_DOT>), dim3(count), dim3(BLOCK), 0, st, n, (int)n_chunks(n), a, b,
                       part, gate, chunk_list);
}

void launch_partials_norm1(hipStream_t st, int32_t n, const double *a, double *part)
{
    const int nc = (int)n_chunks(n);
    if (nc == 0) return;
    hipLaunchKernelGGL((k_partials<P_NORM1>), dim3(nc), dim3(BLOCK), 0, st, n, nc, a, nullptr,
                       part, nullptr, nullptr);
}

void launch_partials_normfactor(hipStream_t st, int32_t n, const double *b, const double *w,
                                const double *r, double *part)
{
    const int nc = (int)n_chunks(n);
    if (nc == 0) return;
    hipLaunchKernelGGL(k_partials_normfactor, dim3(nc), dim3(BLOCK), 0, st, n, b, w, r, part);
}

void launch_cg_rho_norm(hipStream_t st, int32_t n, const double *r, const double *inv_diag,
                        double *part_rho, double *part_norm, const DevScalars *gate)
{
    const int nc = (int)n_chunks(n);
    if (nc == 0) return;
    hipLaunchKernelGGL(k_cg_rho_norm, dim3(nc), dim3(BLOCK), 0, st, n, r, inv_diag, part_rho,
                       part_norm, gate);
}

void launch_cg_step1(hipStream_t st, int32_t n, double *p, const double *r, const double *inv_diag,
                     const DevScalars *s)
{
    const int nc = (int)n_chunks(n);
    if (nc == 0) return;
    hipLaunchKernelGGL(k_cg_step1, dim3(nc), dim3(BLOCK), 0, st, n, p, r, inv_diag, s);
}

void launch_cg_step1x(hipStream_t st, int32_t n, double *p, double *x, const double *r,
                      const double *inv_diag, const DevScalars *s, const HaloPutFused *put)
{
    const int nc = (int)n_chunks(n);
    if (nc == 0) return;
    if (put && put->chunk_sptr)
        hipLaunchKernelGGL((k_cg_step1x<true>), dim3(nc), dim3(BLOCK), 0, st, n, p, x, r, inv_diag, s, *put);
    else
        hipLaunchKernelGGL((k_cg_step1x<false>), dim3(nc), dim3(BLOCK), 0, st, n, p, x, r, inv_diag, s,
                           HaloPutFused{});
}

void launch_cg_step2r(hipStream_t st, int32_t n, double *r, const double *q, const double *inv_diag,
                      double *part_rho, double *part_norm, const DevScalars *s, double *z_out,
                      const HaloPutFused *put)
{
    const int nc = (int)n_chunks(n);
    if (nc == 0) return;
    if (put && put->chunk_sptr)
        hipLaunchKernelGGL((k_cg_step2r<true>), dim3(nc), dim3(BLOCK), 0, st, n, r, q, inv_diag, part_rho,
                           part_norm, s, z_out, *put);
    else
        hipLaunchKernelGGL((k_cg_step2r<false>), dim3(nc), dim3(BLOCK), 0, st, n, r, q, inv_diag, part_rho,
                           part_norm, s, z_out, HaloPutFused{});
}

void launch_cg_step1x_fin(hipStream_t st, int32_t n, double *p, double *x, const double *r, const double *inv_diag,
                          const DevScalars *sin, DevScalars *sout, const double *part_rho,
                          const double *part_norm, double *history, int first, const LeadBox &lead, double *p_out,
                          const PRing &ring)
{
    const int nc = (int)n_chunks(n);
    if (nc == 0) return;
    if (!p_out) p_out = p;
    const bool led = lead.box && nc >= 3 * FIN_WAVES;
    const LeadBox lb = led ? lead : LeadBox{};
#define OGL_STEP1X(LEAD, K)                                                                                               \
    hipLaunchKernelGGL((k_cg_step1x_fin<LEAD, K>), dim3(nc), dim3(BLOCK), 0, st, n, p, x, r, inv_diag, sin, sout,        \
                       part_rho, part_norm, nc, history, first, lb, p_out, ring)
#define OGL_STEP1X_K(LEAD)                                                                                                \
    switch (ring.k) {                                                                                                     \
    case 2: OGL_STEP1X(LEAD, 2); break;                                                                                   \
    case 4: OGL_STEP1X(LEAD, 4); break;                                                                                   \
    case 8: OGL_STEP1X(LEAD, 8); break;                                                                                   \
    default: OGL_STEP1X(LEAD, 0); break;                                                                                  \
    }
    if (led) {
        OGL_STEP1X_K(true)
    } else {
        OGL_STEP1X(false, 0);  // (every workgroup its own finaliser: x every turn, whatever buffers p and p_out are)
    }
#undef OGL_STEP1X_K
#undef OGL_STEP1X
}

void launch_cg_step2r_fin(hipStream_t st, int32_t n, double *r, const double *q, const double *inv_diag,
                          double *part_rho, double *part_norm, const DevScalars *sin, DevScalars *sout,
                          const double *part_beta, double *z_out, const LeadBox &lead)
{
    const int nc = (int)n_chunks(n);
    if (nc == 0) return;
    if (lead.box && nc >= 3 * FIN_WAVES)
        hipLaunchKernelGGL(k_cg_step2r_fin<true>, dim3(nc), dim3(BLOCK), 0, st, n, r, q, inv_diag, part_rho, part_norm, sin,
                           sout, part_beta, nc, z_out, lead);
    else
        hipLaunchKernelGGL(k_cg_step2r_fin<false>, dim3(nc), dim3(BLOCK), 0, st, n, r, q, inv_diag, part_rho, part_norm, sin,
                           sout, part_beta, nc, z_out, LeadBox{});
}

void launch_cg_step2(hipStream_t st, int32_t n, double *x, double *r, const double *p,
                     const double *q, const double *inv_diag, double *part_rho, double *part_norm,
                     const DevScalars *s)
{
    const int nc = (int)n_chunks(n);
    if (nc == 0) return;
    hipLaunchKernelGGL(k_cg_step2, dim3(nc), dim3(BLOCK), 0, st, n, x, r, p, q, inv_diag, part_rho,
                       part_norm, s);
}

void launch_bicg_step1(hipStream_t st, int32_t n, double *p, const double *r, const double *v,
                       const double *inv_diag, double *y, const DevScalars *s)
{
    const int nc = (int)n_chunks(n);
    if (nc == 0) return;
    hipLaunchKernelGGL(k_bicg_step1, dim3(nc), dim3(BLOCK), 0, st, n, p, r, v, inv_diag, y, s);
}

void launch_bicg_step2(hipStream_t st, int32_t n, const double *r, const double *v, double *sv,
                       const double *inv_diag, double *z, double *part_norm, const DevScalars *s)
{
    const int nc = (int)n_chunks(n);
    if (nc == 0) return;
    hipLaunchKernelGGL(k_bicg_step2, dim3(nc), dim3(BLOCK), 0, st, n, r, v, sv, inv_diag, z,
                       part_norm, s);
}

void launch_bicg_step3(hipStream_t st, int32_t n, double *x, double *r, const double *sv,
                       const double *t, const double *y, const double *z, const double *rr,
                       double *part_rho, double *part_norm, const DevScalars *s, int turn)
{
    const int nc = (int)n_chunks(n);
    if (nc == 0) return;
    hipLaunchKernelGGL(k_bicg_step3, dim3(nc), dim3(BLOCK), 0, st, n, x, r, sv, t, y, z, rr,
                       part_rho, part_norm, s, turn);
}

void launch_gmres_mgs_fold(hipStream_t st, int32_t n, double *w, const double *vprev, double *h_out, const double *vdot,
                           const double *part_in, double *part_out, const DevScalars *gate, const LeadBox &lead, uint32_t tag)
{
    const int nc = (int)n_chunks(n);
    if (nc == 0) return;
    if (lead.box && nc >= 3 * FIN_WAVES)
        hipLaunchKernelGGL(k_gmres_mgs_fold<true>, dim3(nc), dim3(BLOCK), 0, st, n, w, vprev, h_out, vdot, part_in, nc,
                           part_out, gate, lead, tag);
    else
        hipLaunchKernelGGL(k_gmres_mgs_fold<false>, dim3(nc), dim3(BLOCK), 0, st, n, w, vprev, h_out, vdot, part_in, nc,
                           part_out, gate, LeadBox{}, 0u);
}

void launch_bicg_fold1(hipStream_t st, int32_t n, double *p, const double *r, const double *v, const double *inv_diag,
                       double *y, const DevScalars *sin, DevScalars *sout, const double *part_rho,
                       const double *part_norm, double *history, const LeadBox &lead)
{
    const int nc = (int)n_chunks(n);
    if (nc == 0) return;
    if (lead.box && nc >= 3 * FIN_WAVES)
        hipLaunchKernelGGL(k_bicg_fold1<true>, dim3(nc), dim3(BLOCK), 0, st, n, p, r, v, inv_diag, y, sin, sout, part_rho,
                           part_norm, nc, history, lead);
    else
        hipLaunchKernelGGL(k_bicg_fold1<false>, dim3(nc), dim3(BLOCK), 0, st, n, p, r, v, inv_diag, y, sin, sout, part_rho,
                           part_norm, nc, history, LeadBox{});
}

void launch_bicg_fold2(hipStream_t st, int32_t n, const double *r, const double *v, double *sv, const double *inv_diag,
                       double *z, double *part_norm_out, const DevScalars *sin, DevScalars *sout,
                       const double *part_beta, const LeadBox &lead)
{
    const int nc = (int)n_chunks(n);
    if (nc == 0) return;
    if (lead.box && nc >= 3 * FIN_WAVES)
        hipLaunchKernelGGL(k_bicg_fold2<true>, dim3(nc), dim3(BLOCK), 0, st, n, r, v, sv, inv_diag, z, part_norm_out, sin,
                           sout, part_beta, nc, lead);
    else
        hipLaunchKernelGGL(k_bicg_fold2<false>, dim3(nc), dim3(BLOCK), 0, st, n, r, v, sv, inv_diag, z, part_norm_out, sin,
                           sout, part_beta, nc, LeadBox{});
}

void launch_bicg_fold3(hipStream_t st, int32_t n, double *x, double *r, const double *sv, const double *t,
                       const double *y, const double *z, const double *rr, double *part_rho_out, double *part_norm_out,
                       const DevScalars *sin, DevScalars *sout, const double *part_gamma, const double *part_tt,
                       const double *part_snorm, double *history, int turn, const LeadBox &lead)
{
    const int nc = (int)n_chunks(n);
    if (nc == 0) return;
    if (lead.box && nc >= 3 * FIN_WAVES)
        hipLaunchKernelGGL(k_bicg_fold3<true>, dim3(nc), dim3(BLOCK), 0, st, n, x, r, sv, t, y, z, rr, part_rho_out,
                           part_norm_out, sin, sout, part_gamma, part_tt, part_snorm, nc, history, turn, lead);
    else
        hipLaunchKernelGGL(k_bicg_fold3<false>, dim3(nc), dim3(BLOCK), 0, st, n, x, r, sv, t, y, z, rr, part_rho_out,
                           part_norm_out, sin, sout, part_gamma, part_tt, part_snorm, nc, history, turn, LeadBox{});
}

void launch_gmres_scale(hipStream_t st, int32_t n, double *out, const double *in,
                        const double *denom, const DevScalars *gate)
{
    const int nc = (int)n_chunks(n);
    if (nc == 0) return;
    hipLaunchKernelGGL(k_gmres_scale, dim3(nc), dim3(BLOCK), 0, st, n, out, in, denom, gate);
}

void launch_gmres_scale_mul(hipStream_t st, int32_t n, double *out, const double *in, const double *denom,
                            const double *inv_diag, double *w, const DevScalars *gate)
{
    const int nc = (int)n_chunks(n);
    if (nc == 0) return;
    hipLaunchKernelGGL(k_gmres_scale_mul, dim3(nc), dim3(BLOCK), 0, st, n, out, in, denom, inv_diag, w, gate);
}

void launch_gmres_mgs(hipStream_t st, int32_t n, double *w, const double *vprev,
                      const double *hprev, const double *vdot, double *part,
                      const DevScalars *gate)
{
    const int nc = (int)n_chunks(n);
    if (nc == 0) return;
    hipLaunchKernelGGL(k_gmres_mgs, dim3(nc), dim3(BLOCK), 0, st, n, w, vprev, hprev, vdot, part, gate);
}

void launch_gmres_update_x(hipStream_t st, int32_t n, const double *V, int64_t ld, const double *y,
                           int32_t it, const double *inv_diag, double *x, double *before,
                           const DevScalars *gate)
{
    const int nc = (int)n_chunks(n);
    if (nc == 0) return;
    hipLaunchKernelGGL(k_gmres_update_x, dim3(nc), dim3(BLOCK), 0, st, n, V, (long)ld, y, it,
                       inv_diag, x, before, gate);
}

void launch_mul(hipStream_t st, int32_t n, double *out, const double *in, const double *inv_diag,
                const DevScalars *gate)
{
    const int nc = (int)n_chunks(n);
    if (nc == 0) return;
    hipLaunchKernelGGL(k_mul, dim3(nc), dim3(BLOCK), 0, st, n, out, in, inv_diag, gate);
}

void launch_add(hipStream_t st, int32_t n, double *x, const double *a, const DevScalars *gate)
{
    const int nc = (int)n_chunks(n);
    if (nc == 0) return;
    hipLaunchKernelGGL(k_add, dim3(nc), dim3(BLOCK), 0, st, n, x, a, gate);
}

void launch_finalize(hipStream_t st, int phase, DevScalars *s, const FinArgs &a)
{
    const dim3 grid(1), block(FIN_BLOCK);
    switch (phase) {
    case FIN_MEAN:
        hipLaunchKernelGGL((k_finalize<FIN_MEAN>), grid, block, 0, st, s, a);
        break;
    case FIN_NORMFACTOR:
        hipLaunchKernelGGL((k_finalize<FIN_NORMFACTOR>), grid, block, 0, st, s, a);
        break;
    case FIN_CG_CHECK:
        hipLaunchKernelGGL((k_finalize<FIN_CG_CHECK>), grid, block, 0, st, s, a);
        break;
    case FIN_BETA:
        hipLaunchKernelGGL((k_finalize<FIN_BETA>), grid, block, 0, st, s, a);
        break;
    case FIN_BICG_ALPHA:
        hipLaunchKernelGGL((k_finalize<FIN_BICG_ALPHA>), grid, block, 0, st, s, a);
        break;
    case FIN_BICG_CHECK2:
        hipLaunchKernelGGL((k_finalize<FIN_BICG_CHECK2>), grid, block, 0, st, s, a);
        break;
    case FIN_BICG_OMEGA:
        hipLaunchKernelGGL((k_finalize<FIN_BICG_OMEGA>), grid, block, 0, st, s, a);
        break;
    case FIN_BICG_CHECK2_OMEGA:
        hipLaunchKernelGGL((k_finalize<FIN_BICG_CHECK2_OMEGA>), grid, block, 0, st, s, a);
        break;
    case FIN_GMRES_RESTART:
        hipLaunchKernelGGL((k_finalize<FIN_GMRES_RESTART>), grid, block, 0, st, s, a);
        break;
    case FIN_GMRES_H:
        hipLaunchKernelGGL((k_finalize<FIN_GMRES_H>), grid, block, 0, st, s, a);
        break;
    case FIN_GMRES_COL:
        hipLaunchKernelGGL((k_finalize<FIN_GMRES_COL>), grid, block, 0, st, s, a);
        break;
    case FIN_GMRES_SOLVE:
        hipLaunchKernelGGL((k_finalize<FIN_GMRES_SOLVE>), grid, block, 0, st, s, a);
        break;
    default:
        hipLaunchKernelGGL((k_finalize<FIN_RAW>), grid, block, 0, st, s, a);
        break;
    }
}

void launch_reset_scalars(hipStream_t st, DevScalars *s, const DevCriterion &crit)
{
    hipLaunchKernelGGL(k_reset_scalars, dim3(1), dim3(64), 0, st, s, crit);
}

}  // namespace ogl
