// kernels_spmv_sym.hip -- half-storage SpMV of banded symmetric patterns and the GKOCG turn kernels built on it
// (geometry, reduction tree and the -ffp-contract=off rule: device_common.hpp)
#include "device_common.hpp"

namespace ogl {

namespace {

// ------------------------------------------------------------------------------------------
// Half storage of a symmetric matrix on a banded pattern (SymLayout, host_matrix.hpp).  ND planes per
// chunk: plane j holds A(r, r + d[j]), d[0] = 0.  A lower entry A(r, r - d[j]) is read where its twin
// lives: plane j of row r - d[j] -- a coalesced strip of values that some workgroup reads (or has read) as
// upper entries, so DRAM delivers every coefficient once; the second reader finds it in L2 / the Infinity
// Cache.  8 ND + 1 bytes per row instead of 8.1 per stored entry: measured 115 us against 139 us for the
// pattern-coded full storage on the 216^3 matrix (tools/sym_tune.hip, profiles/spmv_tune_r02.txt).  Rows are
// summed in ascending column order -- furthest lower entry first, diagonal, upper entries -- so y and the
// fused dot partials have the same bits as k_spmv_sell / k_spmv_stream.
// ------------------------------------------------------------------------------------------
struct SymOffsets {
    int d[SYM_MAX_OFFSETS];
};
// FAST: d[1] == 1 and every further distance even (a box with an even line length) -- known at compile
// time, so the kernel stays straight-line code (a run-time test of the parity splits the loads into basic
// blocks that wait for each other: 130 us instead of 113, tools/sym_tune.hip var1/var2).  Then the two rows
// of a lane are an aligned pair in every strip: x and the lower values of the even distances come as one
// 16-byte load per pair, the d = 1 neighbours from the diagonal pair and the lane's own plane-1 value.
// STREAM (planes + vectors larger than the Infinity Cache): the planes that are read exactly once per launch are
// streamed past the caches -- the diagonal always, and with FAST also plane 1, whose second reader (the d = 1
// lower entry of the next row) is the neighbouring lane: a lane shuffle instead of a load.
template <int MODE, int NDOT, int ND, bool FAST, bool STREAM>
__global__ __launch_bounds__(BLOCK) void k_spmv_sym(int n_rows, int n_chunks, SymOffsets off,
                                                    const uint8_t *__restrict__ mask,
                                                    const double *__restrict__ planes,
                                                    const double *__restrict__ x, const double *__restrict__ b,
                                                    double *__restrict__ y, const double *__restrict__ w,
                                                    double *__restrict__ dot_partials,
                                                    double *__restrict__ dot2_partials, const DevScalars *gate,
                                                    const int *__restrict__ block_order, HaloFused hf)
{
    __shared__ double slot[N_WAVES];
    __shared__ double ys[CHUNK_ROWS];
    if (gate && gate->stop) return;
    // banded patterns: the host's order puts the chunks of rows r and r +- d[ND-1] on one XCD (band_block_order)
    const int chunk = block_order ? block_order[blockIdx.x] : xcd_chunk(blockIdx.x);
    if (chunk < 0 || chunk >= n_chunks) return;
    const int t = threadIdx.x;
    const RowPair rp = my_rows(chunk, n_rows);
    const int row = rp.row;
    // which of the 2 ND - 1 entries the two rows have: bit (ND-1-j) = the one at -d[j], bit (ND-1+j) = at +d[j]
    const unsigned mm = *reinterpret_cast<const unsigned short *>(mask + row);
    const unsigned m0 = mm & 0xffu, m1 = mm >> 8;
    double2 acc;
    acc.x = acc.y = 0.0;
    if (MODE == SPMV_RESIDUAL) acc = ld2(b, rp);
    // own planes: diagonal and upper entries of the two rows
    double2 up[ND];
    const double *own = planes + (long)chunk * (ND * CHUNK_ROWS) + t * ROWS_PER_THREAD;
#pragma unroll
    for (int j = 0; j < ND; ++j)
        up[j] = (STREAM && (j == 0 || (FAST && j == 1))) ? ld_pair_stream(own + (long)j * CHUNK_ROWS)
                                                         : *reinterpret_cast<const double2 *>(own + (long)j * CHUNK_ROWS);
    const double2 xd = ld2(x, rp);
    // lower entries: plane j at rows row - d[j], row + 1 - d[j].  Every load is issued whatever the mask says, at an
    // index clamped into its array (the mask decides below what is used): the loads do not wait for the mask
    const int last = n_rows - 1, last_pair = last & ~1;  // (a pair load at the last even row: the vectors are allocated two past n_rows)
    double2 lo[ND];
#pragma unroll
    for (int j = 1; j < ND; ++j) {
        const int r0 = max(row - off.d[j], 0), r1 = max(row + 1 - off.d[j], 0);
        const long a0 = (long)(r0 >> 9) * (ND * CHUNK_ROWS) + (long)j * CHUNK_ROWS + (r0 & (CHUNK_ROWS - 1));
        const long a1 = (long)(r1 >> 9) * (ND * CHUNK_ROWS) + (long)j * CHUNK_ROWS + (r1 & (CHUNK_ROWS - 1));
        if (FAST && j >= 2) {  // even distance: rows r0, r0 + 1 are an aligned pair of one chunk's plane
            lo[j] = *reinterpret_cast<const double2 *>(planes + a0);
        } else if (FAST) {     // d = 1: A(row + 1, row) is this lane's own upper entry of row
            if (STREAM) {      // A(row, row - 1) is the previous lane's second plane-1 value (lane 0: from memory)
                lo[j].x = __shfl_up(up[1].y, 1, WAVE);
                if ((t & (WAVE - 1)) == 0) lo[j].x = planes[a0];
            } else {
                lo[j].x = planes[a0];
            }
            lo[j].y = up[1].x;
        } else {
            lo[j].x = planes[a0];
            lo[j].y = planes[a1];
        }
    }
    static_assert(CHUNK_ROWS == 512, "row >> 9 above");
    double2 xl[ND], xu[ND];
#pragma unroll
    for (int j = 1; j < ND; ++j) {
        if (FAST && j >= 2) {
            xl[j] = *reinterpret_cast<const double2 *>(x + min(max(row - off.d[j], 0), last_pair));
            xu[j] = *reinterpret_cast<const double2 *>(x + min(row + off.d[j], last_pair));
        } else if (FAST) {     // the neighbours of a pair at distance 1: the pair itself + one on each side
            xl[j].x = x[min(max(row - 1, 0), last)];
            xl[j].y = xd.x;
            xu[j].x = xd.y;
            xu[j].y = x[min(row + 2, last)];
        } else {
            xl[j].x = x[min(max(row - off.d[j], 0), last)];
            xl[j].y = x[min(max(row + 1 - off.d[j], 0), last)];
            xu[j].x = x[min(row + off.d[j], last)];
            xu[j].y = x[min(row + 1 + off.d[j], last)];
        }
    }
#pragma unroll
    for (int j = ND - 1; j >= 1; --j) {  // ascending columns: the furthest lower entry first
        if ((m0 >> (ND - 1 - j)) & 1u) {
            const double p = lo[j].x * xl[j].x;
            acc.x = (MODE == SPMV_RESIDUAL) ? acc.x - p : acc.x + p;
        }
        if ((m1 >> (ND - 1 - j)) & 1u) {
            const double p = lo[j].y * xl[j].y;
            acc.y = (MODE == SPMV_RESIDUAL) ? acc.y - p : acc.y + p;
        }
    }
    if ((m0 >> (ND - 1)) & 1u) {
        const double p = up[0].x * xd.x;
        acc.x = (MODE == SPMV_RESIDUAL) ? acc.x - p : acc.x + p;
    }
    if ((m1 >> (ND - 1)) & 1u) {
        const double p = up[0].y * xd.y;
        acc.y = (MODE == SPMV_RESIDUAL) ? acc.y - p : acc.y + p;
    }
#pragma unroll
    for (int j = 1; j < ND; ++j) {
        if ((m0 >> (ND - 1 + j)) & 1u) {
            const double p = up[j].x * xu[j].x;
            acc.x = (MODE == SPMV_RESIDUAL) ? acc.x - p : acc.x + p;
        }
        if ((m1 >> (ND - 1 + j)) & 1u) {
            const double p = up[j].y * xu[j].y;
            acc.y = (MODE == SPMV_RESIDUAL) ? acc.y - p : acc.y + p;
        }
    }
    if (hf.chunk_bptr) halo_fused_add<MODE>(hf, chunk, acc.x, acc.y, ys);
    st2(y, rp, acc);
    if (NDOT >= 1) {
        const double2 vw = ld2(w, rp);
        double d = 0.0, d2 = 0.0;
        if (rp.n > 0) {
            d += vw.x * acc.x;
            d2 += acc.x * acc.x;
        }
        if (rp.n > 1) {
            d += vw.y * acc.y;
            d2 += acc.y * acc.y;
        }
        const double s = block_sum(d, slot);
        if (threadIdx.x == 0) dot_partials[chunk] = s;
        if (NDOT >= 2) {
            const double s2 = block_sum(d2, slot);
            if (threadIdx.x == 0) dot2_partials[chunk] = s2;
        }
    }
}

// ------------------------------------------------------------------------------------------
// Small systems on half storage, two launches per GKOCG turn: step_1x_fin and the SpMV in one kernel.  The SpMV
// needs p_new = z + (rho/rho') p at the columns of its rows, which other workgroups own -- but p_new is an
// elementwise function of z (which step_2r_fin leaves behind for this; r itself without a preconditioner) and the
// old p, so every workgroup recomputes it for the columns it gathers (same expression, same rounding:
// -ffp-contract=off) instead of waiting for a kernel boundary.  The new p of the
// own rows goes to the other of two p buffers (neighbours still read the old one).  Everything else -- check of
// the previous turn, pending x update, row sums in ascending column order, partial of p.q -- is what
// k_cg_step1x_fin followed by k_spmv_sym<SPMV_PLAIN, 1> does, bit for bit.
//   turn = [this kernel] -> k_cg_step2r_fin
// ------------------------------------------------------------------------------------------
// what a thread of the merged kernels holds of its two rows: [0] p, [1] z at the rows and at the gathered columns,
// own planes (diagonal, upper entries) and the twins of the lower entries
template <int ND>
struct TurnSymRegs {
    double2 vd[2], vl[2][ND], vu[2][ND], up[ND], lo[ND];
};
// Every load is issued without waiting for the mask (mask -> gathers would be two round trips, and a small system
// is all latency), at an index clamped into the vector; the mask decides later what is used.
template <int ND, bool FAST, bool STREAM>
__device__ __forceinline__ void turn_sym_load(TurnSymRegs<ND> &R, int chunk, const RowPair &rp, int n_rows,
                                              const SymOffsets &off, const double *__restrict__ planes,
                                              const double *__restrict__ p_in, const double *__restrict__ z)
{
    const int row = rp.row;
    const double *src[2] = {p_in, z};
    const int last = n_rows - 1, last_pair = last & ~1;  // (a pair load at the last even row: the vectors are allocated two past n_rows)
#pragma unroll
    for (int a = 0; a < 2; ++a) {
        const double *__restrict__ v = src[a];
        R.vd[a] = ld2(v, rp);
#pragma unroll
        for (int j = 1; j < ND; ++j) {
            if (FAST && j >= 2) {
                R.vl[a][j] = *reinterpret_cast<const double2 *>(v + min(max(row - off.d[j], 0), last_pair));
                R.vu[a][j] = *reinterpret_cast<const double2 *>(v + min(row + off.d[j], last_pair));
            } else if (FAST) {
                R.vl[a][j].x = v[min(max(row - 1, 0), last)];
                R.vl[a][j].y = R.vd[a].x;
                R.vu[a][j].x = R.vd[a].y;
                R.vu[a][j].y = v[min(row + 2, last)];
            } else {
                R.vl[a][j].x = v[min(max(row - off.d[j], 0), last)];
                R.vl[a][j].y = v[min(max(row + 1 - off.d[j], 0), last)];
                R.vu[a][j].x = v[min(row + off.d[j], last)];
                R.vu[a][j].y = v[min(row + 1 + off.d[j], last)];
            }
        }
    }
    // the matrix, as k_spmv_sym reads it (STREAM: the planes read once per launch go past the caches)
    const double *own = planes + (long)chunk * (ND * CHUNK_ROWS) + threadIdx.x * ROWS_PER_THREAD;
#pragma unroll
    for (int j = 0; j < ND; ++j)
        R.up[j] = (STREAM && (j == 0 || (FAST && j == 1))) ? ld_pair_stream(own + (long)j * CHUNK_ROWS)
                                                           : *reinterpret_cast<const double2 *>(own + (long)j * CHUNK_ROWS);
#pragma unroll
    for (int j = 1; j < ND; ++j) {
        const int r0 = max(row - off.d[j], 0), r1 = max(row + 1 - off.d[j], 0);
        const long a0 = (long)(r0 >> 9) * (ND * CHUNK_ROWS) + (long)j * CHUNK_ROWS + (r0 & (CHUNK_ROWS - 1));
        const long a1 = (long)(r1 >> 9) * (ND * CHUNK_ROWS) + (long)j * CHUNK_ROWS + (r1 & (CHUNK_ROWS - 1));
        if (FAST && j >= 2) {
            R.lo[j] = *reinterpret_cast<const double2 *>(planes + a0);
        } else if (FAST) {
            if (STREAM) {  // A(row, row - 1) is the previous lane's second plane-1 value (lane 0: from memory)
                const double prev = __shfl_up(R.up[1].y, 1, WAVE);
                R.lo[j].x = prev;
                if ((threadIdx.x & (WAVE - 1)) == 0) R.lo[j].x = planes[a0];
            } else {
                R.lo[j].x = planes[a0];
            }
            R.lo[j].y = R.up[1].x;
        } else {
            R.lo[j].x = planes[a0];
            R.lo[j].y = planes[a1];
        }
    }
}
// p_new = z + tmp p for the own rows (-> xd) and for every gathered column, then the two row sums in ascending
// column order (k_spmv_sym's)
template <int ND>
__device__ __forceinline__ double2 turn_sym_rows(const TurnSymRegs<ND> &R, unsigned m0, unsigned m1, double tmp,
                                                 double2 &xd)
{
    double2 xl[ND], xu[ND];
    xd.x = R.vd[1].x + tmp * R.vd[0].x;
    xd.y = R.vd[1].y + tmp * R.vd[0].y;
#pragma unroll
    for (int j = 1; j < ND; ++j) {
        xl[j].x = R.vl[1][j].x + tmp * R.vl[0][j].x;
        xl[j].y = R.vl[1][j].y + tmp * R.vl[0][j].y;
        xu[j].x = R.vu[1][j].x + tmp * R.vu[0][j].x;
        xu[j].y = R.vu[1][j].y + tmp * R.vu[0][j].y;
    }
    double2 acc;
    acc.x = acc.y = 0.0;
#pragma unroll
    for (int j = ND - 1; j >= 1; --j) {  // ascending columns: the furthest lower entry first
        if ((m0 >> (ND - 1 - j)) & 1u) acc.x = acc.x + R.lo[j].x * xl[j].x;
        if ((m1 >> (ND - 1 - j)) & 1u) acc.y = acc.y + R.lo[j].y * xl[j].y;
    }
    if ((m0 >> (ND - 1)) & 1u) acc.x = acc.x + R.up[0].x * xd.x;
    if ((m1 >> (ND - 1)) & 1u) acc.y = acc.y + R.up[0].y * xd.y;
#pragma unroll
    for (int j = 1; j < ND; ++j) {
        if ((m0 >> (ND - 1 + j)) & 1u) acc.x = acc.x + R.up[j].x * xu[j].x;
        if ((m1 >> (ND - 1 + j)) & 1u) acc.y = acc.y + R.up[j].y * xu[j].y;
    }
    return acc;
}

// LEAD (more than FUSED_FIN_MAX_CHUNKS chunks): the first 32 workgroups of the launch are the finaliser's wavefronts, every
// workgroup fetches their sums from the mailbox (leader finalisation, device_common.hpp; k_cg_step1x_fin<true>)
template <int ND, bool FAST, bool LEAD>
__global__ __launch_bounds__(BLOCK) void k_cg_turn_sym(int n_rows, int n_chunks, SymOffsets off,
                                                       const uint8_t *__restrict__ mask,
                                                       const double *__restrict__ planes,
                                                       const double *__restrict__ p_in, double *__restrict__ p_out,
                                                       double *__restrict__ x, const double *__restrict__ z,
                                                       double *__restrict__ q,
                                                       double *__restrict__ part_beta, const DevScalars *sin,
                                                       DevScalars *sout, const double *__restrict__ part_rho,
                                                       const double *__restrict__ part_norm, int n_part,
                                                       double *history, int first,
                                                       const int *__restrict__ block_order, LeadBox lbox)
{
    __shared__ double red[2 * FIN_WAVES];
    __shared__ double sh[4];
    __shared__ int sh_stop;
    __shared__ double slot[N_WAVES];
    __shared__ double lead_words[LEAD ? LEAD_BOX_WORDS / 2 : 1];
    __shared__ double lead_stage[LEAD ? LEAD_STAGE : 1];
    __shared__ int lead_timed_out;
    const uint32_t seq = LEAD ? sin->launch_seq : 0u;
    if (LEAD) lead_leaders<2>(lbox, seq, part_rho, part_norm, nullptr, n_part, lead_stage);  // (by launch position, not by chunk)
    const int chunk = block_order ? block_order[blockIdx.x] : xcd_chunk(blockIdx.x);
    if (chunk < 0 || chunk >= n_chunks) return;
    const bool lead = chunk == 0;  // the workgroup that stores the scalars and the history entry
    // scalars field by field (k_cg_step1x_fin), and every load of the kernel asked for before the first wait
    const int stopped = sin->stop;
    const double s_rho = sin->rho, s_beta = sin->beta, s_nf = sin->norm_factor, s_init = sin->init_res;
    const int s_iter = sin->iter, s_evals = sin->n_evals;
    const double c_tol = sin->crit.tolerance, c_rel = sin->crit.rel_tol;
    const int c_min = sin->crit.min_iter, c_max = sin->crit.max_iter, c_freq = sin->crit.frequency,
              c_exp = sin->crit.export_res;
    if (lead && threadIdx.x < sizeof(DevScalars) / 8)
        reinterpret_cast<unsigned long long *>(sout)[threadIdx.x] =
            reinterpret_cast<const unsigned long long *>(sin)[threadIdx.x];
    const RowPair rp = my_rows(chunk, n_rows);
    const unsigned mm = *reinterpret_cast<const unsigned short *>(mask + rp.row);
    const unsigned m0 = mm & 0xffu, m1 = mm >> 8;
    double pv[2][FIN_VT];
    if (!LEAD) load_partials_as_finaliser<2>(part_rho, part_norm, n_part, pv);
    double2 vx = ld2(x, rp);
    TurnSymRegs<ND> R;
    turn_sym_load<ND, FAST, false>(R, chunk, rp, n_rows, off, planes, p_in, z);
    if (stopped) return;  // (the solve has ended: the lead workgroup has handed the scalars on)
    double v[2] = {0.0, 0.0};
    if (LEAD) {
        if (!lead_wait(lbox, 4 * FIN_WAVES, seq, lead_words, &lead_timed_out)) {
            if (threadIdx.x == 0) sout->comm_error = sout->stop = 1;
            return;
        }
        if (threadIdx.x == 0) {
            v[0] = lead_total(lead_words, 0);
            v[1] = lead_total(lead_words, 1);
        }
    } else {
        reduce_partials_as_finaliser<2>(pv, n_part, red, v);
    }
    if (threadIdx.x == 0) {
        // FIN_CG_CHECK as in k_cg_step1x_fin (StoppingCriterion.C:71-151)
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
            if (c_exp && history && lead) history[iter] = res;  // :115-117
            if (iter >= c_max) stop = 1;                        // :124
            if (res < c_tol) stop = 1;                          // :128
            if (c_rel > 0 && res < c_rel * init_res) stop = 1;  // :132-136
            iter += 1;                                          // :143
        }
        sh[0] = s_beta;
        sh[1] = prev_rho;
        sh[2] = rho;
        sh_stop = stop;
        if (lead) {
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
    const double beta = sh[0], prev = sh[1], rho = sh[2];
    const int stop = sh_stop;
    if (!first && beta != 0.0) {  // x += t_j p of the turn this check closed
        const double t = prev / beta;
        vx.x += t * R.vd[0].x;
        vx.y += t * R.vd[0].y;
        st2(x, rp, vx);
    }
    if (stop) return;
    const double tmp = (prev == 0.0) ? 0.0 : rho / prev;
    double2 xd;
    const double2 acc = turn_sym_rows<ND>(R, m0, m1, tmp, xd);
    st2(p_out, rp, xd);
    st2(q, rp, acc);
    double d = 0.0;
    if (rp.n > 0) d += xd.x * acc.x;
    if (rp.n > 1) d += xd.y * acc.y;
    const double sd = block_sum(d, slot);
    if (threadIdx.x == 0) part_beta[chunk] = sd;
}

// The same merge for systems of any size, between the single-workgroup finalisers of the five-launch turn (which
// becomes four): the scalars are read where k_cg_step1x reads them, the pending x update included.  Per turn the
// vectors cost 8 N bytes less than step_1x + SpMV + step_2r (p is read once, z written once and read once instead of
// r and 1/d read twice), and one kernel boundary goes.
// HALO (several ranks, peer-put transport): the neighbours' step_2r has put z of the halo columns; a workgroup whose
// chunk holds boundary rows waits for it, forms p_new at those columns from the old halo p it keeps (ph_in -> ph_out)
// and continues its boundary rows over their non-local entries -- no put and no wait for a put of THIS launch here.
template <int ND, bool FAST, bool STREAM, bool HALO>
__global__ __launch_bounds__(BLOCK) void k_cg_turn_sym_big(int n_rows, int n_chunks, SymOffsets off,
                                                           const uint8_t *__restrict__ mask,
                                                           const double *__restrict__ planes,
                                                           const double *__restrict__ p_in,
                                                           double *__restrict__ p_out, double *__restrict__ x,
                                                           const double *__restrict__ z, double *__restrict__ q,
                                                           double *__restrict__ part_beta, const DevScalars *s,
                                                           const int *__restrict__ block_order, HaloFused hf,
                                                           const double *__restrict__ ph_in,
                                                           double *__restrict__ ph_out)
{
    __shared__ double slot[N_WAVES];
    const int stop = s->stop;
    const bool pending = s->x_pending != 0;
    if (stop && !pending) return;
    const int chunk = block_order ? block_order[blockIdx.x] : xcd_chunk(blockIdx.x);
    if (chunk < 0 || chunk >= n_chunks) return;
    const RowPair rp = my_rows(chunk, n_rows);
    if (stop) {  // the solve has ended with an update still to apply: that only
        const double beta = s->beta;
        if (beta != 0.0) {
            const double t = s->prev_rho / beta;
            const double2 vp = ld2(p_in, rp);
            double2 vx = ld2_stream(x, rp);
            vx.x += t * vp.x;
            vx.y += t * vp.y;
            st2_stream(x, rp, vx);
        }
        return;
    }
    const unsigned mm = *reinterpret_cast<const unsigned short *>(mask + rp.row);
    const unsigned m0 = mm & 0xffu, m1 = mm >> 8;
    TurnSymRegs<ND> R;
    turn_sym_load<ND, FAST, STREAM>(R, chunk, rp, n_rows, off, planes, p_in, z);
    if (pending) {
        const double beta = s->beta;
        if (beta != 0.0) {
            const double t = s->prev_rho / beta;
            double2 vx = ld2_stream(x, rp);  // x is touched once per turn
            vx.x += t * R.vd[0].x;
            vx.y += t * R.vd[0].y;
            st2_stream(x, rp, vx);
        }
    }
    const double rho = s->rho, prev = s->prev_rho;
    const double tmp = (prev == 0.0) ? 0.0 : rho / prev;
    double2 xd;
    double2 acc = turn_sym_rows<ND>(R, m0, m1, tmp, xd);
    st2(p_out, rp, xd);
    if (HALO) {
        __shared__ double ys[CHUNK_ROWS];
        if (hf.chunk_bptr) halo_fused_add<SPMV_PLAIN, true>(hf, chunk, acc.x, acc.y, ys, tmp, ph_in, ph_out);
    }
    st2(q, rp, acc);
    double d = 0.0;
    if (rp.n > 0) d += xd.x * acc.x;
    if (rp.n > 1) d += xd.y * acc.y;
    const double sd = block_sum(d, slot);
    if (threadIdx.x == 0) part_beta[chunk] = sd;
}

}  // namespace

// ------------------------------------------------------------------------------------------
// launchers
// ------------------------------------------------------------------------------------------
void launch_spmv_sym(hipStream_t st, const DevSym &A, int mode, const double *x, const double *b, double *y,
                     const SpmvDots &dots, const DevScalars *gate, const HaloFused &hf)
{
    if (A.n_rows == 0) return;
    const int nc = (int)n_chunks(A.n_rows);
    const dim3 grid(A.block_order ? A.n_blocks : xcd_grid(nc)), block(BLOCK);
    SymOffsets off;
    for (int j = 0; j < SYM_MAX_OFFSETS; ++j) off.d[j] = A.d[j];
    // d[1] == 1 and the further distances even: the straight-line pair-load instantiation
    bool fast = A.nd >= 2 && A.d[1] == 1;
    for (int j = 2; j < A.nd; ++j) fast = fast && (A.d[j] % 2 == 0);
#define OGL_SYM_K(MODE, NDOT, ND, FAST, STREAM)                                                                          \
    hipLaunchKernelGGL((k_spmv_sym<MODE, NDOT, ND, FAST, STREAM>), grid, block, 0, st, A.n_rows, nc, off, A.mask, A.planes, \
                       x, b, y, dots.with, dots.part, dots.part_yy, gate, A.block_order, hf)
#define OGL_SYM_ND(MODE, NDOT, ND)                   \
    do {                                             \
        if (fast && A.stream)                        \
            OGL_SYM_K(MODE, NDOT, ND, true, true);   \
        else if (fast)                               \
            OGL_SYM_K(MODE, NDOT, ND, true, false);  \
        else if (A.stream)                           \
            OGL_SYM_K(MODE, NDOT, ND, false, true);  \
        else                                         \
            OGL_SYM_K(MODE, NDOT, ND, false, false); \
    } while (0)
#define OGL_SYM(MODE, NDOT)                    \
    do {                                       \
        if (A.nd == 2)                         \
            OGL_SYM_ND(MODE, NDOT, 2);         \
        else if (A.nd == 3)                    \
            OGL_SYM_ND(MODE, NDOT, 3);         \
        else                                   \
            OGL_SYM_ND(MODE, NDOT, 4);         \
    } while (0)
    static_assert(SYM_MAX_OFFSETS == 4, "instantiations above");
    if (mode == SPMV_RESIDUAL) {
        OGL_SYM(SPMV_RESIDUAL, 0);
    } else if (dots.part && dots.part_yy) {
        OGL_SYM(SPMV_PLAIN, 2);
    } else if (dots.part) {
        OGL_SYM(SPMV_PLAIN, 1);
    } else {
        OGL_SYM(SPMV_PLAIN, 0);
    }
#undef OGL_SYM
#undef OGL_SYM_ND
#undef OGL_SYM_K
}

void launch_cg_turn_sym(hipStream_t st, const DevSym &A, const double *p_in, double *p_out, double *x, const double *z,
                        double *q, double *part_beta, const DevScalars *sin, DevScalars *sout,
                        const double *part_rho, const double *part_norm, double *history, int first, const LeadBox &lead)
{
    if (A.n_rows == 0) return;
    const int nc = (int)n_chunks(A.n_rows);
    const dim3 grid(A.block_order ? A.n_blocks : xcd_grid(nc)), block(BLOCK);
    const bool led = lead.box && nc >= 3 * FIN_WAVES;
    SymOffsets off;
    for (int j = 0; j < SYM_MAX_OFFSETS; ++j) off.d[j] = A.d[j];
    bool fast = A.nd >= 2 && A.d[1] == 1;
    for (int j = 2; j < A.nd; ++j) fast = fast && (A.d[j] % 2 == 0);
#define OGL_TURN_K(ND, FAST)                                                                                                  \
    do {                                                                                                                      \
        if (led)                                                                                                              \
            hipLaunchKernelGGL((k_cg_turn_sym<ND, FAST, true>), grid, block, 0, st, A.n_rows, nc, off, A.mask, A.planes, p_in, \
                               p_out, x, z, q, part_beta, sin, sout, part_rho, part_norm, nc, history, first, A.block_order,  \
                               lead);                                                                                         \
        else                                                                                                                  \
            hipLaunchKernelGGL((k_cg_turn_sym<ND, FAST, false>), grid, block, 0, st, A.n_rows, nc, off, A.mask, A.planes,     \
                               p_in, p_out, x, z, q, part_beta, sin, sout, part_rho, part_norm, nc, history, first,           \
                               A.block_order, LeadBox{});                                                                     \
    } while (0)
#define OGL_TURN_ND(ND)              \
    do {                             \
        if (fast)                    \
            OGL_TURN_K(ND, true);    \
        else                         \
            OGL_TURN_K(ND, false);   \
    } while (0)
    if (A.nd == 2)
        OGL_TURN_ND(2);
    else if (A.nd == 3)
        OGL_TURN_ND(3);
    else
        OGL_TURN_ND(4);
#undef OGL_TURN_ND
#undef OGL_TURN_K
}

void launch_cg_turn_sym_big(hipStream_t st, const DevSym &A, const double *p_in, double *p_out, double *x,
                            const double *z, double *q, double *part_beta, const DevScalars *s,
                            const HaloFused &hf, const double *p_halo_in, double *p_halo_out)
{
    if (A.n_rows == 0) return;
    const int nc = (int)n_chunks(A.n_rows);
    const dim3 grid(A.block_order ? A.n_blocks : xcd_grid(nc)), block(BLOCK);
    SymOffsets off;
    for (int j = 0; j < SYM_MAX_OFFSETS; ++j) off.d[j] = A.d[j];
    bool fast = A.nd >= 2 && A.d[1] == 1;
    for (int j = 2; j < A.nd; ++j) fast = fast && (A.d[j] % 2 == 0);
#define OGL_TURN_K(ND, FAST, STREAM)                                                                                \
    do {                                                                                                            \
        if (hf.chunk_bptr)                                                                                          \
            hipLaunchKernelGGL((k_cg_turn_sym_big<ND, FAST, STREAM, true>), grid, block, 0, st, A.n_rows, nc, off,  \
                               A.mask, A.planes, p_in, p_out, x, z, q, part_beta, s, A.block_order, hf, p_halo_in,  \
                               p_halo_out);                                                                         \
        else                                                                                                        \
            hipLaunchKernelGGL((k_cg_turn_sym_big<ND, FAST, STREAM, false>), grid, block, 0, st, A.n_rows, nc, off, \
                               A.mask, A.planes, p_in, p_out, x, z, q, part_beta, s, A.block_order, HaloFused{},    \
                               nullptr, nullptr);                                                                   \
    } while (0)
#define OGL_TURN_ND(ND)                     \
    do {                                    \
        if (fast && A.stream)               \
            OGL_TURN_K(ND, true, true);     \
        else if (fast)                      \
            OGL_TURN_K(ND, true, false);    \
        else if (A.stream)                  \
            OGL_TURN_K(ND, false, true);    \
        else                                \
            OGL_TURN_K(ND, false, false);   \
    } while (0)
    if (A.nd == 2)
        OGL_TURN_ND(2);
    else if (A.nd == 3)
        OGL_TURN_ND(3);
    else
        OGL_TURN_ND(4);
#undef OGL_TURN_ND
#undef OGL_TURN_K
}

}  // namespace ogl
