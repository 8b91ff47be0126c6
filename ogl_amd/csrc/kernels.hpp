// kernels.hpp -- launchers of the hand-written gfx950 kernels (kernels_*.hip; what they share: device_common.hpp).
// Everything the Krylov loop of the reference delegates to Ginkgo.
#pragma once
#include <hip/hip_runtime.h>

#include "common.hpp"

namespace ogl {

// OpenFOAMDistStoppingCriterion parameters (StoppingCriterion.H:32-72)
struct DevCriterion {
    double tolerance, rel_tol;
    int32_t min_iter, max_iter, frequency, export_res;
};

// Solver scalars and criterion state, resident in device memory for the whole solve so that the
// loop never synchronises with the host (the reference syncs D2H on every evaluated check,
// StoppingCriterion.C:95-97).
struct DevScalars {
    double rho, prev_rho, beta;         // CG  ([UPSTREAM] gko::solver::Cg scalars)
    double alpha, omega, gamma;         // BiCGStab extras
    double norm_factor;                 // StoppingCriterion.H:136
    double init_res, res;               // init_normalised_res_norm_, normalised_res_norm_
    double xbar;                        // mean(x) for the norm factor (StoppingCriterion.C:17-19)
    double sums[4];                     // rank-local sums staged for the all-reduce
    int32_t iter;                       // iter_ : number of check_impl calls so far
    int32_t stop;                       // criterion said stop; later kernels become no-ops
    int32_t n_evals;                    // checks that evaluated the norm
    int32_t stop_phase;                 // BiCGStab: 1 = stopped at the mid-step check (finalize x)
    int32_t stop_turn;                  // BiCGStab: turn index of that stop
    int32_t comm_error;                 // peer all-reduce timed out (a rank is gone): solve fails
    double stale_norm;                  // GMRES: sum|r| of the last restart (what the criterion sees)
    DevCriterion crit;                  // this solve's criterion (kernel arguments stay solve-independent)
    int32_t x_pending;                  // GKOCG: step_2r's x update is still to be applied by a step_1x
    // GKOCG with x touched every K-th turn (PRing): bit i of defer_valid = the head of ring position i left
    // t_ring[i] * (its old p, still intact in ring buffer i) pending
    int32_t defer_valid;
    double t_ring[8];
    // Where a multi-rank turn waits (per solve; wall_clock64 ticks of 10 ns): halo_wait_ticks = sum over the workgroups
    // that waited for the neighbours' puts of the longest of their flag waits (halo_waits of them: boundary workgroups
    // of the SpMV, or the single waiter of peerSafeWait / the separate finish kernel); reduce_wait_ticks = sum over
    // the in-finaliser all-reduces of the longest mailbox wait (reduce_waits of them).  bench.py reports both per turn
    // and rank, so that a scaling curve below DESIGN.md section 6's table can be attributed.
    uint32_t halo_waits;
    unsigned long long halo_wait_ticks, reduce_wait_ticks;
    uint32_t reduce_waits;
    uint32_t launch_seq;  // leader finalisation (LeadBox): tag of the next leader launch's mailbox words
};

// GKOCG, three-launch leader turn: K search-direction buffers used in turn (the head of turn j reads b[j % K] and writes
// b[(j + 1) % K]), so that x is read and written by every K-th head only (k_cg_step1x_fin).  k == 0: p in place, x every turn.
constexpr int P_RING_MAX = 8;
struct PRing {
    double *b[P_RING_MAX] = {nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};
    int32_t k = 0;      // 0 | 2 | 4 | 8
    int32_t phase = 0;  // turn % k of the head this is handed to
};

// Persistent device CSR ("<field>_matrix", CsrMatrixWrapper.H:163-210) + halo part.
struct DevCsr {
    int32_t n_rows = 0;
    int32_t nnz = 0;
    const int32_t *row_ptrs = nullptr;  // [n_rows + 1]
    const int32_t *cols = nullptr;      // [nnz + NNZ_PAD]
    const double *vals = nullptr;       // [nnz + NNZ_PAD]
    // the matrix does not fit the Infinity Cache next to the solver's vectors: its values and columns are
    // streamed past the caches (non-temporal loads); a matrix that fits stays cached from turn to turn
    bool stream = false;
    // consecutive chunks one XCD takes before the next XCD's group starts (0: the built-in 4).  Large groups
    // (slabs of tens of thousands of rows) keep the x window of an irregular pattern in ONE L2
    int32_t xcd_group = 0;
    // packed columns (Stream21Chunk, common.hpp): when set, the kernel reads these instead of `cols`
    const Stream21Chunk *chunks21 = nullptr;
    const uint4 *codes21 = nullptr;
    const int32_t *far_idx21 = nullptr, *far_col21 = nullptr;  // the chunks' far entries (Stream21Chunk::far_off / far_n)
    // banded patterns: the order in which the workgroups take the chunks (band_block_order, host_matrix.hpp: the
    // chunks of rows r and r +- band on one XCD; -1 = no chunk), n_blocks entries; nullptr = the XCD groups above
    const int32_t *block_order = nullptr;
    int32_t n_blocks = 0;
    // CSR-stream kernel: rounds in which a pass's 4096 products go through LDS (2: half the LDS, six workgroups per CU
    // instead of four -- measured 199 against 196 us at 216^3: the kernel does not wait for wavefronts; property spmvLdsRounds)
    int32_t lds_rounds = 1;
};

// Rows that own non-local entries, for "y += A_non_local * recv" (distributed::Matrix::apply).
struct DevHalo {
    int32_t n_boundary_rows = 0;            // distinct rows with non-local entries
    const int32_t *boundary_rows = nullptr; // [n_boundary_rows]
    const int32_t *entry_ptrs = nullptr;    // [n_boundary_rows + 1] into cols/vals
    const int32_t *cols = nullptr;          // position in the receive buffer
    const double *vals = nullptr;           // -bouCoeffs, row-sorted (HostMatrix.C:708-732)
    int32_t n_send = 0;
    const int32_t *send_idxs = nullptr;     // rows gathered into the send buffer
};

// Mailbox of the leader finalisation (device_common.hpp): a few 8-byte words in fine-grained (uncached) device memory,
// written by workgroup 0 of a launch and polled by the others.  box == nullptr: every workgroup reduces the partials
// itself (small systems) -- or the finalisers are launches of their own.
struct LeadBox {
    unsigned long long *box = nullptr;
    long long timeout_ticks = 0;  // wall_clock64 ticks (10 ns) a polling workgroup waits before it gives the solve up
    int32_t early_loads = 0;      // 1: a polling workgroup asks for its rows before it polls (else after)
};
constexpr int LEAD_BOX_WORDS = 96;  // 16 wavefront sums x up to 3 arrays x 2 half-words
constexpr int LEAD_REPLICAS = 16;   // copies of the mailbox (workgroup b polls copy b % 16), LEAD_REPLICA_STRIDE words apart
constexpr int LEAD_REPLICA_STRIDE = 544;  // 4352 bytes: 4 KiB + 256, so that the copies fall into different channels

enum SpmvMode { SPMV_PLAIN = 0, SPMV_RESIDUAL = 1 };

// y = A x (PLAIN) or y = b - A x (RESIDUAL: accumulator starts at b_i and subtracts the products
// in stored order, which is what Ginkgo's advanced apply with alpha=-1, beta=1 evaluates).
// dots.part != nullptr: also one partial of sum_i w_i*y_i (and y_i*y_i) per chunk of CHUNK_ROWS rows.
// `gate` (may be nullptr): kernel returns immediately when gate->stop is set.
struct SpmvDots {
    const double *with = nullptr;  // w in sum_i w_i*y_i (CG: x itself; BiCGStab: rr or s)
    double *part = nullptr;        // per-chunk partials of w.y, or nullptr
    double *part_yy = nullptr;     // per-chunk partials of y.y (needs `part`), or nullptr
};
// Non-local part of the product inside the LOCAL kernel (peer-put transport): the workgroup of a chunk that
// holds boundary rows waits for the neighbours' flags of this SpMV, then continues the accumulators of its
// boundary rows over their non-local entries in stored order -- local first, then non-local, as
// distributed::Matrix::apply does -- before y and the fused dot partials are formed.  chunk_bptr == nullptr: none.
struct HaloFused {
    const int32_t *chunk_bptr = nullptr;     // [n_chunks + 1] ranges of boundary_rows per chunk
    const int32_t *boundary_rows = nullptr;  // as DevHalo
    const int32_t *entry_ptrs = nullptr;
    const int32_t *cols = nullptr;
    const double *vals = nullptr;
    const double *recv = nullptr;                    // this SpMV's receive block
    const unsigned long long *local_flag = nullptr;  // this rank's flags, one per neighbour
    int32_t n_neigh = 0;
    uint32_t seq = 0;
    long long timeout_ticks = 0;
    DevScalars *s = nullptr;                 // receives comm_error / stop when a neighbour does not show up
};
void launch_spmv(hipStream_t st, const DevCsr &A, int mode, const double *x, const double *b,
                 double *y, const SpmvDots &dots, const DevScalars *gate, const HaloFused &hf = HaloFused{});

// matrixFormat Ell (CsrMatrixWrapper.H:146-149): `width` slots per row, slot-major (column-major in
// Ginkgo's terms) with leading dimension `stride` >= n_rows; padding slots carry column -1 and are
// skipped, so a row is summed in the same stored order as in the CSR kernel (bit-identical y).
struct DevEll {
    int32_t n_rows = 0;
    int32_t width = 0;
    int64_t stride = 0;
    const int32_t *cols = nullptr;  // [width * stride]
    const double *vals = nullptr;   // [width * stride]
    bool stream = false;            // as DevCsr::stream
};
void launch_spmv_ell(hipStream_t st, const DevEll &A, int mode, const double *x, const double *b,
                     double *y, const SpmvDots &dots, const DevScalars *gate, const HaloFused &hf = HaloFused{});
// Index-compressed chunked ELL (SellChunk, common.hpp), device view.
struct DevSell {
    int32_t n_rows = 0;
    const SellChunk *chunks = nullptr;  // [n_chunks]
    const int32_t *dict = nullptr;      // column - row offsets, ascending per chunk
    const uint8_t *codes = nullptr;     // thread-major: [thread][slot][row of the pair]
    const double *vals = nullptr;
    // spill (nullptr: none): tails of the rows longer than their chunk's cap, added by the chunk's
    // workgroup after the planes.  spill_chunk_ptr[c] .. [c + 1] = this chunk's range of spill_rows;
    // spill_ptrs = their entry ranges in spill_cols / spill_vals
    const int32_t *spill_chunk_ptr = nullptr, *spill_rows = nullptr, *spill_ptrs = nullptr, *spill_cols = nullptr;
    const double *spill_vals = nullptr;
    bool stream = false;  // as DevCsr::stream, for the value planes and the 16 / 32-bit code words
    int32_t xcd_group = 0;  // as DevCsr::xcd_group
    // rows of a wavefront's window stored in an order of their own (longest first: SellDev::build, sort_windows):
    // rmap[chunk * CHUNK_ROWS + slot row] = the row of the chunk whose entries the slot row holds.  The workgroup
    // hands the sums back to the rows' owners through LDS before y and the dot partials are formed, so y, the
    // per-row order of the products and the partials are those of the plain layout.  PLAIN mode, no spill, no halo.
    const uint16_t *rmap = nullptr;
    const int32_t *block_order = nullptr;  // as DevCsr::block_order
    int32_t n_blocks = 0;
};
void launch_spmv_sell(hipStream_t st, const DevSell &A, int mode, const double *x, const double *b,
                      double *y, const SpmvDots &dots, const DevScalars *gate, const HaloFused &hf = HaloFused{});
// Half storage of a symmetric matrix on a banded pattern (SymLayout, host_matrix.hpp), device view.
struct DevSym {
    int32_t n_rows = 0;
    int32_t nd = 0;                 // planes per chunk: diagonal + distances
    int32_t d[4] = {0, 0, 0, 0};    // the distances, ascending, d[0] = 0
    const uint8_t *mask = nullptr;  // [n_chunks * CHUNK_ROWS] which entries a row has
    const double *planes = nullptr; // [n_chunks][nd][CHUNK_ROWS]
    bool stream = false;            // as DevCsr::stream, for the planes that are read once per launch
    // workgroup b takes chunk block_order[b] (-1: none), n_blocks workgroups (band_block_order); nullptr: default map
    const int32_t *block_order = nullptr;
    int32_t n_blocks = 0;
};
void launch_spmv_sym(hipStream_t st, const DevSym &A, int mode, const double *x, const double *b, double *y,
                     const SpmvDots &dots, const DevScalars *gate, const HaloFused &hf = HaloFused{});
// Half storage with per-chunk distances and explicit exceptions (SymxChunk, common.hpp), device view.
struct DevSymx {
    int32_t n_rows = 0;
    // headers in dispatch order (symx_block_order): workgroup b works on chunks[b].chunk.  Two lists: the chunks the
    // lean kernel takes (no explicit entries or simple ones) and the ones the general kernel takes
    const SymxChunk *chunks = nullptr, *chunks_general = nullptr;
    int32_t n_blocks_general = 0;
    const uint8_t *mask = nullptr;      // [n_chunks * CHUNK_ROWS]
    const double *planes = nullptr;
    const int32_t *ex_rowptr = nullptr, *ex_cols = nullptr;  // explicit entries (per-chunk row pointers, columns)
    const int32_t *ex_lrow = nullptr;                        // ... their rows within the chunk | SYMX_BEHIND_BIT
    const double *ex_vals = nullptr;
    bool stream = false;
    bool fast = false;  // every chunk: first distance 1, further distances even (pair-load instantiation)
    int32_t xcd_group = 0;
    int32_t n_blocks = 0;
};
void launch_spmv_symx(hipStream_t st, const DevSymx &A, int mode, const double *x, const double *b, double *y,
                      const SpmvDots &dots, const DevScalars *gate, const HaloFused &hf = HaloFused{});
// out[i] = map[i] >= 0 ? source[map[i]] : 0   (coefficient permutation into the padded ELL slots)
void launch_gather_coeffs_masked(hipStream_t st, int64_t n, const int32_t *map, const double *source,
                                 double *out);
// the same into the value planes of a chunked layout (SellChunk), one workgroup per chunk
void launch_gather_sell(hipStream_t st, int32_t n_chunks, const SellChunk *chunks, const int32_t *map,
                        const double *source, double *out);

// y[row] (+/-)= sum_k vals[k] * recv[cols[k]] over the boundary rows, continuing y's accumulator.
void launch_spmv_non_local(hipStream_t st, const DevHalo &H, int mode, const double *recv,
                           double *y, const DevScalars *gate);
void launch_pack(hipStream_t st, const DevHalo &H, const double *x, double *send,
                 const DevScalars *gate);

// coeffs[i] = source[ldu_mapping[i]]   (row_gather, HostMatrix.C:700-703)
void launch_gather_coeffs(hipStream_t st, int32_t nnz, const int32_t *ldu_mapping,
                          const double *source, double *coeffs);
// inv_diag[i] = 1 / A(i,i)   (Jacobi generate with max_block_size 1)
void launch_jacobi_generate(hipStream_t st, const DevCsr &A, double *inv_diag);
// the same with diag_pos[i] = position of row i's first diagonal entry (-1: none), found once per pattern
void launch_jacobi_generate_pos(hipStream_t st, const DevCsr &A, const int32_t *diag_pos, double *inv_diag);
// ... from a contiguous copy of the diagonal in device-row order (the staged lduMatrix source)
void launch_jacobi_generate_diag(hipStream_t st, int32_t n_rows, const double *diag, double *inv_diag);

// Block Jacobi with maxBlockSize k > 1 (Preconditioner.H:91-105): inverted diagonal blocks,
// row-major, `stride` x `stride` doubles per block.
struct DevBlockJacobi {
    int32_t n_rows = 0;
    int32_t n_blocks = 0;
    int32_t stride = 0;                  // = maxBlockSize
    const int32_t *block_ptrs = nullptr; // [n_blocks + 1] first row of each block
    const int32_t *row_block = nullptr;  // [n_rows] block of each row
    double *blocks = nullptr;            // [n_blocks * stride * stride]
    // every block but the last holds exactly `stride` rows (what agglomeration gives on a mesh without
    // repeated row patterns): block and first row follow from the row index, no index loads in the apply
    int32_t uniform = 0;
    // renumbered device copy: the blocks are those of the CALLER's numbering (Preconditioner.H:91-105 generates on
    // the matrix OpenFOAM hands over) -- block_ptrs / row_block refer to positions there, rows[position] = device
    // row, pos[device row] = position.  nullptr: the device copy is in the caller's numbering.
    const int32_t *rows = nullptr, *pos = nullptr;
    // ... and the block rows are stored at their device rows (rows[position] * stride; the direct apply) instead of
    // block-major in the caller's order (the staged apply)
    int32_t by_device_row = 0;
    // k_bj_apply_perm: consecutive 512-position ranges one XCD takes (0: chosen from the size, 4 .. 256)
    int32_t perm_xcd_group = 0;
};
constexpr int MAX_JACOBI_BLOCK = 32;
// blocks[b] = inverse of A(block b, block b) by Gauss-Jordan with partial pivoting
void launch_bj_generate(hipStream_t st, const DevCsr &A, const DevBlockJacobi &J,
                        bool group_lanes = true);  // blocks of <= 8 rows: 4 | 8 lanes per block instead of a thread
// out = M^-1 in : per row, the block row times the block's slice of `in`, summed left to right;
// dot_part != nullptr: also the per-chunk partials of sum_i in_i * out_i
void launch_bj_apply(hipStream_t st, const DevBlockJacobi &J, const double *in, double *out,
                     double *dot_part, const DevScalars *gate);
// the same through a permutation in three launches (J.rows / J.pos set, blocks block-major in the caller's order):
// in -> caller's order (one gather per row), contiguous apply, back to the device order + the dot partials.
// tmp_in / tmp_out: two scratch vectors of n_rows doubles
void launch_bj_apply_staged(hipStream_t st, const DevBlockJacobi &J, const double *in, double *out, double *dot_part,
                            const DevScalars *gate, double *tmp_in, double *tmp_out);

// ISAI / GISAI (Preconditioner.H:225-258): row i of the approximate inverse W lives on the pattern J of
// row i of S^sparsityPower (S = tril(A) resp. A, built on the host) and solves a dense system over it by
// Gaussian elimination with partial pivoting.  spd: A(J,J) y = e_i, W(i,J) = y / sqrt(y_i); general:
// A(J,J)^T y = e_i, W(i,J) = y.  Rows of up to ISAI_THREAD_ROW entries: one thread per row (per-thread
// arrays); longer ones, up to MAX_ISAI_ROW: one wavefront per row, the system in LDS, lane = column --
// the same operations on every element in the same order, so both give the oracle's bits.
// Rows beyond MAX_ISAI_ROW, up to MAX_ISAI_HUGE_ROW (2048: the square of a polyhedral mesh's pattern): one WORKGROUP per row, the system in global scratch
// (bs x bs doubles per row), the same elimination, the back substitution column by column (the oracle's
// solve_dense_wide: a row-wise walk would be one dependent chain of bs^2 / 2 operations).  [UPSTREAM] Ginkgo solves
// rows beyond its in-kernel limit through an iterative "excess system" (GMRES to 1e-6): an approximation of this.
constexpr int ISAI_THREAD_ROW = 32;
constexpr int MAX_ISAI_ROW = 64;
constexpr int MAX_ISAI_HUGE_ROW = 2048;
// max_row = longest row of W (selects the per-thread scratch size: 8, 16 or 32);
// wide_rows[n_wide] = the rows with ISAI_THREAD_ROW < entries <= MAX_ISAI_ROW
void launch_isai_generate(hipStream_t st, const DevCsr &A, int spd, const int32_t *w_row_ptrs,
                          const int32_t *w_cols, double *w_vals, int32_t max_row,
                          const int32_t *wide_rows, int32_t n_wide,
                          bool group_lanes = true);  // rows of <= 8 entries: 4 | 8 lanes per row instead of a thread
// huge_rows[first .. first + count): rows wider than MAX_ISAI_ROW; scratch_off[k] = where row huge_rows[k]'s
// system starts in `scratch` (doubles)
void launch_isai_generate_huge(hipStream_t st, const DevCsr &A, int spd, const int32_t *w_row_ptrs,
                               const int32_t *w_cols, double *w_vals, const int32_t *huge_rows,
                               const int64_t *scratch_off, int32_t first, int32_t count, double *scratch);

// renumbering (keyword `renumber`): host vectors arrive in the caller's cell order
//   scatter: out[new_id[i]] = in[i]   (b, x on upload)      gather: out[i] = in[new_id[i]]   (x on copy-back)
void launch_permute_scatter(hipStream_t st, int32_t n, const int32_t *new_id, const double *in, double *out);
void launch_permute_gather(hipStream_t st, int32_t n, const int32_t *new_id, const double *in, double *out);

// b *= scaling (lduLduBase.H:244-252)
void launch_scale(hipStream_t st, int32_t n, double *v, double factor);
// v[i] = s->xbar
void launch_fill_xbar(hipStream_t st, int32_t n, double *v, const DevScalars *s);

// --- partial sums, one per chunk ---
void launch_partials_sum(hipStream_t st, int32_t n, const double *a, double *part);
void launch_partials_dot(hipStream_t st, int32_t n, const double *a, const double *b, double *part,
                         const DevScalars *gate);
// same for the listed chunks only (multi-rank: the chunks that hold boundary rows)
void launch_partials_dot_chunks(hipStream_t st, int32_t n, const double *a, const double *b,
                                double *part, const DevScalars *gate, const int32_t *chunk_list,
                                int32_t count);
void launch_partials_norm1(hipStream_t st, int32_t n, const double *a, double *part);
// part[c] = sum over chunk of |(b - w) - r| + |b - w|   (StoppingCriterion.C:53-61)
void launch_partials_normfactor(hipStream_t st, int32_t n, const double *b, const double *w,
                                const double *r, double *part);

// --- CG steps ([UPSTREAM] cg::initialize / step_1 / step_2) ---
// partials of rho = sum r_i z_i (z = r * inv_diag, or r when inv_diag == nullptr) and sum |r_i|
void launch_cg_rho_norm(hipStream_t st, int32_t n, const double *r, const double *inv_diag,
                        double *part_rho, double *part_norm, const DevScalars *gate);
// p = z + (rho / prev_rho) p
void launch_cg_step1(hipStream_t st, int32_t n, double *p, const double *r, const double *inv_diag,
                     const DevScalars *s);
// x += (rho/beta) p ; r -= (rho/beta) q ; then the partials of the next rho and sum |r|
void launch_cg_step2(hipStream_t st, int32_t n, double *x, double *r, const double *p,
                     const double *q, const double *inv_diag, double *part_rho, double *part_norm,
                     const DevScalars *s);

// The same steps with the x update deferred by one turn (p is then read once per turn):
//   step_2r: r -= (rho/beta) q + partials;  step_1x(turn): x += (prev_rho/beta) p_old, then step_1.
// step_1x applies the pending update of turn-1 also when the solve has just stopped.
struct HaloPutFused;  // (below, after PeerHalo)
void launch_cg_step1x(hipStream_t st, int32_t n, double *p, double *x, const double *r,
                      const double *inv_diag, const DevScalars *s, const HaloPutFused *put = nullptr);
// put (chunk_sptr != nullptr): z of the send rows goes to the neighbours (multi-rank merged turn)
void launch_cg_step2r(hipStream_t st, int32_t n, double *r, const double *q, const double *inv_diag,
                      double *part_rho, double *part_norm, const DevScalars *s, double *z_out = nullptr,
                      const HaloPutFused *put = nullptr);

// Small systems: the same pair with the finalisers folded in (kernels_krylov.hip).  Every workgroup reduces the
// per-chunk partials itself in the finaliser's order; the scalars are read from `sin` and written to `sout`.
//   step_1x_fin: check on (part_rho, part_norm) of the previous step_2r (first: of the initial residual), the
//                pending x update (not when `first`), then step_1
//   step_2r_fin: beta from part_beta, then step_2r
// lead.box != nullptr (any number of chunks): workgroup 0 alone reduces and publishes, the others poll (LeadBox)
void launch_cg_step1x_fin(hipStream_t st, int32_t n, double *p, double *x, const double *r, const double *inv_diag,
                          const DevScalars *sin, DevScalars *sout, const double *part_rho,
                          const double *part_norm, double *history, int first, const LeadBox &lead = LeadBox{},
                          double *p_out = nullptr, const PRing &ring = PRing{});  // (ring.k > 0: x touched every k-th turn, see the kernel)
void launch_cg_step2r_fin(hipStream_t st, int32_t n, double *r, const double *q, const double *inv_diag,
                          double *part_rho, double *part_norm, const DevScalars *sin, DevScalars *sout,
                          const double *part_beta, double *z_out = nullptr,  // z_out: z = r / d kept for k_cg_turn_sym
                          const LeadBox &lead = LeadBox{});
constexpr int FUSED_FIN_MAX_CHUNKS = 1024;  // up to 524,288 rows: one partial per virtual finaliser thread
// step_1x_fin and the SpMV on half storage in one launch (p_new = z + (rho/rho') p recomputed at the gathered
// columns; it goes to p_out != p_in for the own rows): a turn is this + step_2r_fin.  z: what step_2r_fin's z_out
// (or, before the first turn, launch_mul) has left; r itself without a preconditioner
void launch_cg_turn_sym(hipStream_t st, const DevSym &A, const double *p_in, double *p_out, double *x, const double *z,
                        double *q, double *part_beta, const DevScalars *sin, DevScalars *sout,
                        const double *part_rho, const double *part_norm, double *history, int first,
                        const LeadBox &lead = LeadBox{});
// ... and between the single-workgroup finalisers of larger systems (scalars as k_cg_step1x reads them):
// turn = this | FIN_BETA | step_2r (z_out) | FIN_CG_CHECK
// Several ranks (hf.chunk_bptr != nullptr; peer-put transport): the neighbours have put the z of the halo columns
// (step_2r's `put`, launch_pack_put_signal before the first turn); p_halo_in = this rank's copy of the OLD p at its
// halo columns (zero before the first turn), p_halo_out receives the new one:
// turn = this (waits for z, nothing to put) | FIN_BETA | step_2r (z_out, puts z) | FIN_CG_CHECK
void launch_cg_turn_sym_big(hipStream_t st, const DevSym &A, const double *p_in, double *p_out, double *x,
                            const double *z, double *q, double *part_beta, const DevScalars *s,
                            const HaloFused &hf = HaloFused{}, const double *p_halo_in = nullptr,
                            double *p_halo_out = nullptr);

// --- BiCGStab steps ([UPSTREAM] bicgstab::step_1/2/3, finalize) ---
// a Gram-Schmidt link with the finaliser of the PREVIOUS link folded in (<= FUSED_FIN_MAX_CHUNKS chunks, one rank):
// vprev != nullptr: H = sum(part_in) -> *h_out, w -= H vprev; then the partials of w . vdot (w . w when vdot is nullptr)
void launch_gmres_mgs_fold(hipStream_t st, int32_t n, double *w, const double *vprev, double *h_out, const double *vdot,
                           const double *part_in, double *part_out, const DevScalars *gate, const LeadBox &lead = LeadBox{},
                           uint32_t tag = 0);  // (lead.box: leader finalisation, tag = a number no earlier launch of the solve used)
// GKOBiCGStab with the finalisers folded into the step kernels (<= FUSED_FIN_MAX_CHUNKS chunks, one rank; kernels_krylov.hip):
// scalars go sin -> sout; a kernel never writes a partial array it reads
void launch_bicg_fold1(hipStream_t st, int32_t n, double *p, const double *r, const double *v, const double *inv_diag,
                       double *y, const DevScalars *sin, DevScalars *sout, const double *part_rho,
                       const double *part_norm, double *history, const LeadBox &lead = LeadBox{});
void launch_bicg_fold2(hipStream_t st, int32_t n, const double *r, const double *v, double *sv, const double *inv_diag,
                       double *z, double *part_norm_out, const DevScalars *sin, DevScalars *sout,
                       const double *part_beta, const LeadBox &lead = LeadBox{});
void launch_bicg_fold3(hipStream_t st, int32_t n, double *x, double *r, const double *sv, const double *t,
                       const double *y, const double *z, const double *rr, double *part_rho_out, double *part_norm_out,
                       const DevScalars *sin, DevScalars *sout, const double *part_gamma, const double *part_tt,
                       const double *part_snorm, double *history, int turn, const LeadBox &lead = LeadBox{});
void launch_bicg_step1(hipStream_t st, int32_t n, double *p, const double *r, const double *v,
                       const double *inv_diag, double *y, const DevScalars *s);
void launch_bicg_step2(hipStream_t st, int32_t n, const double *r, const double *v, double *sv,
                       const double *inv_diag, double *z, double *part_norm, const DevScalars *s);
void launch_bicg_step3(hipStream_t st, int32_t n, double *x, double *r, const double *sv,
                       const double *t, const double *y, const double *z, const double *rr,
                       double *part_rho, double *part_norm, const DevScalars *s, int turn);
// (step_3 of the turn whose mid-step check stopped the solve applies bicgstab::finalize,
//  x += alpha y, instead of its own work)

// --- GMRES vector kernels ([UPSTREAM] gmres::restart / finish_arnoldi / solve_krylov) ---
// out = in / *denom
// out = in / *denom and w = out * inv_diag in one pass (scalar Jacobi: the scaling of a new basis vector waits for the
// turn that applies the preconditioner to it)
void launch_gmres_scale_mul(hipStream_t st, int32_t n, double *out, const double *in, const double *denom,
                            const double *inv_diag, double *w, const DevScalars *gate);
void launch_gmres_scale(hipStream_t st, int32_t n, double *out, const double *in,
                        const double *denom, const DevScalars *gate);
// modified Gram-Schmidt link: if (vprev) w -= (*hprev) * vprev ; partial of w . vdot (vdot == nullptr: w . w)
void launch_gmres_mgs(hipStream_t st, int32_t n, double *w, const double *vprev,
                      const double *hprev, const double *vdot, double *part,
                      const DevScalars *gate);
// t_i = sum_{j < it} V_j[i] * y[j] ; with `before` != nullptr the sum is stored there, otherwise
// x_i += t_i * inv_diag_i (inv_diag == nullptr: x_i += t_i)
void launch_gmres_update_x(hipStream_t st, int32_t n, const double *V, int64_t ld, const double *y,
                           int32_t it, const double *inv_diag, double *x, double *before,
                           const DevScalars *gate);
// out = in * inv_diag (scalar Jacobi apply), x += a
void launch_mul(hipStream_t st, int32_t n, double *out, const double *in, const double *inv_diag,
                const DevScalars *gate);
void launch_add(hipStream_t st, int32_t n, double *x, const double *a, const DevScalars *gate);

// --- single-workgroup finalisers: reduce per-chunk partials, then scalar logic -------------
// phases: what the scalar logic does with the reduced sums.
enum FinPhase {
    FIN_MEAN = 0,        // xbar = sum/n * n/global_n (all-reduced)        StoppingCriterion.C:17-19
    FIN_NORMFACTOR = 1,  // norm_factor = sum + SMALL                      StoppingCriterion.C:62-68
    FIN_CG_CHECK = 2,    // rho <- sum0, criterion check on sum1           StoppingCriterion.C:71-151
    FIN_BETA = 3,        // beta <- sum0
    FIN_RAW = 4,         // sums[] only (test / reduce entry point)
    FIN_BICG_ALPHA = 5,  // beta <- sum0 ; alpha = rho / beta
    FIN_BICG_CHECK2 = 6, // mid-turn criterion check on sum0 = sum|s|
    FIN_BICG_OMEGA = 7,  // gamma <- sum0 ; beta <- sum1 ; omega = gamma / beta
    // (the criterion check of a GKOGMRES turn -- on stale_norm, the sum|r| of the last restart -- has no launch of its own:
    //  FinArgs::check_after runs it at the end of the finaliser before it, restart or column)
    FIN_GMRES_RESTART = 8,  // rn = sqrt(sum0) -> rnc[0], beta ; stale_norm = sum1   (gmres::restart)
    FIN_GMRES_H = 9,        // H(k, it) = sum0                                        (finish_arnoldi)
    FIN_GMRES_COL = 10,     // H(it+1, it) = sqrt(sum0) -> beta ; Givens ; rnc        (givens_rotation)
    FIN_GMRES_SOLVE = 12,   // y = R^-1 rnc for `turn` columns                        (solve_krylov)
    // single rank: FIN_BICG_CHECK2 and FIN_BICG_OMEGA in one launch, after the second SpMV (which then runs even when
    // the mid-turn check stops the solve: its result is not used); sums: s.t, t.t and, from part_extra, sum|s|
    FIN_BICG_CHECK2_OMEGA = 13
};
// GMRES small dense state in one device array of doubles:
//   H[(m+1) x m] column-major | givens_sin[m] | givens_cos[m] | rnc[m+1] | y[m]
inline size_t gmres_state_len(int m) { return (size_t)(m + 1) * m + 2 * (size_t)m + (m + 1) + m; }
// Peer-write all-reduce over xGMI (SURVEY.md §8e "peer-write mesh all-reduce"): every rank owns a
// small mailbox in fine-grained device memory that all ranks have mapped (hipIpc).  All-reduce
// number `seq`: a rank stores its values into column `rank` of slot seq % PEER_SLOTS of EVERY
// mailbox, as 64-bit words (32 data bits | seq << 32) so that data and flag arrive in one atomic
// store, then reads its own mailbox until every column carries `seq`, and adds the columns in rank
// order -- every rank gets the same bits.  It runs INSIDE the finaliser kernel: no extra launch,
// no host, no collective library on the latency path of the two reductions per CG turn.
constexpr int PEER_MAX_RANKS = 16;
// a rank that does not show up within this time fails the solve with OGL_ERR_COMM instead of
// hanging the GPU (OGL_PEER_TIMEOUT_S overrides the 60 s)
constexpr long long PEER_DEFAULT_TIMEOUT_TICKS = 60LL * 100000000LL;
constexpr int PEER_SLOTS = 4;
constexpr int PEER_ELEMS = 4;  // two doubles as four half-words
constexpr size_t PEER_BOX_WORDS = (size_t)PEER_SLOTS * PEER_MAX_RANKS * PEER_ELEMS;
struct PeerArgs {
    int32_t world = 0;  // 0 / 1: no exchange
    int32_t rank = 0;
    uint32_t seq = 0;   // never 0 (the mailboxes start zeroed)
    long long timeout_ticks = PEER_DEFAULT_TIMEOUT_TICKS;  // of the 100 MHz wall clock
    unsigned long long *box[PEER_MAX_RANKS] = {};  // mailbox of rank q as mapped in this process
};
// vals[0..n) (n <= 2, device memory) summed over the ranks in place; *error set on timeout
void launch_peer_allreduce(hipStream_t st, const PeerArgs &pa, double *vals, int n, int32_t *error);

// Peer-put halo exchange (SURVEY.md §8e "pack x[send_idxs] then peer-to-peer put to each neighbour
// over xGMI").  A rank's IPC allocation is [mailbox | control slots | arena]; every solver with
// processor interfaces owns an arena block [flags 2 x n_neigh | recv parity 0 | recv parity 1].
// SpMV number `seq` of a solver: k_pack_put stores x[send_idxs] straight into the neighbours' recv
// segments of parity seq & 1, k_halo_signal then stores `seq` into their flag for this rank; the
// local SpMV runs; k_halo_wait spins until every neighbour's flag carries `seq`; the non-local
// kernel reads the recv block.  Two parities: a neighbour can be at most one SpMV ahead.
constexpr size_t PEER_CTRL_WORDS = (size_t)PEER_MAX_RANKS * 4;  // per source rank: epoch + 3 words
constexpr size_t PEER_ARENA_OFF = PEER_BOX_WORDS + PEER_CTRL_WORDS;  // 8-byte words from the base
constexpr int PEER_MAX_NEIGH = 16;
struct PeerHalo {
    int32_t n_neigh = 0;
    uint32_t seq = 0;
    long long timeout_ticks = PEER_DEFAULT_TIMEOUT_TICKS;
    int32_t send_off[PEER_MAX_NEIGH + 1] = {};             // send buffer blocks, by neighbour
    double *remote_recv[PEER_MAX_NEIGH] = {};              // neighbour i's segment for this rank
    unsigned long long *remote_flag[PEER_MAX_NEIGH] = {};  // neighbour i's flag for this rank
    const unsigned long long *local_flag = nullptr;        // this rank's flags, one per neighbour
};
// launch_cg_step1x's `put` (chunk_sptr != nullptr): the workgroups whose chunks hold send rows store the p they
// have just formed straight into the neighbours' receive blocks, and the last of them signals -- the pack kernel
// of the SpMV that follows is folded into the kernel that produces its input.
struct HaloPutFused {
    PeerHalo P;
    const int32_t *chunk_sptr = nullptr;  // [n_chunks + 1] ranges of send_pos per chunk
    const int32_t *send_pos = nullptr;    // positions in the send list, grouped by the chunk of their row
    const int32_t *send_idxs = nullptr;   // the send list (rows), as DevHalo
    unsigned *ticket = nullptr;
    int32_t n_put_chunks = 0;             // chunks that hold send rows
};
void launch_pack_put(hipStream_t st, const DevHalo &H, const PeerHalo &P, const double *x,
                     const DevScalars *gate);
void launch_halo_signal(hipStream_t st, const PeerHalo &P, const DevScalars *gate);
// the same three steps in two launches: [pack + put + signal by the last workgroup] and, after the
// local SpMV, [wait + y += A_non_local recv + dot partials of the chunks that hold boundary rows]
// (one workgroup per such chunk; chunk_row_ptr = ranges of H.boundary_rows per listed chunk)
void launch_pack_put_signal(hipStream_t st, const DevHalo &H, const PeerHalo &P, const double *x,
                            const DevScalars *gate, unsigned *ticket);
void launch_halo_finish(hipStream_t st, const DevHalo &H, int mode, int32_t n_rows,
                        const int32_t *chunk_list, const int32_t *chunk_row_ptr, int32_t n_chunks_b,
                        const double *recv, double *y, const SpmvDots &dots, const PeerHalo &P,
                        const DevScalars *gate, DevScalars *s);
// `s` receives comm_error / stop when a neighbour does not show up within the timeout
void launch_halo_wait(hipStream_t st, const PeerHalo &P, const DevScalars *gate, DevScalars *s);
// control message to another rank: dst[1..3] = w1..w3, then dst[0] = w0 (the epoch)
void launch_peer_post(hipStream_t st, unsigned long long *dst, unsigned long long w0,
                      unsigned long long w1, unsigned long long w2, unsigned long long w3);

struct FinArgs {
    PeerArgs peer{};  // world > 1: all-reduce the sums inside the kernel (do_reduce && do_logic)
    const double *part[2] = {nullptr, nullptr};
    const double *part_extra = nullptr;  // third partial array (FIN_BICG_CHECK2_OMEGA)
    int32_t n_part = 0;     // entries per partial array
    int32_t n_sums = 1;     // 1 or 2 (3 with part_extra)
    int32_t do_reduce = 1;  // reduce partials -> s->sums
    int32_t do_logic = 1;   // scalar logic from s->sums (after the all-reduce when multi-rank)
    double n_local = 0, n_global = 0;  // FIN_MEAN
    double *history = nullptr;
    int32_t turn = 0;  // BiCGStab turn index (FIN_BICG_CHECK2); GMRES: column `it`
    double *gm = nullptr;  // GMRES dense state
    int32_t m = 0;         // GMRES krylov_dim
    int32_t k = 0;         // GMRES row of H (FIN_GMRES_H)
    int32_t check_after = 0;  // FIN_GMRES_RESTART / FIN_GMRES_COL: the criterion check of the turn that follows, in the same launch
};
void launch_finalize(hipStream_t st, int phase, DevScalars *s, const FinArgs &a);

// cg::initialize scalars: rho = 0? (unused), prev_rho = 1, iter = 0, stop = 0, norm_factor = 1
void launch_reset_scalars(hipStream_t st, DevScalars *s, const DevCriterion &crit);

}  // namespace ogl
