/*
 * ogl_oracle.h -- CPU restatement ("oracle") of the hpsim/OGL hot path.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing under oracle/ is part of the product.
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may
 * load this library, and only as the checker / the reported CPU baseline.
 * The product (ogl_amd/) never links, imports or falls back to it.
 *
 * PARITY STATUS
 *   - LDU -> row-major COO/CSR conversion, coefficient update functions:
 *       PINNED by the reference's gtest known-answer vectors
 *       (unitTests/test_HostMatrix.C:8-107) -> tests/golden/host_matrix_kat.json.
 *   - Krylov arithmetic (SpMV order, dots, CG / BiCGStab step order, Jacobi):
 *       PARITY UNPINNED.  It lives in Ginkgo (git fc86d48b78ce..., fetched by
 *       the reference's CMakeLists.txt:51-53, not vendored, not installed).
 *       Restated here from Ginkgo's published reference-executor algorithm and
 *       anchored on OGL's call sites (lduLduBase/lduLduBase.H:272-276,
 *       Solver/CG/GKOCG.H:45-61, StoppingCriterion/StoppingCriterion.C:11-151).
 *
 * All file:line citations are relative to the reference tree (hpsim/OGL @ 2024-10-16).
 */
#ifndef OGL_ORACLE_H
#define OGL_ORACLE_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef int32_t orc_label;  /* WM_LABEL_SIZE=32 (unitTests/CMakeLists.txt:26-27) */
typedef double orc_scalar;  /* WM_DP */

/* OpenFOAM SMALL for double precision (added at StoppingCriterion.C:68). */
#define ORC_SMALL 1.0e-15

/* ------------------------------------------------------------------ */
/* HostMatrix/HostMatrixFreeFunctions.C                                */
/* ------------------------------------------------------------------ */

/* HostMatrixFreeFunctions.C:105-201 */
void orc_init_local_sparsity(orc_label nrows, orc_label upper_nnz, int is_symmetric,
                             const orc_label *upper, const orc_label *lower,
                             orc_label *rows, orc_label *cols, orc_label *permute);

/* HostMatrixFreeFunctions.C:21-30 (operator-precedence quirk kept: scale is ignored) */
void orc_symmetric_update(orc_label total_nnz, orc_label upper_nnz, const orc_label *permute,
                          orc_scalar scale, const orc_scalar *diag, const orc_scalar *upper,
                          orc_scalar *out);

/* HostMatrixFreeFunctions.C:32-56 */
void orc_symmetric_update_w_interface(orc_label total_nnz, orc_label diag_nnz,
                                      orc_label upper_nnz, const orc_label *permute,
                                      orc_scalar scale, const orc_scalar *diag,
                                      const orc_scalar *upper, const orc_scalar *iface,
                                      orc_scalar *out);

/* HostMatrixFreeFunctions.C:58-82 */
void orc_non_symmetric_update_w_interface(orc_label total_nnz, orc_label diag_nnz,
                                          orc_label upper_nnz, const orc_label *permute,
                                          orc_scalar scale, const orc_scalar *diag,
                                          const orc_scalar *upper, const orc_scalar *lower,
                                          const orc_scalar *iface, orc_scalar *out);

/* HostMatrixFreeFunctions.C:85-102 */
void orc_non_symmetric_update(orc_label total_nnz, orc_label upper_nnz,
                              const orc_label *permute, orc_scalar scale,
                              const orc_scalar *diag, const orc_scalar *upper,
                              const orc_scalar *lower, orc_scalar *out);

/* ------------------------------------------------------------------ */
/* HostMatrix/HostMatrix.C -- interfaces                                */
/* ------------------------------------------------------------------ */

enum { ORC_IFACE_PROCESSOR = 0, ORC_IFACE_CYCLIC = 1 };

/* Plain view of one lduInterfaceField (what HostMatrix.C reads from it). */
typedef struct {
    int kind;                      /* isA<processorLduInterface> / cyclicFvPatch */
    orc_label neighb_proc;         /* processorFvPatch::neighbProcNo()   (:266) */
    orc_label neighb_patch;        /* cyclic: neighbPatchID() -> index into this array (:319-324) */
    orc_label size;                /* interface().faceCells().size() */
    const orc_label *face_cells;   /* interface().faceCells() */
    const orc_scalar *bou_coeffs;  /* interfaceBouCoeffs[i] */
} orc_iface;

/* HostMatrix.C:159-178 */
orc_label orc_count_interface_nnz(const orc_iface *ifaces, orc_label n_ifaces, int proc_interfaces);

/* HostMatrix.C:180-207: concatenate bouCoeffs of (non-)processor interfaces, times -1 */
void orc_collect_interface_coeffs(const orc_iface *ifaces, orc_label n_ifaces, int local,
                                  orc_scalar *out);

/* HostMatrix.C:251-306.  Outputs ordered by ascending neighbour rank (std::map).
 * target_ids/target_sizes need room for n_ifaces entries, send_idxs for the
 * processor-interface nnz.  Returns the number of neighbour ranks. */
orc_label orc_create_communication_pattern(const orc_iface *ifaces, orc_label n_ifaces,
                                           orc_label *target_ids, orc_label *target_sizes,
                                           orc_label *send_idxs);

/* HostMatrix.C:412-466.  rows = faceCell, cols = permute = running interface index,
 * sorted by row.  (The reference uses the unstable std::sort keyed on row only, so the
 * order of equal rows is unspecified there; the oracle uses a stable sort.) */
void orc_init_non_local_sparsity(const orc_iface *ifaces, orc_label n_ifaces,
                                 orc_label *rows, orc_label *cols, orc_label *permute);

/* HostMatrix.C:468-589: init_local_sparsity + in-order merge of cyclic interfaces.
 * rows/cols/permute have nrows + 2*upper_nnz + local_interface_nnz entries. */
void orc_init_local_sparsity_pattern(orc_label nrows, orc_label upper_nnz, int is_symmetric,
                                     const orc_label *upper, const orc_label *lower,
                                     const orc_iface *ifaces, orc_label n_ifaces,
                                     orc_label *rows, orc_label *cols, orc_label *permute);

/* HostMatrix.C:634-704 (default device branch): concat [upper|lower(asym)|diag|local-iface]
 * then row_gather(ldu_mapping).  No scaling is applied on this path. */
void orc_update_local_matrix_data(orc_label nrows, orc_label upper_nnz, int is_symmetric,
                                  const orc_scalar *diag, const orc_scalar *upper,
                                  const orc_scalar *lower, const orc_iface *ifaces,
                                  orc_label n_ifaces, const orc_label *permute,
                                  orc_label total_nnz, orc_scalar *out);

/* HostMatrix.C:608-633 (reorderOnHost true): the free update functions, scaling applied
 * (except for the symmetric_update quirk). */
void orc_update_local_matrix_data_host(orc_label nrows, orc_label upper_nnz, int is_symmetric,
                                       orc_scalar scaling, const orc_scalar *diag,
                                       const orc_scalar *upper, const orc_scalar *lower,
                                       const orc_iface *ifaces, orc_label n_ifaces,
                                       const orc_label *permute, orc_label total_nnz,
                                       orc_scalar *out);

/* HostMatrix.C:708-732 */
void orc_update_non_local_matrix_data(const orc_iface *ifaces, orc_label n_ifaces,
                                      const orc_label *permute, orc_label nnz, orc_scalar *out);

/* ------------------------------------------------------------------ */
/* Arithmetic ([UPSTREAM] Ginkgo reference-executor semantics)          */
/* ------------------------------------------------------------------ */

/* COO rows (sorted) -> CSR row pointers (what Csr::read does with device_matrix_data,
 * CsrMatrixWrapper.H:181-188). */
void orc_rowptr_from_rows(orc_label nrows, orc_label nnz, const orc_label *rows, orc_label *rowptr);

/* y = A x : per row, sum starts at 0, accumulates val*x in stored order. */
void orc_spmv(orc_label n, const orc_label *rowptr, const orc_label *cols, const orc_scalar *vals,
              const orc_scalar *x, orc_scalar *y);
/* y = alpha A x + beta y : sum starts at beta*y, accumulates (alpha*val)*x in stored order. */
void orc_spmv_adv(orc_label n, const orc_label *rowptr, const orc_label *cols,
                  const orc_scalar *vals, orc_scalar alpha, const orc_scalar *x, orc_scalar beta,
                  orc_scalar *y);

/* Reduction order.  SEQUENTIAL = reference executor (left to right).  BLOCKED = the fixed
 * tree the HIP kernels use (chunks of `chunk_rows` rows, 256 threads, 64-lane xor tree), so
 * that a HIP result can be checked bit-for-bit; it is NOT a reference semantic.  EXACT = every dot / norm1 / sum and
 * every SpMV row sum accumulated with error-free transformations (TwoSum, TwoProduct) and rounded once: the arbiter
 * that says which of two summation orders is closer to the exact result (tests/test_gpu_exact_arbiter.py); neither a
 * reference semantic nor the device's. */
enum { ORC_REDUCE_SEQUENTIAL = 0, ORC_REDUCE_BLOCKED = 1, ORC_REDUCE_EXACT = 2 };
void orc_set_reduction(int mode, orc_label chunk_rows);

orc_scalar orc_dot(orc_label n, const orc_scalar *a, const orc_scalar *b);
orc_scalar orc_norm1(orc_label n, const orc_scalar *a);
orc_scalar orc_sum(orc_label n, const orc_scalar *a);

/* ------------------------------------------------------------------ */
/* Distributed matrix = local CSR + non-local CSR + halo exchange       */
/* (CsrMatrixWrapper.H:163-210, Partition.H:57-70)                      */
/* ------------------------------------------------------------------ */

typedef void (*orc_exchange_fn)(void *user, const orc_scalar *send, orc_scalar *recv);
typedef void (*orc_allreduce_fn)(void *user, orc_scalar *v, orc_label n);

typedef struct {
    orc_label n;                   /* local rows */
    const orc_label *rowptr;       /* local CSR */
    const orc_label *cols;
    const orc_scalar *vals;
    orc_label n_halo;              /* columns of the non-local matrix (0 = none) */
    const orc_label *nl_rowptr;    /* non-local CSR, n x n_halo */
    const orc_label *nl_cols;
    const orc_scalar *nl_vals;
    orc_label n_send;              /* total send entries */
    const orc_label *send_idxs;    /* concatenated in ascending neighbour rank */
    orc_exchange_fn exchange;      /* send[n_send] -> recv[n_halo]; NULL when n_halo == 0 */
    orc_allreduce_fn allreduce;    /* in-place SUM over ranks; NULL = single rank */
    void *user;
    int64_t global_n;              /* Partition.H:118-121 */
} orc_dist_matrix;

/* y = A_local x ; y += A_non_local halo(x)   (distributed::Matrix::apply) */
void orc_dist_spmv(const orc_dist_matrix *A, const orc_scalar *x, orc_scalar *y);

/* ------------------------------------------------------------------ */
/* Stopping criterion (StoppingCriterion/StoppingCriterion.{H,C})       */
/* ------------------------------------------------------------------ */

typedef struct {
    orc_scalar tolerance;  /* openfoam_absolute_tolerance */
    orc_scalar rel_tol;    /* openfoam_relative_tolerance */
    orc_label min_iter;
    orc_label max_iter;    /* already doubled for BiCGStab (StoppingCriterion.H:188) */
    orc_label frequency;
    int export_res;
} orc_criterion;

typedef struct {
    orc_scalar init_residual;  /* init_normalised_res_norm_ */
    orc_scalar residual;       /* normalised_res_norm_ */
    orc_scalar norm_factor;
    orc_label iter;            /* iter_ : number of check_impl calls */
    orc_label n_evals;         /* how many checks actually evaluated the norm */
    orc_scalar *history;       /* residual_norms (needs max_iter+1 entries), may be NULL */
} orc_criterion_state;

/* StoppingCriterion.H:191-209: adaptive minIter / frequency. */
void orc_adapt_criterion(orc_label min_iter, orc_label frequency, int export_res,
                         orc_label prev_solve_iters, int adapt_min_iter,
                         orc_scalar relaxation_factor, orc_label norm_eval_limit,
                         orc_scalar prev_rel_cost, orc_label *min_iter_out,
                         orc_label *frequency_out);

/* StoppingCriterion.C:11-69 */
orc_scalar orc_compute_normfactor(const orc_dist_matrix *A, const orc_scalar *r,
                                  const orc_scalar *x, const orc_scalar *b);

/* ------------------------------------------------------------------ */
/* Preconditioner (Preconditioner.H:91-105, Ginkgo Jacobi [UPSTREAM])   */
/* ------------------------------------------------------------------ */

/* max_block_size == 1: inv_diag[i] = 1 / A_local(i,i) */
void orc_jacobi_generate_scalar(orc_label n, const orc_label *rowptr, const orc_label *cols,
                                const orc_scalar *vals, orc_scalar *inv_diag);

/* max_block_size > 1: block pointers (returns the number of blocks; block_ptrs needs n+1
 * entries), then the inverted diagonal blocks, row-major, `stride` x `stride` doubles each. */
orc_label orc_jacobi_find_blocks(orc_label n, const orc_label *rowptr, const orc_label *cols,
                                 orc_label max_block_size, orc_label *block_ptrs);
void orc_jacobi_generate_blocks(orc_label n, const orc_label *rowptr, const orc_label *cols,
                                const orc_scalar *vals, orc_label n_blocks,
                                const orc_label *block_ptrs, orc_label stride, orc_scalar *blocks);

enum { ORC_PRECOND_NONE = 0, ORC_PRECOND_SCALAR = 1, ORC_PRECOND_BLOCK = 2,
       ORC_PRECOND_ISAI_SPD = 3, ORC_PRECOND_ISAI_GENERAL = 4 };
typedef struct {
    int kind;
    const orc_scalar *inv_diag;   /* SCALAR */
    orc_label n_blocks;           /* BLOCK  */
    const orc_label *block_ptrs;
    const orc_scalar *blocks;
    orc_label stride;
    /* ISAI (Preconditioner.H:225-258): approximate inverse W in CSR; SPD: z = W^T (W r) */
    const orc_label *w_rowptr, *w_cols;
    const orc_scalar *w_vals;
    const orc_label *wt_rowptr, *wt_cols;
    const orc_scalar *wt_vals;
    /* BLOCK on a system whose rows were renumbered after the blocks were formed (the backend's `renumber`): the
     * blocks are those of the caller's numbering (Preconditioner.H:91-105 generates on the matrix OpenFOAM hands
     * over); block_rows[r0 + i] = row of the block's i-th member in the system at hand.  NULL: r0 + i itself. */
    const orc_label *block_rows;
} orc_precond;

/* ISAI with sparsityPower 1 ([UPSTREAM] gko::preconditioner::Isai).  spd != 0: W has the pattern
 * of tril(A); row i solves A(J,J) y = e_i and stores y / sqrt(y_i) (FSAI), M^-1 = W^T W.
 * spd == 0 (GISAI): W has A's pattern; row i solves A(J,J)^T y = e_i, M^-1 = W.
 * Step 1 (vals == NULL): returns nnz(W) and fills w_rowptr[n+1]; step 2: fills w_cols / w_vals.
 * Rows of up to 64 pattern entries: one dense solve with the row-wise back substitution; up to 2048: the same
 * elimination with the column-wise back substitution (see solve_dense_wide); wider: returns -1. */
/* the same on the pattern of S^power (keyword sparsityPower, Preconditioner.H:227) */
orc_label orc_isai_generate_p(orc_label n, const orc_label *rowptr, const orc_label *cols,
                              const orc_scalar *vals, int spd, int power, orc_label *w_rowptr,
                              orc_label *w_cols, orc_scalar *w_vals);
/* ... with the triangle of the spd variant taken in ANOTHER numbering of the same rows: S(r, c) is kept when
 * key[c] <= key[r] (key = the caller's index of every row; NULL = the row index itself): P tril(A) P^T of a
 * system renumbered by the backend, so that W is the reference's operator in the new numbering */
orc_label orc_isai_generate_pk(orc_label n, const orc_label *rowptr, const orc_label *cols,
                               const orc_scalar *vals, int spd, int power, const orc_label *key,
                               orc_label *w_rowptr, orc_label *w_cols, orc_scalar *w_vals);
orc_label orc_isai_generate(orc_label n, const orc_label *rowptr, const orc_label *cols,
                            const orc_scalar *vals, int spd, orc_label *w_rowptr, orc_label *w_cols,
                            orc_scalar *w_vals);
/* CSR transpose (pattern + values), rows of the result sorted by column */
void orc_csr_transpose(orc_label n, const orc_label *rowptr, const orc_label *cols,
                       const orc_scalar *vals, orc_label *t_rowptr, orc_label *t_cols,
                       orc_scalar *t_vals);

/* ------------------------------------------------------------------ */
/* Solvers ([UPSTREAM] gko::solver::Cg / Bicgstab, lduLduBase.H:272-276) */
/* ------------------------------------------------------------------ */

/* inv_diag == NULL -> identity preconditioner.  Returns st->iter. */
orc_label orc_cg(const orc_dist_matrix *A, const orc_scalar *b, orc_scalar *x,
                 const orc_scalar *inv_diag, const orc_criterion *crit,
                 orc_criterion_state *st);

orc_label orc_bicgstab(const orc_dist_matrix *A, const orc_scalar *b, orc_scalar *x,
                       const orc_scalar *inv_diag, const orc_criterion *crit,
                       orc_criterion_state *st);

/* same with a general preconditioner object (NULL = identity) */
orc_label orc_cg_p(const orc_dist_matrix *A, const orc_scalar *b, orc_scalar *x,
                   const orc_precond *P, const orc_criterion *crit, orc_criterion_state *st);
orc_label orc_bicgstab_p(const orc_dist_matrix *A, const orc_scalar *b, orc_scalar *x,
                         const orc_precond *P, const orc_criterion *crit,
                         orc_criterion_state *st);

/* GKOGMRES: restarted GMRES(krylov_dim), krylov_dim <= 0 -> Ginkgo's default 100 */
orc_label orc_gmres_p(const orc_dist_matrix *A, const orc_scalar *b, orc_scalar *x,
                      const orc_precond *P, const orc_criterion *crit, orc_criterion_state *st,
                      orc_label krylov_dim);

/* "omp executor" baseline: row-parallel SpMV, parallel AXPYs and reductions, single rank.
 * Same step order as orc_cg; reductions are OpenMP tree sums (not bit-comparable).
 * Only built when compiled with -fopenmp; returns -1 otherwise. */
orc_label orc_cg_omp(const orc_dist_matrix *A, const orc_scalar *b, orc_scalar *x,
                     const orc_scalar *inv_diag, const orc_criterion *crit,
                     orc_criterion_state *st, int n_threads);
int orc_omp_max_threads(void);
/* the same solve with set-up (allocation + first-touch copy) and loop timed apart, in seconds */
/* seconds the last orc_cg_omp_timed spent in [0] the x/r update + reductions, [1] the p update, [2] SpMV + p.q */
void orc_cg_omp_phases(double out[3]);
orc_label orc_cg_omp_timed(const orc_dist_matrix *A, const orc_scalar *b, orc_scalar *x,
                           const orc_scalar *inv_diag, const orc_criterion *crit,
                           orc_criterion_state *st, int n_threads, double *t_setup_s,
                           double *t_loop_s);
/* STREAM-like triad over 3 x n doubles, first-touch placed; GB/s of the best of `reps` passes */
double orc_stream_triad_omp(long n, int reps, int n_threads);

#ifdef __cplusplus
}
#endif
#endif
