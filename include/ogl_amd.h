/*
 * ogl_amd.h -- C ABI of the MI355X-native sparse Krylov backend (libogl_amd.so).
 *
 * Drop-in boundary for the hot path of hpsim/OGL: the OpenFOAM lduMatrix::solver plug-ins
 * GKOCG / GKOBiCGStab / GKOGMRES.  The reference has no FFI of its own: it is C++ that calls
 * Ginkgo directly.  Every entry point below therefore names the reference C++ interface it
 * replaces (file:line relative to the reference tree); INTEGRATION.md shows the binding a
 * maintainer adds on the OpenFOAM side.
 *
 * Conventions
 *   - plain C types only: label = int32_t (WM_LABEL_SIZE=32), scalar = double (WM_DP);
 *   - host pointers are BORROWED for the duration of the call (OpenFOAM owns them);
 *   - every function returns OGL_OK (0) or a negative ogl_status; the message is in
 *     ogl_last_error() (thread local).  Nothing throws across this boundary;
 *   - a solver handle is NOT thread-safe; one handle per (rank, field), as in the reference
 *     (one MPI rank <-> one device, DevicePersistent/ExecutorHandler/ExecutorHandler.H:33,90-91);
 *   - there is no CPU fallback: without a gfx950 device every compute call fails with
 *     OGL_ERR_NO_DEVICE.  Only the ogl_host_* functions (pure host logic) run without a GPU.
 */
#ifndef OGL_AMD_H
#define OGL_AMD_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* 4: ogl_ldu_view gained `cell_centres` (a caller built against 3 hands over a shorter struct: the plug-in compares
 * ogl_abi_version() with this value in its constructor, OGLAdapter.H); ogl_memory_ledger, ogl_registry_mem_info. */
#define OGL_AMD_ABI_VERSION 4

typedef int32_t ogl_label;
typedef double ogl_scalar;

typedef enum {
    OGL_OK = 0,
    OGL_ERR_INVALID = -1,     /* bad argument / unsupported keyword value (FatalError in the reference) */
    OGL_ERR_NO_DEVICE = -2,   /* no usable gfx950 device */
    OGL_ERR_HIP = -3,         /* a HIP runtime call failed */
    OGL_ERR_COMM = -4,        /* RCCL / host transport failure */
    OGL_ERR_STATE = -5,       /* call order violated (e.g. solve before set_matrix) */
    OGL_ERR_UNSUPPORTED = -6, /* e.g. AMI/ACMI interfaces (HostMatrix.C:339-341,367-369) */
    OGL_ERR_COMM_SELFTEST = -7 /* a transport's collectives completed but delivered wrong numbers: the caller may step down */
} ogl_status;

const char *ogl_last_error(void);
int ogl_abi_version(void);

/* ------------------------------------------------------------------------------------ */
/* fvSolution keywords (defaults are the CODE's defaults)           */
/* ------------------------------------------------------------------------------------ */

typedef enum { OGL_SOLVER_CG = 0, OGL_SOLVER_BICGSTAB = 1, OGL_SOLVER_GMRES = 2 } ogl_solver_kind;
/* Solver/CG/GKOCG.H:120, Solver/BiCGStab/GKOBiCGStab.H, Solver/GMRES/GKOGMRES.H (TypeName) */

typedef enum {
    OGL_PRECOND_NONE = 0, /* Preconditioner.H:342 */
    OGL_PRECOND_BJ = 1,   /* Preconditioner.H:91-105 (Schwarz-wrapped Jacobi on the local matrix) */
    OGL_PRECOND_ISAI = 2, /* Preconditioner.H:225-241  isai_type::spd     (M^-1 = W^T W)           */
    OGL_PRECOND_GISAI = 3 /* Preconditioner.H:242-258  isai_type::general (M^-1 = W)               */
} ogl_precond_kind;

typedef enum { OGL_FORMAT_COO = 0, OGL_FORMAT_CSR = 1, OGL_FORMAT_ELL = 2 } ogl_matrix_format;
/* CsrMatrixWrapper.H:138-161.  Coo and Csr select the persistent device CSR (row-sorted triplets
 * and CSR accumulate a row in the same stored order in the reference executor); Ell adds a
 * slot-major padded copy and its own SpMV kernel.  All three give bit-identical products. */

typedef struct ogl_config {
    int32_t solver;             /* ogl_solver_kind                               "solver"            */
    int32_t preconditioner;     /* ogl_precond_kind                              "preconditioner"    */
    int32_t max_block_size;     /* 1     Preconditioner.H:94                     "maxBlockSize"      */
    int32_t caching;            /* 0     Preconditioner.H:405                    "caching"           */
    double tolerance;           /* 1e-6  StoppingCriterion.H:167                 "tolerance"         */
    double rel_tol;             /* 1e-6  StoppingCriterion.H:168                 "relTol"            */
    int32_t max_iter;           /* 1000  StoppingCriterion.H:165                 "maxIter"           */
    int32_t min_iter;           /* 0     StoppingCriterion.H:166                 "minIter"           */
    int32_t eval_frequency;     /* 1     StoppingCriterion.H:173                 "evalFrequency"     */
    int32_t norm_eval_limit;    /* 100   StoppingCriterion.H:171-172             "normEvalLimit"     */
    double relaxation_factor;   /* 0.6   StoppingCriterion.H:174-175             "relaxationFactor"  */
    int32_t adapt_min_iter;     /* 1     StoppingCriterion.H:176-177             "adaptMinIter"      */
    int32_t matrix_format;      /* COO   CsrMatrixWrapper.H:250                  "matrixFormat"      */
    int32_t regenerate;         /* 0     CsrMatrixWrapper.H:257                  "regenerate"        */
    int32_t update_sys_matrix;  /* 1     CsrMatrixWrapper.H:259                  "updateSysMatrix"   */
    int32_t update_rhs;         /* 1     lduLduBase.H:224                        "updateRHS"         */
    int32_t update_init_guess;  /* 0     lduLduBase.H:235                        "updateInitGuess"   */
    double scaling;             /* 1.0   HostMatrix.C:33, lduLduBase.H:242-252   "scaling"           */
    int32_t reorder_on_host;    /* 0     HostMatrix.C:32                         "reorderOnHost"     */
    int32_t export_res;         /* 0     CsrMatrixWrapper.H:247                  "export"            */
    int32_t verbose;            /* 0     lduLduBase.H:49                         "verbose"           */
    int32_t force_host_buffer;  /* 0     ExecutorHandler.H:136-139               "forceHostBuffer"   */
    int32_t ranks_per_gpu;      /* 1     ExecutorHandler.H:135 (only 1 works)    "ranksPerGPU"       */
    int32_t krylov_dim;         /* 0 = Ginkgo default (100); GMRES only; NOT a reference keyword   */
    int32_t sparsity_power;     /* 1     Preconditioner.H:227 (rows of W <= 2048)  "sparsityPower"     */
    int32_t profile_kernels;    /* 0; k > 0 = hipEvent-time the in-loop SpMV of every k-th turn
                                   (bench.py roofline leg)                                         */
    int32_t compress_indices;   /* 1; Coo/Csr formats: run the SpMV on the index-compressed chunked
                                   ELL copy of the matrix when the pattern qualifies (same bits in
                                   y; 8.1-10 instead of 12 bytes per entry).  0 off; 1 on -- for an
                                   irregular pattern (unstructured mesh) both kernels are timed once
                                   per pattern and the faster one runs; 2 on without that measurement.
                                   NOT a reference keyword: "compressIndices" (false | true | force) */
    int32_t renumber;           /* 2; 0 off, 1 on, 2 auto.  The device matrix, b and x live in a
                                   bandwidth-reducing (reverse Cuthill-McKee) numbering of the cells,
                                   computed once per sparsity pattern -- what `renumberMesh` does for
                                   a case, done by the backend for itself.  auto: only for >= 16384
                                   rows and when the numbering it is handed makes the SpMV gather x badly (the
                                   compressed layout does not qualify and > 0.25 sectors of x per
                                   entry) and RCM improves that.  psi / source keep the caller's
                                   order at the boundary.  Rows are still summed in stored (new)
                                   column order, so results differ from the un-renumbered run at
                                   rounding level, exactly as after renumberMesh.  NOT a reference
                                   keyword: "renumber"                                              */
    int32_t symmetric_half;     /* 1; a symmetric lduMatrix (no `lower`) on a banded pattern (a
                                   structured mesh: at most 3 distances from the diagonal) is kept
                                   the OpenFOAM way on the device too -- diagonal + upper coefficients
                                   only; the SpMV reads A(r, r-d) where it reads A(r-d, r).  Same bits
                                   in y, a third fewer bytes from DRAM.  A pattern that is banded only
                                   locally (multi-block mesh, refinement shell) gets the distances per
                                   chunk of 512 rows and explicit entries for what breaks the bands,
                                   and is timed once against the full-storage copy (the faster stays).
                                   Needs compress_indices; 0 = full storage.  NOT a reference keyword:
                                   "symmetricStorage"                                               */
} ogl_config;

/* Fill with the reference code's defaults. */
void ogl_config_default(ogl_config *cfg);

/* ------------------------------------------------------------------------------------ */
/* lduMatrix view (what HostMatrix.C reads from `matrix`, `interfaces`, `interfaceBouCoeffs`) */
/* ------------------------------------------------------------------------------------ */

typedef enum { OGL_IFACE_PROCESSOR = 0, OGL_IFACE_CYCLIC = 1 } ogl_iface_kind;

typedef struct ogl_interface {
    int32_t kind;                  /* isA<processorLduInterface> / cyclicFvPatch  HostMatrix.C:170,400 */
    ogl_label neighb_proc;         /* processorFvPatch::neighbProcNo()            HostMatrix.C:266     */
    ogl_label neighb_patch;        /* cyclic: index (in this array) of neighbPatchID()  HostMatrix.C:319-324 */
    ogl_label size;                /* interface().faceCells().size()                                    */
    const ogl_label *face_cells;   /* interface().faceCells()                                           */
    const ogl_scalar *bou_coeffs;  /* interfaceBouCoeffs[i]  (true off-diagonal = -bouCoeffs, :204)     */
} ogl_interface;

typedef struct ogl_ldu_view {
    ogl_label n_cells;             /* matrix.diag().size()                        HostMatrix.C:34   */
    ogl_label n_faces;             /* matrix.lduAddr().upperAddr().size()         HostMatrix.C:36   */
    const ogl_label *lower_addr;   /* lduAddr().lowerAddr()                       HostMatrix.C:477  */
    const ogl_label *upper_addr;   /* lduAddr().upperAddr()                       HostMatrix.C:481  */
    const ogl_scalar *diag;        /* matrix.diag()                               HostMatrix.C:600  */
    const ogl_scalar *upper;       /* matrix.upper()                              HostMatrix.C:598  */
    const ogl_scalar *lower;       /* matrix.lower(); NULL <=> matrix.symmetric() HostMatrix.C:473  */
    ogl_label n_interfaces;
    const ogl_interface *interfaces;
    /* Optional (NULL = not given; this build's addition, the reference reads no geometry): the cell centres,
     * mesh.C() -- x, y, z per cell, 3 n_cells doubles.  With `renumber auto | on` the library then has a second
     * candidate for the numbering of its device copy next to reverse Cuthill-McKee: the cells along a Hilbert curve
     * through their centres, which on polyhedral meshes gathers x from fewer cache lines (host_matrix.hpp
     * NumberingHooks).  Read during the ogl_solver_set_matrix call that builds the pattern only. */
    const ogl_scalar *cell_centres;
} ogl_ldu_view;

/* solverPerformance fields the reference fills (lduLduBase.H:283-285) + its statistics block
 * (lduLduBase.H:280-305). */
typedef struct ogl_perf {
    double initial_residual;
    double final_residual;
    int32_t n_iterations;   /* CG/GMRES: number of criterion checks (= steps + 1, StoppingCriterion.C:143);
                               BiCGStab: checks / 2 (GKOBiCGStab.H:114) */
    int32_t n_norm_evals;   /* checks that evaluated the residual norm */
    double norm_factor;
    double t_update_matrix_ms;  /* H2D + gather of the last ogl_solver_set_matrix */
    double t_upload_ms;         /* b (and x) H2D */
    double t_solve_ms;          /* delta_t_solve: solver->apply only (lduLduBase.H:275-276) */
    double t_copy_back_ms;      /* x D2H (lduLduBase.H:278-279) */
    double spmv_avg_ms;         /* mean in-loop SpMV kernel time (profile_kernels=1), else 0 */
    int32_t spmv_launches;
    int32_t reserved0;
    double t_res_norm_us;       /* time_for_res_norm_eval: one evaluated criterion check, measured (lduLduBase.H:287) */
    double n_global_rows;       /* partition.get_total_size() (lduLduBase.H:294-295): rows over all ranks */
} ogl_perf;

/* ------------------------------------------------------------------------------------ */
/* Registry = objectRegistry analogue (DevicePersistent/Base/Base.H:53-137)             */
/* ------------------------------------------------------------------------------------ */

typedef struct ogl_registry ogl_registry;
typedef struct ogl_solver ogl_solver;

/* Replaces ExecutorHandler (ExecutorHandler.H:83-93: device id) + DeviceIdGuard
 * (DeviceIdGuard.H:15-43).  The device used is device_id % n_devices, as in the reference
 * (ExecutorHandler.H:90-91: the adapter passes rank / ranksPerGPU); device_id < 0 => device 0.  `hip_stream` may carry an existing hipStream_t (e.g. torch's current
 * stream) so the caller's events see the work; NULL => the registry creates its own stream. */
int ogl_registry_create(ogl_registry **out, int device_id, void *hip_stream);
void ogl_registry_destroy(ogl_registry *reg);

/* --- communicators (ExecutorHandler.H:29-32,140-144,167-172) ----------------------- */

/* Host-buffer transport (forceHostBuffer, ExecutorHandler.H:136-139): the host side owns the
 * message passing (MPI in OpenFOAM); device data is staged through pinned host buffers. */
typedef void (*ogl_allreduce_sum_fn)(void *user, double *values, int32_t n);
/* send/recv are blocked by neighbour in ascending rank order; counts[i] entries each way. */
typedef void (*ogl_neighbour_exchange_fn)(void *user, int32_t n_neighbours,
                                          const int32_t *neighbour_ranks, const int32_t *counts,
                                          const double *send, double *recv);
int ogl_registry_set_host_comm(ogl_registry *reg, int32_t rank, int32_t n_ranks,
                               ogl_allreduce_sum_fn allreduce, ogl_neighbour_exchange_fn exchange,
                               void *user);

/* Device transport: RCCL over xGMI (replaces the GPU-aware MPI communicator).  The 128-byte
 * unique id is produced on rank 0 and broadcast by the host (MPI_Bcast in OpenFOAM). */
#define OGL_RCCL_ID_BYTES 128
int ogl_rccl_unique_id(void *id_out);
/* Local, non-collective: can this rank enter ogl_registry_init_rccl at all (librccl loads, the device can be
 * selected)?  init_rccl is collective (ncclCommInitRank, then a self-test) and has no time-out: the host agrees
 * on this answer over its own message passing FIRST (a min-reduce in OpenFOAM), so that a rank which cannot
 * join makes every rank step down to the host-buffer transport instead of leaving the others inside the
 * collective (the reference aborts the job in that situation: a Ginkgo exception, ExecutorHandler.H:83-110). */
int ogl_registry_rccl_ready(ogl_registry *reg);
int ogl_registry_init_rccl(ogl_registry *reg, int32_t rank, int32_t n_ranks, const void *id);

/* Peer mesh over xGMI (hipIpc): the scalar reductions of the Krylov loop (the MPI_Allreduce behind
 * gko's distributed dot / norm1, StoppingCriterion.C:19,54-63,94) run inside the finaliser kernels
 * through peer-written mailboxes, and the halo values of every SpMV (sparse_communicator,
 * CsrMatrixWrapper.H:195-204) are put straight into the neighbours' receive blocks by the pack
 * kernel.  Optional, on top of either transport above (which remains the fallback and the
 * bootstrap): every rank exports a 64-byte IPC handle of its mailbox + arena, the host all-gathers
 * them (MPI_Allgather in OpenFOAM), every rank connects BEFORE the first set_matrix of any field.
 * With the mesh up, building a field's sparsity pattern is collective over the ranks.
 * peer_connect is COLLECTIVE: it runs a self-test all-reduce and fails on every rank alike if the
 * mesh does not work; call ogl_registry_peer_disable then and the transport's own all-reduce is
 * used.  At most 16 ranks, one node. */
#define OGL_PEER_HANDLE_BYTES 64
int ogl_registry_peer_handle(ogl_registry *reg, void *handle_out);
int ogl_registry_peer_connect(ogl_registry *reg, int32_t rank, int32_t n_ranks, const void *handles);
int ogl_registry_peer_disable(ogl_registry *reg);

/* What the registry's communicator really is (bench.py --selfcheck prints it): transport 0 = none
 * (single rank), 1 = host-buffer callbacks, 2 = RCCL; ranks_seen = what the transport itself
 * reports (RCCL: ncclCommCount); peer_mesh = 1 when the hipIpc peer mesh passed its self-test. */
typedef struct ogl_comm_info {
    int32_t transport, rank, n_ranks, ranks_seen, peer_mesh, device;
} ogl_comm_info;
int ogl_registry_comm_info(ogl_registry *reg, ogl_comm_info *info);

/* What the library holds, process-wide (all registries): every hipMalloc / hipHostMalloc it makes is booked, every
 * hipFree / hipHostFree unbooked (csrc/ledger.hpp).  The reference's device objects belong to the objectRegistry and
 * live as long as it does (DevicePersistent/Base/Base.H:53-137, HostMatrix.C:79-95: created once per field, updated in
 * place afterwards); a time-step loop must therefore leave device_bytes / pinned_bytes exactly where they were after
 * the first steps, and closing the last registry must bring all of them to 0.  Also readable per solver as the
 * properties deviceBytesInUse / pinnedBytesInUse (process-wide numbers, same source). */
typedef struct ogl_memory_ledger {
    int64_t device_bytes, device_blocks, device_peak_bytes, device_alloc_calls;
    int64_t pinned_bytes, pinned_blocks, pinned_peak_bytes, pinned_alloc_calls;
    int64_t streams, events, graph_execs;       /* live runtime objects the library created */
    int64_t events_created, graph_execs_created; /* ... and how many it has created so far */
    int64_t unknown_frees;                       /* frees of pointers the ledger never booked (always 0) */
} ogl_memory_ledger;
int ogl_memory_ledger_read(ogl_memory_ledger *out);
/* hipMemGetInfo of the registry's device (what the DRIVER sees in use: the library's blocks plus the runtime's own
 * pools and every other user of the device). */
int ogl_registry_mem_info(ogl_registry *reg, int64_t *free_bytes, int64_t *total_bytes);

/* ------------------------------------------------------------------------------------ */
/* The plug-in path                                                                      */
/* ------------------------------------------------------------------------------------ */

/* GKOCG/GKOBiCGStab/GKOGMRES constructor (Solver/CG/GKOCG.H:132-139): lookup-or-create by field
 * name, like PersistentBase (Base.H:75-115).  The reference constructs a fresh solver object per
 * solve and finds its device state in the registry; so does this. `cfg` is re-read on every call
 * (the dictionary may change between time steps). */
int ogl_solver_get_or_create(ogl_registry *reg, const char *field_name, const ogl_config *cfg,
                             ogl_solver **out);

/* HostMatrixWrapper constructor (HostMatrix.C:15-96): first call builds and uploads the sparsity
 * pattern + ldu_mapping (init_local_sparsity_pattern :468-589, init_non_local_sparsity_pattern
 * :438-466, create_communication_pattern :251-306); every call stages upper/lower/diag/interface
 * coefficients through pinned buffers and permutes them on the device (update_local_matrix_data
 * :592-705, update_non_local_matrix_data :708-732, MatrixInitFunctor::update
 * CsrMatrixWrapper.H:74-136). */
int ogl_solver_set_matrix(ogl_solver *s, const ogl_ldu_view *ldu);

/* The components of a vector field: fvMatrix::solveSegregated constructs the solvers of Ux, Uy, Uz on ONE lduMatrix --
 * the same upper() / lower() storage with the same values, only diag() and the interface coefficients differ per
 * component -- and the reference uploads all of it three times (HostMatrix.C:644-682 runs once per solver object,
 * lduLduBase.H:224-237).  Like ogl_solver_set_matrix, but the off-diagonal coefficients are taken from `donor`'s device
 * copy (a device-to-device copy) instead of crossing PCIe again, when ALL of this holds: same registry, same face
 * addressing (fingerprint of lowerAddr / upperAddr / interfaces), ldu->upper / ldu->lower are the very host arrays
 * `donor` uploaded from (pointer identity), a checksum over 8192 sampled entries of them still equals what `donor`
 * recorded, and neither solver reorders on the host.  Otherwise it IS ogl_solver_set_matrix.  (When, in addition, the
 * addressing arrays are the very ones `donor` was handed -- same pointers, same counts -- their hash is taken over from
 * `donor` instead of reading them again.)  The caller vouches that
 * nothing wrote to those arrays in between (the plug-in asks for this only for the sibling component solved
 * immediately before, in the same time step: OGLAdapter.H).  Property offDiagReused tells what happened. */
int ogl_solver_set_matrix_like(ogl_solver *s, const ogl_ldu_view *ldu, ogl_solver *donor);

/* lduMatrix::solver::solve(psi, source, cmpt) (GKOCG.H:149-153 -> lduLduBase.H:189-308):
 * upload source (updateRHS) and psi (first call / updateInitGuess), scale the RHS, (re)generate
 * the preconditioner, run the Krylov loop, copy x back into psi, fill `perf`. */
int ogl_solver_solve(ogl_solver *s, const ogl_scalar *source, ogl_scalar *psi, ogl_perf *perf);

/* Residual history recorded with `export 1` (StoppingCriterion.C:115-117; the reference stores it
 * but never writes it out).  Returns the number of entries copied (<= capacity) or a negative
 * status.  Entry i = normalised residual at criterion check i. */
int ogl_solver_history(ogl_solver *s, double *out, int32_t capacity);

/* `debug true` at a write time (lduLduBase.H:259-264): MatrixMarket dump of the persistent system
 * as the DEVICE holds it, into `directory` (the reference uses processor?/<time>/):
 *   <field>_A_local.mtx, <field>_A_non_local.mtx   coordinate real general, 1-based, row-major,
 *                                                  15 significant digits (common.C:31-58,
 *                                                  CsrMatrixWrapper.H:273-290)
 *   <field>_rhs_b_.mtx                              array real general (Vector.H:173-176)
 *   <field>_res_norms.mtx                           residual history of the last solve when `export`
 *                                                  was set (the reference records it, never writes it) */
int ogl_solver_export_system(ogl_solver *s, const char *directory);

/* Per-field solver properties kept between solves (common/common.C:75-146):
 * prevSolveIters(_final), _prev_solve (relative residual-evaluation cost), preconditionerCaching. */
int ogl_solver_get_property(ogl_solver *s, const char *key, double *value);
int ogl_solver_set_property(ogl_solver *s, const char *key, double value);

/* ------------------------------------------------------------------------------------ */
/* Device-resident entry points (benchmark / parity harness)                             */
/* ------------------------------------------------------------------------------------ */

/* solver->apply(b, x) alone on the resident vectors (lduLduBase.H:275-276): what the reference
 * times as delta_t_solve. */
int ogl_solver_apply_resident(ogl_solver *s, ogl_perf *perf);
/* Overwrite the resident solution / RHS (PersistentVector update, Vector.H:52-62); NULL => zeros. */
int ogl_solver_upload_solution(ogl_solver *s, const ogl_scalar *psi);
int ogl_solver_upload_rhs(ogl_solver *s, const ogl_scalar *source);
int ogl_solver_download_solution(ogl_solver *s, ogl_scalar *psi);

/* y = A x through the in-loop SpMV kernel (dist_mtx::apply, StoppingCriterion.C:29). */
int ogl_solver_spmv(ogl_solver *s, const ogl_scalar *x, ogl_scalar *y);
/* `repeats` back-to-back in-loop SpMVs (fused with the p.q dot, as in the CG loop) on resident
 * vectors, timed with HIP events on the solver's stream; avg_ms = per launch. */
int ogl_solver_time_spmv(ogl_solver *s, int32_t repeats, double *avg_ms);
/* Device reductions as used in the loop (deterministic fixed tree): dot, sum|a|, sum a. */
int ogl_solver_reduce(ogl_solver *s, int32_t op /*0 dot,1 norm1,2 sum*/, const ogl_scalar *a,
                      const ogl_scalar *b, double *out);
/* Rows per reduction chunk of the device tree (the oracle's BLOCKED mode mirrors it in tests). */
int ogl_reduction_chunk_rows(void);

/* Persistent device matrix read-back (<field>_local_{rows,cols,ldu_map,coeffs} and the non-local
 * twins, HostMatrix.C:40-69).  Sizes first (NULL arrays), then data. */
typedef struct ogl_matrix_dims {
    ogl_label n_rows;
    ogl_label local_nnz;       /* nrows + 2*upper_nnz + local_interface_nnz  (HostMatrix.C:38-39) */
    ogl_label non_local_nnz;   /* HostMatrix.C:55 */
    ogl_label n_halo;          /* columns of the non-local matrix (CsrMatrixWrapper.H:185-188) */
    ogl_label n_neighbours;
    ogl_label n_send;
} ogl_matrix_dims;
int ogl_solver_matrix_dims(ogl_solver *s, ogl_matrix_dims *dims);
int ogl_solver_get_local_matrix(ogl_solver *s, ogl_label *row_ptrs, ogl_label *cols,
                                ogl_label *ldu_mapping, ogl_scalar *coeffs);
int ogl_solver_get_non_local_matrix(ogl_solver *s, ogl_label *rows, ogl_label *cols,
                                    ogl_label *ldu_mapping, ogl_scalar *coeffs);
int ogl_solver_get_comm_pattern(ogl_solver *s, ogl_label *target_ids, ogl_label *target_sizes,
                                ogl_label *send_idxs);
/* Renumbering in use for this field's pattern (config `renumber`): returns 1 and fills
 * new_id[n_rows] (cell c of the lduMatrix is row new_id[c] of the arrays the three read-backs above
 * return; new_id may be NULL), 0 if the caller's numbering is used, or a negative status. */
int ogl_solver_get_renumbering(ogl_solver *s, ogl_label *new_id);

/* ------------------------------------------------------------------------------------ */
/* Pure host logic (runs without a GPU): HostMatrix/HostMatrixFreeFunctions.C:21-201     */
/* ("free functions - for unit testing" in the reference, same argument order)           */
/* ------------------------------------------------------------------------------------ */

void ogl_host_init_local_sparsity(ogl_label nrows, ogl_label upper_nnz, int is_symmetric,
                                  const ogl_label *upper, const ogl_label *lower, ogl_label *rows,
                                  ogl_label *cols, ogl_label *permute);
void ogl_host_symmetric_update(ogl_label total_nnz, ogl_label upper_nnz, const ogl_label *permute,
                               ogl_scalar scale, const ogl_scalar *diag, const ogl_scalar *upper,
                               ogl_scalar *out);
void ogl_host_symmetric_update_w_interface(ogl_label total_nnz, ogl_label diag_nnz,
                                           ogl_label upper_nnz, const ogl_label *permute,
                                           ogl_scalar scale, const ogl_scalar *diag,
                                           const ogl_scalar *upper, const ogl_scalar *iface,
                                           ogl_scalar *out);
void ogl_host_non_symmetric_update_w_interface(ogl_label total_nnz, ogl_label diag_nnz,
                                               ogl_label upper_nnz, const ogl_label *permute,
                                               ogl_scalar scale, const ogl_scalar *diag,
                                               const ogl_scalar *upper, const ogl_scalar *lower,
                                               const ogl_scalar *iface, ogl_scalar *out);
void ogl_host_non_symmetric_update(ogl_label total_nnz, ogl_label upper_nnz,
                                   const ogl_label *permute, ogl_scalar scale,
                                   const ogl_scalar *diag, const ogl_scalar *upper,
                                   const ogl_scalar *lower, ogl_scalar *out);

/* Whole host-side pattern of one lduMatrix (HostMatrix.C:159-178,251-306,412-589) without touching
 * a device: sizes with NULL arrays, then data.  Used by the CPU tests of the sharded path. */
int ogl_host_pattern(const ogl_ldu_view *ldu, ogl_matrix_dims *dims, ogl_label *local_rows,
                     ogl_label *local_cols, ogl_label *local_ldu_mapping, ogl_label *nl_rows,
                     ogl_label *nl_cols, ogl_label *nl_ldu_mapping, ogl_label *target_ids,
                     ogl_label *target_sizes, ogl_label *send_idxs);

/* The renumbering pieces on their own (pure host).  ogl_host_rcm: reverse Cuthill-McKee order of a
 * row-major pattern, new_id[old] = new.  ogl_host_gather_sector_ratio: distinct 64-byte sectors of x
 * per stored entry over groups of 256 consecutive entries (the locality measure the auto policy
 * uses); new_id NULL = the pattern's own numbering.  ogl_host_pattern_renumbered: ogl_host_pattern
 * followed by the policy of config `renumber` (mode) -- same outputs in the chosen numbering plus
 * new_id[n_cells] (identity when the caller's numbering was kept); returns 1 if renumbered. */
int ogl_host_rcm(ogl_label n_rows, const ogl_label *row_ptrs, const ogl_label *cols, ogl_label *new_id);
/* ... and the second candidate when ogl_ldu_view::cell_centres is given: new_id[old] = position of cell `old` along the
 * Hilbert curve (16 bits per axis over the bounding box) through the centres; ties keep the caller's order. */
int ogl_host_hilbert_order(ogl_label n_cells, const ogl_scalar *centres, ogl_label *new_id);
double ogl_host_gather_sector_ratio(ogl_label n_rows, const ogl_label *row_ptrs, const ogl_label *cols,
                                    const ogl_label *new_id);
int ogl_host_pattern_renumbered(const ogl_ldu_view *ldu, int32_t mode, int32_t compress_indices,
                                ogl_matrix_dims *dims, ogl_label *local_rows, ogl_label *local_cols,
                                ogl_label *local_ldu_mapping, ogl_label *nl_rows, ogl_label *nl_cols,
                                ogl_label *nl_ldu_mapping, ogl_label *target_ids,
                                ogl_label *target_sizes, ogl_label *send_idxs, ogl_label *new_id);

/* Hash of the WHOLE addressing of a view (every face, every interface cell): what decides, next to
 * the counts, whether a field's persistent pattern is rebuilt (the reference never rebuilds,
 * HostMatrix.C:79-87). */
uint64_t ogl_host_addressing_fingerprint(const ogl_ldu_view *ldu);

/* StoppingCriterion::build_dist_stopping_criterion's adaptive policy (StoppingCriterion.H:197-209). */
void ogl_host_adapt_criterion(const ogl_config *cfg, ogl_label prev_solve_iters,
                              ogl_scalar prev_rel_cost, ogl_label *min_iter, ogl_label *frequency);

/* Index-compressed chunked ELL layout (compress_indices) of a row-major sorted CSR pattern: builds
 * it, decodes it back and compares with the input.  stats[0] = 1 if the pattern qualifies (else 0
 * and the rest is 0), stats[1] = padded value slots, stats[2] = dictionary entries, stats[3] = code
 * bytes, stats[4] / stats[5] = chunks coded with 16-bit deltas / with plain 32-bit columns (the
 * others use 1-byte pattern or offset codes), stats[6] = value slots the kernel reads (a wavefront
 * runs to the longest of ITS 128 rows; stats[1] counts what is allocated, to the chunk's longest
 * row), stats[7] = entries spilled (tails of rows longer than their chunk's cap, added by a second
 * small kernel).  OGL_ERR_STATE if the decoded pattern differs from the input. */
int ogl_host_sell_check(ogl_label n_rows, const ogl_label *row_ptrs, const ogl_label *cols,
                        int64_t stats[8]);

/* Half storage of a symmetric matrix on a banded pattern (symmetric_half): builds the layout of a
 * row-major sorted CSR pattern and walks every row the way the kernel does.  stats[0] = 1 if the
 * pattern qualifies (else 0 and the rest is 0), stats[1] = planes (diagonal + distances), stats[2..5]
 * = the distances (ascending, the first is 0), stats[6] = plane slots, stats[7] = slots in use.
 * OGL_ERR_STATE if the walk does not reproduce the input. */
int ogl_host_sym_check(ogl_label n_rows, const ogl_label *row_ptrs, const ogl_label *cols,
                       int64_t stats[8]);

/* Half storage with per-chunk distances and explicit exceptions (symmetric matrices that are banded only
 * locally: multi-block structured meshes, refinement shells): builds the layout of a row-major sorted CSR pattern and
 * walks every row the way the kernel does, explicit entries merged by column.  stats[0] = 1 if the layout is worth
 * using (>= 80 % of the entries in planes), stats[1] = plane slots, stats[2] = entries served from planes, stats[3] =
 * explicit entries, stats[4] = chunks holding explicit entries, stats[5] = chunks.  OGL_ERR_STATE if the walk does
 * not reproduce the input. */
int ogl_host_symx_check(ogl_label n_rows, const ogl_label *row_ptrs, const ogl_label *cols,
                        int64_t stats[8]);

#ifdef __cplusplus
}
#endif
#endif /* OGL_AMD_H */
