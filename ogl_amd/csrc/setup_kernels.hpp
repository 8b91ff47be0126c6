// setup_kernels.hpp -- once-per-pattern work of HostMatrixWrapper done on the device (setup_kernels.hip):
// the lduMatrix addressing becomes the persistent row-major pattern + ldu_mapping without the host ever
// holding the 12 bytes per entry (init_local_sparsity_pattern, HostMatrix.C:468-589, whose tuple sorts take
// longer than a hundred solver turns at 10 M cells), and the layouts the SpMV kernels read are derived
// from it in place.  host_matrix.cpp keeps the same algorithms as the checked fallback (non-conforming
// addressing, OGL_HOST_SETUP=1) and as the reference the CPU tests pin against the reference's own vectors.
#pragma once
#include <hip/hip_runtime.h>

#include "common.hpp"

namespace ogl {

// out[0..n] = exclusive prefix sums of in[0..n) (out[n] = total); in may alias out.  tmp: at least
// scan_tmp_len(n) ints.
size_t scan_tmp_len(int64_t n);
void launch_exclusive_scan(hipStream_t st, const int32_t *in, int32_t *out, int64_t n, int32_t *tmp);

// ---- lduMatrix addressing -> row-major sorted pattern + ldu_mapping (HostMatrixFreeFunctions.C:105-201
// + the cyclic merge of HostMatrix.C:506-586).  Row r holds its face entries, its diagonal and its
// same-rank interface entries ordered by (column, position in the coefficient source) -- for conforming
// addressing (owner < neighbour on every face) exactly what the reference's [lower | diag | upper] rows
// with interface entries merged behind equal columns are.
struct PatternBuild {
    int32_t n_rows = 0, n_faces = 0, n_iface = 0;
    int32_t symmetric = 1;
    const int32_t *lower_addr = nullptr, *upper_addr = nullptr;  // [n_faces], device
    const int32_t *if_rows = nullptr, *if_cols = nullptr;        // [n_iface] same-rank interface entries, interface order
    int32_t *row_ptrs = nullptr;     // [n_rows + 1]
    int32_t *cols = nullptr;         // [nnz]
    int32_t *ldu_mapping = nullptr;  // [nnz]
    int32_t *diag_pos = nullptr;     // [n_rows] position of each row's first (r, r) entry; doubles as scratch
    int32_t *counts = nullptr;       // [n_rows] scratch (row lengths)
    int32_t *scan_tmp = nullptr;     // [scan_tmp_len(n_rows)]
    int32_t *flags = nullptr;        // [PATTERN_FLAGS] zeroed by the caller
};
enum PatternFlag { PAT_FLAG_NONCONFORMING = 0, PAT_FLAG_OUT_OF_RANGE = 1, PATTERN_FLAGS = 4 };
// enqueues everything; the caller reads flags (both must be 0) before trusting the arrays
void launch_build_pattern(hipStream_t st, const PatternBuild &b);

// ---- half storage of a symmetric matrix (SymLayout, host_matrix.hpp), from the device pattern
constexpr int SYM_TABLE = 8;  // slots of the distance table (>= SYM_MAX_OFFSETS + 1, so that an overflow shows)
enum SymFlag { SYM_FLAG_TOO_MANY = 0, SYM_FLAG_UNSORTED_ROW = 1, SYM_FLAGS = 4 };
// table[SYM_TABLE] must hold SYM_EMPTY on entry; on return (after a stream sync) the distinct distances
// column - row >= 0 of the pattern in arbitrary order, SYM_EMPTY in the unused slots
constexpr int32_t SYM_EMPTY = INT32_MIN;
void launch_sym_distances(hipStream_t st, int32_t n_rows, const int32_t *row_ptrs, const int32_t *cols,
                          int32_t *table, int32_t *flags);
// mask [n_chunks * CHUNK_ROWS (+16)] zeroed and map [n_chunks * nd * CHUNK_ROWS (+2)] filled with -1 by the caller
struct SymDistances {
    int32_t nd;
    int32_t d[SYM_MAX_OFFSETS];
};
void launch_sym_fill(hipStream_t st, int32_t n_rows, const int32_t *row_ptrs, const int32_t *cols,
                     SymDistances dist, uint8_t *mask, int32_t *map, int32_t *flags);

}  // namespace ogl

namespace ogl {

// ---- reverse Cuthill-McKee on the device: the same order as rcm_order (host_matrix.cpp), level by level ----
// State per node: lvl (-1 untouched, >= 0 level / placed), key (position of the earliest parent in the
// current frontier).  One BFS level = mark (atomicMin of the parents' positions) -> count -> scan -> emit.
struct RcmWork {
    int32_t n_rows = 0;
    const int32_t *row_ptrs = nullptr, *cols = nullptr;
    int32_t *lvl = nullptr;      // [n_rows]
    int32_t *key = nullptr;      // [n_rows]
    int32_t *order = nullptr;    // [n_rows] Cuthill-McKee order, filled front to back
    int32_t *scratch = nullptr;  // [n_rows] BFS order of the start-node sweep
    int32_t *cnt = nullptr;      // [n_rows + 1] children per frontier node, then their offsets
    int32_t *scan_tmp = nullptr; // [scan_tmp_len(n_rows)]
    unsigned long long *cell = nullptr;  // [4] reduction cells (seed search, far-node search)
};
constexpr int32_t RCM_NO_KEY = INT32_MAX;
// lvl = -1, key = RCM_NO_KEY for all nodes
void launch_rcm_init(hipStream_t st, const RcmWork &w);
// cell[0] = smallest index with lvl == -1 (or ~0 when none)
void launch_rcm_find_seed(hipStream_t st, const RcmWork &w);
// put `node` at list[pos] with lvl = level (single thread)
void launch_rcm_start(hipStream_t st, const RcmWork &w, int32_t *list, int32_t pos, int32_t node, int32_t level);
// one level: frontier = list[begin, end) (level `level`); children go to list[end, end + total), total -> cnt[end - begin]
// by_degree: children of a parent ordered by (degree, index) (Cuthill-McKee); else by index (= stored order)
void launch_rcm_level(hipStream_t st, const RcmWork &w, int32_t *list, int32_t begin, int32_t end, int32_t level,
                      bool by_degree);
// cell[1] = (degree << 32 | position) minimum over list[begin, end)
void launch_rcm_far_node(hipStream_t st, const RcmWork &w, const int32_t *list, int32_t begin, int32_t end);
// lvl = -1, key = none for the nodes of list[0, count) (undo a sweep)
void launch_rcm_reset(hipStream_t st, const RcmWork &w, const int32_t *list, int32_t count);
// new_id[order[n - 1 - k]] = k
void launch_rcm_finish(hipStream_t st, const RcmWork &w, int32_t *new_id);

// ---- the pattern in another numbering (renumber_pattern, host_matrix.cpp): row new_id[r] holds row r's
// entries with columns renamed, ordered by new column (stable)
struct RenumberWork {
    int32_t n_rows = 0;
    const int32_t *new_id = nullptr;                                  // [n_rows]
    const int32_t *row_ptrs = nullptr, *cols = nullptr, *map = nullptr;   // the pattern as it is
    int32_t *old_of = nullptr;                                        // [n_rows] out: inverse permutation
    int32_t *row_ptrs_out = nullptr, *cols_out = nullptr, *map_out = nullptr, *diag_pos_out = nullptr;
    int32_t *scan_tmp = nullptr;
};
void launch_renumber_pattern(hipStream_t st, const RenumberWork &w);

// ---- packed columns of the CSR-stream kernel (Stream21Chunk, common.hpp) ----
struct Stream21Build {
    int32_t n_rows = 0;
    const int32_t *row_ptrs = nullptr, *cols = nullptr;
    Stream21Chunk *chunks = nullptr;  // [n_chunks] out
    int32_t *words = nullptr;         // [n_chunks + 1] scratch: code words per chunk, then their offsets; [n_chunks] = total
    int32_t *scan_tmp = nullptr;      // [scan_tmp_len(n_chunks)]
    int32_t *flags = nullptr;         // [1] zeroed by the caller (unused since the far lists: kept for the launch signature)
    int32_t *far = nullptr;           // [n_chunks + 1] scratch: far entries per chunk, then their offsets; [n_chunks] = total
};
// per chunk: window base, code words needed, far entries; offsets by two scans (read words[n_chunks] and
// far[n_chunks] afterwards)
void launch_stream21_plan(hipStream_t st, const Stream21Build &b);
// codes and the far lists (far_idx / far_col: far[n_chunks] entries each; may be nullptr when that is 0)
void launch_stream21_fill(hipStream_t st, const Stream21Build &b, uint4 *codes, int32_t *far_idx, int32_t *far_col);

// ---- the cells along a Hilbert curve through their centres (hilbert_order, host_matrix.cpp: Skilling's transpose, 16 bits
// per axis), on the device: keys, a stable radix sort of (key, cell), the inverse permutation.  The same numbering as the
// host's: the keys are formed by the same double operations (lo and scale come from the host), ties keep the caller's order.
// keys / cells: [n] each, in and out buffers of the sort; temp: hilbert_sort_temp_bytes(n) bytes
size_t hilbert_sort_temp_bytes(int32_t n);
int hilbert_order_device(hipStream_t st, int32_t n, const double *centres, const double lo[3], double scale,
                         unsigned long long *keys, unsigned long long *keys_out, int32_t *cells, int32_t *cells_out,
                         void *temp, size_t temp_bytes, int32_t *new_id);
// entries of the pattern (caller's numbering: row_ptrs / cols) that would fall outside their chunk's 2^21-column window
// in the numbering new_id / old_of (what choose_numbering counts before it takes the curve: STREAM21_MAX_FAR); *far_out
// must be zero on entry
void launch_curve_far_count(hipStream_t st, int32_t n_rows, const int32_t *row_ptrs, const int32_t *cols,
                            const int32_t *new_id, const int32_t *old_of, unsigned long long *far_out);

}  // namespace ogl
