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
