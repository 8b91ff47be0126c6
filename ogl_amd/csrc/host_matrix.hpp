// host_matrix.hpp -- host-side data preparation: lduMatrix -> row-major sparsity pattern,
// ldu_mapping permutation, halo (non-local) pattern and communication pattern.
// Re-implements what HostMatrixWrapper builds once per field (reference HostMatrix/HostMatrix.C,
// HostMatrix/HostMatrixFreeFunctions.C).  Pure host code: no device, no oracle.
#pragma once
#include <vector>

#include "common.hpp"

namespace ogl {

// PersistentSparsityPattern (HostMatrix.H:21-64) x2 + CommunicationPattern (HostMatrix.H:67-79)
struct HostPattern {
    uint64_t fingerprint = 0;  // addressing_fingerprint() of the view the pattern was built from
    ogl_label n_rows = 0;
    ogl_label upper_nnz = 0;
    bool symmetric = true;
    ogl_label local_iface_nnz = 0;  // local_interface_nnz_      HostMatrix.C:35
    ogl_label local_nnz = 0;        // local_matrix_w_interfaces_nnz_  HostMatrix.C:39

    // "<field>_local_{rows,cols,ldu_map}"
    std::vector<ogl_label> rows, cols, ldu_mapping;
    std::vector<ogl_label> row_ptrs;  // CSR view of `rows` (Csr::read of the sorted triplets)

    // "<field>_non_local_{rows,cols,ldu_map}"
    ogl_label non_local_nnz = 0;  // HostMatrix.C:55
    std::vector<ogl_label> nl_rows, nl_cols, nl_ldu_mapping;

    // communication pattern, ascending neighbour rank (HostMatrix.C:251-306)
    std::vector<ogl_label> target_ids, target_sizes, send_idxs;

    // length of the unsorted coefficient source [upper | lower(asym) | diag | local-iface]
    // that ldu_mapping indexes (HostMatrix.C:644-682)
    int64_t source_len() const
    {
        return (symmetric ? 1 : 2) * (int64_t)upper_nnz + n_rows + local_iface_nnz;
    }
    ogl_label diag_start() const { return symmetric ? upper_nnz : 2 * upper_nnz; }
};

// HostMatrixFreeFunctions.C:105-201
void init_local_sparsity(ogl_label nrows, ogl_label upper_nnz, bool is_symmetric,
                         const ogl_label *upper, const ogl_label *lower, ogl_label *rows,
                         ogl_label *cols, ogl_label *permute);

// Validates the view and fills `p`.  Returns OGL_OK or a negative status (message set).
int build_host_pattern(const ogl_ldu_view &ldu, HostPattern &p);

// Cheap identity check used to decide whether a cached pattern still matches a new view
// ("For now we assume columns and rows to be constant", HostMatrix.H:33).
uint64_t addressing_fingerprint(const ogl_ldu_view &ldu);
// same counts and same (sampled) addressing as the view the pattern was built from
bool same_shape(const ogl_ldu_view &ldu, const HostPattern &p);

// Jacobi block pointers for maxBlockSize > 1 ([UPSTREAM] gko::preconditioner::Jacobi
// find_blocks): natural blocks = runs of consecutive rows with identical column pattern (capped at
// max_block_size), adjacent natural blocks agglomerated while the merged size <= max_block_size.
void find_jacobi_blocks(const HostPattern &p, ogl_label max_block_size,
                        std::vector<ogl_label> &block_ptrs, std::vector<ogl_label> &row_block);

// Index-compressed chunked ELL of a row-major sorted pattern (SellChunk, common.hpp).  Returns false
// (and leaves `out` unusable) when the pattern does not qualify; the CSR-stream kernel runs then.
struct SellLayout {
    std::vector<SellChunk> chunks;
    std::vector<int32_t> dict;   // ascending (column - row) offsets, chunk after chunk
    std::vector<uint8_t> codes;  // thread-major code bytes (+16 bytes of padding)
    std::vector<int32_t> map;    // value slot -> position in the CSR value array, -1 = padding
    int64_t n_slots = 0;         // padded value slots
};
bool build_sell_layout(ogl_label n_rows, const ogl_label *row_ptrs, const ogl_label *cols,
                       SellLayout &out);

// HostMatrix.C:180-207: concatenated bouCoeffs of the (non-)processor interfaces, times -1.
void collect_interface_coeffs(const ogl_ldu_view &ldu, bool local, ogl_scalar *out);

}  // namespace ogl
