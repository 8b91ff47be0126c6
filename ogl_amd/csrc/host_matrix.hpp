// host_matrix.hpp -- host-side data preparation: lduMatrix -> row-major sparsity pattern,
// ldu_mapping permutation, halo (non-local) pattern and communication pattern.
// Re-implements what HostMatrixWrapper builds once per field (reference HostMatrix/HostMatrix.C,
// HostMatrix/HostMatrixFreeFunctions.C).  Pure host code: no device, no oracle.
#pragma once
#include <functional>
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
    // false: the four arrays above were built on the device (setup_kernels.hip) and exist only there; the
    // solver downloads cols / ldu_mapping / row_ptrs (never `rows`) when a host consumer asks for them
    bool local_on_host = true;

    // "<field>_non_local_{rows,cols,ldu_map}"
    ogl_label non_local_nnz = 0;  // HostMatrix.C:55
    std::vector<ogl_label> nl_rows, nl_cols, nl_ldu_mapping;

    // communication pattern, ascending neighbour rank (HostMatrix.C:251-306)
    std::vector<ogl_label> target_ids, target_sizes, send_idxs;

    // Renumbering (this build's addition, keyword `renumber`): empty = the caller's numbering;
    // otherwise cell c of the lduMatrix is row/column new_id[c] of everything above (rows, cols,
    // nl_rows, send_idxs are in the NEW numbering; ldu_mapping / nl_ldu_mapping still address the
    // caller's coefficient arrays) and old_of is the inverse map.
    std::vector<ogl_label> new_id, old_of;
    bool renumbered() const { return !new_id.empty(); }

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
// The same in two halves, for the device set-up path: everything that is small (validation, counts, non-local
// pattern, communication pattern; no fingerprint) ...
int build_host_pattern_meta(const ogl_ldu_view &ldu, HostPattern &p, bool check_faces);
// ... and the local pattern arrays (rows, cols, ldu_mapping, row_ptrs) on the host
void build_local_pattern(const ogl_ldu_view &ldu, HostPattern &p);
// (row, col) of every same-rank interface face in interface order (HostMatrix.C:385-410)
void local_interface_entries(const ogl_ldu_view &ldu, std::vector<ogl_label> &rows, std::vector<ogl_label> &cols);

// Cheap identity check used to decide whether a cached pattern still matches a new view
// ("For now we assume columns and rows to be constant", HostMatrix.H:33).
uint64_t addressing_fingerprint(const ogl_ldu_view &ldu);
// same counts and same addressing (full hash) as the view the pattern was built from
bool same_shape(const ogl_ldu_view &ldu, const HostPattern &p);
// the cheap half of it: sizes and symmetry only
bool same_counts(const ogl_ldu_view &ldu, const HostPattern &p);

// Jacobi block pointers for maxBlockSize > 1 ([UPSTREAM] gko::preconditioner::Jacobi
// find_blocks): natural blocks = runs of consecutive rows with identical column pattern (capped at
// max_block_size), adjacent natural blocks agglomerated while the merged size <= max_block_size.
// On a renumbered pattern the blocks are those of the CALLER's numbering (block_ptrs / row_block refer to positions
// there, position i = row p.new_id[i] here) unless caller_numbering is false (an A/B switch: round-3 behaviour).
void find_jacobi_blocks(const HostPattern &p, ogl_label max_block_size,
                        std::vector<ogl_label> &block_ptrs, std::vector<ogl_label> &row_block,
                        bool caller_numbering = true);

// Index-compressed chunked ELL of a row-major sorted pattern (SellChunk, common.hpp).  Returns false
// (and leaves `out` unusable) when the pattern does not qualify; the CSR-stream kernel runs then.
struct SellLayout {
    std::vector<SellChunk> chunks;
    std::vector<int32_t> dict;   // ascending (column - row) offsets, chunk after chunk
    std::vector<uint8_t> codes;  // thread-major code bytes (+16 bytes of padding)
    std::vector<int32_t> map;    // value slot -> position in the CSR value array, -1 = padding
    int64_t n_slots = 0;         // padded value slots
    int64_t n_delta16 = 0, n_col32 = 0;  // chunks coded with 16-bit deltas / plain 32-bit columns
    int64_t read_slots = 0;  // value slots in the 128-byte lines the kernel reads (lanes stop at their own rows' ends)
    // spill: the tails of the rows that are longer than their chunk's cap, row-sorted; the workgroup of the
    // chunk adds them after the planes, continuing every row's sum in stored order
    std::vector<int32_t> spill_rows, spill_ptrs, spill_cols, spill_map;  // map = position in the CSR values
    std::vector<int32_t> spill_chunk_ptr;  // [n_chunks + 1] range of spill_rows that belongs to each chunk
};
// allow_spill false: every row keeps all its entries in the planes (layouts of matrices whose apply has no
// second pass, e.g. the ISAI factors)
bool build_sell_layout(ogl_label n_rows, const ogl_label *row_ptrs, const ogl_label *cols,
                       SellLayout &out, bool allow_spill = true);

// ---- half storage of a symmetric matrix on a banded pattern (config `symmetric_half`) ----
// A symmetric lduMatrix carries every off-diagonal coefficient once (`upper`); the reference expands it to a
// full CSR and so does the compressed layout above.  Where the pattern is banded -- every entry sits at one
// of a handful of distances from the diagonal, a structured mesh -- the device can keep the OpenFOAM way:
// plane j (of nd <= SYM_MAX_OFFSETS) of a chunk holds A(r, r + d[j]) for the chunk's rows, d[0] = 0 being
// the diagonal, and the SpMV reads the lower entry A(r, r - d[j]) as plane j of row r - d[j]: the same
// bytes it reads as an upper entry of that row, one coalesced strip further back.  DRAM sees every
// coefficient once: 8 nd + 1 bytes per row instead of 8.1 per stored entry.  One byte per row tells which
// of its 2 nd - 1 possible entries exist; rows are summed in ascending column order, so y and the fused
// dot partials have the same bits as with full storage.
// (SYM_MAX_OFFSETS, SYM_MAX_PADDING: common.hpp)
struct SymLayout {
    int32_t nd = 0;
    int32_t d[SYM_MAX_OFFSETS] = {};  // ascending, d[0] = 0
    std::vector<uint8_t> mask;        // [n_chunks * CHUNK_ROWS] bit (nd-1-j): entry at -d[j] (j >= 1); bit (nd-1+j): at +d[j]
    std::vector<int32_t> map;         // [n_chunks * nd * CHUNK_ROWS] plane slot -> position in the CSR values, -1 = none
};
// false: the pattern does not qualify (more distances than planes, a lower entry without its upper twin,
// too much padding).  The VALUES being symmetric is the caller's knowledge (lduMatrix without `lower`).
bool build_sym_layout(ogl_label n_rows, const ogl_label *row_ptrs, const ogl_label *cols, SymLayout &out);
// ---- half storage with per-chunk distances and explicit exceptions (SymxChunk, common.hpp) ----
// For symmetric matrices whose pattern is banded only locally: a multi-block structured mesh (every block has its
// own line and plane lengths, the block interfaces couple at arbitrary distances), a hex mesh with a refinement
// shell.  Per chunk the three most frequent upper distances get planes; an upper entry is planar when its distance
// is one of them, a lower entry when, in addition, the chunk of its twin holds that distance too (the twin's plane
// is where it is read); everything else -- later duplicates of a column included -- is an explicit entry
// (column + value, per-chunk row pointers) that the kernel merges into the row sum by column, so y keeps the bits of
// a row-major walk.
struct SymxLayout {
    std::vector<SymxChunk> chunks;
    std::vector<uint8_t> mask;        // [n_chunks * CHUNK_ROWS (+16)]
    std::vector<int32_t> map;         // plane slot -> position in the CSR values, -1 = none  (+2)
    std::vector<int32_t> ex_rowptr;   // (CHUNK_ROWS + 1) pointers per chunk that has explicit entries
    std::vector<int32_t> ex_cols, ex_map;  // explicit entries: column, position in the CSR values
    std::vector<int32_t> ex_lrow;          // ... and the row, counted from the chunk's first
    int64_t planar = 0;               // entries served from planes
    bool all_fast = false;            // every chunk with distances: the first is 1, the others are even
};
// false: not worth it (fewer than SYMX_MIN_PLANAR of the entries planar, or more plane slots than
// SYM_MAX_PADDING x the diagonal + upper entries they hold)
bool build_symx_layout(ogl_label n_rows, const ogl_label *row_ptrs, const ogl_label *cols, SymxLayout &out);

// Order of the SpMV's workgroups for a pattern whose furthest leg couples row r with r +- band: workgroup b
// works on chunk order[b] (-1: none).  Workgroups go to the XCDs round robin (b % N_XCD); the order gives XCD
// k the chunks whose first row lies in the k-th eighth of its band period, ascending -- so the chunk of row
// r and the chunks of r +- band share an XCD, hence an L2: the x strips and (half storage) the value planes
// that two of them read are fetched over the fabric once.  Empty when the band is too short to give every
// XCD a chunk per period or too long to recur.
void band_block_order(ogl_label n_rows, int64_t band, std::vector<int32_t> &order);
// the same with every chunk's own band (its largest distance): multi-block meshes (chunks without a band worth it:
// groups of 4 per XCD, as the default map of the other kernels).  k_spmv_symx reads its headers in this order.
// general = false: the chunks the lean kernel takes (no explicit entries, or simple ones: SymxChunk::merge == 0),
// true: the others.  Empty when there is no such chunk.
void symx_block_order(const SymxLayout &L, bool general, std::vector<int32_t> &order);

// ---- renumbering (no reference counterpart: OpenFOAM users run `renumberMesh`; here the backend
// does it for itself when the numbering it is handed gathers x badly) ----
// Reverse Cuthill-McKee order of the graph of a row-major pattern (George-Liu pseudo-peripheral
// start per connected component, neighbours by ascending degree, index as tie break):
// new_id[old] = new, a permutation of [0, n_rows).
void rcm_order(ogl_label n_rows, const ogl_label *row_ptrs, const ogl_label *cols,
               std::vector<ogl_label> &new_id);
// How well the x gather of an SpMV coalesces under a numbering: distinct 64-byte sectors of x per
// stored entry, over groups of 256 consecutive stored entries (what one wavefront instruction of the
// CSR-stream kernel gathers), sampled.  1/8 is the floor (a dense run), ~1 is a random gather.
// new_id == nullptr: the pattern's own numbering.
double gather_sector_ratio(ogl_label n_rows, const ogl_label *row_ptrs, const ogl_label *cols,
                           const ogl_label *new_id, const ogl_label *old_of);
// The same per instruction of the compressed layout's kernel (slot-major: the s-th entry of the 64 even
// or odd rows of a wavefront's 128 rows).  0.25 on a hex mesh in natural order.
double slot_gather_sector_ratio(ogl_label n_rows, const ogl_label *row_ptrs, const ogl_label *cols,
                                const ogl_label *new_id, const ogl_label *old_of);
// The two heavy steps of the renumbering may be done elsewhere (the solver does them on the device,
// setup_kernels.hip, with the same results): a hook that returns false leaves the step to the host code.
struct NumberingHooks {
    // reverse Cuthill-McKee order of p's graph: new_id[old] = new
    std::function<bool(const HostPattern &p, std::vector<ogl_label> &new_id)> rcm;
    // p.row_ptrs / p.cols / p.ldu_mapping rewritten into the numbering new_id (p.rows is left alone)
    std::function<bool(HostPattern &p, const std::vector<ogl_label> &new_id)> renumber_local;
    // cell centres (x, y, z per cell; ogl_ldu_view::cell_centres), nullptr = not given: a second candidate for the
    // order at large -- the cells along a Hilbert curve through their centres -- next to reverse Cuthill-McKee
    const double *centres = nullptr;
    // ... and the two heavy steps of THAT candidate: the order along the curve (hilbert_order) and the count of entries that
    // would fall outside their chunk's window of packed columns (both: false = left to the host code)
    std::function<bool(ogl_label n, const double *centres, std::vector<ogl_label> &new_id)> curve;
    std::function<bool(const HostPattern &p, const std::vector<ogl_label> &new_id, const std::vector<ogl_label> &old_of,
                       int64_t &far)> curve_far;
};
// new_id[old] = position of cell `old` along the Hilbert curve (16 bits per axis over the bounding box) through the
// cell centres; ties keep the caller's order
void hilbert_order(ogl_label n, const double *centres, std::vector<ogl_label> &new_id);
// Rewrites `p` (built by build_host_pattern in the caller's numbering) into the numbering new_id.
// Rows keep their entries; within a row the entries are ordered by NEW column (stable).
void renumber_pattern(HostPattern &p, std::vector<ogl_label> new_id, const NumberingHooks *hooks = nullptr);
// true: one of a few chunks sampled over the matrix needs 16-bit delta / 32-bit column codes in the compressed
// layout (an irregular pattern for sure); false: none of the sampled ones does
bool sell_pattern_is_irregular_sampled(ogl_label n_rows, const ogl_label *row_ptrs, const ogl_label *cols);
struct SellLayout;
struct RenumberReport {
    bool applied = false;
    bool sorted_by_length = false;  // rows of a wavefront reordered longest first (compressed layout)
    bool sell_natural = false, sell_used = false;  // compressed layout qualifies (only when tried)
    double ratio_natural = 0.0, ratio_used = 0.0;  // gather_sector_ratio before / after
    double ratio_rcm = -1.0, ratio_curve = -1.0;    // ... of the two candidates (-1: not formed)
    bool curve_used = false;                        // the Hilbert order through the cell centres was taken
    bool curve_packable = true;                     // ... the packed columns of the CSR-stream kernel stay possible along the curve
    int64_t curve_far_entries = 0;                  // entries outside their chunk's 2^21-column window along the curve
    double slot_ratio = 0.0;  // slot_gather_sector_ratio of the numbering at large (0: not needed)
};
// mode 0: keep the caller's numbering; 1: always RCM; 2 (default, "auto"): keep it when the
// compressed layout qualifies with 1-byte codes throughout (structured mesh); otherwise RCM when the
// gather coalesces badly (ratio > 0.25) and RCM cuts the sector ratio by >= 10 %.  With the compressed
// layout in play (try_sell), modes 1 and 2 then put the rows of every wavefront (SELL_WAVE_ROWS consecutive
// rows) longest first when row lengths are mixed and the layout's slot-major gather is local enough to have
// a chance (slot_gather_sector_ratio <= SELL_SORT_MAX_SLOT_RATIO): the lanes of the kernel stop loading at
// the end of their own rows, so sorted rows leave (almost) no padding in the lines that are read.  No row
// leaves its wavefront.  (Sorting over a whole chunk was tried and dropped: it scatters the gather,
// profiles/r02_unstructured_proxy.txt.)
// `sell_out` (may be null) receives the compressed layout of the numbering that was chosen when
// one was built on the way (sell_built tells), so the caller does not derive it twice.
int choose_numbering(HostPattern &p, int mode, bool try_sell, SellLayout *sell_out, bool *sell_built,
                     RenumberReport &rep, const NumberingHooks *hooks = nullptr);

// Pattern of the ISAI approximate inverse W (keyword sparsityPower, Preconditioner.H:227): rows of
// S^power in ascending column order, S = tril(A) (spd = ISAI) or A (general = GISAI).  Returns false
// (first_wide_row set) when a row would get more than `max_row` entries.
// spd on a renumbered pattern: the triangle is taken by the caller's index (P tril(A) P^T) unless caller_numbering
// is false.
bool isai_pattern(const HostPattern &p, bool spd, int power, int max_row, std::vector<ogl_label> &w_row_ptrs,
                  std::vector<ogl_label> &w_cols, ogl_label &first_wide_row, bool caller_numbering = true);

// HostMatrix.C:180-207: concatenated bouCoeffs of the (non-)processor interfaces, times -1.
void collect_interface_coeffs(const ogl_ldu_view &ldu, bool local, ogl_scalar *out);

}  // namespace ogl
