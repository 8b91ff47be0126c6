// solver.hpp -- persistent per-(rank, field) device state and the Krylov drivers.
// Re-implements the reference's L2-L4: DevicePersistent/*, HostMatrixWrapper's
// device half, Preconditioner caching, StoppingCriterion policy and lduLduBase orchestration.
#pragma once
#include <hip/hip_runtime.h>

#include <algorithm>
#include <map>
#include <memory>
#include <string>
#include <vector>

#include "comm.hpp"
#include "common.hpp"
#include "host_matrix.hpp"
#include "kernels.hpp"
#include "setup_kernels.hpp"

namespace ogl {

#define OGL_HIP_CHECK(expr)                                                                \
    do {                                                                                   \
        hipError_t e_ = (expr);                                                            \
        if (e_ != hipSuccess)                                                              \
            return ::ogl::fail(OGL_ERR_HIP, "%s failed: %s (%s:%d)", #expr,                \
                               hipGetErrorString(e_), __FILE__, __LINE__);                 \
    } while (0)
#define OGL_TRY(expr)                 \
    do {                              \
        int rc_ = (expr);             \
        if (rc_ != OGL_OK) return rc_; \
    } while (0)

// PersistentArray<T> (DevicePersistent/Array/Array.H:91-229): a named device array that lives as
// long as its registry.
template <class T>
struct DevBuf {
    T *p = nullptr;
    size_t n = 0;    // elements in use
    size_t cap = 0;  // elements allocated (>= n)
    DevBuf() = default;
    DevBuf(const DevBuf &) = delete;
    DevBuf &operator=(const DevBuf &) = delete;
    ~DevBuf() { release(); }
    void release()
    {
        ledger::dev_free(p);
        p = nullptr;
        n = cap = 0;
    }
    void swap(DevBuf &o)
    {
        std::swap(p, o.p);
        std::swap(n, o.n);
        std::swap(cap, o.cap);
    }
    // `count` elements, zero-filled when the size changes.  A block that is large enough (and not more than four times
    // too large) is kept: sizes that go back and forth from solve to solve -- the registry-wide preconditioner store
    // taking scalar Jacobi, blocks, W in turn (Preconditioner.H:357: one key for all fields), a residual history whose
    // length follows the adaptive evaluation frequency -- then cost no hipFree / hipMalloc pair per time step, and the
    // same pointers come back (a captured hipGraph stays valid).
    int alloc(size_t count, hipStream_t st)
    {
        if (count == n && p) return OGL_OK;
        if (p && count > 0 && count <= cap && count >= cap / 4) {
            OGL_HIP_CHECK(hipMemsetAsync(p, 0, count * sizeof(T), st));
            n = count;
            return OGL_OK;
        }
        release();
        if (count == 0) return OGL_OK;
        OGL_HIP_CHECK(ledger::dev_malloc(reinterpret_cast<void **>(&p), count * sizeof(T)));
        OGL_HIP_CHECK(hipMemsetAsync(p, 0, count * sizeof(T), st));
        n = cap = count;
        return OGL_OK;
    }
};

// Pinned double-buffered staging for pageable host arrays (K10/K12: "pinned async copies").
// Pageable host arrays <-> device through a ring of pinned buffers.  The copy between the caller's array and a
// pinned buffer is what limits a coefficient refresh (one core moves 10-25 GB/s, the PCIe 5 x16 link takes 55): it
// is split over a small pool of persistent helper threads (OGL_STAGE_THREADS, default 8) that store past the
// caches (non-temporal: the DMA engine -- or, coming down, the caller -- reads the data from DRAM anyway, and a
// plain store would first read the destination line), while the DMA of the previous buffers is in flight.
class CopyPool;
class Stager {
public:
    static constexpr int NBUF = 4;
    ~Stager();
    int init(size_t chunk_bytes);
    int h2d(void *dst, const void *src, size_t bytes, hipStream_t st);
    int d2h(void *dst, const void *src, size_t bytes, hipStream_t st);  // returns after completion

private:
    void *pin_[NBUF] = {};
    hipEvent_t ev_[NBUF] = {};
    bool busy_[NBUF] = {};
    size_t chunk_ = 0;
    int next_ = 0;
    CopyPool *pool_ = nullptr;
};

class Stager;

// Device copy of an index-compressed chunked ELL (SellChunk, common.hpp) of some CSR matrix whose
// values live elsewhere: pattern once (`build`), values by `refresh` from the CSR value array.
struct SellDev {
    DevBuf<SellChunk> chunks;
    DevBuf<int32_t> dict, map;
    DevBuf<uint8_t> codes;
    DevBuf<double> vals;
    int64_t slots = 0, read_slots = 0;
    bool ready = false;  // false: the pattern does not qualify (or build was never called)
    // rows of each wavefront's window stored longest first (sort_windows; the kernel undoes it: DevSell::rmap)
    DevBuf<uint16_t> rmap;
    bool sorted = false;
    int build(ogl_label n_rows, const ogl_label *row_ptrs, const ogl_label *cols, Stager &stager,
              hipStream_t st, bool sort_windows = false);
    void refresh(const double *csr_vals, hipStream_t st)
    {
        if (ready) launch_gather_sell(st, (int32_t)chunks.n, chunks.p, map.p, csr_vals, vals.p);
    }
    DevSell view(int32_t n_rows, bool stream) const
    {
        DevSell S;
        S.n_rows = n_rows;
        S.chunks = chunks.p;
        S.dict = dict.p;
        S.codes = codes.p;
        S.vals = vals.p;
        S.rmap = sorted ? rmap.p : nullptr;
        S.stream = stream;
        return S;
    }
};

// A generated preconditioner ("Cached_preconditinoner" holds one of these, Preconditioner.H:357)
struct PrecondData {
    int kind = 0;  // 0 none, 1 scalar Jacobi (inverse diagonal), 2 block Jacobi
    size_t n_rows = 0;
    int stride = 0;  // block Jacobi: maxBlockSize
    int32_t n_blocks = 0;
    bool uniform_blocks = false;  // block Jacobi: every block but the last has exactly `stride` rows
    bool through_perm = false;    // block Jacobi: block_ptrs / row_block are positions in the caller's numbering
    uint64_t perm_pat_id = 0;     // ... of THIS pattern's permutation
    bool by_device_row = false;   // block rows stored at their device rows (direct apply) instead of block-major (staged)
    DevBuf<double> values;  // inverse diagonal (n_rows + 2) or inverted blocks
    DevBuf<int32_t> block_ptrs, row_block;
    // ISAI (kind 3: spd, M^-1 = W^T W; kind 4: general, M^-1 = W): CSR arrays padded like the
    // system matrix so the CSR-stream SpMV kernel applies them; wt_map = position in W of every
    // entry of W^T
    DevBuf<int32_t> w_row_ptrs, w_cols, wt_row_ptrs, wt_cols, wt_map;
    DevBuf<double> w_vals, wt_vals;
    int32_t w_nnz = 0, w_max_row = 0;
    DevBuf<int32_t> wide_rows;  // rows of W with ISAI_THREAD_ROW < entries <= MAX_ISAI_ROW (one wavefront each)
    int32_t n_wide_rows = 0;
    // rows of W with more than MAX_ISAI_ROW entries (one workgroup each, dense system in global scratch):
    // huge_off[k] = start of row huge_rows[k]'s system in huge_scratch; huge_batches = runs of rows that share
    // the scratch at one time
    DevBuf<int32_t> huge_rows;
    DevBuf<int64_t> huge_off;
    DevBuf<double> huge_scratch;  // allocated for a generation, released after it
    int64_t huge_scratch_len = 0;
    std::vector<int32_t> huge_batches;
    int32_t n_huge_rows = 0;
    SellDev w_sell, wt_sell;  // compressed copies the apply runs on when compress_indices is set
    // the pattern-only part (block pointers / W and W^T patterns) is kept for as long as it was
    // derived from the same sparsity pattern: only the values are regenerated per solve
    uint64_t struct_pat_id = 0;
    int struct_kind = 0, struct_stride = 0;
    bool struct_caller_numbering = true;
    bool has_structure(uint64_t id, int k, int st) const
    {
        return id != 0 && struct_pat_id == id && struct_kind == k && struct_stride == st;
    }
    bool matches(int k, size_t n, int st) const { return kind == k && n_rows == n && stride == st; }
    // Which numbering the VALUES are laid out in.  The store is shared by all fields (Preconditioner.H:357), so the
    // object may be applied by a solver other than the one that generated it: that is sound when both see the caller's
    // numbering, or when the object is a block Jacobi kept block-major in the caller's order (the applying solver carries
    // the vectors through ITS permutation); everything else -- inverse diagonal, W / W^T, block rows stored by device
    // row, the backend's own blocks -- belongs to the generating pattern's device numbering.
    uint64_t gen_pat_id = 0;
    bool gen_device_numbering = false;
    bool caller_order_blocks() const { return kind == 2 && !by_device_row && !gen_device_numbering; }
    bool foreign_to(uint64_t id, bool renumbered) const
    {
        if (gen_pat_id == id) return false;
        return gen_device_numbering || (renumbered && !caller_order_blocks());
    }
};

}  // namespace ogl

struct ogl_solver;

// objectRegistry analogue (DevicePersistent/Base/Base.H:53-137) + ExecutorHandler
struct ogl_registry {
    int device = -1;
    hipStream_t stream = nullptr;
    bool own_stream = false;
    // halo exchange runs on its own stream so that it overlaps the local SpMV (K3)
    hipStream_t comm_stream = nullptr;
    hipEvent_t ev_packed = nullptr, ev_received = nullptr;
    std::unique_ptr<ogl::Comm> comm;
    std::map<std::string, std::unique_ptr<ogl_solver>> solvers;
    ogl::Stager stager;
    // "Cached_preconditinoner" (sic) -- one registry-wide slot (Preconditioner.H:357)
    ogl::PrecondData cached_precond;
    bool has_cached_precond = false;
    // Peer-write all-reduce mesh (PeerArgs, kernels.hpp): own mailbox + the other ranks' mailboxes
    // mapped through hipIpc.  When `peer_ready`, the scalar all-reduces run inside the finaliser
    // kernels instead of through comm->allreduce (the halo exchange stays with `comm`).
    unsigned long long *peer_local = nullptr;
    void *peer_mapped[ogl::PEER_MAX_RANKS] = {};  // hipIpcOpenMemHandle results (closed on destroy)
    int32_t *peer_error = nullptr;                // device flag for the stand-alone all-reduce kernel
    ogl::PeerArgs peer{};                         // world/rank/box[]; seq is stamped per call
    uint32_t peer_seq = 0;
    bool peer_ready = false;
    bool peer_shared_device = false;  // two ranks of the mesh report the same PCI bus id (peer_connect)
    // peer-put halo arena behind the mailbox in the same IPC allocation (PeerHalo, kernels.hpp):
    // solvers take blocks at pattern-build time; every rank builds patterns in the same order, so
    // `halo_epoch` names the same handshake on all ranks
    size_t arena_words = 0, arena_used = 0;
    uint32_t halo_epoch = 0;
    int peer_export(void *handle_out);
    int peer_connect(int rank, int n_ranks, const void *handles);
    void peer_close();
    // next all-reduce's arguments (every rank calls this the same number of times, in the same order)
    ogl::PeerArgs peer_next()
    {
        ogl::PeerArgs p = peer;
        if (++peer_seq == 0) ++peer_seq;
        p.seq = peer_seq;
        return p;
    }
    // in-place SUM of n <= 2 doubles over the ranks, on `stream`
    int allreduce(double *dev, int n);
    ~ogl_registry();
};

struct ogl_solver {
    ogl_registry *reg = nullptr;
    std::string field;
    ogl_config cfg{};

    // ---- HostMatrixWrapper state ----
    ogl::HostPattern pat;
    bool have_pattern = false;
    uint64_t pat_id = 0;  // unique per built sparsity pattern (keys the preconditioner structure)
    bool matrix_set = false;
    ogl::DevBuf<int32_t> d_row_ptrs, d_cols, d_ldu_mapping;  // "<field>_local_*"
    ogl::DevBuf<double> d_vals;                              // "<field>_matrix" values
    ogl::DevBuf<double> d_source;                            // unsorted [upper|lower|diag|iface]
    ogl::DevBuf<int32_t> d_diag_pos;  // position of each row's first diagonal entry (scalar Jacobi)
    // matrixFormat Ell: slot-major copy of the local matrix (built on demand, refreshed from vals)
    ogl::DevBuf<int32_t> d_ell_cols, d_ell_map;
    ogl::DevBuf<double> d_ell_vals;
    int32_t ell_width = 0;
    int64_t ell_stride = 0;
    bool ell_ready = false, ell_values_stale = true;
    int build_ell();
    // half storage of a symmetric matrix on a banded pattern (SymLayout, host_matrix.hpp): what the Coo/Csr
    // formats run on when cfg.compress_indices and cfg.symmetric_half are set, the lduMatrix has no `lower`
    // and no same-rank (cyclic) interface, the device copy keeps the caller's numbering and the pattern
    // qualifies.  Takes the place of the compressed copy below (sym_state as sell_state).
    ogl::DevBuf<uint8_t> d_sym_mask;
    ogl::DevBuf<int32_t> d_sym_map, d_sym_order;  // (order: band_block_order, may be empty)
    bool band_order_off = std::getenv("OGL_NO_BAND_ORDER") != nullptr;  // (A/B switch for measurements)
    ogl::DevBuf<double> d_sym_planes;
    int32_t sym_nd = 0, sym_d[4] = {0, 0, 0, 0};
    int sym_state = 0;
    bool sym_values_stale = true;
    int build_sym(const ogl::SymLayout &L);
    int finish_sym(int nd, const int32_t *d);
    // device set-up (setup_kernels.hip)
    int build_pattern_on_device(const ogl_ldu_view &ldu, ogl::HostPattern &np, bool *built);
    int build_sym_on_device(const ogl::HostPattern &np, ogl::SymDistances *sd_out, bool *done);
    int download_local_pattern(ogl::HostPattern &hp);
    // reverse Cuthill-McKee of the device pattern (same order as rcm_order); new_id stays empty when the graph
    // is not one for a level-synchronous search (very many components or levels): the host does it then
    int rcm_on_device(const ogl::HostPattern &hp, std::vector<ogl_label> &new_id);
    // device pattern rewritten into the numbering new_id (and downloaded into hp for the host-side layout code)
    int renumber_on_device(ogl::HostPattern &hp, const std::vector<ogl_label> &new_id);
    // the Hilbert-curve candidate of the numbering policy on the device: keys + radix sort, and the far-entry count
    int curve_on_device(ogl_label n, const double *centres, std::vector<ogl_label> &new_id);
    int curve_far_on_device(const ogl::HostPattern &hp, const std::vector<ogl_label> &new_id,
                            const std::vector<ogl_label> &old_of, int64_t &far);
    ogl::DevSym sym() const;
    bool use_sym() const
    {
        return cfg.matrix_format != OGL_FORMAT_ELL && cfg.compress_indices && sym_state == 1 && !sym_values_stale;
    }
    // half storage with per-chunk distances and explicit exceptions (SymxLayout, host_matrix.hpp): symmetric
    // matrices that are banded only locally (multi-block meshes, refinement shells) -- tried when the global
    // half storage above does not qualify, before the compressed full-storage copy below
    ogl::DevBuf<ogl::SymxChunk> d_symx_chunks, d_symx_chunks_general;  // (dispatch order: lean kernel's list, general one's)
    ogl::DevBuf<uint8_t> d_symx_mask;
    ogl::DevBuf<int32_t> d_symx_map, d_symx_ex_rowptr, d_symx_ex_cols, d_symx_ex_map, d_symx_ex_lrow;
    ogl::DevBuf<double> d_symx_planes, d_symx_ex_vals;
    int symx_state = 0;  // 0 not tried, 1 built, -1 not worth it
    bool symx_fast = false;
    // built next to the compressed full-storage copy and still to be timed against it (once per pattern, systems
    // of >= SPMV_TUNE_MIN_ROWS rows with compress_indices 1): the faster one stays
    bool symx_tune_pending = false;
    int tune_symx();
    bool symx_values_stale = true;
    double symx_bytes = 0.0;
    int build_symx();
    ogl::DevSymx symx() const;
    bool use_symx() const
    {
        return cfg.matrix_format != OGL_FORMAT_ELL && cfg.compress_indices && symx_state == 1 && !symx_values_stale;
    }
    // index-compressed chunked ELL copy (SellChunk, common.hpp): what the Coo/Csr formats run on when
    // cfg.compress_indices is set and the pattern qualifies.  sell_state: 0 = not tried for this
    // pattern, 1 = built, -1 = pattern does not qualify (CSR-stream kernel runs)
    ogl::DevBuf<ogl::SellChunk> d_sell_chunks;
    ogl::DevBuf<int32_t> d_sell_dict, d_sell_map;
    ogl::DevBuf<uint8_t> d_sell_codes;
    ogl::DevBuf<double> d_sell_vals;
    int64_t sell_slots = 0;
    double sell_bytes = 0.0;  // bytes one SpMV reads of the compressed copy (decides the cache policy of its loads)
    int sell_state = 0;
    // spill of the compressed copy: tails of the rows longer than their chunk's cap (SellLayout)
    ogl::DevBuf<int32_t> d_spill_rows, d_spill_ptrs, d_spill_cols, d_spill_map, d_spill_chunks;
    ogl::DevBuf<double> d_spill_vals;
    int32_t n_spill_rows = 0, n_spill = 0;   // (d_spill_chunks holds the per-chunk ranges of d_spill_rows)
    bool sell_values_stale = true;
    // Patterns with irregular chunks (16-bit delta / 32-bit column codes: unstructured meshes) are timed on
    // both kernels once per pattern (tune_spmv_layout): the compressed layout moves fewer bytes, but its
    // slot-major gather -- one entry of 64 different rows per instruction -- only pays where neighbouring
    // rows have neighbouring columns; on a polyhedral mesh the CSR-stream kernel's row-major gather wins.
    // sell_tuned: 0 = not measured, 1 = compressed layout is faster (or cfg.compress_indices == 2: forced),
    // -1 = the CSR-stream kernel is.  The results are bit-identical either way.
    bool sell_irregular = false;
    int sell_tuned = 0;
    bool layout_tuned = false;  // tune_spmv_layout has run for this pattern
    int tune_spmv_layout();
    // packed columns for the CSR-stream kernel (Stream21Chunk, common.hpp): built for irregular patterns of
    // >= SPMV_TUNE_MIN_ROWS rows when compress_indices is set; s21_use: the in-loop CSR-stream SpMV reads them
    // (it won the one-off timing, or compress_indices = force and the chunked ELL does not qualify)
    ogl::DevBuf<ogl::Stream21Chunk> d_s21_chunks;
    ogl::DevBuf<uint4> d_s21_codes;
    ogl::DevBuf<int32_t> d_s21_far_idx, d_s21_far_col;  // the chunks' entries outside their 2^21-column windows
    int s21_state = 0;  // 0 not tried for this pattern, 1 built, -1 a chunk's columns span 2^21 or more
    bool s21_use = false;
    int build_stream21();
    // `pre` != nullptr: the layout choose_numbering already derived for this pattern
    // (`pre_qualifies` tells whether it is usable)
    int build_sell(ogl::SellLayout *pre = nullptr, bool pre_qualifies = false);
    ogl::DevSell sell() const;
    // renumbering (config `renumber`): pat.new_id on the device + a staging vector, so that host
    // vectors cross the boundary in the caller's cell order
    ogl::DevBuf<int32_t> d_new_id, d_old_of;
    ogl::DevBuf<double> d_perm_tmp;
    int pat_renumber_mode = -1;  // cfg.renumber / layout eligibility the pattern was built under
    bool pat_try_sell = false, pat_try_sym = false;
    ogl::DevBuf<double> d_flag;  // 2 doubles: cross-rank agreement on pattern rebuilds
    // host -> device / device -> host of one row vector, through the renumbering when there is one
    int upload_rows(double *dst, const double *src);
    int download_rows(double *dst, const double *src);
    // peer-put halo exchange (PeerHalo, kernels.hpp): agreed per sparsity pattern by all ranks
    struct PeerNeighbour {
        size_t block = 0;   // the neighbour's arena block (words from its arena start)
        int32_t n_neigh = 0, my_index = 0, n_halo = 0, my_seg = 0;  // its layout, this rank's place
    };
    bool peer_halo = false;
    size_t peer_block = 0;  // this solver's arena block
    size_t peer_block_words = 0;
    std::vector<PeerNeighbour> peer_nb;
    uint32_t halo_seq = 0;
    ogl::DevBuf<int32_t> d_boundary_chunk_ptr;  // ranges of boundary_rows per boundary chunk
    // the same ranges for EVERY chunk (HaloFused: the local SpMV kernel adds the non-local part itself) and the
    // send list grouped by the chunk of its rows (HaloPutFused: step_1x puts the halo values it has just formed)
    ogl::DevBuf<int32_t> d_chunk_bptr, d_chunk_sptr, d_send_pos;
    int32_t n_put_chunks = 0;
    ogl::PeerHalo cur_halo{};   // arguments of the SpMV whose halo values a producer kernel has already put
    ogl::HaloPutFused begin_halo_put();
    ogl::HaloFused halo_fused_args(const ogl::PeerHalo &ph) const;
    ogl::DevBuf<unsigned> d_ticket;             // last-workgroup ticket of k_pack_put_signal
    // a full batch of single-rank GKOCG turns captured as a hipGraph (run_krylov)
    hipGraphExec_t cg_graph = nullptr;
    uint64_t cg_graph_key = 0;  // hash of everything the captured launches bake in (launch_key.hpp)
    void drop_cg_graph();  // (every pattern / layout rebuild)
    int setup_peer_halo();
    ogl::PeerHalo peer_halo_args(uint32_t seq) const;
    double *peer_recv(uint32_t seq) const;
    bool use_sell() const
    {
        return cfg.matrix_format != OGL_FORMAT_ELL && cfg.compress_indices && sell_state == 1 &&
               !sell_values_stale && sell_tuned >= 0;
    }
    ogl::DevEll ell() const;
    // halo part
    std::vector<int32_t> boundary_rows, boundary_ptrs;
    ogl::DevBuf<int32_t> d_boundary_rows, d_boundary_ptrs, d_nl_cols, d_send_idxs;
    ogl::DevBuf<int32_t> d_boundary_chunks;  // chunks (of CHUNK_ROWS rows) that hold boundary rows
    int32_t n_boundary_chunks = 0;
    ogl::DevBuf<double> d_nl_vals, d_send, d_recv;
    std::vector<double> h_nl_vals;
    std::vector<int> neighbours, counts;

    // ---- vectors: "<field>_rhs", "<field>_solution" + Krylov work vectors ----
    ogl::DevBuf<double> d_x, d_b, d_r, d_p, d_q, d_w, d_inv_diag;
    ogl::DevBuf<double> d_p2;  // second p buffer of the 2-launch turn (k_cg_turn_sym)
    ogl::DevBuf<double> d_pring[ogl::P_RING_MAX - 2];  // further p buffers of the leader turn's ring (PRing, deferX)
    ogl::DevBuf<double> d_p_halo;  // multi-rank merged turn: old / new p at the halo columns
    ogl::DevBuf<double> d_bj_tmp0, d_bj_tmp1;  // block Jacobi through a permutation, staged apply: in / out in the caller's order
    ogl::DevBuf<double> d_v, d_s, d_t, d_y, d_z, d_rr;  // BiCGStab
    ogl::DevBuf<double> d_V, d_gm;                      // GMRES: Krylov bases, dense state
    ogl::DevBuf<double> d_isai_tmp;                     // ISAI(spd): W r before W^T
    ogl::DevBuf<double> d_part0, d_part1, d_part2;  // (part2: beta partials of the fused-finaliser turn)
    int64_t band_order_rows = 0, sell_band_rows = 0;
    bool source_diag_valid = false;  // d_source's diagonal segment is the diagonal of d_vals (set by the coefficient update)
    int64_t csr_band_rows = 0;   // band of the device CSR arrays (csr_band), valid for pattern csr_band_pat
    uint64_t csr_band_pat = 0;
    int csr_band(int64_t *band);
    ogl::DevBuf<int32_t> d_band_order;  // band-aware workgroup order of the CSR-stream / compressed kernels (property spmvBandRows)
    ogl::DevBuf<double> d_part3, d_part4, d_part5;  // (the folded GKOBiCGStab turn: sum|s|, s.t, t.t)
    ogl::DevBuf<ogl::DevScalars> d_scal;
    ogl::DevBuf<double> d_history;
    ogl::DevScalars *h_scal = nullptr;  // pinned, 2 slots
    hipEvent_t poll_ev[2] = {nullptr, nullptr};
    unsigned long long *lead_box = nullptr;  // LeadBox of the leader finalisation (fine-grained, LEAD_BOX_WORDS words)
    hipEvent_t chk_ev[2] = {nullptr, nullptr};  // brackets one evaluated criterion check per solve (time_for_res_norm_eval)
    bool x_resident = false, b_resident = false;
    ogl::PrecondData own_precond;              // regenerated-for-this-solve preconditioner
    const ogl::PrecondData *precond_data = nullptr;  // the one in use (own or the registry's)
    const double *precond = nullptr;  // scalar Jacobi: inverse diagonal (fused path); else nullptr

    // ---- per-field properties (common/common.C:75-146) ----
    std::map<std::string, double> props;

    // ---- last solve ----
    std::vector<double> history;
    double t_update_matrix_ms = 0;
    // profile_kernels
    std::vector<hipEvent_t> prof_ev;

    ~ogl_solver();

    int set_matrix(const ogl_ldu_view &ldu);
    int solve(const double *source, double *psi, ogl_perf *perf);
    int apply_resident(ogl_perf *perf);
    int upload_vec(ogl::DevBuf<double> &dst, const double *src);
    int ensure_vectors();
    int init_preconditioner();
    int generate_preconditioner(ogl::PrecondData &P);
    // out = M^-1 in for the block-Jacobi / ISAI / GISAI kinds; dot_part != nullptr: also the per-chunk
    // partials of sum_i in_i * out_i (CG's rho), out of the same kernel
    void apply_preconditioner(const double *in, double *out, const ogl::DevScalars *gate,
                              double *dot_part = nullptr);
    // prepacked: the kernel that produced x has put the halo values already (begin_halo_put)
    int dist_spmv(int mode, const double *x, const double *b, double *y, const ogl::SpmvDots &dots,
                  const ogl::DevScalars *gate, bool prepacked = false);
    int finalize(int phase, ogl::FinArgs &a);
    int run_cg(ogl_perf *perf);
    int run_bicgstab(ogl_perf *perf);
    int run_krylov(ogl_perf *perf);
    // run_krylov = plan -> prepare -> loop -> finish; one function per solver x turn shape (solver.cpp)
    struct KrylovRun;
    int krylov_plan(KrylovRun &k);
    int krylov_prepare(KrylovRun &k);
    int krylov_loop(KrylovRun &k);
    int krylov_enqueue(KrylovRun &k, int count);
    int krylov_finish(KrylovRun &k, ogl_perf *perf);
    int gmres_restart(KrylovRun &k, const ogl::DevScalars *gate);
    int gmres_update_x(KrylovRun &k, int cols, const ogl::DevScalars *gate);
    int turn_gmres(KrylovRun &k, int enq, int pe);
    int turn_cg_generic(KrylovRun &k, int enq, int pe);
    int turn_cg_generic_led(KrylovRun &k, int enq, int pe);
    int turn_cg_two_launch(KrylovRun &k, int enq, int pe);
    int turn_cg_three_launch(KrylovRun &k, int enq, int pe);
    int turn_cg_merged(KrylovRun &k, int enq, int pe);
    int turn_cg_five_launch(KrylovRun &k, int enq, int pe);
    int turn_bicg_folded(KrylovRun &k, int enq, int pe);
    int turn_bicg(KrylovRun &k, int enq, int pe);
    int time_spmv(int repeats, double *avg_ms);
    ogl::DevCsr csr() const;
    ogl::DevHalo halo() const;
    double prop(const std::string &key, double dflt) const;
    // ogl_solver_set_matrix_like: the solver whose device copy of upper / lower this set_matrix may take (only during
    // that call), and what THIS solver's copy was uploaded from (host arrays + sampled checksum) for a later taker
    const ogl_solver *share_from = nullptr;
    const double *offdiag_upper = nullptr, *offdiag_lower = nullptr;
    uint64_t offdiag_sum = 0;
    bool offdiag_valid = false;
    // the addressing arrays of the last set_matrix (a sibling on the same arrays need not hash them again)
    const ogl_label *seen_lower_addr = nullptr, *seen_upper_addr = nullptr;
    ogl_label seen_faces = -1;
    struct SeenIface {  // (everything addressing_fingerprint mixes in per interface, by identity / value)
        const ogl_label *face_cells;
        ogl_label size, kind, neighb_proc, neighb_patch;
    };
    std::vector<SeenIface> seen_iface_cells;
    bool saw_addressing(const ogl_ldu_view &ldu) const;
    bool peer_safe_wait() const;
    double stream_above_bytes() const;
    double turn_extra_bytes() const;
    int32_t xcd_group() const;
    int32_t pat_xcd_group = 0;  // chosen per pattern (0 = the built-in group of 4 chunks)
};
