// host_matrix.cpp -- see host_matrix.hpp.  Citations: reference HostMatrix/*.C.
#include "host_matrix.hpp"

#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstddef>
#include <cstdlib>
#include <cstring>
#include <map>
#include <thread>
#include <tuple>

namespace ogl {

namespace {

// threads for the once-per-pattern host work that splits over chunks / rows (OGL_SETUP_THREADS; default: the
// hardware's, at most 16)
int setup_threads()
{
    static const int n = [] {
        const char *e = std::getenv("OGL_SETUP_THREADS");
        const int hw = (int)std::thread::hardware_concurrency();
        return std::max(1, std::min(64, e ? atoi(e) : std::min(16, hw > 0 ? hw : 4)));
    }();
    return n;
}

// fn(begin, end) over [0, n) in contiguous parts, one per thread
template <class F>
void parallel_ranges(int64_t n, int64_t min_per_thread, F fn)
{
    const int nt = (int)std::max<int64_t>(1, std::min<int64_t>(setup_threads(), n / std::max<int64_t>(1, min_per_thread)));
    if (nt <= 1) {
        fn((int64_t)0, n);
        return;
    }
    const int64_t part = (n + nt - 1) / nt;
    std::vector<std::thread> helpers;
    for (int t = 1; t < nt; ++t)
        if ((int64_t)t * part < n) helpers.emplace_back(fn, (int64_t)t * part, std::min<int64_t>(n, (int64_t)(t + 1) * part));
    fn((int64_t)0, std::min(part, n));
    for (auto &h : helpers) h.join();
}

// Order one row segment by column; entries arrive in face order, so equal columns keep it.
inline void sort_segment(ogl_label *cols, ogl_label *perm, ogl_label len)
{
    for (ogl_label i = 1; i < len; ++i) {
        const ogl_label c = cols[i], p = perm[i];
        ogl_label j = i;
        while (j > 0 && cols[j - 1] > c) {
            cols[j] = cols[j - 1];
            perm[j] = perm[j - 1];
            --j;
        }
        cols[j] = c;
        perm[j] = p;
    }
}

}  // namespace

// HostMatrixFreeFunctions.C:105-201.  The reference sorts two face lists by (row, col) and then
// interleaves them row by row as [lower entries | diagonal | upper entries].  Here every row's
// three segments are sized by a counting pass, filled in face order and ordered by column, which
// yields the same arrays in O(F) for OpenFOAM's upper-triangular face order.
void init_local_sparsity(ogl_label nrows, ogl_label upper_nnz, bool is_symmetric,
                         const ogl_label *upper, const ogl_label *lower, ogl_label *rows,
                         ogl_label *cols, ogl_label *permute)
{
    const ogl_label after_neighbours = is_symmetric ? upper_nnz : 2 * upper_nnz;  // :116
    std::vector<ogl_label> n_low(nrows, 0), n_up(nrows, 0);
    for (ogl_label f = 0; f < upper_nnz; ++f) {
        ++n_up[lower[f]];   // upper-triangle entry lives in row lower[f]   (:123-124)
        ++n_low[upper[f]];  // its transpose lives in row upper[f]          (:139-140)
    }
    std::vector<int64_t> seg_low(nrows), seg_up(nrows);
    int64_t off = 0;
    for (ogl_label r = 0; r < nrows; ++r) {
        seg_low[r] = off;
        off += n_low[r];
        rows[off] = r;  // diagonal (:179-182)
        cols[off] = r;
        permute[off] = after_neighbours + r;
        ++off;
        seg_up[r] = off;
        off += n_up[r];
    }
    std::vector<int64_t> fill_low(seg_low), fill_up(seg_up);
    for (ogl_label f = 0; f < upper_nnz; ++f) {
        int64_t e = fill_up[lower[f]]++;
        rows[e] = lower[f];
        cols[e] = upper[f];
        permute[e] = f;  // :188
        e = fill_low[upper[f]]++;
        rows[e] = upper[f];
        cols[e] = lower[f];
        permute[e] = is_symmetric ? f : upper_nnz + f;  // :164-165
    }
    for (ogl_label r = 0; r < nrows; ++r) {
        sort_segment(cols + seg_low[r], permute + seg_low[r], n_low[r]);
        sort_segment(cols + seg_up[r], permute + seg_up[r], n_up[r]);
    }
}

void collect_interface_coeffs(const ogl_ldu_view &ldu, bool local, ogl_scalar *out)
{
    int64_t k = 0;
    for (ogl_label i = 0; i < ldu.n_interfaces; ++i) {
        const ogl_interface &itf = ldu.interfaces[i];
        const bool is_proc = itf.kind == OGL_IFACE_PROCESSOR;
        if (local ? is_proc : !is_proc) continue;
        for (ogl_label f = 0; f < itf.size; ++f) out[k++] = itf.bou_coeffs[f] * -1.0;  // :204
    }
}

void find_jacobi_blocks(const HostPattern &p, ogl_label max_block_size,
                        std::vector<ogl_label> &block_ptrs, std::vector<ogl_label> &row_block, bool caller_numbering)
{
    const ogl_label n = p.n_rows;
    block_ptrs.assign(1, 0);
    row_block.assign(n, 0);
    if (n == 0) return;
    // A renumbered pattern: the blocks are formed on the CALLER's numbering, position i there being row
    // p.new_id[i] here (two rows have the same column set in one numbering exactly when they have it in the other,
    // and both are stored in ascending order of THIS numbering)
    const bool rn = p.renumbered() && caller_numbering;
    auto same_pattern = [&](ogl_label a, ogl_label b) {
        if (rn) {
            a = p.new_id[a];
            b = p.new_id[b];
        }
        const ogl_label la = p.row_ptrs[a + 1] - p.row_ptrs[a], lb = p.row_ptrs[b + 1] - p.row_ptrs[b];
        if (la != lb) return false;
        return std::equal(p.cols.begin() + p.row_ptrs[a], p.cols.begin() + p.row_ptrs[a + 1],
                          p.cols.begin() + p.row_ptrs[b]);
    };
    std::vector<ogl_label> nat{0};  // natural blocks
    ogl_label cur = 1;
    for (ogl_label i = 1; i < n; ++i) {
        if (same_pattern(i - 1, i) && cur < max_block_size) {
            ++cur;
        } else {
            nat.push_back(i);
            cur = 1;
        }
    }
    nat.push_back(n);
    cur = nat[1] - nat[0];  // agglomerate
    for (size_t i = 1; i + 1 < nat.size(); ++i) {
        const ogl_label bs = nat[i + 1] - nat[i];
        if (cur + bs <= max_block_size) {
            cur += bs;
        } else {
            block_ptrs.push_back(nat[i]);
            cur = bs;
        }
    }
    block_ptrs.push_back(n);
    for (size_t b = 0; b + 1 < block_ptrs.size(); ++b)
        for (ogl_label r = block_ptrs[b]; r < block_ptrs[b + 1]; ++r) row_block[r] = (ogl_label)b;
}

// Fingerprint of the WHOLE addressing (every face, every interface cell): the reference keeps the
// pattern of a field for ever once built (HostMatrix.C:79-87); this build also rebuilds it when the
// addressing changed under the same counts.  240 MB at 30 M faces: four independent multiply-mix
// lanes per thread and a few threads bring that to a few milliseconds (a byte-wise FNV would take
// longer than the coefficient upload it guards).
namespace {

inline uint64_t mix64(uint64_t h, uint64_t v)
{
    h = (h ^ v) * 0x9E3779B97F4A7C15ull;
    return h ^ (h >> 29);
}

uint64_t hash_labels_serial(const ogl_label *a, int64_t n)
{
    uint64_t h0 = 0x243F6A8885A308D3ull, h1 = 0x13198A2E03707344ull, h2 = 0xA4093822299F31D0ull,
             h3 = 0x082EFA98EC4E6C89ull;
    int64_t i = 0;
    for (; i + 8 <= n; i += 8) {
        uint64_t w[4];
        std::memcpy(w, a + i, sizeof(w));
        h0 = mix64(h0, w[0]);
        h1 = mix64(h1, w[1]);
        h2 = mix64(h2, w[2]);
        h3 = mix64(h3, w[3]);
    }
    for (; i < n; ++i) h0 = mix64(h0, (uint64_t)(uint32_t)a[i]);
    return mix64(mix64(mix64(mix64(h0, h1), h2), h3), (uint64_t)n);
}

uint64_t hash_labels(const ogl_label *a, int64_t n)
{
    if (!a || n <= 0) return 0x452821E638D01377ull;
    static const int n_threads = [] {
        const char *e = std::getenv("OGL_STAGE_THREADS");
        return std::max(1, std::min(16, e ? atoi(e) : 4));
    }();
    if (n_threads == 1 || n < (int64_t(1) << 20)) return hash_labels_serial(a, n);
    // fixed split (multiples of 8 labels) so the value does not depend on scheduling
    const int64_t part = ((n + n_threads - 1) / n_threads + 7) / 8 * 8;
    std::vector<uint64_t> hs((size_t)n_threads, 0);
    std::vector<std::thread> helpers;
    for (int t = 1; t < n_threads; ++t) {
        const int64_t off = (int64_t)t * part;
        if (off >= n) break;
        helpers.emplace_back([&, t, off] { hs[(size_t)t] = hash_labels_serial(a + off, std::min(part, n - off)); });
    }
    hs[0] = hash_labels_serial(a, std::min(part, n));
    for (auto &h : helpers) h.join();
    uint64_t h = 0xBE5466CF34E90C6Cull;
    for (uint64_t v : hs) h = mix64(h, v);
    return mix64(h, (uint64_t)n_threads);
}

}  // namespace

uint64_t addressing_fingerprint(const ogl_ldu_view &ldu)
{
    uint64_t h = 1469598103934665603ull;
    h = mix64(h, hash_labels(ldu.lower_addr, ldu.n_faces));
    h = mix64(h, hash_labels(ldu.upper_addr, ldu.n_faces));
    for (ogl_label i = 0; i < ldu.n_interfaces; ++i) {
        const ogl_interface &itf = ldu.interfaces[i];
        h = mix64(h, (uint64_t)(uint32_t)itf.kind);
        h = mix64(h, (uint64_t)(uint32_t)(itf.kind == OGL_IFACE_PROCESSOR ? itf.neighb_proc : itf.neighb_patch));
        h = mix64(h, hash_labels(itf.face_cells, itf.size));
    }
    return h;
}

bool same_counts(const ogl_ldu_view &ldu, const HostPattern &p)
{
    if (ldu.n_cells != p.n_rows || ldu.n_faces != p.upper_nnz) return false;
    if ((ldu.lower == nullptr) != p.symmetric) return false;
    int64_t loc = 0, nl = 0;
    for (ogl_label i = 0; i < ldu.n_interfaces; ++i)
        (ldu.interfaces[i].kind == OGL_IFACE_PROCESSOR ? nl : loc) += ldu.interfaces[i].size;
    return loc == p.local_iface_nnz && nl == p.non_local_nnz;
}

bool same_shape(const ogl_ldu_view &ldu, const HostPattern &p)
{
    return same_counts(ldu, p) && addressing_fingerprint(ldu) == p.fingerprint;
}

int build_host_pattern_meta(const ogl_ldu_view &ldu, HostPattern &p, bool check_faces)
{
    if (ldu.n_cells < 0 || ldu.n_faces < 0 || ldu.n_interfaces < 0)
        return fail(OGL_ERR_INVALID, "negative size in ldu view");
    if (ldu.n_cells > 0 && !ldu.diag) return fail(OGL_ERR_INVALID, "diag is NULL");
    if (ldu.n_faces > 0 && (!ldu.lower_addr || !ldu.upper_addr || !ldu.upper))
        return fail(OGL_ERR_INVALID, "face addressing / upper is NULL");
    if (ldu.n_interfaces > 0 && !ldu.interfaces) return fail(OGL_ERR_INVALID, "interfaces is NULL");
    const ogl_label N = ldu.n_cells, F = ldu.n_faces;
    if (check_faces)  // (the device build checks the faces while it counts them)
        for (ogl_label f = 0; f < F; ++f)
            if (ldu.lower_addr[f] < 0 || ldu.lower_addr[f] >= N || ldu.upper_addr[f] < 0 ||
                ldu.upper_addr[f] >= N)
                return fail(OGL_ERR_INVALID, "face %d addresses a cell outside [0,%d)", f, N);

    p = HostPattern{};
    p.n_rows = N;
    p.upper_nnz = F;
    p.symmetric = (ldu.lower == nullptr);  // matrix.symmetric()  HostMatrix.C:473
    int64_t loc = 0, nl = 0;
    for (ogl_label i = 0; i < ldu.n_interfaces; ++i) {  // count_interface_nnz  :159-178
        const ogl_interface &itf = ldu.interfaces[i];
        if (itf.size < 0 || (itf.size > 0 && (!itf.face_cells || !itf.bou_coeffs)))
            return fail(OGL_ERR_INVALID, "interface %d: bad size or NULL arrays", i);
        for (ogl_label f = 0; f < itf.size; ++f)
            if (itf.face_cells[f] < 0 || itf.face_cells[f] >= N)
                return fail(OGL_ERR_INVALID, "interface %d: faceCell outside [0,%d)", i, N);
        if (itf.kind == OGL_IFACE_PROCESSOR) {
            if (itf.neighb_proc < 0)
                return fail(OGL_ERR_INVALID, "interface %d: negative neighbProcNo", i);
            nl += itf.size;
        } else if (itf.kind == OGL_IFACE_CYCLIC) {
            if (itf.neighb_patch < 0 || itf.neighb_patch >= ldu.n_interfaces ||
                ldu.interfaces[itf.neighb_patch].size != itf.size)
                return fail(OGL_ERR_INVALID, "interface %d: cyclic neighbour patch mismatch", i);
            loc += itf.size;
        } else {
            // cyclicAMI / cyclicACMI abort in the reference (HostMatrix.C:339-341, :367-369);
            // other coupled kinds would corrupt its pattern.
            return fail(OGL_ERR_UNSUPPORTED, "interface %d: unsupported coupled patch kind %d", i,
                        itf.kind);
        }
    }
    const int64_t total = (int64_t)N + 2 * (int64_t)F + loc;
    if (total > INT32_MAX - NNZ_PAD || nl > INT32_MAX - NNZ_PAD)
        return fail(OGL_ERR_INVALID, "matrix too large for 32-bit labels");
    p.local_iface_nnz = (ogl_label)loc;
    p.non_local_nnz = (ogl_label)nl;
    p.local_nnz = (ogl_label)total;
    p.local_on_host = false;

    // ---- non-local pattern (HostMatrix.C:412-466) ----
    // row = faceCell, col = ldu_mapping = running index over processor-interface faces in
    // interface order; ordered by row.  The reference's std::sort is keyed on the row only and is
    // not stable, so the order of entries sharing a row is unspecified there; a stable sort keeps
    // them in interface order.
    p.nl_rows.resize(nl);
    p.nl_cols.resize(nl);
    p.nl_ldu_mapping.resize(nl);
    {
        std::vector<ogl_label> row_of(nl), order(nl);
        ogl_label k = 0;
        for (ogl_label i = 0; i < ldu.n_interfaces; ++i) {
            const ogl_interface &itf = ldu.interfaces[i];
            if (itf.kind != OGL_IFACE_PROCESSOR) continue;
            for (ogl_label f = 0; f < itf.size; ++f) {
                row_of[k] = itf.face_cells[f];
                order[k] = k;
                ++k;
            }
        }
        std::stable_sort(order.begin(), order.end(),
                         [&](ogl_label a, ogl_label b) { return row_of[a] < row_of[b]; });
        for (ogl_label e = 0; e < (ogl_label)nl; ++e) {
            p.nl_rows[e] = row_of[order[e]];
            p.nl_cols[e] = order[e];
            p.nl_ldu_mapping[e] = order[e];
        }
    }

    // ---- communication pattern (HostMatrix.C:251-306): std::map keyed by neighbour rank ----
    {
        std::map<ogl_label, std::vector<ogl_label>> by_rank;
        for (ogl_label i = 0; i < ldu.n_interfaces; ++i) {
            const ogl_interface &itf = ldu.interfaces[i];
            if (itf.kind != OGL_IFACE_PROCESSOR) continue;
            auto &v = by_rank[itf.neighb_proc];
            v.insert(v.end(), itf.face_cells, itf.face_cells + itf.size);
        }
        for (const auto &[rank, cells] : by_rank) {
            p.target_ids.push_back(rank);
            p.target_sizes.push_back((ogl_label)cells.size());
            p.send_idxs.insert(p.send_idxs.end(), cells.begin(), cells.end());
        }
    }
    return OGL_OK;
}

// collect_local_interface_indices (HostMatrix.C:385-410): (row, col) of every same-rank (cyclic) interface
// face in interface order; the column is the face cell of the neighbour patch (:324-327)
void local_interface_entries(const ogl_ldu_view &ldu, std::vector<ogl_label> &rows, std::vector<ogl_label> &cols)
{
    rows.clear();
    cols.clear();
    for (ogl_label i = 0; i < ldu.n_interfaces; ++i) {
        const ogl_interface &itf = ldu.interfaces[i];
        if (itf.kind != OGL_IFACE_CYCLIC) continue;
        const ogl_label *nbr = ldu.interfaces[itf.neighb_patch].face_cells;
        for (ogl_label f = 0; f < itf.size; ++f) {
            rows.push_back(itf.face_cells[f]);
            cols.push_back(nbr[f]);
        }
    }
}

// ---- local pattern (init_local_sparsity_pattern, HostMatrix.C:468-589) on the host ----
void build_local_pattern(const ogl_ldu_view &ldu, HostPattern &p)
{
    const ogl_label N = p.n_rows, F = p.upper_nnz;
    const int64_t total = p.local_nnz, loc = p.local_iface_nnz;
    p.rows.resize(total);
    p.cols.resize(total);
    p.ldu_mapping.resize(total);
    init_local_sparsity(N, F, p.symmetric, ldu.upper_addr, ldu.lower_addr, p.rows.data(),
                        p.cols.data(), p.ldu_mapping.data());
    if (loc) {
        std::vector<ogl_label> ir, ic;
        local_interface_entries(ldu, ir, ic);
        std::vector<std::tuple<ogl_label, ogl_label, ogl_label>> ifc;  // (row, col, idx)
        ifc.reserve(loc);
        for (ogl_label k = 0; k < (ogl_label)ir.size(); ++k) ifc.emplace_back(ir[(size_t)k], ic[(size_t)k], k);
        std::stable_sort(ifc.begin(), ifc.end(), [](const auto &a, const auto &b) {  // :510-515
            return std::tie(std::get<0>(a), std::get<1>(a)) < std::tie(std::get<0>(b), std::get<1>(b));
        });
        const int64_t base_nnz = (int64_t)N + 2 * (int64_t)F;
        const ogl_label iface_base = p.diag_start() + N;  // after_neighbours + nrows_  (:574)
        std::vector<ogl_label> r2(total), c2(total), m2(total);
        int64_t cur = 0, tot = 0;
        for (const auto &[r, c, idx] : ifc) {  // :539-576
            while (cur < base_nnz && (p.rows[cur] < r || (p.rows[cur] == r && p.cols[cur] <= c))) {
                r2[tot] = p.rows[cur];
                c2[tot] = p.cols[cur];
                m2[tot] = p.ldu_mapping[cur];
                ++cur;
                ++tot;
            }
            r2[tot] = r;
            c2[tot] = c;
            m2[tot] = iface_base + idx;
            ++tot;
        }
        for (; cur < base_nnz; ++cur, ++tot) {  // :580-585
            r2[tot] = p.rows[cur];
            c2[tot] = p.cols[cur];
            m2[tot] = p.ldu_mapping[cur];
        }
        p.rows.swap(r2);
        p.cols.swap(c2);
        p.ldu_mapping.swap(m2);
    }
    p.row_ptrs.assign((size_t)N + 1, 0);
    for (int64_t e = 0; e < total; ++e) ++p.row_ptrs[p.rows[e] + 1];
    for (ogl_label r = 0; r < N; ++r) p.row_ptrs[r + 1] += p.row_ptrs[r];
    p.local_on_host = true;
}

int build_host_pattern(const ogl_ldu_view &ldu, HostPattern &p)
{
    if (int rc = build_host_pattern_meta(ldu, p, /*check_faces*/ true)) return rc;
    build_local_pattern(ldu, p);
    p.fingerprint = addressing_fingerprint(ldu);
    return OGL_OK;
}

bool isai_pattern(const HostPattern &p, bool spd, int power, int max_row, std::vector<ogl_label> &w_row_ptrs,
                  std::vector<ogl_label> &w_cols, ogl_label &first_wide_row, bool caller_numbering)
{
    const ogl_label N = p.n_rows;
    w_row_ptrs.assign((size_t)N + 1, 0);
    w_cols.clear();
    w_cols.reserve((size_t)p.local_nnz);
    std::vector<ogl_label> seen((size_t)N, -1), row, next;
    // spd: tril(A) of the matrix OpenFOAM hands over (Preconditioner.H:225-241) -- on a renumbered pattern the
    // triangle is taken by the caller's index, i.e. W gets the pattern P tril(A) P^T
    const bool rn = p.renumbered() && caller_numbering;
    auto in_s = [&](ogl_label r, ogl_label c) {
        return !spd || (rn ? p.old_of[(size_t)c] <= p.old_of[(size_t)r] : c <= r);
    };
    for (ogl_label i = 0; i < N; ++i) {
        row.clear();
        for (ogl_label k = p.row_ptrs[i]; k < p.row_ptrs[i + 1]; ++k) {
            const ogl_label c = p.cols[k];
            if (in_s(i, c) && seen[(size_t)c] != i) {  // (cyclic patches can repeat a column)
                seen[(size_t)c] = i;
                row.push_back(c);
            }
        }
        for (int pw = 1; pw < power && (int)row.size() <= max_row; ++pw) {
            next = row;
            for (ogl_label j : row)
                for (ogl_label k = p.row_ptrs[j]; k < p.row_ptrs[j + 1]; ++k) {
                    const ogl_label c = p.cols[k];
                    if (in_s(j, c) && seen[(size_t)c] != i) {
                        seen[(size_t)c] = i;
                        next.push_back(c);
                    }
                }
            row.swap(next);
        }
        if ((int)row.size() > max_row) {
            first_wide_row = i;
            return false;
        }
        std::sort(row.begin(), row.end());
        w_cols.insert(w_cols.end(), row.begin(), row.end());
        w_row_ptrs[(size_t)i + 1] = (ogl_label)w_cols.size();
    }
    return true;
}

// ---------------------------------------------------------------------------------------
// Renumbering
// ---------------------------------------------------------------------------------------
void rcm_order(ogl_label n_rows, const ogl_label *row_ptrs, const ogl_label *cols,
               std::vector<ogl_label> &new_id)
{
    const ogl_label n = n_rows;
    new_id.assign((size_t)n, 0);
    if (n == 0) return;
    auto degree = [&](ogl_label v) { return row_ptrs[v + 1] - row_ptrs[v]; };
    std::vector<ogl_label> order((size_t)n);   // Cuthill-McKee order, filled front to back
    std::vector<uint8_t> placed((size_t)n, 0);
    std::vector<ogl_label> stamp((size_t)n, 0), work;  // scratch BFS of the start-node search
    ogl_label epoch = 0;
    work.reserve(1024);
    // BFS over the not yet placed part of the graph from `root`; returns the eccentricity and the
    // first node of minimum degree in the last level
    auto sweep = [&](ogl_label root, ogl_label &far_node) {
        ++epoch;
        work.clear();
        work.push_back(root);
        stamp[(size_t)root] = epoch;
        size_t level_begin = 0, level_end = 1;
        int depth = 0;
        for (;;) {
            for (size_t i = level_begin; i < level_end; ++i) {
                const ogl_label v = work[i];
                for (ogl_label k = row_ptrs[v]; k < row_ptrs[v + 1]; ++k) {
                    const ogl_label c = cols[k];
                    if (c < 0 || c >= n || placed[(size_t)c] || stamp[(size_t)c] == epoch) continue;
                    stamp[(size_t)c] = epoch;
                    work.push_back(c);
                }
            }
            if (work.size() == level_end) break;
            level_begin = level_end;
            level_end = work.size();
            ++depth;
        }
        far_node = work[level_begin];
        for (size_t i = level_begin; i < level_end; ++i)
            if (degree(work[i]) < degree(far_node)) far_node = work[i];
        return depth;
    };
    size_t filled = 0;
    for (ogl_label seed = 0; seed < n; ++seed) {
        if (placed[(size_t)seed]) continue;
        // start of this component: the node of smallest degree in the last level of a breadth-first
        // sweep from the seed (one step of the George-Liu pseudo-peripheral search; more sweeps gain
        // little on mesh graphs and each costs a traversal of the whole component)
        ogl_label start = seed;
        if (degree(seed) > 1) {  // (isolated cells and chain ends are peripheral already)
            ogl_label far_node = seed;
            (void)sweep(seed, far_node);
            start = far_node;
        }
        // Cuthill-McKee from `start`
        size_t head = filled;
        order[filled++] = start;
        placed[(size_t)start] = 1;
        while (head < filled) {
            const ogl_label v = order[head++];
            const size_t tail0 = filled;
            for (ogl_label k = row_ptrs[v]; k < row_ptrs[v + 1]; ++k) {
                const ogl_label c = cols[k];
                if (c < 0 || c >= n || placed[(size_t)c]) continue;
                placed[(size_t)c] = 1;
                order[filled++] = c;
            }
            for (size_t i = tail0 + 1; i < filled; ++i) {  // ascending (degree, index)
                const ogl_label c = order[i];
                const ogl_label d = degree(c);
                size_t j = i;
                while (j > tail0 && (degree(order[j - 1]) > d ||
                                     (degree(order[j - 1]) == d && order[j - 1] > c))) {
                    order[j] = order[j - 1];
                    --j;
                }
                order[j] = c;
            }
        }
    }
    for (ogl_label k = 0; k < n; ++k) new_id[(size_t)order[(size_t)(n - 1 - k)]] = k;  // reverse
}

// Hilbert index of a point of the 2^16 x 2^16 x 2^16 lattice (J. Skilling, "Programming the Hilbert curve", AIP Conf.
// Proc. 707 (2004): axes -> transpose -> interleave)
static uint64_t hilbert_key(uint32_t x, uint32_t y, uint32_t z)
{
    constexpr int BITS = 16;
    uint32_t X[3] = {x, y, z};
    for (uint32_t Q = 1u << (BITS - 1); Q > 1; Q >>= 1) {
        const uint32_t P = Q - 1;
        for (int i = 0; i < 3; ++i) {
            if (X[i] & Q) {
                X[0] ^= P;
            } else {
                const uint32_t t = (X[0] ^ X[i]) & P;
                X[0] ^= t;
                X[i] ^= t;
            }
        }
    }
    for (int i = 1; i < 3; ++i) X[i] ^= X[i - 1];
    uint32_t t = 0;
    for (uint32_t Q = 1u << (BITS - 1); Q > 1; Q >>= 1)
        if (X[2] & Q) t ^= Q - 1;
    for (int i = 0; i < 3; ++i) X[i] ^= t;
    uint64_t key = 0;
    for (int b = BITS - 1; b >= 0; --b)
        for (int i = 0; i < 3; ++i) key = (key << 1) | ((X[i] >> b) & 1u);
    return key;
}

void hilbert_order(ogl_label n, const double *centres, std::vector<ogl_label> &new_id)
{
    new_id.assign((size_t)std::max<ogl_label>(n, 0), 0);
    if (n <= 0) return;
    double lo[3] = {centres[0], centres[1], centres[2]}, hi[3] = {centres[0], centres[1], centres[2]};
    for (ogl_label c = 0; c < n; ++c)
        for (int d = 0; d < 3; ++d) {
            lo[d] = std::min(lo[d], centres[3 * (size_t)c + d]);
            hi[d] = std::max(hi[d], centres[3 * (size_t)c + d]);
        }
    // one scale for the three axes: the curve's cubes stay cubes in space
    double ext = 0.0;
    for (int d = 0; d < 3; ++d) ext = std::max(ext, hi[d] - lo[d]);
    const double scale = ext > 0.0 ? 65535.0 / ext : 0.0;
    std::vector<std::pair<uint64_t, ogl_label>> keyed((size_t)n);
    parallel_ranges(n, 1 << 16, [&](int64_t c0, int64_t c1) {
        for (int64_t c = c0; c < c1; ++c) {
            uint32_t q[3];
            for (int d = 0; d < 3; ++d) {
                const double v = (centres[3 * (size_t)c + d] - lo[d]) * scale;
                q[d] = (uint32_t)std::min(65535.0, std::max(0.0, v));
            }
            keyed[(size_t)c] = {hilbert_key(q[0], q[1], q[2]), (ogl_label)c};
        }
    });
    std::sort(keyed.begin(), keyed.end());  // (key, caller's index): ties keep the caller's order
    for (ogl_label k = 0; k < n; ++k) new_id[(size_t)keyed[(size_t)k].second] = k;
}

double gather_sector_ratio(ogl_label n_rows, const ogl_label *row_ptrs, const ogl_label *cols,
                           const ogl_label *new_id, const ogl_label *old_of)
{
    if (n_rows <= 0 || row_ptrs[n_rows] == 0) return 0.0;
    constexpr int GROUP = 256;    // entries one wavefront instruction of k_spmv_stream gathers
    constexpr int ROWS = 64;      // rows examined per sample
    const int64_t n_samples = std::min<int64_t>(4096, (n_rows + ROWS - 1) / ROWS);
    const double step = (double)n_rows / (double)n_samples;
    std::vector<ogl_label> sect;
    sect.reserve(GROUP);
    int64_t entries = 0, sectors = 0;
    auto flush = [&] {
        std::sort(sect.begin(), sect.end());
        sectors += std::unique(sect.begin(), sect.end()) - sect.begin();
        entries += (int64_t)sect.size();
        sect.clear();
    };
    for (int64_t smp = 0; smp < n_samples; ++smp) {
        const ogl_label k0 = (ogl_label)((double)smp * step);
        for (ogl_label k = k0; k < std::min<int64_t>(n_rows, (int64_t)k0 + ROWS); ++k) {
            const ogl_label r = old_of ? old_of[k] : k;
            for (ogl_label e = row_ptrs[r]; e < row_ptrs[r + 1]; ++e) {
                sect.push_back((new_id ? new_id[cols[e]] : cols[e]) >> 3);
                if ((int)sect.size() == GROUP) flush();
            }
        }
        if (!sect.empty()) flush();
    }
    return entries ? (double)sectors / (double)entries : 0.0;
}

// The same for the compressed layout's kernel, whose gather is slot-major: one instruction fetches the
// s-th entry of the 64 even (or odd) rows of a wavefront's SELL_WAVE_ROWS rows.  0.25 on a hex mesh in
// natural order (the s-th neighbours of consecutive rows are consecutive cells), towards 1 where the s-th
// neighbours of neighbouring rows have nothing to do with each other (polyhedral meshes: there the
// CSR-stream kernel's row-major gather touches fewer lines, gather_sector_ratio above).
double slot_gather_sector_ratio(ogl_label n_rows, const ogl_label *row_ptrs, const ogl_label *cols,
                                const ogl_label *new_id, const ogl_label *old_of)
{
    if (n_rows <= 0 || row_ptrs[n_rows] == 0) return 0.0;
    const int64_t n_waves = (n_rows + SELL_WAVE_ROWS - 1) / SELL_WAVE_ROWS;
    const int64_t n_samples = std::min<int64_t>(1024, n_waves);
    const double step = (double)n_waves / (double)n_samples;
    std::vector<ogl_label> sect;
    sect.reserve(WAVE);
    int64_t entries = 0, sectors = 0;
    for (int64_t smp = 0; smp < n_samples; ++smp) {
        const ogl_label k0 = (ogl_label)((int64_t)((double)smp * step) * SELL_WAVE_ROWS);
        const ogl_label k1 = (ogl_label)std::min<int64_t>(n_rows, (int64_t)k0 + SELL_WAVE_ROWS);
        for (int which = 0; which < ROWS_PER_THREAD; ++which)
            for (ogl_label s = 0;; ++s) {
                sect.clear();
                for (ogl_label k = k0 + which; k < k1; k += ROWS_PER_THREAD) {
                    const ogl_label r = old_of ? old_of[k] : k;
                    if (row_ptrs[r] + s < row_ptrs[r + 1]) {
                        const ogl_label c = cols[row_ptrs[r] + s];
                        sect.push_back((new_id ? new_id[c] : c) >> 3);
                    }
                }
                if (sect.empty()) break;
                std::sort(sect.begin(), sect.end());
                sectors += std::unique(sect.begin(), sect.end()) - sect.begin();
                entries += (int64_t)sect.size();
            }
    }
    return entries ? (double)sectors / (double)entries : 0.0;
}

void renumber_pattern(HostPattern &p, std::vector<ogl_label> new_id, const NumberingHooks *hooks)
{
    const ogl_label N = p.n_rows;
    const int64_t nnz = p.local_nnz;
    std::vector<ogl_label> old_of((size_t)N);
    for (ogl_label c = 0; c < N; ++c) old_of[(size_t)new_id[(size_t)c]] = c;
    if (!(hooks && hooks->renumber_local && hooks->renumber_local(p, new_id))) {
    std::vector<ogl_label> rp((size_t)N + 1, 0);
    for (ogl_label k = 0; k < N; ++k) {
        const ogl_label r = old_of[(size_t)k];
        rp[(size_t)k + 1] = rp[(size_t)k] + (p.row_ptrs[r + 1] - p.row_ptrs[r]);
    }
    std::vector<ogl_label> rows((size_t)nnz), cols((size_t)nnz), map((size_t)nnz);
    auto fill_rows = [&](int64_t k0, int64_t k1) {  // new rows [k0, k1): independent of each other
        for (ogl_label k = (ogl_label)k0; k < (ogl_label)k1; ++k) {
            const ogl_label r = old_of[(size_t)k];
            const ogl_label len = p.row_ptrs[r + 1] - p.row_ptrs[r];
            ogl_label *c = cols.data() + rp[(size_t)k], *m = map.data() + rp[(size_t)k];
            for (ogl_label i = 0; i < len; ++i) {
                rows[(size_t)rp[(size_t)k] + i] = k;
                c[i] = new_id[(size_t)p.cols[(size_t)p.row_ptrs[r] + i]];
                m[i] = p.ldu_mapping[(size_t)p.row_ptrs[r] + i];
            }
            sort_segment(c, m, len);  // stable: equal columns keep the reference's order
        }
    };
    parallel_ranges(N, 1 << 16, fill_rows);
    p.rows.swap(rows);
    p.cols.swap(cols);
    p.ldu_mapping.swap(map);
    p.row_ptrs.swap(rp);
    }
    // non-local part: rows renamed, row-sorted again (stable, so entries of one row keep their order)
    const size_t nl = (size_t)p.non_local_nnz;
    if (nl) {
        std::vector<ogl_label> ord(nl);
        for (size_t e = 0; e < nl; ++e) ord[e] = (ogl_label)e;
        std::stable_sort(ord.begin(), ord.end(), [&](ogl_label a, ogl_label b) {
            return new_id[(size_t)p.nl_rows[(size_t)a]] < new_id[(size_t)p.nl_rows[(size_t)b]];
        });
        std::vector<ogl_label> r2(nl), c2(nl), m2(nl);
        for (size_t e = 0; e < nl; ++e) {
            r2[e] = new_id[(size_t)p.nl_rows[(size_t)ord[e]]];
            c2[e] = p.nl_cols[(size_t)ord[e]];
            m2[e] = p.nl_ldu_mapping[(size_t)ord[e]];
        }
        p.nl_rows.swap(r2);
        p.nl_cols.swap(c2);
        p.nl_ldu_mapping.swap(m2);
    }
    for (auto &c : p.send_idxs) c = new_id[(size_t)c];  // the neighbours expect the same order
    p.new_id = std::move(new_id);
    p.old_of = std::move(old_of);
}

// Slots a chunk's rows get in the value planes (its "cap"), chosen from the row lengths alone: the cap
// that minimises  slots in the 128-byte lines the kernel reads (a lane loads up to the longer of its two
// rows; a line of SELL_LINE_ROWS rows is fetched when one of them is that long) + SELL_ISSUE_COST x slots
// the wavefronts step over (each runs to its own longest row) + SELL_SPILL_COST x entries spilled beyond
// the cap.  Equal to the longest row when lengths are uniform.  lens[n], n <= CHUNK_ROWS, in the order
// the rows take in the chunk.  touched_out: the first term at the chosen cap.
static int32_t sell_chunk_cap_lens(const int32_t *lens, int n, bool allow_spill, double *cost_out,
                                   int64_t *touched_out)
{
    constexpr int N_LINES = CHUNK_ROWS / SELL_LINE_ROWS;
    int32_t hist_max = 0, wave_max[SELL_WAVES] = {}, line_max[N_LINES] = {};
    for (int i = 0; i < n; ++i) {
        hist_max = std::max(hist_max, lens[i]);
        wave_max[i / SELL_WAVE_ROWS] = std::max(wave_max[i / SELL_WAVE_ROWS], lens[i]);
        line_max[i / SELL_LINE_ROWS] = std::max(line_max[i / SELL_LINE_ROWS], lens[i]);
    }
    auto touched_of = [&](int32_t w) {
        int64_t t = 0;
        for (int l = 0; l < N_LINES; ++l) t += (int64_t)std::min(line_max[l], w) * SELL_LINE_ROWS;
        return t;
    };
    if (!allow_spill) {  // (the caller refuses rows beyond SELL_MAX_WIDTH)
        if (touched_out) *touched_out = touched_of(hist_max);
        if (cost_out) *cost_out = (double)touched_of(hist_max);
        return hist_max;
    }
    std::vector<int64_t> longer((size_t)hist_max + 2, 0);  // longer[w] = sum max(0, len - w)
    std::vector<int32_t> cnt((size_t)hist_max + 2, 0);
    for (int i = 0; i < n; ++i) ++cnt[(size_t)lens[i]];
    int64_t rows_longer = 0;
    for (int32_t w = hist_max; w >= 0; --w) {  // longer[w] = longer[w + 1] + #rows with len > w
        longer[(size_t)w] = longer[(size_t)w + 1] + rows_longer;
        rows_longer += cnt[(size_t)w];
    }
    auto cost_of = [&](int32_t w) {
        double issue = 0;
        for (int wv = 0; wv < SELL_WAVES; ++wv) issue += (double)std::min(wave_max[wv], w) * SELL_WAVE_ROWS;
        return (double)touched_of(w) + SELL_ISSUE_COST * issue + SELL_SPILL_COST * (double)longer[(size_t)w];
    };
    const int32_t w_top = std::min<int32_t>(hist_max, SELL_MAX_WIDTH);  // a length must fit a byte
    const double full = hist_max ? cost_of(w_top) : 0.0;
    int32_t best_w = w_top;
    double best = full;
    for (int32_t w = w_top - 1; w >= 1; --w) {
        const double cost = cost_of(w);
        if (cost < best) {
            best = cost;
            best_w = w;
        }
    }
    // spilling is for heavy tails (a few long rows among many): a saving below 10 % is not worth the
    // extra phase, the chunk keeps all its entries in the planes
    if (best > 0.9 * full) {
        best = full;
        best_w = w_top;
    }
    if (cost_out) *cost_out = best;
    if (touched_out) *touched_out = touched_of(best_w);
    return best_w;
}

static int32_t sell_chunk_cap(const ogl_label *row_ptrs, ogl_label r0, ogl_label r1, bool allow_spill,
                              int64_t *touched_out)
{
    int32_t lens[CHUNK_ROWS];
    for (ogl_label r = r0; r < r1; ++r) lens[r - r0] = row_ptrs[r + 1] - row_ptrs[r];
    return sell_chunk_cap_lens(lens, (int)(r1 - r0), allow_spill, nullptr, touched_out);
}

// Cost of the compressed layout per stored entry (sell_chunk_cap_lens, summed over the chunks) when the
// rows are taken in the order order[0], order[1], ... (order == nullptr: the pattern's own).  1 = no
// padding read, nothing spilled, every wavefront's rows equally long.
static double sell_cost_ratio(ogl_label n_rows, const ogl_label *row_ptrs, const ogl_label *order)
{
    if (n_rows <= 0 || row_ptrs[n_rows] == 0) return 1.0;
    double total = 0;
    int32_t lens[CHUNK_ROWS];
    for (ogl_label k0 = 0; k0 < n_rows; k0 += CHUNK_ROWS) {
        const int n = (int)std::min<int64_t>(CHUNK_ROWS, (int64_t)n_rows - k0);
        for (int i = 0; i < n; ++i) {
            const ogl_label r = order ? order[k0 + i] : k0 + i;
            lens[i] = row_ptrs[r + 1] - row_ptrs[r];
        }
        double cost = 0;
        (void)sell_chunk_cap_lens(lens, n, true, &cost, nullptr);
        total += cost;
    }
    return total / (double)row_ptrs[n_rows];
}

namespace {
// OGL_TIME_SETUP=1: the phases of the first set_matrix of a pattern on stderr (development aid)
struct PhaseTimer {
    bool on = std::getenv("OGL_TIME_SETUP") != nullptr;
    std::chrono::steady_clock::time_point t = std::chrono::steady_clock::now();
    void lap(const char *what)
    {
        if (!on) return;
        const auto n = std::chrono::steady_clock::now();
        std::fprintf(stderr, "[ogl setup] %-34s %8.3f s\n", what, std::chrono::duration<double>(n - t).count());
        t = n;
    }
};
}  // namespace

int choose_numbering(HostPattern &p, int mode, bool try_sell, SellLayout *sell_out, bool *sell_built,
                     RenumberReport &rep, const NumberingHooks *hooks)
{
    PhaseTimer tm;
    rep = RenumberReport{};
    if (sell_built) *sell_built = false;
    if (mode < 0 || mode > 2) return fail(OGL_ERR_INVALID, "renumber %d outside {0 off, 1 on, 2 auto}", mode);
    const ogl_label N = p.n_rows;
    if (mode == 0 || N < 2) return OGL_OK;
    // auto leaves small systems alone: below this a solve is bound by launch latency, not by the
    // x gather, and renumbering would only cost set-up time
    if (mode == 2 && N < RENUMBER_AUTO_MIN_ROWS) return OGL_OK;
    auto hand_over = [&](SellLayout &L, bool ok) {
        rep.sell_used = ok;
        if (sell_out && sell_built) {
            *sell_out = std::move(L);
            *sell_built = true;
        }
    };
    SellLayout natural;
    bool have_natural = false;
    rep.ratio_natural = rep.ratio_used =
        gather_sector_ratio(N, p.row_ptrs.data(), p.cols.data(), nullptr, nullptr);
    if (mode == 2 && try_sell) {
        // "structured" = the compressed layout qualifies with 1-byte codes in every chunk: kept as it
        // is.  Patterns that only fit the 16-bit delta / 32-bit column codes are judged by their
        // gather locality and by the padding they make the kernel read.
        // (a look at a few chunks first: one that needs 16 / 32-bit codes settles "not structured" without
        //  laying out the whole matrix; none found -> the full layout decides, as it always did)
        if (!sell_pattern_is_irregular_sampled(N, p.row_ptrs.data(), p.cols.data())) {
            rep.sell_natural = build_sell_layout(N, p.row_ptrs.data(), p.cols.data(), natural);
            have_natural = true;
            tm.lap("sell layout, caller's numbering");
            if (rep.sell_natural && natural.n_delta16 + natural.n_col32 == 0) {
                hand_over(natural, true);
                return OGL_OK;
            }
        }
    }
    // ---- step 1: the order of the rows at large: the caller's, or reverse Cuthill-McKee
    std::vector<ogl_label> new_id, old_of;  // empty = the caller's numbering
    if (mode == 1 || rep.ratio_natural > 0.25) {
        std::vector<ogl_label> cand, cand_old((size_t)N);
        if (!(hooks && hooks->rcm && hooks->rcm(p, cand))) rcm_order(N, p.row_ptrs.data(), p.cols.data(), cand);
        tm.lap("rcm_order");
        for (ogl_label c = 0; c < N; ++c) cand_old[(size_t)cand[(size_t)c]] = c;
        double r = gather_sector_ratio(N, p.row_ptrs.data(), p.cols.data(), cand.data(),
                                       cand_old.data());
        rep.ratio_rcm = r;
        if (hooks && hooks->centres) {
            // second candidate: the cells along a Hilbert curve through their centres.  Consecutive rows are then
            // neighbours in SPACE (a chunk of 512 rows is a compact blob, most of whose neighbours lie in the blob),
            // where RCM makes them neighbours in a breadth-first front (a chunk is a strip of a front whose neighbours lie
            // in the two adjacent fronts): on a Voronoi mesh a chunk touches 0.029 distinct 64-byte sectors of x per
            // entry against 0.052 in RCM order.  The better of the two by the gather measure is taken.
            std::vector<ogl_label> curve, curve_old((size_t)N);
            if (!(hooks->curve && hooks->curve(N, hooks->centres, curve) && (ogl_label)curve.size() == N))
                hilbert_order(N, hooks->centres, curve);
            tm.lap("hilbert_order");
            for (ogl_label c = 0; c < N; ++c) curve_old[(size_t)curve[(size_t)c]] = c;
            rep.ratio_curve = gather_sector_ratio(N, p.row_ptrs.data(), p.cols.data(), curve.data(), curve_old.data());
            // ... unless it would cost the packed columns: along the curve most of a chunk's neighbours sit in the chunk's
            // own blob, but a few sit in blobs anywhere in the numbering.  The CSR-stream kernel packs its columns as 21-bit
            // offsets into a window of 2^21 columns around the chunk's rows and lists the entries outside it ("far") apart;
            // beyond STREAM21_MAX_FAR of the entries it gives the packing up, and the plain 12-byte CSR-stream kernel
            // loses more than the curve gains (Voronoi 3 M cells: 151 us against 133 in RCM order with packed columns).
            bool packable = true;
            if ((int64_t)N > ((int64_t)1 << STREAM21_BITS)) {
                int64_t far = 0;
                const bool counted = hooks->curve_far && hooks->curve_far(p, curve, curve_old, far);
                for (ogl_label k0 = 0; !counted && k0 < N; k0 += CHUNK_ROWS) {
                    ogl_label lo = N, hi = 0;
                    const ogl_label k1 = (ogl_label)std::min<int64_t>(N, (int64_t)k0 + CHUNK_ROWS);
                    for (ogl_label k = k0; k < k1; ++k) {
                        const ogl_label row = curve_old[(size_t)k];
                        for (ogl_label e = p.row_ptrs[row]; e < p.row_ptrs[row + 1]; ++e) {
                            const ogl_label c = curve[(size_t)p.cols[e]];
                            lo = std::min(lo, c);
                            hi = std::max(hi, c);
                        }
                    }
                    if (hi < lo || (int64_t)hi - lo < ((int64_t)1 << STREAM21_BITS)) continue;
                    const int64_t base = stream21_window_base(k0, N);
                    for (ogl_label k = k0; k < k1; ++k) {
                        const ogl_label row = curve_old[(size_t)k];
                        for (ogl_label e = p.row_ptrs[row]; e < p.row_ptrs[row + 1]; ++e) {
                            const int64_t d = (int64_t)curve[(size_t)p.cols[e]] - base;
                            far += (d < 0 || d >= ((int64_t)1 << STREAM21_BITS)) ? 1 : 0;
                        }
                    }
                }
                packable = (double)far <= STREAM21_MAX_FAR * (double)p.row_ptrs[N];
                rep.curve_far_entries = far;
            }
            rep.curve_packable = packable;
            // (and only where it gathers clearly better than RCM -- a hex mesh: 0.091 against 0.102, but the compressed
            //  layout's 16-bit deltas do not survive the curve and the 12-byte CSR-stream kernel would run: 79.9 against
            //  77.3 us per CG turn at 128^3 shuffled; a Voronoi mesh: 0.12 against 0.175)
            if (rep.ratio_curve <= 0.8 * r && packable) {
                cand.swap(curve);
                cand_old.swap(curve_old);
                r = rep.ratio_curve;
                rep.curve_used = true;
            }
        }
        if (mode == 1 || r <= 0.9 * rep.ratio_natural) {
            new_id.swap(cand);
            old_of.swap(cand_old);
            rep.ratio_used = r;
        } else {
            rep.curve_used = false;
        }
    }
    tm.lap("gather ratio of the candidate");
    // ---- step 2 (compressed layout only): inside every wavefront's SELL_WAVE_ROWS rows, longest rows
    // first.  The lanes of the SpMV stop loading at the end of their own rows, so with the long rows of a
    // wavefront next to each other the 128-byte lines it reads hold (almost) no padding.  The x gather is
    // untouched -- a wavefront still works on the same SELL_WAVE_ROWS rows (sorting over a whole chunk
    // was tried: it scatters the gather, profiles/r02_unstructured_proxy.txt).  Matters for meshes with
    // mixed cell types (polyhedral, hex-dominant); a no-op on hex meshes.
    // (not on top of the Hilbert order: there the rows of a window are neighbours in space whatever their lengths, the
    //  CSR-stream kernel with packed columns is what runs, and a length sort inside the windows undoes part of the
    //  locality the curve was taken for -- Voronoi 3 M cells: 155 us on the sorted numbering against 133 in plain RCM order)
    if (try_sell && !rep.curve_used) {
        const ogl_label *base = old_of.empty() ? nullptr : old_of.data();
        const double before = sell_cost_ratio(N, p.row_ptrs.data(), base);
        // ... and only where the compressed layout's slot-major gather has a chance against the
        // CSR-stream kernel's row-major one (a polyhedral mesh stays in plain RCM order, which is what
        // the CSR-stream kernel wants: consecutive rows = neighbouring cells)
        if (before > 1.15)  // (1 + SELL_ISSUE_COST = rows of equal length everywhere)
            rep.slot_ratio = slot_gather_sector_ratio(N, p.row_ptrs.data(), p.cols.data(),
                                                      new_id.empty() ? nullptr : new_id.data(), base);
        if (before > 1.15 && rep.slot_ratio <= SELL_SORT_MAX_SLOT_RATIO) {
            std::vector<ogl_label> order((size_t)N);
            for (ogl_label k = 0; k < N; ++k) order[(size_t)k] = base ? base[k] : k;
            auto len = [&](ogl_label r) { return p.row_ptrs[r + 1] - p.row_ptrs[r]; };
            for (ogl_label k0 = 0; k0 < N; k0 += SELL_WAVE_ROWS)
                std::stable_sort(order.begin() + k0, order.begin() + std::min<int64_t>(N, (int64_t)k0 + SELL_WAVE_ROWS),
                                 [&](ogl_label a, ogl_label b) { return len(a) > len(b); });
            const double after = sell_cost_ratio(N, p.row_ptrs.data(), order.data());
            if (after <= 0.95 * before) {
                new_id.assign((size_t)N, 0);
                for (ogl_label k = 0; k < N; ++k) new_id[(size_t)order[(size_t)k]] = k;
                rep.sorted_by_length = true;
            }
        }
    }
    if (new_id.empty()) {  // the caller's numbering stays
        if (have_natural) hand_over(natural, rep.sell_natural);
        return OGL_OK;
    }
    tm.lap("row-length sort policy");
    renumber_pattern(p, std::move(new_id), hooks);
    tm.lap("renumber_pattern");
    rep.applied = true;
    rep.ratio_used = gather_sector_ratio(N, p.row_ptrs.data(), p.cols.data(), nullptr, nullptr);
    if (try_sell && sell_out && sell_built) {
        SellLayout L;
        const bool ok = build_sell_layout(N, p.row_ptrs.data(), p.cols.data(), L);
        tm.lap("sell layout, new numbering");
        hand_over(L, ok);
    }
    return OGL_OK;
}

// ---------------------------------------------------------------------------------------
// Index-compressed chunked ELL (SellChunk, common.hpp).  The pattern qualifies when every chunk of
// CHUNK_ROWS rows uses at most SELL_MAX_DICT distinct (column - row) offsets -- true for structured
// and banded finite-volume meshes, where a chunk sees a handful of diagonals -- and the padding to
// the chunk's longest row stays below SELL_MAX_PADDING x nnz.
// ---------------------------------------------------------------------------------------
namespace {

// What one chunk of the compressed layout looks like, from its rows alone (pass 1 of build_sell_layout):
// the chunks are independent of each other until their offsets into the common arrays are assigned.
struct SellChunkPlan {
    bool ok = true;             // false: a row longer than SELL_MAX_WIDTH without the spill
    int mode = SELL_MODE_COL32;
    int32_t wave_w[SELL_WAVES] = {0, 0, 0, 0};
    int32_t width = 0, base = 0;
    int64_t touched = 0;        // value slots in the 128-byte lines the kernel reads
    std::vector<int32_t> table; // pattern mode: patterns x width offsets; offset mode: the dictionary
    std::vector<uint8_t> pid;   // pattern mode: pattern id of each of the CHUNK_ROWS rows
    std::vector<int32_t> spill_rows, spill_len;  // rows whose tails spill, entries spilled per row
};

void plan_sell_chunk(int64_t c, ogl_label n_rows, const ogl_label *row_ptrs, const ogl_label *cols, bool allow_spill,
                     SellChunkPlan &P)
{
    const ogl_label r0 = (ogl_label)(c * CHUNK_ROWS);
    const ogl_label r1 = (ogl_label)std::min<int64_t>(n_rows, (c + 1) * (int64_t)CHUNK_ROWS);
    // Cap of the chunk: the slots its rows get in the planes.  Entries beyond it (the tails of a
    // few long rows: split / polyhedral cells of a hex-dominant mesh) are "spilled" into a short
    // row-sorted list that the chunk's workgroup adds after the planes, continuing every row's sum
    // in stored order -- so one long row does not make 127 others read padding.
    const int32_t cap = sell_chunk_cap(row_ptrs, r0, r1, allow_spill, &P.touched);
    if (cap > SELL_MAX_WIDTH) {  // (only without the spill: a length must fit a byte)
        P.ok = false;
        return;
    }
    auto row_end = [&](ogl_label r) { return std::min(row_ptrs[r + 1], row_ptrs[r] + cap); };
    int32_t width = 0;
    for (int wv = 0; wv < SELL_WAVES; ++wv) {  // what each wavefront has to run to
        int32_t ww = 0;
        for (ogl_label r = r0 + wv * SELL_WAVE_ROWS; r < std::min<int64_t>(r1, (int64_t)r0 + (wv + 1) * SELL_WAVE_ROWS); ++r)
            ww = std::max(ww, row_end(r) - row_ptrs[r]);
        P.wave_w[wv] = ww;
        width = std::max(width, ww);
    }
    P.width = width;
    for (ogl_label r = r0; r < r1; ++r)
        if (row_end(r) < row_ptrs[r + 1]) {
            P.spill_rows.push_back(r);
            P.spill_len.push_back(row_ptrs[r + 1] - row_end(r));
        }
    // (a) row patterns: one byte per row.  Known patterns are found through a small hash table
    // (an irregular chunk would otherwise compare every row with up to 256 patterns before giving up)
    bool pat_mode = width > 0;
    std::vector<int32_t> &pats = P.table;
    std::vector<int32_t> pat;
    P.pid.assign(CHUNK_ROWS, 0);
    size_t last_pat = 0;
    constexpr int PAT_HASH = 1024;  // > 2 x 256 patterns, power of two
    int16_t pat_slot[PAT_HASH];
    for (int i = 0; i < PAT_HASH; ++i) pat_slot[i] = -1;
    for (ogl_label lr = 0; lr < CHUNK_ROWS && pat_mode; ++lr) {
        const ogl_label r = r0 + lr;
        pat.assign((size_t)width, SELL_PAD_OFFSET);
        uint32_t hsh = 2166136261u;
        if (r < r1)
            for (ogl_label k = row_ptrs[r], s = 0; k < row_end(r); ++k, ++s) {
                pat[(size_t)s] = cols[k] - r;
                hsh = (hsh ^ (uint32_t)pat[(size_t)s]) * 16777619u;
            }
        const size_t n_pat = pats.size() / (size_t)width;
        auto same = [&](size_t i) {
            return std::equal(pat.begin(), pat.end(), pats.begin() + (std::ptrdiff_t)(i * width));
        };
        size_t id = n_pat;
        if (n_pat && same(last_pat)) {
            id = last_pat;
        } else {
            uint32_t slot = (hsh ^ (hsh >> 15)) & (PAT_HASH - 1);
            while (pat_slot[slot] >= 0 && !same((size_t)pat_slot[slot])) slot = (slot + 1) & (PAT_HASH - 1);
            if (pat_slot[slot] >= 0) {
                id = (size_t)pat_slot[slot];
            } else {
                if (n_pat == 256 || (n_pat + 1) * (size_t)width > (size_t)SELL_TABLE_INTS) {
                    pat_mode = false;
                    break;
                }
                pat_slot[slot] = (int16_t)n_pat;
                pats.insert(pats.end(), pat.begin(), pat.end());
            }
        }
        last_pat = id;
        P.pid[(size_t)lr] = (uint8_t)id;
    }
    if (pat_mode) {
        P.mode = SELL_MODE_PATTERN;
        return;
    }
    P.pid.clear();
    // (b) otherwise offsets: one byte per (row, slot), when the chunk has <= 255 distinct ones
    std::vector<int32_t> &ds = P.table;
    ds.clear();
    bool off8_mode = width > 0;
    if (off8_mode) {
        size_t last = 0;
        for (ogl_label r = r0; r < r1 && off8_mode; ++r)
            for (ogl_label k = row_ptrs[r]; k < row_end(r); ++k) {
                const int32_t d = cols[k] - r;
                if (!ds.empty()) {
                    if (ds[last] == d) continue;
                    size_t i = 0;
                    while (i < ds.size() && ds[i] != d) ++i;
                    if (i < ds.size()) {
                        last = i;
                        continue;
                    }
                }
                if (ds.size() == (size_t)SELL_MAX_DICT) {
                    off8_mode = false;
                    break;
                }
                ds.push_back(d);
                last = ds.size() - 1;
            }
        if (off8_mode) std::sort(ds.begin(), ds.end());
    }
    if (off8_mode) {
        P.mode = SELL_MODE_OFFSET8;
        return;
    }
    ds.clear();
    // (c) otherwise 16-bit deltas along the row: first code = (first column - row) - base, then
    // column[s] - column[s-1] (rows are stored in ascending column order); 0xFFFF = padding
    bool d16_mode = width > 0;
    if (d16_mode) {
        int64_t lo = INT64_MAX, hi = INT64_MIN;
        for (ogl_label r = r0; r < r1 && d16_mode; ++r) {
            if (row_ptrs[r] == row_end(r)) continue;
            const int64_t first = (int64_t)cols[row_ptrs[r]] - r;
            lo = std::min(lo, first);
            hi = std::max(hi, first);
            for (ogl_label k = row_ptrs[r] + 1; k < row_end(r); ++k) {
                const int64_t d = (int64_t)cols[k] - cols[k - 1];
                if (d < 0 || d > SELL_MAX_DELTA16) d16_mode = false;
            }
        }
        if (lo == INT64_MAX) lo = hi = 0;
        if (hi - lo > SELL_MAX_DELTA16) d16_mode = false;
        P.base = (int32_t)lo;
    }
    // (d) otherwise plain 32-bit columns (-1 = padding): always possible (also what an empty chunk gets:
    // nothing is ever read)
    P.mode = d16_mode ? SELL_MODE_DELTA16 : SELL_MODE_COL32;
}

}  // namespace

bool sell_pattern_is_irregular_sampled(ogl_label n_rows, const ogl_label *row_ptrs, const ogl_label *cols)
{
    const int64_t nc = n_chunks(n_rows);
    const int64_t n_samples = std::min<int64_t>(64, nc);
    for (int64_t i = 0; i < n_samples; ++i) {
        SellChunkPlan P;
        plan_sell_chunk(i * nc / n_samples, n_rows, row_ptrs, cols, true, P);
        if (P.ok && P.width > 0 && P.mode >= SELL_MODE_DELTA16) return true;
    }
    return false;
}

bool build_sell_layout(ogl_label n_rows, const ogl_label *row_ptrs, const ogl_label *cols,
                       SellLayout &out, bool allow_spill)
{
    out = SellLayout{};
    const int64_t nc = n_chunks(n_rows);
    const int64_t nnz = n_rows > 0 ? row_ptrs[n_rows] : 0;
    out.chunks.resize((size_t)nc);
    // pass 1: per chunk the width, the coding mode, the dictionary / pattern table (independent: in parallel)
    std::vector<SellChunkPlan> plans((size_t)nc);
    parallel_ranges(nc, 64, [&](int64_t c0, int64_t c1) {
        for (int64_t c = c0; c < c1; ++c) plan_sell_chunk(c, n_rows, row_ptrs, cols, allow_spill, plans[(size_t)c]);
    });
    // ... then their places in the common arrays, chunk after chunk
    int64_t val_len = 0, code_len = 0;
    std::vector<int64_t> pid_off((size_t)nc, -1);
    for (int64_t c = 0; c < nc; ++c) {
        const SellChunkPlan &P = plans[(size_t)c];
        if (!P.ok) return false;
        out.read_slots += P.touched;
        out.spill_chunk_ptr.push_back((int32_t)out.spill_rows.size());
        for (size_t i = 0; i < P.spill_rows.size(); ++i) {
            const ogl_label r = P.spill_rows[i];
            out.spill_rows.push_back(r);
            out.spill_ptrs.push_back((int32_t)out.spill_cols.size());
            for (ogl_label k = row_ptrs[r + 1] - P.spill_len[i]; k < row_ptrs[r + 1]; ++k) {
                out.spill_cols.push_back(cols[k]);
                out.spill_map.push_back(k);
            }
        }
        SellChunk &h = out.chunks[(size_t)c];
        h.val_off = val_len;
        if (P.mode != SELL_MODE_PATTERN) code_len += SELL_LEN_BYTES;  // the row lengths, one byte each, in front of the codes
        h.code_off = code_len;
        h.dict_off = (int32_t)out.dict.size();
        if (P.mode == SELL_MODE_PATTERN || P.mode == SELL_MODE_OFFSET8) {
            h.set(P.mode, (int)P.table.size(), P.wave_w);
            out.dict.insert(out.dict.end(), P.table.begin(), P.table.end());
        } else if (P.mode == SELL_MODE_DELTA16) {
            // group-major 16-byte words: word (g, t) holds SELL_D16_GROUP slots of thread t's two rows;
            // lane t of a wavefront reads consecutive words
            h.set(SELL_MODE_DELTA16, 0, P.wave_w);
            h.dict_off = P.base;
            out.n_delta16 += 1;
        } else {
            h.set(SELL_MODE_COL32, 0, P.wave_w);
            h.dict_off = 0;
            if (P.width > 0) out.n_col32 += 1;
        }
        const int cs = h.code_stride();
        code_len += (int64_t)(cs > 0 ? cs : 16) * BLOCK;
        val_len += (int64_t)P.width * CHUNK_ROWS;
        code_len = (code_len + 15) / 16 * 16;
        // padding that is READ (a wavefront runs to its own longest row) must stay below what CSR's
        // indices cost; padding that is only allocated (to the chunk's longest row) is bounded too
        if ((double)out.read_slots + SELL_SPILL_COST * (double)out.spill_cols.size() >
            SELL_MAX_PADDING * (double)nnz + 8.0 * CHUNK_ROWS)
            return false;
        if ((double)val_len > SELL_MAX_ALLOC * (double)nnz + 8.0 * CHUNK_ROWS) return false;
    }
    out.n_slots = val_len;
    out.spill_ptrs.push_back((int32_t)out.spill_cols.size());
    out.spill_chunk_ptr.push_back((int32_t)out.spill_rows.size());
    out.codes.assign((size_t)code_len + 16, (uint8_t)255);
    out.map.assign((size_t)val_len + 2, -1);
    // pass 2: codes and the value map (every chunk writes its own ranges: in parallel)
    parallel_ranges(nc, 64, [&](int64_t c0, int64_t c1) {
      for (int64_t c = c0; c < c1; ++c) {
        const SellChunk &h = out.chunks[(size_t)c];
        const ogl_label r0 = (ogl_label)(c * CHUNK_ROWS);
        const ogl_label r1 = (ogl_label)std::min<int64_t>(n_rows, (c + 1) * (int64_t)CHUNK_ROWS);
        const bool d16_mode = h.mode() == SELL_MODE_DELTA16, c32_mode = h.mode() == SELL_MODE_COL32;
        const bool pat_mode = h.mode() == SELL_MODE_PATTERN;
        const int code_stride = h.code_stride();
        if (pat_mode)  // rows 2t, 2t+1 of thread t are adjacent bytes
            std::copy(plans[(size_t)c].pid.begin(), plans[(size_t)c].pid.end(), out.codes.begin() + h.code_off);
        const int32_t *d0 = (d16_mode || c32_mode) ? nullptr : out.dict.data() + h.dict_off;
        if (!pat_mode) std::fill_n(out.codes.begin() + (h.code_off - SELL_LEN_BYTES), SELL_LEN_BYTES, (uint8_t)0);
        for (ogl_label r = r0; r < r1; ++r) {
            const int32_t lr = r - r0, t = lr / ROWS_PER_THREAD, which = lr % ROWS_PER_THREAD;
            uint8_t *code = out.codes.data() + h.code_off + (int64_t)t * code_stride;
            const ogl_label k_end = std::min(row_ptrs[r + 1], row_ptrs[r] + h.width());  // (the rest is spilled)
            if (!pat_mode) out.codes[(size_t)(h.code_off - SELL_LEN_BYTES + lr)] = (uint8_t)(k_end - row_ptrs[r]);
            for (ogl_label k = row_ptrs[r], s = 0; k < k_end; ++k, ++s) {
                out.map[(size_t)(h.val_off + (int64_t)s * CHUNK_ROWS + lr)] = k;
                if (pat_mode) continue;
                if (d16_mode) {
                    const int64_t v = s == 0 ? (int64_t)cols[k] - r - h.dict_off : (int64_t)cols[k] - cols[k - 1];
                    uint8_t *w = out.codes.data() + h.code_off +
                                 ((int64_t)(s / SELL_D16_GROUP) * BLOCK + t) * 16 +
                                 ((s % SELL_D16_GROUP) * ROWS_PER_THREAD + which) * 2;
                    const uint16_t v16 = (uint16_t)v;
                    std::memcpy(w, &v16, 2);
                } else if (c32_mode) {
                    uint8_t *w = out.codes.data() + h.code_off +
                                 ((int64_t)(s / SELL_C32_GROUP) * BLOCK + t) * 16 +
                                 ((s % SELL_C32_GROUP) * ROWS_PER_THREAD + which) * 4;
                    const int32_t v32 = cols[k];
                    std::memcpy(w, &v32, 4);
                } else {
                    const int32_t d = cols[k] - r;
                    code[ROWS_PER_THREAD * s + which] =
                        (uint8_t)(std::lower_bound(d0, d0 + h.dict_len(), d) - d0);
                }
            }
        }
      }
    });
    return true;
}

void band_block_order(ogl_label n_rows, int64_t band, std::vector<int32_t> &order)
{
    order.clear();
    if (band < (int64_t)N_XCD * CHUNK_ROWS || band * 4 > n_rows) return;
    const int64_t nc = n_chunks(n_rows);
    std::vector<std::vector<int32_t>> lists(N_XCD);
    for (int64_t c = 0; c < nc; ++c) {
        const int64_t ph = (c * CHUNK_ROWS) % band;  // position of the chunk's first row in its band period
        lists[(size_t)std::min<int64_t>(N_XCD - 1, ph * N_XCD / band)].push_back((int32_t)c);
    }
    size_t longest = 0;
    for (auto &l : lists) longest = std::max(longest, l.size());
    order.assign(longest * N_XCD, -1);
    for (int x = 0; x < N_XCD; ++x)
        for (size_t i = 0; i < lists[(size_t)x].size(); ++i) order[i * N_XCD + (size_t)x] = lists[(size_t)x][i];
}

void symx_block_order(const SymxLayout &L, bool general, std::vector<int32_t> &order)
{
    order.clear();
    const int64_t nc = (int64_t)L.chunks.size();
    std::vector<std::vector<int32_t>> lists(N_XCD);
    int64_t banded = 0, taken = 0;
    auto wanted = [&](const SymxChunk &h) { return (h.ex_rp_off >= 0 && h.merge != 0) == general; };
    for (int64_t c = 0; c < nc; ++c) {
        const SymxChunk &h = L.chunks[(size_t)c];
        if (!wanted(h)) continue;
        ++taken;
        const int64_t band = h.nd >= 2 ? h.d[h.nd - 2] : 0;
        int x = (int)((c / 4) % N_XCD);  // as the default map does (groups of 4)
        if (band >= (int64_t)N_XCD * CHUNK_ROWS) {
            const int64_t ph = (c * CHUNK_ROWS) % band;  // the chunks of rows r and r +- band share a phase, hence an XCD
            x = (int)std::min<int64_t>(N_XCD - 1, ph * N_XCD / band);
            ++banded;
        }
        lists[(size_t)x].push_back((int32_t)c);
    }
    if (banded * 2 < taken) {  // mostly short bands: the default map is as good
        for (auto &l : lists) l.clear();
        for (int64_t c = 0; c < nc; ++c)
            if (wanted(L.chunks[(size_t)c])) lists[(size_t)((c / 4) % N_XCD)].push_back((int32_t)c);
    }
    size_t longest = 0;
    for (auto &l : lists) longest = std::max(longest, l.size());
    order.assign(longest * N_XCD, -1);
    for (int x = 0; x < N_XCD; ++x)
        for (size_t i = 0; i < lists[(size_t)x].size(); ++i) order[i * N_XCD + (size_t)x] = lists[(size_t)x][i];
}

bool build_sym_layout(ogl_label n_rows, const ogl_label *row_ptrs, const ogl_label *cols, SymLayout &out)
{
    out = SymLayout{};
    if (n_rows <= 0) return false;
    const int64_t nc = n_chunks(n_rows);
    // the distances from the diagonal that occur (upper side), ascending; 0 always takes plane 0
    int64_t dist[SYM_MAX_OFFSETS] = {0};
    int nd = 1;
    int64_t upper_entries = 0;
    for (ogl_label r = 0; r < n_rows; ++r)
        for (ogl_label k = row_ptrs[r]; k < row_ptrs[r + 1]; ++k) {
            const int64_t d = (int64_t)cols[k] - r;
            if (d < 0) continue;
            ++upper_entries;
            int j = 0;
            while (j < nd && dist[j] != d) ++j;
            if (j < nd) continue;
            if (nd == SYM_MAX_OFFSETS) return false;
            dist[nd++] = d;
        }
    std::sort(dist, dist + nd);
    if (nd < 2 || dist[0] != 0 || dist[nd - 1] > INT32_MAX / 2) return false;
    if ((double)nd * (double)nc * CHUNK_ROWS > SYM_MAX_PADDING * (double)upper_entries + 8.0 * CHUNK_ROWS) return false;
    out.nd = nd;
    for (int j = 0; j < nd; ++j) out.d[j] = (int32_t)dist[j];
    out.mask.assign((size_t)nc * CHUNK_ROWS + 16, 0);
    out.map.assign((size_t)nc * nd * CHUNK_ROWS + 2, -1);
    auto plane_of = [&](int64_t d) {
        int j = 0;
        while (j < nd && dist[j] != d) ++j;
        return j;
    };
    auto slot_of = [&](int j, int64_t r) { return (size_t)((r / CHUNK_ROWS) * nd + j) * CHUNK_ROWS + (size_t)(r % CHUNK_ROWS); };
    for (ogl_label r = 0; r < n_rows; ++r) {
        int64_t prev = INT64_MIN;
        for (ogl_label k = row_ptrs[r]; k < row_ptrs[r + 1]; ++k) {
            const int64_t d = (int64_t)cols[k] - r;
            if (d <= prev) return false;  // the kernel sums in ascending column order: one entry per column
            prev = d;
            if (d < 0) continue;
            const int j = plane_of(d);
            out.map[slot_of(j, r)] = k;
            out.mask[(size_t)r] |= (uint8_t)(1u << (nd - 1 + j));
        }
    }
    // every lower entry must have its upper twin (that is where its value is read)
    for (ogl_label r = 0; r < n_rows; ++r)
        for (ogl_label k = row_ptrs[r]; k < row_ptrs[r + 1] && cols[k] < r; ++k) {
            const int j = plane_of((int64_t)r - cols[k]);
            if (j == nd || out.map[slot_of(j, cols[k])] < 0) return false;
            out.mask[(size_t)r] |= (uint8_t)(1u << (nd - 1 - j));
        }
    return true;
}

bool build_symx_layout(ogl_label n_rows, const ogl_label *row_ptrs, const ogl_label *cols, SymxLayout &out)
{
    out = SymxLayout{};
    if (n_rows <= 0) return false;
    const int64_t nc = n_chunks(n_rows);
    const int64_t nnz = row_ptrs[n_rows];
    out.chunks.assign((size_t)nc, SymxChunk{});
    // pass 1: the distances of every chunk -- the three most frequent upper distances of its rows (ties: the
    // shorter one), ascending -- and the place of its planes.  pair_only: only distance 1 and even distances are
    // eligible, and none without distance 1 (what the pair-load instantiation of the kernel can address).
    int64_t slots = 0, upper_entries = 0;
    auto pair_shaped = [](const SymxChunk &h) {
        if (h.nd >= 2 && h.d[0] != 1) return false;
        for (int j = 2; j < h.nd; ++j)
            if (h.d[j - 1] % 2 != 0) return false;
        return true;
    };
    auto pick_distances = [&](int64_t c, bool pair_only, std::vector<std::pair<int32_t, int32_t>> &hist) {
        const ogl_label r0 = (ogl_label)(c * CHUNK_ROWS), r1 = (ogl_label)std::min<int64_t>(n_rows, (c + 1) * (int64_t)CHUNK_ROWS);
        hist.clear();
        for (ogl_label r = r0; r < r1; ++r)
            for (ogl_label k = row_ptrs[r]; k < row_ptrs[r + 1]; ++k) {
                const int64_t d = (int64_t)cols[k] - r;
                if (d <= 0 || d > INT32_MAX / 2) continue;
                if (pair_only && d != 1 && d % 2 != 0) continue;
                size_t i = 0;
                while (i < hist.size() && hist[i].first != (int32_t)d) ++i;
                if (i == hist.size()) hist.emplace_back((int32_t)d, 0);
                ++hist[i].second;
            }
        std::sort(hist.begin(), hist.end(), [](const auto &a, const auto &b) {
            return a.second != b.second ? a.second > b.second : a.first < b.first;
        });
        SymxChunk &h = out.chunks[(size_t)c];
        h.nd = 1 + (int32_t)std::min<size_t>(3, hist.size());
        int32_t dd[3] = {0, 0, 0};
        for (int j = 0; j + 1 < h.nd; ++j) dd[j] = hist[(size_t)j].first;
        std::sort(dd, dd + (h.nd - 1));
        for (int j = 0; j < 3; ++j) h.d[j] = dd[j];
        if (pair_only && h.nd >= 2 && h.d[0] != 1) {  // no distance 1 among them: nothing planar but the diagonal
            h.nd = 1;
            h.d[0] = h.d[1] = h.d[2] = 0;
        }
        h.ex_rp_off = -1;
    };
    parallel_ranges(nc, 64, [&](int64_t c0, int64_t c1) {
        std::vector<std::pair<int32_t, int32_t>> hist;  // (distance, count)
        for (int64_t c = c0; c < c1; ++c) pick_distances(c, false, hist);
    });
    {   // a few chunks out of shape (block seams) must not cost all the others the pair-load instantiation: they
        // keep what fits it and hold the rest as explicit entries
        int64_t odd = 0;
        for (const SymxChunk &h : out.chunks) odd += !pair_shaped(h);
        if (odd > 0 && odd * 20 <= nc) {
            std::vector<std::pair<int32_t, int32_t>> hist;
            for (int64_t c = 0; c < nc; ++c)
                if (!pair_shaped(out.chunks[(size_t)c])) pick_distances(c, true, hist);
        }
    }
    for (int64_t c = 0; c < nc; ++c) {
        out.chunks[(size_t)c].val_off = slots;
        slots += (int64_t)out.chunks[(size_t)c].nd * CHUNK_ROWS;
    }
    auto plane_of = [&](int64_t c, int64_t d) {  // plane of distance d in chunk c, 0 = none
        const SymxChunk &h = out.chunks[(size_t)c];
        for (int j = 1; j < h.nd; ++j)
            if (h.d[j - 1] == d) return j;
        return 0;
    };
    for (int64_t c = 0; c < nc; ++c) {  // where the lower entries of a chunk find their twins
        SymxChunk &h = out.chunks[(size_t)c];
        for (int j = 1; j < 4; ++j)
            for (int w = 0; w < 2; ++w) {
                h.lo_base[j - 1][w] = -1;
                if (j >= h.nd) continue;
                const int64_t first = c * CHUNK_ROWS - h.d[j - 1];
                const int64_t cs = (first >= 0 ? first / CHUNK_ROWS : -((-first + CHUNK_ROWS - 1) / CHUNK_ROWS)) + w;
                if (cs < 0 || cs >= nc) continue;
                const int pj = plane_of(cs, h.d[j - 1]);
                if (pj) h.lo_base[j - 1][w] = out.chunks[(size_t)cs].val_off + (int64_t)pj * CHUNK_ROWS;
            }
    }
    out.mask.assign((size_t)nc * CHUNK_ROWS + 16, 0);
    out.map.assign((size_t)slots + 2, -1);
    // pass 2: planes first (a lower entry is planar only if its twin has taken a plane slot), rows independent
    parallel_ranges(n_rows, 1 << 14, [&](int64_t ra, int64_t rb) {
        for (ogl_label r = (ogl_label)ra; r < (ogl_label)rb; ++r) {
            const int64_t c = r / CHUNK_ROWS;
            const SymxChunk &h = out.chunks[(size_t)c];
            unsigned m = 0;
            for (ogl_label k = row_ptrs[r]; k < row_ptrs[r + 1]; ++k) {
                const int64_t d = (int64_t)cols[k] - r;
                if (d < 0) continue;
                const int j = d == 0 ? 0 : plane_of(c, d);
                if (d > 0 && !j) continue;
                if (m & (1u << (3 + j))) continue;  // a later entry of the same column: explicit
                out.map[(size_t)(h.val_off + (int64_t)j * CHUNK_ROWS + r % CHUNK_ROWS)] = k;
                m |= 1u << (3 + j);
            }
            out.mask[(size_t)r] = (uint8_t)m;
        }
    });
    // pass 3: lower entries and the explicit ones, chunk after chunk (the explicit lists are appended in row order)
    for (int64_t c = 0; c < nc; ++c) {
        SymxChunk &h = out.chunks[(size_t)c];
        const ogl_label r0 = (ogl_label)(c * CHUNK_ROWS), r1 = (ogl_label)std::min<int64_t>(n_rows, (c + 1) * (int64_t)CHUNK_ROWS);
        const size_t ex_begin = out.ex_cols.size();
        std::vector<int32_t> rp(CHUNK_ROWS + 1, 0);
        for (ogl_label r = r0; r < r1; ++r) {
            unsigned m = out.mask[(size_t)r];
            rp[(size_t)(r - r0)] = (int32_t)out.ex_cols.size();
            for (ogl_label k = row_ptrs[r]; k < row_ptrs[r + 1]; ++k) {
                const int64_t col = cols[k], d = col - r;
                bool planar = false;
                if (d >= 0) {
                    const int j = d == 0 ? 0 : plane_of(c, d);
                    planar = (d == 0 || j) && out.map[(size_t)(h.val_off + (int64_t)j * CHUNK_ROWS + r % CHUNK_ROWS)] == k;
                    if (planar) ++upper_entries;
                } else {
                    const int j = plane_of(c, -d);
                    if (j && !(m & (1u << (3 - j)))) {
                        const int64_t first = (int64_t)r0 - h.d[j - 1];
                        const int64_t cs0 = first >= 0 ? first / CHUNK_ROWS : -((-first + CHUNK_ROWS - 1) / CHUNK_ROWS);
                        const int w = (int)(col / CHUNK_ROWS - cs0);
                        const int64_t base = (w == 0 || w == 1) ? h.lo_base[j - 1][w] : -1;
                        // the twin (col, r) must be the entry its row keeps in that plane
                        if (base >= 0) {
                            const int32_t tk = out.map[(size_t)(base + col % CHUNK_ROWS)];
                            planar = tk >= 0 && cols[tk] == r;
                        }
                        if (planar) m |= 1u << (3 - j);
                    }
                }
                if (planar) {
                    ++out.planar;
                } else {
                    out.ex_cols.push_back((int32_t)col);
                    out.ex_map.push_back(k);
                    out.ex_lrow.push_back((int32_t)(r - r0));
                    m |= SYMX_EXTRAS_BIT;
                }
            }
            out.mask[(size_t)r] = (uint8_t)m;
        }
        for (ogl_label lr = r1 - r0; lr <= CHUNK_ROWS; ++lr) rp[(size_t)lr] = (int32_t)out.ex_cols.size();
        if (out.ex_cols.size() > ex_begin) {
            h.ex_rp_off = (int32_t)out.ex_rowptr.size();
            h.ex_begin = (int32_t)ex_begin;
            h.ex_count = (int32_t)(out.ex_cols.size() - ex_begin);
            out.ex_rowptr.insert(out.ex_rowptr.end(), rp.begin(), rp.end());
            // simple chunk: every row has at most one explicit entry before its first planar entry and at most one
            // behind its last, none in between (the coupling across a block face) -- anything else: h.merge
            for (ogl_label r = r0; r < r1 && !h.merge; ++r) {
                const unsigned m = out.mask[(size_t)r];
                if (!(m & SYMX_EXTRAS_BIT)) continue;
                int64_t first = INT64_MAX, last = INT64_MIN;  // columns of the row's first / last planar entry
                for (int j = 1; j < h.nd; ++j) {
                    if (m & (1u << (3 - j))) first = std::min<int64_t>(first, (int64_t)r - h.d[j - 1]);
                    if (m & (1u << (3 + j))) last = std::max<int64_t>(last, (int64_t)r + h.d[j - 1]);
                }
                if (m & (1u << 3)) {
                    first = std::min<int64_t>(first, r);
                    last = std::max<int64_t>(last, r);
                }
                for (int j = 1; j < h.nd; ++j) {
                    if (m & (1u << (3 - j))) last = std::max<int64_t>(last, (int64_t)r - h.d[j - 1]);
                    if (m & (1u << (3 + j))) first = std::min<int64_t>(first, (int64_t)r + h.d[j - 1]);
                }
                int ahead = 0, behind = 0;
                for (int32_t e = rp[(size_t)(r - r0)]; e < rp[(size_t)(r - r0) + 1]; ++e) {
                    const int64_t col = out.ex_cols[(size_t)e];
                    if (first == INT64_MAX ? col < r : col < first) {
                        ++ahead;
                    } else if (first == INT64_MAX || col > last) {
                        ++behind;
                        out.ex_lrow[(size_t)e] |= SYMX_BEHIND_BIT;  // (what the lean kernel goes by)
                    } else {
                        h.merge = 1;  // (a repeated column of a planar entry included)
                    }
                }
                if (ahead > 1 || behind > 1) h.merge = 1;
            }
        }
    }
    out.all_fast = true;
    for (const SymxChunk &h : out.chunks) out.all_fast = out.all_fast && pair_shaped(h);
    if ((double)out.planar < SYMX_MIN_PLANAR * (double)nnz) return false;
    if ((double)slots > SYM_MAX_PADDING * 1.5 * (double)upper_entries + 8.0 * CHUNK_ROWS) return false;
    return true;
}

}  // namespace ogl

// ---------------------------------------------------------------------------------------
// C ABI: pure host logic
// ---------------------------------------------------------------------------------------
using namespace ogl;

extern "C" void ogl_host_init_local_sparsity(ogl_label nrows, ogl_label upper_nnz, int is_symmetric,
                                             const ogl_label *upper, const ogl_label *lower,
                                             ogl_label *rows, ogl_label *cols, ogl_label *permute)
{
    init_local_sparsity(nrows, upper_nnz, is_symmetric != 0, upper, lower, rows, cols, permute);
}

// HostMatrixFreeFunctions.C:21-30.  In the reference `scale * (pos >= upper_nnz) ? a : b` binds as
// `(scale * (pos >= upper_nnz)) ? a : b`: scale is a truth value, never a factor.  Kept, so that a
// case run with reorderOnHost gives the reference's numbers.
extern "C" void ogl_host_symmetric_update(ogl_label total_nnz, ogl_label upper_nnz,
                                          const ogl_label *permute, ogl_scalar scale,
                                          const ogl_scalar *diag, const ogl_scalar *upper,
                                          ogl_scalar *out)
{
    for (ogl_label i = 0; i < total_nnz; ++i) {
        const ogl_label pos = permute[i];
        const bool pick_diag = (scale * static_cast<ogl_scalar>(pos >= upper_nnz)) != 0.0;
        out[i] = pick_diag ? diag[pos - upper_nnz] : upper[pos];
    }
}

// HostMatrixFreeFunctions.C:32-56
extern "C" void ogl_host_symmetric_update_w_interface(ogl_label total_nnz, ogl_label diag_nnz,
                                                      ogl_label upper_nnz, const ogl_label *permute,
                                                      ogl_scalar scale, const ogl_scalar *diag,
                                                      const ogl_scalar *upper,
                                                      const ogl_scalar *iface, ogl_scalar *out)
{
    const ogl_label d0 = upper_nnz, i0 = upper_nnz + diag_nnz;
    for (ogl_label i = 0; i < total_nnz; ++i) {
        const ogl_label pos = permute[i];
        const ogl_scalar v = pos < d0 ? upper[pos] : (pos < i0 ? diag[pos - d0] : iface[pos - i0]);
        out[i] = scale * v;
    }
}

// HostMatrixFreeFunctions.C:58-82
extern "C" void ogl_host_non_symmetric_update_w_interface(
    ogl_label total_nnz, ogl_label diag_nnz, ogl_label upper_nnz, const ogl_label *permute,
    ogl_scalar scale, const ogl_scalar *diag, const ogl_scalar *upper, const ogl_scalar *lower,
    const ogl_scalar *iface, ogl_scalar *out)
{
    const ogl_label l0 = upper_nnz, d0 = 2 * upper_nnz, i0 = 2 * upper_nnz + diag_nnz;
    for (ogl_label i = 0; i < total_nnz; ++i) {
        const ogl_label pos = permute[i];
        const ogl_scalar v = pos < l0   ? upper[pos]
                             : pos < d0 ? lower[pos - l0]
                             : pos < i0 ? diag[pos - d0]
                                        : iface[pos - i0];
        out[i] = scale * v;
    }
}

// HostMatrixFreeFunctions.C:85-102
extern "C" void ogl_host_non_symmetric_update(ogl_label total_nnz, ogl_label upper_nnz,
                                              const ogl_label *permute, ogl_scalar scale,
                                              const ogl_scalar *diag, const ogl_scalar *upper,
                                              const ogl_scalar *lower, ogl_scalar *out)
{
    const ogl_label l0 = upper_nnz, d0 = 2 * upper_nnz;
    for (ogl_label i = 0; i < total_nnz; ++i) {
        const ogl_label pos = permute[i];
        out[i] = scale * (pos < l0 ? upper[pos] : pos < d0 ? lower[pos - l0] : diag[pos - d0]);
    }
}

extern "C" int ogl_host_pattern(const ogl_ldu_view *ldu, ogl_matrix_dims *dims,
                                ogl_label *local_rows, ogl_label *local_cols,
                                ogl_label *local_ldu_mapping, ogl_label *nl_rows,
                                ogl_label *nl_cols, ogl_label *nl_ldu_mapping,
                                ogl_label *target_ids, ogl_label *target_sizes,
                                ogl_label *send_idxs)
{
    if (!ldu || !dims) return fail(OGL_ERR_INVALID, "ldu/dims is NULL");
    HostPattern p;
    if (int rc = build_host_pattern(*ldu, p)) return rc;
    dims->n_rows = p.n_rows;
    dims->local_nnz = p.local_nnz;
    dims->non_local_nnz = p.non_local_nnz;
    dims->n_halo = p.non_local_nnz;
    dims->n_neighbours = (ogl_label)p.target_ids.size();
    dims->n_send = (ogl_label)p.send_idxs.size();
    auto put = [](ogl_label *dst, const std::vector<ogl_label> &src) {
        if (dst) std::copy(src.begin(), src.end(), dst);
    };
    put(local_rows, p.rows);
    put(local_cols, p.cols);
    put(local_ldu_mapping, p.ldu_mapping);
    put(nl_rows, p.nl_rows);
    put(nl_cols, p.nl_cols);
    put(nl_ldu_mapping, p.nl_ldu_mapping);
    put(target_ids, p.target_ids);
    put(target_sizes, p.target_sizes);
    put(send_idxs, p.send_idxs);
    return OGL_OK;
}

extern "C" uint64_t ogl_host_addressing_fingerprint(const ogl_ldu_view *ldu)
{
    return ldu ? addressing_fingerprint(*ldu) : 0;
}

extern "C" int ogl_host_rcm(ogl_label n_rows, const ogl_label *row_ptrs, const ogl_label *cols,
                            ogl_label *new_id)
{
    if (n_rows < 0 || !row_ptrs || !new_id || (n_rows > 0 && !cols))
        return fail(OGL_ERR_INVALID, "NULL argument");
    std::vector<ogl_label> v;
    rcm_order(n_rows, row_ptrs, cols, v);
    std::copy(v.begin(), v.end(), new_id);
    return OGL_OK;
}

extern "C" int ogl_host_hilbert_order(ogl_label n_cells, const ogl_scalar *centres, ogl_label *new_id)
{
    if (n_cells < 0 || !new_id || (n_cells > 0 && !centres)) return fail(OGL_ERR_INVALID, "NULL argument");
    std::vector<ogl_label> v;
    hilbert_order(n_cells, centres, v);
    std::copy(v.begin(), v.end(), new_id);
    return OGL_OK;
}

extern "C" double ogl_host_gather_sector_ratio(ogl_label n_rows, const ogl_label *row_ptrs,
                                               const ogl_label *cols, const ogl_label *new_id)
{
    if (n_rows <= 0 || !row_ptrs || !cols) return 0.0;
    if (!new_id) return gather_sector_ratio(n_rows, row_ptrs, cols, nullptr, nullptr);
    std::vector<ogl_label> old_of((size_t)n_rows);
    for (ogl_label c = 0; c < n_rows; ++c) old_of[(size_t)new_id[c]] = c;
    return gather_sector_ratio(n_rows, row_ptrs, cols, new_id, old_of.data());
}

extern "C" int ogl_host_pattern_renumbered(const ogl_ldu_view *ldu, int32_t mode,
                                           int32_t compress_indices, ogl_matrix_dims *dims,
                                           ogl_label *local_rows, ogl_label *local_cols,
                                           ogl_label *local_ldu_mapping, ogl_label *nl_rows,
                                           ogl_label *nl_cols, ogl_label *nl_ldu_mapping,
                                           ogl_label *target_ids, ogl_label *target_sizes,
                                           ogl_label *send_idxs, ogl_label *new_id)
{
    if (!ldu || !dims) return fail(OGL_ERR_INVALID, "ldu/dims is NULL");
    HostPattern p;
    if (int rc = build_host_pattern(*ldu, p)) return rc;
    RenumberReport rep;
    NumberingHooks hooks;
    hooks.centres = ldu->cell_centres;  // (NULL: reverse Cuthill-McKee is the only candidate)
    if (int rc = choose_numbering(p, mode, compress_indices != 0, nullptr, nullptr, rep, &hooks)) return rc;
    dims->n_rows = p.n_rows;
    dims->local_nnz = p.local_nnz;
    dims->non_local_nnz = p.non_local_nnz;
    dims->n_halo = p.non_local_nnz;
    dims->n_neighbours = (ogl_label)p.target_ids.size();
    dims->n_send = (ogl_label)p.send_idxs.size();
    auto put = [](ogl_label *dst, const std::vector<ogl_label> &src) {
        if (dst) std::copy(src.begin(), src.end(), dst);
    };
    put(local_rows, p.rows);
    put(local_cols, p.cols);
    put(local_ldu_mapping, p.ldu_mapping);
    put(nl_rows, p.nl_rows);
    put(nl_cols, p.nl_cols);
    put(nl_ldu_mapping, p.nl_ldu_mapping);
    put(target_ids, p.target_ids);
    put(target_sizes, p.target_sizes);
    put(send_idxs, p.send_idxs);
    if (new_id) {
        if (p.renumbered())
            put(new_id, p.new_id);
        else
            for (ogl_label c = 0; c < p.n_rows; ++c) new_id[c] = c;
    }
    return p.renumbered() ? 1 : 0;
}

// StoppingCriterion.H:197-209
extern "C" void ogl_host_adapt_criterion(const ogl_config *cfg, ogl_label prev_solve_iters,
                                         ogl_scalar prev_rel_cost, ogl_label *min_iter,
                                         ogl_label *frequency)
{
    ogl_label mi = cfg->min_iter, fr = cfg->eval_frequency;
    if (!cfg->export_res && prev_solve_iters > 0 && cfg->adapt_min_iter && prev_rel_cost > 0) {
        mi = static_cast<ogl_label>(prev_solve_iters * cfg->relaxation_factor);
        const double alpha =
            std::sqrt(1.0 / (prev_solve_iters * (1.0 - cfg->relaxation_factor)) * prev_rel_cost);
        fr = std::min<ogl_label>(cfg->norm_eval_limit,
                                 std::max<ogl_label>(1, static_cast<ogl_label>(1 / alpha)));
    }
    *min_iter = mi;
    *frequency = fr;
}

extern "C" int ogl_host_sell_check(ogl_label n_rows, const ogl_label *row_ptrs, const ogl_label *cols,
                                   int64_t stats[8])
{
    if (n_rows < 0 || !row_ptrs || !stats || (n_rows > 0 && !cols))
        return fail(OGL_ERR_INVALID, "NULL argument");
    for (int i = 0; i < 8; ++i) stats[i] = 0;
    SellLayout L;
    if (n_rows == 0 || !build_sell_layout(n_rows, row_ptrs, cols, L)) return OGL_OK;
    // decode exactly as k_spmv_sell does: thread t of chunk c owns rows c*CHUNK_ROWS + 2t, +1
    for (size_t c = 0; c < L.chunks.size(); ++c) {
        const SellChunk &h = L.chunks[c];
        const bool d16_mode = h.mode() == SELL_MODE_DELTA16, c32_mode = h.mode() == SELL_MODE_COL32;
        const bool pat_mode = h.mode() == SELL_MODE_PATTERN;
        const int width = h.width(), code_stride = h.code_stride();
        if (h.code_off % 16 != 0 || h.mode() < SELL_MODE_PATTERN || h.mode() > SELL_MODE_COL32)
            return fail(OGL_ERR_STATE, "chunk %zu: bad header", c);
        if (pat_mode && (width <= 0 || h.dict_len() % width != 0 || h.dict_len() > SELL_TABLE_INTS ||
                         h.dict_len() / width > 256))
            return fail(OGL_ERR_STATE, "chunk %zu: bad pattern table", c);
        if (h.mode() == SELL_MODE_OFFSET8 && h.dict_len() > SELL_MAX_DICT)
            return fail(OGL_ERR_STATE, "chunk %zu: bad dictionary", c);
        for (int t = 0; t < BLOCK; ++t)
            for (int which = 0; which < ROWS_PER_THREAD; ++which) {
                const int64_t row = (int64_t)c * CHUNK_ROWS + t * ROWS_PER_THREAD + which;
                const uint8_t *code = L.codes.data() + h.code_off + (int64_t)t * code_stride;
                ogl_label k = row < n_rows ? row_ptrs[row] : 0;
                // the planes hold the first width() entries of a row; the rest is in the spill list
                const ogl_label k_end = row < n_rows ? std::min(row_ptrs[row + 1], row_ptrs[row] + width) : 0;
                int64_t run = row + h.dict_off;  // delta16: running column
                // the lane loads up to the longer of its two rows (lengths in front of the codes)
                int lane_len = width;
                if (!pat_mode) {
                    const uint8_t *lens = L.codes.data() + h.code_off - SELL_LEN_BYTES + t * ROWS_PER_THREAD;
                    if (lens[which] != k_end - k)
                        return fail(OGL_ERR_STATE, "row %ld: length byte %d, %d entries in the planes", (long)row,
                                    (int)lens[which], (int)(k_end - k));
                    lane_len = std::max<int>(lens[0], lens[1]);
                }
                for (int s = 0; s < width; ++s) {
                    const int32_t m = L.map[(size_t)(h.val_off + (int64_t)s * CHUNK_ROWS +
                                                     t * ROWS_PER_THREAD + which)];
                    int64_t col = -1;  // decoded column, -1 = padding
                    if (s >= lane_len) {
                        // never loaded
                    } else if (d16_mode) {
                        uint16_t v;
                        std::memcpy(&v, L.codes.data() + h.code_off + ((int64_t)(s / SELL_D16_GROUP) * BLOCK + t) * 16 +
                                            ((s % SELL_D16_GROUP) * ROWS_PER_THREAD + which) * 2, 2);
                        if (v != 0xFFFFu) {
                            run += v;
                            col = run;
                        }
                    } else if (c32_mode) {
                        int32_t v;
                        std::memcpy(&v, L.codes.data() + h.code_off + ((int64_t)(s / SELL_C32_GROUP) * BLOCK + t) * 16 +
                                            ((s % SELL_C32_GROUP) * ROWS_PER_THREAD + which) * 4, 4);
                        col = v;
                    } else if (pat_mode) {
                        const int32_t id = code[which];
                        if ((id + 1) * width > h.dict_len())
                            return fail(OGL_ERR_STATE, "row %ld: pattern id out of range", (long)row);
                        const int32_t off = L.dict[(size_t)h.dict_off + (size_t)id * width + s];
                        if (off != SELL_PAD_OFFSET) col = row + off;
                    } else {
                        const uint8_t cd = code[ROWS_PER_THREAD * s + which];
                        if (cd != 255 && cd >= h.dict_len())
                            return fail(OGL_ERR_STATE, "row %ld: code out of range", (long)row);
                        if (cd != 255) col = row + L.dict[(size_t)h.dict_off + cd];
                    }
                    if (col < 0) {
                        if (m != -1) return fail(OGL_ERR_STATE, "row %ld: padding slot is mapped", (long)row);
                        continue;
                    }
                    if (k >= k_end || m != k || col != cols[k])
                        return fail(OGL_ERR_STATE, "row %ld slot %d decodes wrongly", (long)row, s);
                    ++k;
                }
                if (k != k_end) return fail(OGL_ERR_STATE, "row %ld lost entries", (long)row);
            }
    }
    {   // every row is either fully in the planes (within its wavefront's width) or continues in the spill list
        size_t sp = 0;
        for (ogl_label r = 0; r < n_rows; ++r) {
            const SellChunk &h = L.chunks[(size_t)(r / CHUNK_ROWS)];
            const int len = row_ptrs[r + 1] - row_ptrs[r], cap = h.width();
            if (std::min(len, cap) > h.wave_width((r % CHUNK_ROWS) / SELL_WAVE_ROWS))
                return fail(OGL_ERR_STATE, "row %d is longer than its wavefront's width", r);
            if (len <= cap) continue;
            if (sp >= L.spill_rows.size() || L.spill_rows[sp] != r ||
                L.spill_ptrs[sp + 1] - L.spill_ptrs[sp] != len - cap)
                return fail(OGL_ERR_STATE, "row %d: spill list does not hold its tail", r);
            for (int i = 0; i < len - cap; ++i)
                if (L.spill_cols[(size_t)L.spill_ptrs[sp] + i] != cols[row_ptrs[r] + cap + i] ||
                    L.spill_map[(size_t)L.spill_ptrs[sp] + i] != row_ptrs[r] + cap + i)
                    return fail(OGL_ERR_STATE, "row %d: spilled entry %d is wrong", r, i);
            ++sp;
        }
        if (sp != L.spill_rows.size()) return fail(OGL_ERR_STATE, "spill list holds rows that do not spill");
        // the kernel finds a chunk's spilled rows through spill_chunk_ptr
        if (L.spill_chunk_ptr.size() != L.chunks.size() + 1 || L.spill_chunk_ptr.front() != 0 ||
            L.spill_chunk_ptr.back() != (int32_t)L.spill_rows.size())
            return fail(OGL_ERR_STATE, "spill_chunk_ptr does not cover the spill list");
        for (size_t c = 0; c < L.chunks.size(); ++c)
            for (int32_t j = L.spill_chunk_ptr[c]; j < L.spill_chunk_ptr[c + 1]; ++j)
                if (j < 0 || j >= (int32_t)L.spill_rows.size() || L.spill_rows[(size_t)j] / CHUNK_ROWS != (int32_t)c)
                    return fail(OGL_ERR_STATE, "spill_chunk_ptr: row %d is not in chunk %zu", L.spill_rows[(size_t)j], c);
    }
    stats[0] = 1;
    stats[1] = L.n_slots;
    stats[2] = (int64_t)L.dict.size();
    stats[3] = (int64_t)L.codes.size() - 16;
    stats[4] = L.n_delta16;
    stats[5] = L.n_col32;
    stats[6] = L.read_slots;
    stats[7] = (int64_t)L.spill_cols.size();
    return OGL_OK;
}

// Half storage of a symmetric matrix (build_sym_layout): builds it and decodes it exactly as k_spmv_sym
// does.  stats: [0] qualifies, [1] planes (nd), [2..5] the distances d[0..3], [6] plane slots, [7] slots in use
extern "C" int ogl_host_sym_check(ogl_label n_rows, const ogl_label *row_ptrs, const ogl_label *cols,
                                  int64_t stats[8])
{
    if (n_rows < 0 || !row_ptrs || !stats || (n_rows > 0 && !cols))
        return fail(OGL_ERR_INVALID, "NULL argument");
    for (int i = 0; i < 8; ++i) stats[i] = 0;
    SymLayout L;
    if (n_rows == 0 || !build_sym_layout(n_rows, row_ptrs, cols, L)) return OGL_OK;
    const int nd = L.nd;
    int64_t used = 0;
    for (int32_t m : L.map) used += m >= 0;
    for (ogl_label r = 0; r < n_rows; ++r) {
        // the kernel's walk over row r: lower entries from the furthest to the nearest, the diagonal, upper entries
        ogl_label k = row_ptrs[r];
        const unsigned m = L.mask[(size_t)r];
        for (int j = nd - 1; j >= 1; --j)
            if ((m >> (nd - 1 - j)) & 1u) {
                const int64_t rr = (int64_t)r - L.d[j];
                if (rr < 0) return fail(OGL_ERR_STATE, "row %d: lower entry before row 0", r);
                const int32_t src = L.map[(size_t)((rr / CHUNK_ROWS) * nd + j) * CHUNK_ROWS + (size_t)(rr % CHUNK_ROWS)];
                // its value is the twin's: position of (rr, r) in the CSR
                if (k >= row_ptrs[r + 1] || cols[k] != rr || src < 0 || cols[src] != r || src < row_ptrs[rr] ||
                    src >= row_ptrs[rr + 1])
                    return fail(OGL_ERR_STATE, "row %d: lower entry at -%d decodes wrongly", r, L.d[j]);
                ++k;
            }
        for (int j = 0; j < nd; ++j)
            if ((m >> (nd - 1 + j)) & 1u) {
                const int32_t src = L.map[(size_t)((r / CHUNK_ROWS) * nd + j) * CHUNK_ROWS + (size_t)(r % CHUNK_ROWS)];
                if (k >= row_ptrs[r + 1] || cols[k] != r + L.d[j] || src != k)
                    return fail(OGL_ERR_STATE, "row %d: upper entry at +%d decodes wrongly", r, L.d[j]);
                ++k;
            }
        if (k != row_ptrs[r + 1]) return fail(OGL_ERR_STATE, "row %d lost entries", r);
    }
    stats[0] = 1;
    stats[1] = nd;
    for (int j = 0; j < nd; ++j) stats[2 + j] = L.d[j];
    stats[6] = (int64_t)L.map.size() - 2;
    stats[7] = used;
    return OGL_OK;
}

// Half storage with exceptions (build_symx_layout): builds it and walks every row the way k_spmv_symx does,
// merging the explicit entries by column.  stats: [0] qualifies, [1] plane slots, [2] entries served from planes,
// [3] explicit entries, [4] chunks with explicit entries, [5] chunks
extern "C" int ogl_host_symx_check(ogl_label n_rows, const ogl_label *row_ptrs, const ogl_label *cols,
                                   int64_t stats[8])
{
    if (n_rows < 0 || !row_ptrs || !stats || (n_rows > 0 && !cols))
        return fail(OGL_ERR_INVALID, "NULL argument");
    for (int i = 0; i < 8; ++i) stats[i] = 0;
    SymxLayout L;
    const bool ok = n_rows > 0 && build_symx_layout(n_rows, row_ptrs, cols, L);
    if (n_rows == 0) return OGL_OK;
    int64_t ex_chunks = 0;
    for (ogl_label r = 0; r < n_rows; ++r) {
        const int64_t c = r / CHUNK_ROWS;
        const SymxChunk &h = L.chunks[(size_t)c];
        const unsigned m = L.mask[(size_t)r];
        int32_t ek = 0, ee = 0;
        if (m & SYMX_EXTRAS_BIT) {
            if (h.ex_rp_off < 0) return fail(OGL_ERR_STATE, "row %d: explicit entries but the chunk has no row pointers", r);
            ek = L.ex_rowptr[(size_t)h.ex_rp_off + r % CHUNK_ROWS];
            ee = L.ex_rowptr[(size_t)h.ex_rp_off + r % CHUNK_ROWS + 1];
        }
        ogl_label k = row_ptrs[r];
        auto take = [&](int64_t col, int32_t csr_pos, const char *what) -> int {
            if (k >= row_ptrs[r + 1] || cols[k] != col || (csr_pos >= 0 && csr_pos != k))
                return fail(OGL_ERR_STATE, "row %d: %s entry at column %ld decodes wrongly", r, what, (long)col);
            ++k;
            return OGL_OK;
        };
        auto flush_before = [&](int64_t limit) -> int {
            while (ek < ee && L.ex_cols[(size_t)ek] < limit) {
                if (int rc = take(L.ex_cols[(size_t)ek], L.ex_map[(size_t)ek], "explicit")) return rc;
                ++ek;
            }
            return OGL_OK;
        };
        for (int j = h.nd - 1; j >= 1; --j)
            if (m & (1u << (3 - j))) {
                const int64_t rr = (int64_t)r - h.d[j - 1];
                if (int rc = flush_before(rr)) return rc;
                const int64_t first = c * CHUNK_ROWS - h.d[j - 1];
                const int64_t cs0 = first >= 0 ? first / CHUNK_ROWS : -((-first + CHUNK_ROWS - 1) / CHUNK_ROWS);
                const int w = (int)(rr / CHUNK_ROWS - cs0);
                if (rr < 0 || w < 0 || w > 1 || h.lo_base[j - 1][w] < 0)
                    return fail(OGL_ERR_STATE, "row %d: lower entry at -%d has no plane to be read from", r, h.d[j - 1]);
                const int32_t tk = L.map[(size_t)(h.lo_base[j - 1][w] + rr % CHUNK_ROWS)];
                if (tk < 0 || cols[tk] != r || tk < row_ptrs[rr] || tk >= row_ptrs[rr + 1])
                    return fail(OGL_ERR_STATE, "row %d: twin of the lower entry at -%d is not in its plane", r, h.d[j - 1]);
                if (int rc = take(rr, -1, "lower")) return rc;
            }
        for (int j = 0; j < h.nd; ++j)
            if (m & (1u << (3 + j))) {
                const int64_t col = (int64_t)r + (j ? h.d[j - 1] : 0);
                if (int rc = flush_before(col)) return rc;
                if (int rc = take(col, L.map[(size_t)(h.val_off + (int64_t)j * CHUNK_ROWS + r % CHUNK_ROWS)], "plane")) return rc;
            }
        if (int rc = flush_before(INT64_MAX)) return rc;
        if (k != row_ptrs[r + 1] || ek != ee) return fail(OGL_ERR_STATE, "row %d lost entries", r);
    }
    for (const SymxChunk &h : L.chunks) ex_chunks += h.ex_rp_off >= 0;
    stats[0] = ok ? 1 : 0;
    stats[1] = (int64_t)L.map.size() - 2;
    stats[2] = L.planar;
    stats[3] = (int64_t)L.ex_cols.size();
    stats[4] = ex_chunks;
    stats[5] = (int64_t)L.chunks.size();
    stats[6] = L.all_fast ? 1 : 0;
    for (const SymxChunk &h : L.chunks) stats[7] += h.ex_rp_off >= 0 && h.merge != 0;  // chunks the general kernel takes
    return OGL_OK;
}

extern "C" void ogl_config_default(ogl_config *c)
{
    *c = ogl_config{};
    c->solver = OGL_SOLVER_CG;
    c->preconditioner = OGL_PRECOND_NONE;
    c->max_block_size = 1;
    c->caching = 0;
    c->tolerance = 1e-6;
    c->rel_tol = 1e-6;
    c->max_iter = 1000;
    c->min_iter = 0;
    c->eval_frequency = 1;
    c->norm_eval_limit = 100;
    c->relaxation_factor = 0.6;
    c->adapt_min_iter = 1;
    c->matrix_format = OGL_FORMAT_COO;
    c->regenerate = 0;
    c->update_sys_matrix = 1;
    c->update_rhs = 1;
    c->update_init_guess = 0;
    c->scaling = 1.0;
    c->reorder_on_host = 0;
    c->export_res = 0;
    c->verbose = 0;
    c->force_host_buffer = 0;
    c->ranks_per_gpu = 1;
    c->krylov_dim = 0;
    c->sparsity_power = 1;
    c->profile_kernels = 0;
    c->compress_indices = 1;
    c->renumber = 2;
    c->symmetric_half = 1;
}
