/*
 * ogl_oracle.c -- CPU restatement ("oracle") of the hpsim/OGL hot path.
 *
 * TEST INFRASTRUCTURE ONLY (see ogl_oracle.h).  Plain C, sequential, compiled with
 * -ffp-contract=off and without -ffast-math so every product and sum rounds once, as
 * Ginkgo's reference executor does on a baseline x86-64 build.
 *
 * PARITY: LDU conversion pinned by the reference's gtest vectors; Krylov arithmetic
 * "parity unpinned" (Ginkgo absent) -- see the header.
 *
 * Three reduction modes (orc_set_reduction): SEQUENTIAL = the reference executor's left-to-right sums (the
 * semantics under test), BLOCKED = the HIP kernels' fixed tree (bit-for-bit checks of the device), EXACT = error-free
 * transformations rounded once (the arbiter between the two; pinned against rational arithmetic in
 * tests/test_oracle_exact.py).
 */
#include "ogl_oracle.h"

#include <math.h>
#include <stdlib.h>
#include <string.h>

#ifdef _OPENMP
#include <omp.h>
#endif

/* ------------------------------------------------------------------ */
/* small helpers                                                        */
/* ------------------------------------------------------------------ */

static void *xmalloc(size_t n) {
    void *p = malloc(n ? n : 1);
    if (!p) abort();
    return p;
}

/* Stable counting sort of idx[0..m) by key[idx[i]] in [0, nkeys). */
static void stable_sort_by_key(orc_label m, const orc_label *key, orc_label nkeys,
                               const orc_label *idx_in, orc_label *idx_out) {
    int64_t *count = (int64_t *)calloc((size_t)nkeys + 1, sizeof(int64_t));
    if (!count) abort();
    for (orc_label i = 0; i < m; ++i) count[key[idx_in[i]] + 1]++;
    for (orc_label k = 0; k < nkeys; ++k) count[k + 1] += count[k];
    for (orc_label i = 0; i < m; ++i) idx_out[count[key[idx_in[i]]]++] = idx_in[i];
    free(count);
}

/* order[] = indices 0..m-1 sorted by (row[i], col[i]), ties by index (stable). */
static void sort_row_col(orc_label m, const orc_label *row, const orc_label *col, orc_label nkeys,
                         orc_label *order) {
    orc_label *tmp0 = (orc_label *)xmalloc(sizeof(orc_label) * (size_t)m);
    orc_label *tmp1 = (orc_label *)xmalloc(sizeof(orc_label) * (size_t)m);
    for (orc_label i = 0; i < m; ++i) tmp0[i] = i;
    stable_sort_by_key(m, col, nkeys, tmp0, tmp1); /* LSD: minor key first */
    stable_sort_by_key(m, row, nkeys, tmp1, order);
    free(tmp0);
    free(tmp1);
}

/* ------------------------------------------------------------------ */
/* HostMatrixFreeFunctions.C                                            */
/* ------------------------------------------------------------------ */

/* HostMatrixFreeFunctions.C:105-201.
 * Upper entry of face f sits at (row=lower[f], col=upper[f]); its transpose (the lower
 * entry) at (row=upper[f], col=lower[f]).  Both lists are ordered by (row, col)
 * (:120-148), then every row emits its lower entries, the diagonal and its upper
 * entries (:159-200).  permute points into [upper | lower (asym) | diag]. */
void orc_init_local_sparsity(orc_label nrows, orc_label upper_nnz, int is_symmetric,
                             const orc_label *upper, const orc_label *lower, orc_label *rows,
                             orc_label *cols, orc_label *permute) {
    const orc_label after_neighbours = is_symmetric ? upper_nnz : 2 * upper_nnz; /* :116 */
    orc_label *ord_u = (orc_label *)xmalloc(sizeof(orc_label) * (size_t)upper_nnz);
    orc_label *ord_l = (orc_label *)xmalloc(sizeof(orc_label) * (size_t)upper_nnz);
    sort_row_col(upper_nnz, lower, upper, nrows, ord_u); /* rows of the upper triangle */
    sort_row_col(upper_nnz, upper, lower, nrows, ord_l); /* rows of the lower triangle */

    orc_label e = 0, uc = 0, lc = 0;
    for (orc_label row = 0; row < nrows; ++row) {
        while (lc < upper_nnz && upper[ord_l[lc]] == row) { /* :161-176 */
            const orc_label f = ord_l[lc++];
            rows[e] = row;
            cols[e] = lower[f];
            permute[e] = is_symmetric ? f : upper_nnz + f;
            ++e;
        }
        rows[e] = row; /* :179-182 */
        cols[e] = row;
        permute[e] = after_neighbours + row;
        ++e;
        while (uc < upper_nnz && lower[ord_u[uc]] == row) { /* :185-199 */
            const orc_label f = ord_u[uc++];
            rows[e] = row;
            cols[e] = upper[f];
            permute[e] = f;
            ++e;
        }
    }
    free(ord_u);
    free(ord_l);
}

/* HostMatrixFreeFunctions.C:21-30.  The reference expression
 *     scale * (pos >= upper_nnz) ? diag[...] : upper[pos]
 * parses as (scale * (pos >= upper_nnz)) ? diag : upper, so `scale` only acts as a
 * truth value: out = diag when scale*(cond) != 0, else upper.  Kept as is. */
void orc_symmetric_update(orc_label total_nnz, orc_label upper_nnz, const orc_label *permute,
                          orc_scalar scale, const orc_scalar *diag, const orc_scalar *upper,
                          orc_scalar *out) {
    for (orc_label i = 0; i < total_nnz; ++i) {
        const orc_label pos = permute[i];
        const orc_scalar selector = scale * (orc_scalar)(pos >= upper_nnz);
        out[i] = (selector != 0.0) ? diag[pos - upper_nnz] : upper[pos];
    }
}

/* HostMatrixFreeFunctions.C:32-56 */
void orc_symmetric_update_w_interface(orc_label total_nnz, orc_label diag_nnz,
                                      orc_label upper_nnz, const orc_label *permute,
                                      orc_scalar scale, const orc_scalar *diag,
                                      const orc_scalar *upper, const orc_scalar *iface,
                                      orc_scalar *out) {
    for (orc_label i = 0; i < total_nnz; ++i) {
        const orc_label pos = permute[i];
        orc_scalar v;
        if (pos < upper_nnz)
            v = upper[pos];
        else if (pos < upper_nnz + diag_nnz)
            v = diag[pos - upper_nnz];
        else
            v = iface[pos - upper_nnz - diag_nnz];
        out[i] = scale * v;
    }
}

/* HostMatrixFreeFunctions.C:58-82 */
void orc_non_symmetric_update_w_interface(orc_label total_nnz, orc_label diag_nnz,
                                          orc_label upper_nnz, const orc_label *permute,
                                          orc_scalar scale, const orc_scalar *diag,
                                          const orc_scalar *upper, const orc_scalar *lower,
                                          const orc_scalar *iface, orc_scalar *out) {
    for (orc_label i = 0; i < total_nnz; ++i) {
        const orc_label pos = permute[i];
        orc_scalar v;
        if (pos < upper_nnz)
            v = upper[pos];
        else if (pos < 2 * upper_nnz)
            v = lower[pos - upper_nnz];
        else if (pos < 2 * upper_nnz + diag_nnz)
            v = diag[pos - 2 * upper_nnz];
        else
            v = iface[pos - 2 * upper_nnz - diag_nnz];
        out[i] = scale * v;
    }
}

/* HostMatrixFreeFunctions.C:85-102 */
void orc_non_symmetric_update(orc_label total_nnz, orc_label upper_nnz,
                              const orc_label *permute, orc_scalar scale,
                              const orc_scalar *diag, const orc_scalar *upper,
                              const orc_scalar *lower, orc_scalar *out) {
    for (orc_label i = 0; i < total_nnz; ++i) {
        const orc_label pos = permute[i];
        if (pos < upper_nnz)
            out[i] = scale * upper[pos];
        else if (pos < 2 * upper_nnz)
            out[i] = scale * lower[pos - upper_nnz];
        else
            out[i] = scale * diag[pos - 2 * upper_nnz];
    }
}

/* ------------------------------------------------------------------ */
/* HostMatrix.C -- interfaces                                           */
/* ------------------------------------------------------------------ */

/* HostMatrix.C:159-178 */
orc_label orc_count_interface_nnz(const orc_iface *ifaces, orc_label n_ifaces,
                                  int proc_interfaces) {
    orc_label ctr = 0;
    for (orc_label i = 0; i < n_ifaces; ++i) {
        const int is_proc = ifaces[i].kind == ORC_IFACE_PROCESSOR;
        if (proc_interfaces ? is_proc : !is_proc) ctr += ifaces[i].size;
    }
    return ctr;
}

/* HostMatrix.C:180-207 */
void orc_collect_interface_coeffs(const orc_iface *ifaces, orc_label n_ifaces, int local,
                                  orc_scalar *out) {
    orc_label k = 0;
    for (orc_label i = 0; i < n_ifaces; ++i) {
        const int is_proc = ifaces[i].kind == ORC_IFACE_PROCESSOR;
        if (local ? !is_proc : is_proc)
            for (orc_label f = 0; f < ifaces[i].size; ++f) out[k++] = ifaces[i].bou_coeffs[f];
    }
    for (orc_label j = 0; j < k; ++j) out[j] = out[j] * -1.0; /* :204 */
}

/* HostMatrix.C:251-306: std::map<neighbProcNo, faceCells...> walked in key order. */
orc_label orc_create_communication_pattern(const orc_iface *ifaces, orc_label n_ifaces,
                                           orc_label *target_ids, orc_label *target_sizes,
                                           orc_label *send_idxs) {
    orc_label n_procs = 0;
    /* distinct neighbour ranks, ascending */
    for (orc_label i = 0; i < n_ifaces; ++i) {
        if (ifaces[i].kind != ORC_IFACE_PROCESSOR) continue;
        const orc_label p = ifaces[i].neighb_proc;
        orc_label pos = 0;
        int found = 0;
        while (pos < n_procs && target_ids[pos] <= p) {
            if (target_ids[pos] == p) found = 1;
            ++pos;
        }
        if (found) continue;
        for (orc_label j = n_procs; j > pos; --j) target_ids[j] = target_ids[j - 1];
        target_ids[pos] = p;
        ++n_procs;
    }
    orc_label k = 0;
    for (orc_label q = 0; q < n_procs; ++q) {
        orc_label cnt = 0;
        for (orc_label i = 0; i < n_ifaces; ++i) {
            if (ifaces[i].kind != ORC_IFACE_PROCESSOR || ifaces[i].neighb_proc != target_ids[q])
                continue;
            for (orc_label f = 0; f < ifaces[i].size; ++f) send_idxs[k++] = ifaces[i].face_cells[f];
            cnt += ifaces[i].size;
        }
        target_sizes[q] = cnt;
    }
    return n_procs;
}

/* HostMatrix.C:412-466 */
void orc_init_non_local_sparsity(const orc_iface *ifaces, orc_label n_ifaces, orc_label *rows,
                                 orc_label *cols, orc_label *permute) {
    const orc_label nnz = orc_count_interface_nnz(ifaces, n_ifaces, 1);
    orc_label *row_of = (orc_label *)xmalloc(sizeof(orc_label) * (size_t)nnz);
    orc_label *id = (orc_label *)xmalloc(sizeof(orc_label) * (size_t)nnz);
    orc_label *ord = (orc_label *)xmalloc(sizeof(orc_label) * (size_t)nnz);
    orc_label ctr = 0, max_row = 0;
    for (orc_label i = 0; i < n_ifaces; ++i) { /* :418-431 */
        if (ifaces[i].kind != ORC_IFACE_PROCESSOR) continue;
        for (orc_label f = 0; f < ifaces[i].size; ++f) {
            row_of[ctr] = ifaces[i].face_cells[f];
            if (row_of[ctr] > max_row) max_row = row_of[ctr];
            id[ctr] = ctr;
            ++ctr;
        }
    }
    stable_sort_by_key(nnz, row_of, max_row + 1, id, ord); /* :452-457 (by row only) */
    for (orc_label e = 0; e < nnz; ++e) { /* :459-465 */
        rows[e] = row_of[ord[e]];
        cols[e] = ord[e];
        permute[e] = ord[e];
    }
    free(row_of);
    free(id);
    free(ord);
}

/* HostMatrix.C:468-589 */
void orc_init_local_sparsity_pattern(orc_label nrows, orc_label upper_nnz, int is_symmetric,
                                     const orc_label *upper, const orc_label *lower,
                                     const orc_iface *ifaces, orc_label n_ifaces, orc_label *rows,
                                     orc_label *cols, orc_label *permute) {
    const orc_label after_neighbours = is_symmetric ? upper_nnz : 2 * upper_nnz; /* :500 */
    const orc_label local_nnz = nrows + 2 * upper_nnz;
    const orc_label iface_nnz = orc_count_interface_nnz(ifaces, n_ifaces, 0);
    orc_init_local_sparsity(nrows, upper_nnz, is_symmetric, upper, lower, rows, cols, permute);
    if (!iface_nnz) return; /* :506 */

    /* collect_local_interface_indices :385-410 -- only cyclic patches contribute */
    orc_label *irow = (orc_label *)xmalloc(sizeof(orc_label) * (size_t)iface_nnz);
    orc_label *icol = (orc_label *)xmalloc(sizeof(orc_label) * (size_t)iface_nnz);
    orc_label m = 0;
    for (orc_label i = 0; i < n_ifaces; ++i) {
        if (ifaces[i].kind != ORC_IFACE_CYCLIC) continue;
        const orc_label *nbr_cells = ifaces[ifaces[i].neighb_patch].face_cells; /* :324 */
        for (orc_label f = 0; f < ifaces[i].size; ++f) {
            irow[m] = ifaces[i].face_cells[f];
            icol[m] = nbr_cells[f];
            ++m;
        }
    }
    orc_label *ord = (orc_label *)xmalloc(sizeof(orc_label) * (size_t)m);
    sort_row_col(m, irow, icol, nrows, ord); /* :510-515 */

    orc_label *rows_c = (orc_label *)xmalloc(sizeof(orc_label) * (size_t)local_nnz);
    orc_label *cols_c = (orc_label *)xmalloc(sizeof(orc_label) * (size_t)local_nnz);
    orc_label *perm_c = (orc_label *)xmalloc(sizeof(orc_label) * (size_t)local_nnz);
    memcpy(rows_c, rows, sizeof(orc_label) * (size_t)local_nnz);
    memcpy(cols_c, cols, sizeof(orc_label) * (size_t)local_nnz);
    memcpy(perm_c, permute, sizeof(orc_label) * (size_t)local_nnz);

    orc_label cur = 0, tot = 0;
    for (orc_label k = 0; k < m; ++k) { /* :539-576 */
        const orc_label idx = ord[k], r = irow[idx], c = icol[idx];
        while (cur < local_nnz &&
               (rows_c[cur] < r || (rows_c[cur] == r && cols_c[cur] <= c))) {
            rows[tot] = rows_c[cur];
            cols[tot] = cols_c[cur];
            permute[tot] = perm_c[cur];
            ++cur;
            ++tot;
        }
        rows[tot] = r;
        cols[tot] = c;
        permute[tot] = after_neighbours + nrows + idx; /* :574 */
        ++tot;
    }
    while (cur < local_nnz) { /* :580-585 */
        rows[tot] = rows_c[cur];
        cols[tot] = cols_c[cur];
        permute[tot] = perm_c[cur];
        ++cur;
        ++tot;
    }
    free(irow);
    free(icol);
    free(ord);
    free(rows_c);
    free(cols_c);
    free(perm_c);
}

/* HostMatrix.C:634-704 */
void orc_update_local_matrix_data(orc_label nrows, orc_label upper_nnz, int is_symmetric,
                                  const orc_scalar *diag, const orc_scalar *upper,
                                  const orc_scalar *lower, const orc_iface *ifaces,
                                  orc_label n_ifaces, const orc_label *permute,
                                  orc_label total_nnz, orc_scalar *out) {
    const orc_label iface_nnz = orc_count_interface_nnz(ifaces, n_ifaces, 0);
    const orc_label diag_start = is_symmetric ? upper_nnz : 2 * upper_nnz; /* :666 */
    const size_t src_n = (size_t)diag_start + (size_t)nrows + (size_t)iface_nnz;
    orc_scalar *src = (orc_scalar *)xmalloc(sizeof(orc_scalar) * src_n);
    memcpy(src, upper, sizeof(orc_scalar) * (size_t)upper_nnz);                   /* :644-650 */
    if (!is_symmetric)
        memcpy(src + upper_nnz, lower, sizeof(orc_scalar) * (size_t)upper_nnz);    /* :653-660 */
    memcpy(src + diag_start, diag, sizeof(orc_scalar) * (size_t)nrows);           /* :663-669 */
    if (iface_nnz)
        orc_collect_interface_coeffs(ifaces, n_ifaces, 1, src + diag_start + nrows); /* :672-682 */
    for (orc_label i = 0; i < total_nnz; ++i) out[i] = src[permute[i]];           /* :700-703 */
    free(src);
}

/* HostMatrix.C:608-633 */
void orc_update_local_matrix_data_host(orc_label nrows, orc_label upper_nnz, int is_symmetric,
                                       orc_scalar scaling, const orc_scalar *diag,
                                       const orc_scalar *upper, const orc_scalar *lower,
                                       const orc_iface *ifaces, orc_label n_ifaces,
                                       const orc_label *permute, orc_label total_nnz,
                                       orc_scalar *out) {
    const orc_label iface_nnz = orc_count_interface_nnz(ifaces, n_ifaces, 0);
    if (iface_nnz) {
        orc_scalar *cc = (orc_scalar *)xmalloc(sizeof(orc_scalar) * (size_t)iface_nnz);
        orc_collect_interface_coeffs(ifaces, n_ifaces, 1, cc);
        if (is_symmetric)
            orc_symmetric_update_w_interface(total_nnz, nrows, upper_nnz, permute, scaling, diag,
                                             upper, cc, out);
        else
            orc_non_symmetric_update_w_interface(total_nnz, nrows, upper_nnz, permute, scaling,
                                                 diag, upper, lower, cc, out);
        free(cc);
    } else if (is_symmetric) {
        orc_symmetric_update(total_nnz, upper_nnz, permute, scaling, diag, upper, out);
    } else {
        orc_non_symmetric_update(total_nnz, upper_nnz, permute, scaling, diag, upper, lower, out);
    }
}

/* HostMatrix.C:708-732 */
void orc_update_non_local_matrix_data(const orc_iface *ifaces, orc_label n_ifaces,
                                      const orc_label *permute, orc_label nnz, orc_scalar *out) {
    orc_scalar *cc = (orc_scalar *)xmalloc(sizeof(orc_scalar) * (size_t)nnz);
    orc_collect_interface_coeffs(ifaces, n_ifaces, 0, cc);
    for (orc_label e = 0; e < nnz; ++e) out[e] = cc[permute[e]];
    free(cc);
}

/* ------------------------------------------------------------------ */
/* Arithmetic                                                           */
/* ------------------------------------------------------------------ */

void orc_rowptr_from_rows(orc_label nrows, orc_label nnz, const orc_label *rows,
                          orc_label *rowptr) {
    for (orc_label r = 0; r <= nrows; ++r) rowptr[r] = 0;
    for (orc_label e = 0; e < nnz; ++e) rowptr[rows[e] + 1]++;
    for (orc_label r = 0; r < nrows; ++r) rowptr[r + 1] += rowptr[r];
}

static int g_reduce_mode = ORC_REDUCE_SEQUENTIAL;
static orc_label g_chunk_rows = 512;

/* ORC_REDUCE_EXACT: every sum of products / sum of terms is carried as an unevaluated pair (hi, lo) with error-free
 * transformations -- TwoSum (Knuth) for the additions, TwoProduct (Dekker) for the products -- and rounded to
 * double once at the end (Ogita, Rump, Oishi: "Accurate sum and dot product", SIAM J. Sci. Comput. 26 (2005), Sum2 /
 * Dot2: the result is as accurate as if computed in twice the working precision, i.e. the correctly rounded exact
 * sum unless the condition number exceeds ~1e16).  The arbiter between two summation orders: NOT a reference
 * semantic, NOT what the device computes. */
typedef struct {
    orc_scalar hi, lo;
} acc2;
static inline void acc2_add(acc2 *a, orc_scalar v) { /* TwoSum(hi, v) */
    const orc_scalar s = a->hi + v;
    const orc_scalar bb = s - a->hi;
    const orc_scalar e = (a->hi - (s - bb)) + (v - bb);
    a->hi = s;
    a->lo += e;
}
static inline void acc2_add_prod(acc2 *a, orc_scalar x, orc_scalar y) { /* TwoProduct, then TwoSum */
    /* TwoProduct after Dekker / Veltkamp (split at 27 bits): p + pe = x * y exactly, barring over- / underflow; no
     * fma(), which a baseline x86-64 build would take from libm one call per entry */
    const orc_scalar p = x * y;
    const orc_scalar cx = 134217729.0 * x, xh = cx - (cx - x), xl = x - xh;
    const orc_scalar cy = 134217729.0 * y, yh = cy - (cy - y), yl = y - yh;
    const orc_scalar pe = xl * yl - (((p - xh * yh) - xl * yh) - xh * yl);
    acc2_add(a, p);
    a->lo += pe;
}
static inline orc_scalar acc2_value(const acc2 *a) { return a->hi + a->lo; }

void orc_spmv(orc_label n, const orc_label *rowptr, const orc_label *cols, const orc_scalar *vals,
              const orc_scalar *x, orc_scalar *y) {
    if (g_reduce_mode == ORC_REDUCE_EXACT) {
        for (orc_label row = 0; row < n; ++row) {
            acc2 a = {0.0, 0.0};
            for (orc_label k = rowptr[row]; k < rowptr[row + 1]; ++k) acc2_add_prod(&a, vals[k], x[cols[k]]);
            y[row] = acc2_value(&a);
        }
        return;
    }
    for (orc_label row = 0; row < n; ++row) {
        orc_scalar sum = 0.0;
        for (orc_label k = rowptr[row]; k < rowptr[row + 1]; ++k) sum += vals[k] * x[cols[k]];
        y[row] = sum;
    }
}

void orc_spmv_adv(orc_label n, const orc_label *rowptr, const orc_label *cols,
                  const orc_scalar *vals, orc_scalar alpha, const orc_scalar *x, orc_scalar beta,
                  orc_scalar *y) {
    if (g_reduce_mode == ORC_REDUCE_EXACT) { /* (alpha = +-1, beta = 1 at every call site: alpha * val is exact) */
        for (orc_label row = 0; row < n; ++row) {
            acc2 a = {0.0, 0.0};
            acc2_add_prod(&a, y[row], beta);
            for (orc_label k = rowptr[row]; k < rowptr[row + 1]; ++k) acc2_add_prod(&a, alpha * vals[k], x[cols[k]]);
            y[row] = acc2_value(&a);
        }
        return;
    }
    for (orc_label row = 0; row < n; ++row) {
        orc_scalar sum = y[row] * beta;
        for (orc_label k = rowptr[row]; k < rowptr[row + 1]; ++k)
            sum += alpha * vals[k] * x[cols[k]];
        y[row] = sum;
    }
}

void orc_set_reduction(int mode, orc_label chunk_rows) {
    g_reduce_mode = mode;
    if (chunk_rows > 0) g_chunk_rows = chunk_rows;
}

/* The fixed reduction tree of the HIP kernels (ogl_amd/csrc/device_common.hpp, block_sum / reduce_partials):
 * a chunk is chunk_rows consecutive rows handled by 256 threads; thread t owns the
 * chunk_rows/256 consecutive rows starting at t*chunk_rows/256, summed in order from 0;
 * 64-lane xor tree (offsets 32,16,8,4,2,1); the 4 wave sums are added left to right.  The
 * per-chunk partials are then summed by one block of 1024 threads: thread t takes partials
 * t, t+1024, ... in order, xor tree per wave, then the 16 wave sums left to right. */
#define ORC_BLOCK 256
#define ORC_WAVE 64
static orc_scalar block_tree(orc_scalar *acc /* [ORC_BLOCK] */) {
    orc_scalar wsum[ORC_BLOCK / ORC_WAVE];
    for (int w = 0; w < ORC_BLOCK / ORC_WAVE; ++w) {
        orc_scalar *v = acc + w * ORC_WAVE, t[ORC_WAVE];
        for (int off = ORC_WAVE / 2; off >= 1; off >>= 1) {
            for (int l = 0; l < ORC_WAVE; ++l) t[l] = v[l] + v[l ^ off];
            memcpy(v, t, sizeof(t));
        }
        wsum[w] = v[0];
    }
    orc_scalar s = wsum[0];
    for (int w = 1; w < ORC_BLOCK / ORC_WAVE; ++w) s += wsum[w];
    return s;
}

typedef orc_scalar (*term_fn)(const orc_scalar *a, const orc_scalar *b, orc_label i);
static orc_scalar term_dot(const orc_scalar *a, const orc_scalar *b, orc_label i) {
    return a[i] * b[i];
}
static orc_scalar term_abs(const orc_scalar *a, const orc_scalar *b, orc_label i) {
    (void)b;
    return fabs(a[i]);
}
static orc_scalar term_id(const orc_scalar *a, const orc_scalar *b, orc_label i) {
    (void)b;
    return a[i];
}

/* finaliser: one block of ORC_FIN_BLOCK threads (16 waves of 64) */
#define ORC_FIN_BLOCK 1024
static orc_scalar reduce_blocked_partials(orc_label m, const orc_scalar *part) {
    orc_scalar acc[ORC_FIN_BLOCK], wsum[ORC_FIN_BLOCK / ORC_WAVE];
    for (int t = 0; t < ORC_FIN_BLOCK; ++t) {
        orc_scalar s = 0.0;
        for (orc_label i = t; i < m; i += ORC_FIN_BLOCK) s += part[i];
        acc[t] = s;
    }
    for (int w = 0; w < ORC_FIN_BLOCK / ORC_WAVE; ++w) {
        orc_scalar *v = acc + w * ORC_WAVE, t[ORC_WAVE];
        for (int off = ORC_WAVE / 2; off >= 1; off >>= 1) {
            for (int l = 0; l < ORC_WAVE; ++l) t[l] = v[l] + v[l ^ off];
            memcpy(v, t, sizeof(t));
        }
        wsum[w] = v[0];
    }
    orc_scalar s = wsum[0];
    for (int w = 1; w < ORC_FIN_BLOCK / ORC_WAVE; ++w) s += wsum[w];
    return s;
}

static orc_scalar reduce_terms(orc_label n, const orc_scalar *a, const orc_scalar *b, term_fn f) {
    if (g_reduce_mode == ORC_REDUCE_EXACT) {
        acc2 s = {0.0, 0.0};
        if (f == term_dot)
            for (orc_label i = 0; i < n; ++i) acc2_add_prod(&s, a[i], b[i]);
        else
            for (orc_label i = 0; i < n; ++i) acc2_add(&s, f(a, b, i));
        return acc2_value(&s);
    }
    if (g_reduce_mode == ORC_REDUCE_SEQUENTIAL) {
        orc_scalar s = 0.0;
        for (orc_label i = 0; i < n; ++i) s += f(a, b, i);
        return s;
    }
    const orc_label R = g_chunk_rows;
    const orc_label n_chunks = (orc_label)(((int64_t)n + R - 1) / R);
    orc_scalar *part = (orc_scalar *)xmalloc(sizeof(orc_scalar) * (size_t)(n_chunks ? n_chunks : 1));
    for (orc_label c = 0; c < n_chunks; ++c) {
        orc_scalar acc[ORC_BLOCK];
        const orc_label rpt = R / ORC_BLOCK; /* consecutive rows owned by one thread */
        for (int t = 0; t < ORC_BLOCK; ++t) {
            orc_scalar s = 0.0;
            for (orc_label j = 0; j < rpt; ++j) {
                const int64_t i = (int64_t)c * R + (int64_t)t * rpt + j;
                if (i < n) s += f(a, b, (orc_label)i);
            }
            acc[t] = s;
        }
        part[c] = block_tree(acc);
    }
    const orc_scalar s = reduce_blocked_partials(n_chunks, part);
    free(part);
    return s;
}

orc_scalar orc_dot(orc_label n, const orc_scalar *a, const orc_scalar *b) {
    return reduce_terms(n, a, b, term_dot);
}
orc_scalar orc_norm1(orc_label n, const orc_scalar *a) { return reduce_terms(n, a, 0, term_abs); }
orc_scalar orc_sum(orc_label n, const orc_scalar *a) { return reduce_terms(n, a, 0, term_id); }

/* ------------------------------------------------------------------ */
/* Distributed pieces                                                   */
/* ------------------------------------------------------------------ */

static void global_sum(const orc_dist_matrix *A, orc_scalar *v, orc_label n) {
    if (A->allreduce) A->allreduce(A->user, v, n);
}

/* distributed::Matrix::apply [UPSTREAM]: y = A_local x_local, then
 * y = 1 * A_non_local * recv + 1 * y with the gathered neighbour values. */
void orc_dist_spmv(const orc_dist_matrix *A, const orc_scalar *x, orc_scalar *y) {
    orc_spmv(A->n, A->rowptr, A->cols, A->vals, x, y);
    if (A->n_halo > 0) {
        orc_scalar *send = (orc_scalar *)xmalloc(sizeof(orc_scalar) * (size_t)A->n_send);
        orc_scalar *recv = (orc_scalar *)xmalloc(sizeof(orc_scalar) * (size_t)A->n_halo);
        for (orc_label i = 0; i < A->n_send; ++i) send[i] = x[A->send_idxs[i]];
        A->exchange(A->user, send, recv);
        orc_spmv_adv(A->n, A->nl_rowptr, A->nl_cols, A->nl_vals, 1.0, recv, 1.0, y);
        free(send);
        free(recv);
    }
}

/* r = b - A x as Ginkgo computes it: r = b; r = (-1) A x + 1 r (advanced apply), local
 * part first, then the non-local part accumulates onto the stored local result. */
static void dist_residual(const orc_dist_matrix *A, const orc_scalar *b, const orc_scalar *x,
                          orc_scalar *r) {
    memcpy(r, b, sizeof(orc_scalar) * (size_t)A->n);
    orc_spmv_adv(A->n, A->rowptr, A->cols, A->vals, -1.0, x, 1.0, r);
    if (A->n_halo > 0) {
        orc_scalar *send = (orc_scalar *)xmalloc(sizeof(orc_scalar) * (size_t)A->n_send);
        orc_scalar *recv = (orc_scalar *)xmalloc(sizeof(orc_scalar) * (size_t)A->n_halo);
        for (orc_label i = 0; i < A->n_send; ++i) send[i] = x[A->send_idxs[i]];
        A->exchange(A->user, send, recv);
        orc_spmv_adv(A->n, A->nl_rowptr, A->nl_cols, A->nl_vals, -1.0, recv, 1.0, r);
        free(send);
        free(recv);
    }
}

static orc_scalar dist_dot(const orc_dist_matrix *A, const orc_scalar *a, const orc_scalar *b) {
    orc_scalar s = orc_dot(A->n, a, b);
    global_sum(A, &s, 1);
    return s;
}

static orc_scalar dist_norm1(const orc_dist_matrix *A, const orc_scalar *a) {
    orc_scalar s = orc_norm1(A->n, a);
    global_sum(A, &s, 1);
    return s;
}

/* ------------------------------------------------------------------ */
/* Stopping criterion                                                   */
/* ------------------------------------------------------------------ */

/* StoppingCriterion.H:197-209 */
void orc_adapt_criterion(orc_label min_iter, orc_label frequency, int export_res,
                         orc_label prev_solve_iters, int adapt_min_iter,
                         orc_scalar relaxation_factor, orc_label norm_eval_limit,
                         orc_scalar prev_rel_cost, orc_label *min_iter_out,
                         orc_label *frequency_out) {
    if (!export_res && prev_solve_iters > 0 && adapt_min_iter && prev_rel_cost > 0) {
        min_iter = (orc_label)(prev_solve_iters * relaxation_factor);
        const orc_scalar alpha =
            sqrt(1.0 / (prev_solve_iters * (1.0 - relaxation_factor)) * prev_rel_cost);
        orc_label f = (orc_label)(1 / alpha);
        if (f < 1) f = 1;
        frequency = norm_eval_limit < f ? norm_eval_limit : f;
    }
    *min_iter_out = min_iter;
    *frequency_out = frequency;
}

/* StoppingCriterion.C:11-69.
 *   xAvg = mean(x)                                     (:17-19; distributed compute_mean
 *          [UPSTREAM]: local mean scaled by local_n/global_n, then summed over ranks)
 *   Axref = A * (xAvg * 1)                             (:24-29)
 *   t = b - Axref ; nf = sum(|t - r| + |t|) + SMALL    (:53-68)                        */
orc_scalar orc_compute_normfactor(const orc_dist_matrix *A, const orc_scalar *r,
                                  const orc_scalar *x, const orc_scalar *b) {
    const orc_label n = A->n;
    orc_scalar mean = orc_sum(n, x);
    mean /= (orc_scalar)n;
    mean *= (orc_scalar)n / (orc_scalar)A->global_n;
    global_sum(A, &mean, 1);

    orc_scalar *xavg = (orc_scalar *)xmalloc(sizeof(orc_scalar) * (size_t)n);
    orc_scalar *w = (orc_scalar *)xmalloc(sizeof(orc_scalar) * (size_t)n);
    for (orc_label i = 0; i < n; ++i) xavg[i] = mean;
    orc_dist_spmv(A, xavg, w); /* w = Axref */
    for (orc_label i = 0; i < n; ++i) {
        orc_scalar t = b[i];
        t -= 1.0 * w[i];                 /* b_sub_xstar = b - Axref        (:53-54) */
        const orc_scalar part2 = fabs(t); /* norm_part2                    (:56)    */
        t -= 1.0 * r[i];                 /* b_sub_xstar -= r               (:58)    */
        t = fabs(t);                     /*                                (:59)    */
        t += 1.0 * part2;                /*                                (:61)    */
        w[i] = t;
    }
    orc_scalar nf = dist_norm1(A, w);    /* (:62-66) */
    free(xavg);
    free(w);
    return nf + ORC_SMALL;               /* (:68) */
}

/* StoppingCriterion.C:71-151.  Returns 1 when the solver has to stop. */
static int criterion_check(const orc_dist_matrix *A, const orc_criterion *c,
                           orc_criterion_state *st, const orc_scalar *residual,
                           const orc_scalar *x, const orc_scalar *b) {
    if (st->iter > 0 && st->iter < c->min_iter) { /* :77-81 */
        st->iter += 1;
        return 0;
    }
    if (st->iter % c->frequency != 0) { /* :84-87 */
        st->iter += 1;
        return 0;
    }
    orc_scalar residual_norm = dist_norm1(A, residual); /* :92-97 */
    st->n_evals += 1;
    int result = 0;
    if (st->iter == 0) { /* :102-111 */
        st->norm_factor = orc_compute_normfactor(A, residual, x, b);
        st->init_residual = residual_norm / st->norm_factor;
    }
    residual_norm /= st->norm_factor;                                  /* :113 */
    if (c->export_res && st->history) st->history[st->iter] = residual_norm; /* :115-117 */
    st->residual = residual_norm;                                      /* :119 */
    if (st->iter >= c->max_iter) result = 1;                           /* :124 */
    if (residual_norm < c->tolerance) result = 1;                      /* :128 */
    if (c->rel_tol > 0 && residual_norm < c->rel_tol * st->init_residual) result = 1; /* :132 */
    st->iter += 1;                                                     /* :143 */
    return result;
}

static void criterion_reset(orc_criterion_state *st) {
    orc_scalar *h = st->history;
    memset(st, 0, sizeof(*st));
    st->history = h;
    st->norm_factor = 1.0; /* StoppingCriterion.H:136 */
}

/* ------------------------------------------------------------------ */
/* Preconditioner                                                       */
/* ------------------------------------------------------------------ */

void orc_jacobi_generate_scalar(orc_label n, const orc_label *rowptr, const orc_label *cols,
                                const orc_scalar *vals, orc_scalar *inv_diag) {
    for (orc_label row = 0; row < n; ++row) {
        orc_scalar d = 0.0;
        for (orc_label k = rowptr[row]; k < rowptr[row + 1]; ++k)
            if (cols[k] == row) {
                d = vals[k]; /* first match, like Csr::extract_diagonal [UPSTREAM] */
                break;
            }
        inv_diag[row] = 1.0 / d;
    }
}

/* ---- block Jacobi, max_block_size > 1 ([UPSTREAM] gko::preconditioner::Jacobi) ----------
 * find_blocks: natural blocks = runs of consecutive rows with the same column pattern (capped
 * at max_block_size), then adjacent natural blocks are agglomerated while the merged size stays
 * <= max_block_size.  Each diagonal block is inverted by Gauss-Jordan elimination with partial
 * (row) pivoting; apply is a dense block mat-vec, rows summed left to right from 0. */
static int same_pattern(const orc_label *rowptr, const orc_label *cols, orc_label a, orc_label b) {
    const orc_label la = rowptr[a + 1] - rowptr[a], lb = rowptr[b + 1] - rowptr[b];
    if (la != lb) return 0;
    for (orc_label k = 0; k < la; ++k)
        if (cols[rowptr[a] + k] != cols[rowptr[b] + k]) return 0;
    return 1;
}

orc_label orc_jacobi_find_blocks(orc_label n, const orc_label *rowptr, const orc_label *cols,
                                 orc_label max_block_size, orc_label *block_ptrs) {
    if (n == 0) {
        block_ptrs[0] = 0;
        return 0;
    }
    /* natural blocks */
    orc_label *nat = (orc_label *)xmalloc(sizeof(orc_label) * ((size_t)n + 1));
    orc_label n_nat = 1, cur = 1;
    nat[0] = 0;
    for (orc_label i = 1; i < n; ++i) {
        if (same_pattern(rowptr, cols, i - 1, i) && cur < max_block_size) {
            ++cur;
        } else {
            nat[n_nat++] = i;
            cur = 1;
        }
    }
    nat[n_nat] = n;
    /* agglomerate */
    orc_label nb = 1;
    block_ptrs[0] = 0;
    cur = nat[1] - nat[0];
    for (orc_label i = 1; i < n_nat; ++i) {
        const orc_label bs = nat[i + 1] - nat[i];
        if (cur + bs <= max_block_size) {
            cur += bs;
        } else {
            block_ptrs[nb++] = nat[i];
            cur = bs;
        }
    }
    block_ptrs[nb] = n;
    free(nat);
    return nb;
}

/* In-place inverse of the bs x bs row-major block `a` (leading dimension ld). */
static void invert_block(orc_label bs, orc_scalar *a, orc_label ld) {
    orc_label perm[64];
    for (orc_label k = 0; k < bs; ++k) perm[k] = k;
    for (orc_label k = 0; k < bs; ++k) {
        orc_label piv = k;
        orc_scalar best = fabs(a[k * ld + k]);
        for (orc_label i = k + 1; i < bs; ++i)
            if (fabs(a[i * ld + k]) > best) {
                best = fabs(a[i * ld + k]);
                piv = i;
            }
        if (piv != k) {
            for (orc_label j = 0; j < bs; ++j) {
                const orc_scalar t = a[k * ld + j];
                a[k * ld + j] = a[piv * ld + j];
                a[piv * ld + j] = t;
            }
            const orc_label t = perm[k];
            perm[k] = perm[piv];
            perm[piv] = t;
        }
        const orc_scalar d = a[k * ld + k];
        a[k * ld + k] = 1.0;
        for (orc_label j = 0; j < bs; ++j) a[k * ld + j] /= d; /* pivot row; a_kk = 1/d */
        for (orc_label i = 0; i < bs; ++i) {
            if (i == k) continue;
            const orc_scalar f = a[i * ld + k];
            a[i * ld + k] = 0.0;
            for (orc_label j = 0; j < bs; ++j) a[i * ld + j] -= f * a[k * ld + j];
        }
    }
    /* undo the row swaps: column perm[k] of the inverse is column k computed above */
    orc_scalar tmp[64 * 64];
    for (orc_label i = 0; i < bs; ++i)
        for (orc_label j = 0; j < bs; ++j) tmp[i * bs + perm[j]] = a[i * ld + j];
    for (orc_label i = 0; i < bs; ++i)
        for (orc_label j = 0; j < bs; ++j) a[i * ld + j] = tmp[i * bs + j];
}

void orc_jacobi_generate_blocks(orc_label n, const orc_label *rowptr, const orc_label *cols,
                                const orc_scalar *vals, orc_label n_blocks,
                                const orc_label *block_ptrs, orc_label stride, orc_scalar *blocks) {
    (void)n;
    for (orc_label b = 0; b < n_blocks; ++b) {
        const orc_label r0 = block_ptrs[b], bs = block_ptrs[b + 1] - r0;
        orc_scalar *a = blocks + (size_t)b * stride * stride;
        for (orc_label i = 0; i < stride * stride; ++i) a[i] = 0.0;
        for (orc_label i = 0; i < bs; ++i)
            for (orc_label k = rowptr[r0 + i]; k < rowptr[r0 + i + 1]; ++k) {
                const orc_label c = cols[k] - r0;
                if (c >= 0 && c < bs) a[i * stride + c] = vals[k];
            }
        invert_block(bs, a, stride);
    }
}

/* ---- ISAI ([UPSTREAM] gko::preconditioner::Isai) --------------------------------------------- */
static orc_scalar csr_entry(const orc_label *rowptr, const orc_label *cols, const orc_scalar *vals,
                            orc_label r, orc_label c) {
    for (orc_label k = rowptr[r]; k < rowptr[r + 1]; ++k)
        if (cols[k] == c) return vals[k];
    return 0.0;
}

/* Gaussian elimination with partial pivoting + back substitution; solution returned in rhs. */
static void solve_dense(orc_label bs, orc_scalar *a, orc_label ld, orc_scalar *rhs) {
    for (orc_label k = 0; k < bs; ++k) {
        orc_label piv = k;
        orc_scalar best = fabs(a[k * ld + k]);
        for (orc_label i = k + 1; i < bs; ++i)
            if (fabs(a[i * ld + k]) > best) {
                best = fabs(a[i * ld + k]);
                piv = i;
            }
        if (piv != k) {
            for (orc_label j = 0; j < bs; ++j) {
                const orc_scalar t = a[k * ld + j];
                a[k * ld + j] = a[piv * ld + j];
                a[piv * ld + j] = t;
            }
            const orc_scalar t = rhs[k];
            rhs[k] = rhs[piv];
            rhs[piv] = t;
        }
        for (orc_label i = k + 1; i < bs; ++i) {
            const orc_scalar f = a[i * ld + k] / a[k * ld + k];
            for (orc_label j = k + 1; j < bs; ++j) a[i * ld + j] -= f * a[k * ld + j];
            rhs[i] -= f * rhs[k];
        }
    }
    for (orc_label i = bs - 1; i >= 0; --i) {
        orc_scalar t = rhs[i];
        for (orc_label j = i + 1; j < bs; ++j) t -= a[i * ld + j] * rhs[j];
        rhs[i] = t / a[i * ld + i];
    }
}

/* Rows wider than ORC_ISAI_NARROW_ROW (below): the same elimination, but the back substitution walks the COLUMNS
 * from the last to the first -- x_r = rhs_r / a_rr, then rhs_i -= a_ir x_r for every i < r -- so that the updates of
 * one step are independent of each other (the product solves such a row with a whole workgroup; the row-wise walk
 * above is one dependent chain of bs^2 / 2 operations).  Same exact solve, the subtractions of a row in descending
 * instead of ascending column order.  [UPSTREAM] Ginkgo hands rows beyond its in-kernel limit of 32 to an "excess
 * system" solved ITERATIVELY (block-Jacobi-preconditioned GMRES to a residual reduction of 1e-6, isai.cpp
 * excess_solver_factory / excess_solver_reduction): an approximation of the solution computed here. */
static void solve_dense_wide(orc_label bs, orc_scalar *a, orc_label ld, orc_scalar *rhs) {
    for (orc_label k = 0; k < bs; ++k) {
        orc_label piv = k;
        orc_scalar best = fabs(a[k * ld + k]);
        for (orc_label i = k + 1; i < bs; ++i)
            if (fabs(a[i * ld + k]) > best) {
                best = fabs(a[i * ld + k]);
                piv = i;
            }
        if (piv != k) {
            for (orc_label j = 0; j < bs; ++j) {
                const orc_scalar t = a[k * ld + j];
                a[k * ld + j] = a[piv * ld + j];
                a[piv * ld + j] = t;
            }
            const orc_scalar t = rhs[k];
            rhs[k] = rhs[piv];
            rhs[piv] = t;
        }
        for (orc_label i = k + 1; i < bs; ++i) {
            const orc_scalar f = a[i * ld + k] / a[k * ld + k];
            for (orc_label j = k + 1; j < bs; ++j) a[i * ld + j] -= f * a[k * ld + j];
            rhs[i] -= f * rhs[k];
        }
    }
    for (orc_label r = bs - 1; r >= 0; --r) {
        rhs[r] = rhs[r] / a[r * ld + r];
        for (orc_label i = 0; i < r; ++i) rhs[i] -= a[i * ld + r] * rhs[r];
    }
}

/* Pattern of S^power, rows in ascending column order, S = tril(A) (spd) or A (general)
 * ([UPSTREAM] isai extend_sparsity; Preconditioner.H:227 `sparsityPower`).  Returns the number of
 * entries, -1 if a row gets more than ORC_ISAI_MAX_ROW of them; p_cols == NULL: sizes only. */
#define ORC_ISAI_MAX_ROW 2048  /* widest row of W handled at all */
#define ORC_ISAI_NARROW_ROW 64 /* up to here: solve_dense (row-wise back substitution), above: solve_dense_wide */
#define ISAI_IN_S(r, c) (!spd || (key ? key[c] <= key[r] : (c) <= (r)))
static orc_label isai_pattern(orc_label n, const orc_label *rowptr, const orc_label *cols, int spd,
                              int power, const orc_label *key, orc_label *p_rowptr, orc_label *p_cols) {
    orc_label *mark = (orc_label *)xmalloc(sizeof(orc_label) * ((size_t)n + 1));
    orc_label row[ORC_ISAI_MAX_ROW + 1], next[ORC_ISAI_MAX_ROW + 1];
    for (orc_label i = 0; i < n; ++i) mark[i] = -1;
    p_rowptr[0] = 0;
    for (orc_label i = 0; i < n; ++i) {
        orc_label len = 0;
        for (orc_label k = rowptr[i]; k < rowptr[i + 1]; ++k)
            if (ISAI_IN_S(i, cols[k]) && mark[cols[k]] != i) { /* S(i,:), duplicates once */
                if (len == ORC_ISAI_MAX_ROW) { free(mark); return -1; }
                mark[cols[k]] = i;
                row[len++] = cols[k];
            }
        for (int pw = 1; pw < power; ++pw) { /* row of S^(pw+1) = union of S(j,:) over the row of S^pw */
            orc_label nl = 0;
            for (orc_label e = 0; e < len; ++e) next[nl++] = row[e];
            for (orc_label e = 0; e < len; ++e) {
                const orc_label j = row[e];
                for (orc_label k = rowptr[j]; k < rowptr[j + 1]; ++k)
                    if (ISAI_IN_S(j, cols[k]) && mark[cols[k]] != i) {
                        if (nl == ORC_ISAI_MAX_ROW) { free(mark); return -1; }
                        mark[cols[k]] = i;
                        next[nl++] = cols[k];
                    }
            }
            len = nl;
            for (orc_label e = 0; e < len; ++e) row[e] = next[e];
        }
        for (orc_label a = 1; a < len; ++a) { /* ascending columns */
            const orc_label c = row[a];
            orc_label bpos = a;
            while (bpos > 0 && row[bpos - 1] > c) { row[bpos] = row[bpos - 1]; --bpos; }
            row[bpos] = c;
        }
        if (p_cols)
            for (orc_label e = 0; e < len; ++e) p_cols[p_rowptr[i] + e] = row[e];
        p_rowptr[i + 1] = p_rowptr[i] + len;
    }
    free(mark);
    return p_rowptr[n];
}

/* ISAI with `sparsityPower` power: row i of W lives on the pattern J of row i of S^power and solves
 * A(J,J) y = e_i (spd; then W(i,J) = y / sqrt(y_i)) or A(J,J)^T y = e_i (general).  Rows of up to
 * ORC_ISAI_MAX_ROW entries, each by one dense solve. */
orc_label orc_isai_generate_p(orc_label n, const orc_label *rowptr, const orc_label *cols,
                              const orc_scalar *vals, int spd, int power, orc_label *w_rowptr,
                              orc_label *w_cols, orc_scalar *w_vals) {
    return orc_isai_generate_pk(n, rowptr, cols, vals, spd, power, 0, w_rowptr, w_cols, w_vals);
}

orc_label orc_isai_generate_pk(orc_label n, const orc_label *rowptr, const orc_label *cols,
                               const orc_scalar *vals, int spd, int power, const orc_label *key,
                               orc_label *w_rowptr, orc_label *w_cols, orc_scalar *w_vals) {
    if (power < 1) return -1;
    if (!w_vals) return isai_pattern(n, rowptr, cols, spd, power, key, w_rowptr, 0);
    if (isai_pattern(n, rowptr, cols, spd, power, key, w_rowptr, w_cols) < 0) return -1;
    enum { LD = ORC_ISAI_MAX_ROW };
    orc_scalar *a = (orc_scalar *)xmalloc(sizeof(orc_scalar) * LD * LD);
    orc_scalar rhs[LD];
    for (orc_label i = 0; i < n; ++i) {
        const orc_label *J = w_cols + w_rowptr[i];
        const orc_label bs = w_rowptr[i + 1] - w_rowptr[i];
        orc_label pos = -1;
        for (orc_label r = 0; r < bs; ++r)
            if (J[r] == i) pos = r;
        const orc_label ld = bs <= ORC_ISAI_NARROW_ROW ? ORC_ISAI_NARROW_ROW : ((bs + 7) & ~7); /* (storage only: same bits) */
        for (orc_label r = 0; r < bs; ++r) {
            for (orc_label c = 0; c < bs; ++c)
                a[r * ld + c] = spd ? csr_entry(rowptr, cols, vals, J[r], J[c])
                                    : csr_entry(rowptr, cols, vals, J[c], J[r]);
            rhs[r] = (r == pos) ? 1.0 : 0.0;
        }
        if (bs <= ORC_ISAI_NARROW_ROW)
            solve_dense(bs, a, ld, rhs);
        else
            solve_dense_wide(bs, a, ld, rhs);
        const orc_scalar scale = spd ? sqrt(rhs[pos]) : 1.0;
        for (orc_label r = 0; r < bs; ++r) w_vals[w_rowptr[i] + r] = spd ? rhs[r] / scale : rhs[r];
    }
    free(a);
    return w_rowptr[n];
}

orc_label orc_isai_generate(orc_label n, const orc_label *rowptr, const orc_label *cols,
                            const orc_scalar *vals, int spd, orc_label *w_rowptr, orc_label *w_cols,
                            orc_scalar *w_vals) {
    return orc_isai_generate_p(n, rowptr, cols, vals, spd, 1, w_rowptr, w_cols, w_vals);
}

void orc_csr_transpose(orc_label n, const orc_label *rowptr, const orc_label *cols,
                       const orc_scalar *vals, orc_label *t_rowptr, orc_label *t_cols,
                       orc_scalar *t_vals) {
    const orc_label nnz = rowptr[n];
    for (orc_label r = 0; r <= n; ++r) t_rowptr[r] = 0;
    for (orc_label k = 0; k < nnz; ++k) t_rowptr[cols[k] + 1]++;
    for (orc_label r = 0; r < n; ++r) t_rowptr[r + 1] += t_rowptr[r];
    orc_label *fill = (orc_label *)xmalloc(sizeof(orc_label) * ((size_t)n + 1));
    memcpy(fill, t_rowptr, sizeof(orc_label) * ((size_t)n + 1));
    for (orc_label r = 0; r < n; ++r)
        for (orc_label k = rowptr[r]; k < rowptr[r + 1]; ++k) {
            const orc_label e = fill[cols[k]]++;
            t_cols[e] = r;
            t_vals[e] = vals[k];
        }
    free(fill);
}

static void precond_apply(orc_label n, const orc_precond *P, const orc_scalar *r, orc_scalar *z) {
    if (P && P->kind == ORC_PRECOND_ISAI_GENERAL) {
        orc_spmv(n, P->w_rowptr, P->w_cols, P->w_vals, r, z);
        return;
    }
    if (P && P->kind == ORC_PRECOND_ISAI_SPD) { /* z = W^T (W r) */
        orc_scalar *t = (orc_scalar *)xmalloc(sizeof(orc_scalar) * (size_t)(n ? n : 1));
        orc_spmv(n, P->w_rowptr, P->w_cols, P->w_vals, r, t);
        orc_spmv(n, P->wt_rowptr, P->wt_cols, P->wt_vals, t, z);
        free(t);
        return;
    }
    if (!P || P->kind == ORC_PRECOND_NONE) {
        memcpy(z, r, sizeof(orc_scalar) * (size_t)n); /* identity: copy */
    } else if (P->kind == ORC_PRECOND_SCALAR) {
        for (orc_label i = 0; i < n; ++i) z[i] = r[i] * P->inv_diag[i];
    } else {
        for (orc_label b = 0; b < P->n_blocks; ++b) {
            const orc_label r0 = P->block_ptrs[b], bs = P->block_ptrs[b + 1] - r0;
            const orc_scalar *a = P->blocks + (size_t)b * P->stride * P->stride;
            const orc_label *rows = P->block_rows;
            for (orc_label i = 0; i < bs; ++i) {
                orc_scalar sum = 0.0;
                for (orc_label j = 0; j < bs; ++j) sum += a[i * P->stride + j] * r[rows ? rows[r0 + j] : r0 + j];
                z[rows ? rows[r0 + i] : r0 + i] = sum;
            }
        }
    }
}

/* ------------------------------------------------------------------ */
/* CG  ([UPSTREAM] gko::solver::Cg::apply_dense_impl)                   */
/* ------------------------------------------------------------------ */
orc_label orc_cg(const orc_dist_matrix *A, const orc_scalar *b, orc_scalar *x,
                 const orc_scalar *inv_diag, const orc_criterion *crit,
                 orc_criterion_state *st) {
    orc_precond P = {inv_diag ? ORC_PRECOND_SCALAR : ORC_PRECOND_NONE, inv_diag, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
    return orc_cg_p(A, b, x, &P, crit, st);
}

orc_label orc_cg_p(const orc_dist_matrix *A, const orc_scalar *b, orc_scalar *x,
                   const orc_precond *inv_diag, const orc_criterion *crit,
                   orc_criterion_state *st) {
    const orc_label n = A->n;
    const size_t bytes = sizeof(orc_scalar) * (size_t)n;
    orc_scalar *r = (orc_scalar *)xmalloc(bytes), *z = (orc_scalar *)calloc(n ? n : 1, sizeof(orc_scalar));
    orc_scalar *p = (orc_scalar *)calloc(n ? n : 1, sizeof(orc_scalar));
    orc_scalar *q = (orc_scalar *)calloc(n ? n : 1, sizeof(orc_scalar));
    if (!z || !p || !q) abort();
    orc_scalar rho = 0.0, prev_rho = 1.0, beta = 0.0; /* initialize: r=b, z=p=q=0 */
    criterion_reset(st);
    dist_residual(A, b, x, r);
    for (;;) {
        precond_apply(n, inv_diag, r, z);
        rho = dist_dot(A, r, z);
        if (criterion_check(A, crit, st, r, x, b)) break;
        { /* step_1: p = z + (rho / prev_rho) p */
            const orc_scalar tmp = (prev_rho == 0.0) ? 0.0 : rho / prev_rho;
            for (orc_label i = 0; i < n; ++i) p[i] = z[i] + tmp * p[i];
        }
        orc_dist_spmv(A, p, q);
        beta = dist_dot(A, p, q);
        if (beta != 0.0) { /* step_2: x += (rho/beta) p ; r -= (rho/beta) q */
            const orc_scalar tmp = rho / beta;
            for (orc_label i = 0; i < n; ++i) {
                x[i] += tmp * p[i];
                r[i] -= tmp * q[i];
            }
        }
        prev_rho = rho; /* swap(prev_rho, rho); rho is recomputed next turn */
    }
    free(r);
    free(z);
    free(p);
    free(q);
    return st->iter;
}

/* ------------------------------------------------------------------ */
/* BiCGStab ([UPSTREAM] gko::solver::Bicgstab::apply_dense_impl)        */
/* Two criterion checks per turn (on r, then on s) -- which is why OGL  */
/* doubles maxIter (StoppingCriterion.H:188) and halves the reported    */
/* iteration count (GKOBiCGStab.H:114).                                 */
/* ------------------------------------------------------------------ */
orc_label orc_bicgstab(const orc_dist_matrix *A, const orc_scalar *b, orc_scalar *x,
                       const orc_scalar *inv_diag, const orc_criterion *crit,
                       orc_criterion_state *st) {
    orc_precond P = {inv_diag ? ORC_PRECOND_SCALAR : ORC_PRECOND_NONE, inv_diag, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
    return orc_bicgstab_p(A, b, x, &P, crit, st);
}

orc_label orc_bicgstab_p(const orc_dist_matrix *A, const orc_scalar *b, orc_scalar *x,
                         const orc_precond *inv_diag, const orc_criterion *crit,
                         orc_criterion_state *st) {
    const orc_label n = A->n;
    const size_t cnt = n ? (size_t)n : 1;
    orc_scalar *r = (orc_scalar *)calloc(cnt, sizeof(orc_scalar));
    orc_scalar *rr = (orc_scalar *)calloc(cnt, sizeof(orc_scalar));
    orc_scalar *y = (orc_scalar *)calloc(cnt, sizeof(orc_scalar));
    orc_scalar *s = (orc_scalar *)calloc(cnt, sizeof(orc_scalar));
    orc_scalar *t = (orc_scalar *)calloc(cnt, sizeof(orc_scalar));
    orc_scalar *z = (orc_scalar *)calloc(cnt, sizeof(orc_scalar));
    orc_scalar *v = (orc_scalar *)calloc(cnt, sizeof(orc_scalar));
    orc_scalar *p = (orc_scalar *)calloc(cnt, sizeof(orc_scalar));
    if (!r || !rr || !y || !s || !t || !z || !v || !p) abort();
    orc_scalar rho = 1.0, prev_rho = 1.0, alpha = 1.0, beta = 1.0, gamma = 1.0, omega = 1.0;
    criterion_reset(st);
    dist_residual(A, b, x, r);
    memcpy(rr, r, sizeof(orc_scalar) * (size_t)n);
    for (;;) {
        rho = dist_dot(A, rr, r);
        if (criterion_check(A, crit, st, r, x, b)) break;
        { /* step_1: p = r + (rho/prev_rho)(alpha/omega) (p - omega v) */
            if (prev_rho * omega != 0.0) {
                const orc_scalar tmp = rho / prev_rho * alpha / omega;
                for (orc_label i = 0; i < n; ++i) p[i] = r[i] + tmp * (p[i] - omega * v[i]);
            } else {
                memcpy(p, r, sizeof(orc_scalar) * (size_t)n);
            }
        }
        precond_apply(n, inv_diag, p, y);
        orc_dist_spmv(A, y, v);
        beta = dist_dot(A, rr, v);
        { /* step_2: alpha = rho / beta ; s = r - alpha v */
            const orc_scalar tmp = (beta == 0.0) ? 0.0 : rho / beta;
            alpha = tmp;
            for (orc_label i = 0; i < n; ++i) s[i] = r[i] - tmp * v[i];
        }
        if (criterion_check(A, crit, st, s, x, b)) {
            for (orc_label i = 0; i < n; ++i) x[i] += alpha * y[i]; /* finalize */
            break;
        }
        precond_apply(n, inv_diag, s, z);
        orc_dist_spmv(A, z, t);
        gamma = dist_dot(A, s, t);
        beta = dist_dot(A, t, t);
        { /* step_3: omega = gamma / beta ; x += alpha y + omega z ; r = s - omega t */
            const orc_scalar tmp = (beta == 0.0) ? 0.0 : gamma / beta;
            omega = tmp;
            for (orc_label i = 0; i < n; ++i) {
                x[i] += alpha * y[i] + tmp * z[i];
                r[i] = s[i] - tmp * t[i];
            }
        }
        prev_rho = rho;
    }
    free(r); free(rr); free(y); free(s); free(t); free(z); free(v); free(p);
    return st->iter;
}

/* ------------------------------------------------------------------ */
/* GMRES ([UPSTREAM] gko::solver::Gmres::apply_dense_impl, restarted,   */
/* right-preconditioned; reference kernels restart / finish_arnoldi     */
/* (modified Gram-Schmidt) / givens_rotation / solve_krylov).           */
/* The criterion receives the residual VECTOR of the last restart (the  */
/* solver only hands the implicit norm forward inside a cycle), so      */
/* OGL's L1 check sees a new value only after a restart (*/
/* unpinned).  krylov_dim <= 0 selects Ginkgo's default, 100.           */
/* ------------------------------------------------------------------ */
static orc_scalar dist_norm2(const orc_dist_matrix *A, const orc_scalar *a) {
    orc_scalar s = orc_dot(A->n, a, a);
    global_sum(A, &s, 1);
    return sqrt(s);
}

orc_label orc_gmres_p(const orc_dist_matrix *A, const orc_scalar *b, orc_scalar *x,
                      const orc_precond *P, const orc_criterion *crit, orc_criterion_state *st,
                      orc_label krylov_dim) {
    const orc_label n = A->n, m = krylov_dim > 0 ? krylov_dim : 100;
    const size_t cnt = n ? (size_t)n : 1;
    orc_scalar *V = (orc_scalar *)calloc(cnt * ((size_t)m + 1), sizeof(orc_scalar));
    orc_scalar *H = (orc_scalar *)calloc(((size_t)m + 1) * (size_t)m, sizeof(orc_scalar));
    orc_scalar *gs = (orc_scalar *)calloc((size_t)m, sizeof(orc_scalar));
    orc_scalar *gc = (orc_scalar *)calloc((size_t)m, sizeof(orc_scalar));
    orc_scalar *rnc = (orc_scalar *)calloc((size_t)m + 1, sizeof(orc_scalar));
    orc_scalar *y = (orc_scalar *)calloc((size_t)m, sizeof(orc_scalar));
    orc_scalar *r = (orc_scalar *)calloc(cnt, sizeof(orc_scalar));
    orc_scalar *w = (orc_scalar *)calloc(cnt, sizeof(orc_scalar));
    orc_scalar *u = (orc_scalar *)calloc(cnt, sizeof(orc_scalar));
    if (!V || !H || !gs || !gc || !rnc || !y || !r || !w || !u) abort();
#define HH(i, j) H[(size_t)(j) * ((size_t)m + 1) + (size_t)(i)]
#define VV(k) (V + (size_t)(k) * cnt)
    criterion_reset(st);
    orc_label it = 0;
    for (int restart = 1;; restart = 0) {
        if (restart) { /* residual + restart kernel */
            dist_residual(A, b, x, r);
            const orc_scalar rn = dist_norm2(A, r);
            rnc[0] = rn;
            for (orc_label i = 0; i < n; ++i) VV(0)[i] = r[i] / rn;
            it = 0;
        }
        if (criterion_check(A, crit, st, r, x, b)) break;
        if (it == m) { /* update x with the cycle's solution, then restart */
            for (orc_label i = m - 1; i >= 0; --i) { /* solve_upper_triangular */
                orc_scalar t = rnc[i];
                for (orc_label j = i + 1; j < m; ++j) t -= HH(i, j) * y[j];
                y[i] = t / HH(i, i);
            }
            for (orc_label i = 0; i < n; ++i) { /* calculate_qy */
                orc_scalar sum = 0.0;
                for (orc_label j = 0; j < m; ++j) sum += VV(j)[i] * y[j];
                w[i] = sum;
            }
            precond_apply(n, P, w, u);
            for (orc_label i = 0; i < n; ++i) x[i] += u[i];
            dist_residual(A, b, x, r);
            const orc_scalar rn = dist_norm2(A, r);
            rnc[0] = rn;
            for (orc_label i = 0; i < n; ++i) VV(0)[i] = r[i] / rn;
            it = 0;
        }
        /* Arnoldi step */
        precond_apply(n, P, VV(it), w);
        orc_dist_spmv(A, w, VV(it + 1));
        orc_scalar *nx = VV(it + 1);
        for (orc_label k = 0; k <= it; ++k) { /* finish_arnoldi: modified Gram-Schmidt */
            const orc_scalar h = dist_dot(A, nx, VV(k));
            HH(k, it) = h;
            for (orc_label i = 0; i < n; ++i) nx[i] -= h * VV(k)[i];
        }
        const orc_scalar hn = dist_norm2(A, nx);
        HH(it + 1, it) = hn;
        for (orc_label i = 0; i < n; ++i) nx[i] /= hn;
        for (orc_label j = 0; j < it; ++j) { /* givens_rotation: previous rotations */
            const orc_scalar t = gc[j] * HH(j, it) + gs[j] * HH(j + 1, it);
            HH(j + 1, it) = -gs[j] * HH(j, it) + gc[j] * HH(j + 1, it);
            HH(j, it) = t;
        }
        if (HH(it, it) == 0.0) { /* calculate_sin_and_cos */
            gc[it] = 0.0;
            gs[it] = 1.0;
        } else {
            const orc_scalar scale = fabs(HH(it, it)) + fabs(HH(it + 1, it));
            const orc_scalar a0 = HH(it, it) / scale, a1 = HH(it + 1, it) / scale;
            const orc_scalar hyp = scale * sqrt(a0 * a0 + a1 * a1);
            gc[it] = HH(it, it) / hyp;
            gs[it] = HH(it + 1, it) / hyp;
        }
        HH(it, it) = gc[it] * HH(it, it) + gs[it] * HH(it + 1, it);
        HH(it + 1, it) = 0.0;
        rnc[it + 1] = -gs[it] * rnc[it]; /* calculate_next_residual_norm */
        rnc[it] = gc[it] * rnc[it];
        ++it;
    }
    if (it > 0) { /* final solve_krylov on the partial cycle */
        for (orc_label i = it - 1; i >= 0; --i) {
            orc_scalar t = rnc[i];
            for (orc_label j = i + 1; j < it; ++j) t -= HH(i, j) * y[j];
            y[i] = t / HH(i, i);
        }
        for (orc_label i = 0; i < n; ++i) {
            orc_scalar sum = 0.0;
            for (orc_label j = 0; j < it; ++j) sum += VV(j)[i] * y[j];
            w[i] = sum;
        }
        precond_apply(n, P, w, u);
        for (orc_label i = 0; i < n; ++i) x[i] += u[i];
    }
#undef HH
#undef VV
    free(V); free(H); free(gs); free(gc); free(rnc); free(y); free(r); free(w); free(u);
    return st->iter;
}

/* ------------------------------------------------------------------ */
/* "omp executor" baseline (single rank)                                */
/* ------------------------------------------------------------------ */
int orc_omp_max_threads(void) {
#ifdef _OPENMP
    return omp_get_max_threads();
#else
    return 0;
#endif
}

/* STREAM-like triad a = b + s*c over three arrays of n doubles placed by first touch with the same
 * static schedule; returns GB/s (24 bytes per element) of the best of `reps` passes.  Run in the same
 * process as orc_cg_omp so the baseline can be read against what the host memory system gives. */
double orc_stream_triad_omp(long n, int reps, int n_threads) {
#ifndef _OPENMP
    (void)n; (void)reps; (void)n_threads;
    return -1.0;
#else
    if (n_threads > 0) omp_set_num_threads(n_threads);
    double *a = (double *)xmalloc(sizeof(double) * (size_t)n);
    double *b = (double *)xmalloc(sizeof(double) * (size_t)n);
    double *c = (double *)xmalloc(sizeof(double) * (size_t)n);
#pragma omp parallel for schedule(static)
    for (long i = 0; i < n; ++i) {
        a[i] = 0.0;
        b[i] = 1.0 + (double)(i & 7);
        c[i] = 0.5;
    }
    double best = 1e30;
    for (int rep = 0; rep < reps; ++rep) {
        const double t0 = omp_get_wtime();
#pragma omp parallel for schedule(static)
        for (long i = 0; i < n; ++i) a[i] = b[i] + 3.0 * c[i];
        const double t = omp_get_wtime() - t0;
        if (t < best) best = t;
    }
    const double chk = a[n / 2];
    free(a); free(b); free(c);
    return chk > 0.0 ? 24.0 * (double)n / best / 1e9 : 0.0;
#endif
}

__attribute__((unused)) static void *orc_skewed_alloc(size_t bytes, void **owned, int *n_owned) {
    void *base = xmalloc(bytes + 8192);
    owned[*n_owned] = base;
    *n_owned += 1;
    return (char *)base + 576 * (size_t)*n_owned;
}

/* seconds the last orc_cg_omp_timed spent in its three passes: [0] x/r update + rho + sum|r|, [1] p update, [2] SpMV + p.q */
static double omp_phase_s[3];
void orc_cg_omp_phases(double out[3]) { out[0] = omp_phase_s[0]; out[1] = omp_phase_s[1]; out[2] = omp_phase_s[2]; }

/* The OpenMP CG with its two phases timed apart: t_setup_s = allocation + first-touch copy of the
 * matrix and vectors (once per matrix in a real run), t_loop_s = initial residual, norm factor and
 * the iterations (what a solve costs).  The norm factor is computed with the same parallel loops. */
orc_label orc_cg_omp_timed(const orc_dist_matrix *A, const orc_scalar *b, orc_scalar *x,
                           const orc_scalar *inv_diag, const orc_criterion *crit,
                           orc_criterion_state *st, int n_threads, double *t_setup_s,
                           double *t_loop_s) {
#ifndef _OPENMP
    (void)A; (void)b; (void)x; (void)inv_diag; (void)crit; (void)st; (void)n_threads;
    (void)t_setup_s; (void)t_loop_s;
    return -1;
#else
    const double t_begin = omp_get_wtime();
    if (A->n_halo > 0 || A->allreduce) return -1;
    if (n_threads > 0) omp_set_num_threads(n_threads);
    const orc_label n = A->n;
    const size_t cnt = n ? (size_t)n : 1;
    /* The vectors of a pass are walked in lock step: allocated page-aligned one after the other they would sit at
     * the same offset within a 4 KiB page, and every store of a pass would alias the loads of the next elements of
     * the other vectors.  Each vector starts a different number of cache lines into its allocation. */
    void *owned[10];
    int n_owned = 0;
#define ORC_SKEWED(T, count) ((T *)orc_skewed_alloc(sizeof(T) * (count), owned, &n_owned))
    orc_scalar *r = ORC_SKEWED(orc_scalar, cnt);
    orc_scalar *z = ORC_SKEWED(orc_scalar, cnt);
    orc_scalar *p = ORC_SKEWED(orc_scalar, cnt);
    orc_scalar *q = ORC_SKEWED(orc_scalar, cnt);
    /* First-touch placement: the caller's arrays were allocated (and touched) by one thread, i.e.
     * on one NUMA node.  Every thread copies the slice it will stream later (same static
     * schedule), so a multi-socket host serves the loop from all of its memory controllers. */
    const orc_label nnz_all = A->rowptr[n];
    orc_label *rowptr = ORC_SKEWED(orc_label, cnt + 1);
    orc_label *cols = ORC_SKEWED(orc_label, (size_t)(nnz_all ? nnz_all : 1));
    orc_scalar *vals = ORC_SKEWED(orc_scalar, (size_t)(nnz_all ? nnz_all : 1));
    orc_scalar *bb = ORC_SKEWED(orc_scalar, cnt);
    orc_scalar *xx = ORC_SKEWED(orc_scalar, cnt);
    orc_scalar *inv_copy = inv_diag ? ORC_SKEWED(orc_scalar, cnt) : 0;
#undef ORC_SKEWED
#pragma omp parallel for schedule(static)
    for (orc_label row = 0; row < n; ++row) {
        rowptr[row] = A->rowptr[row];
        for (orc_label k = A->rowptr[row]; k < A->rowptr[row + 1]; ++k) {
            cols[k] = A->cols[k];
            vals[k] = A->vals[k];
        }
        bb[row] = b[row];
        xx[row] = x[row];
        if (inv_copy) inv_copy[row] = inv_diag[row];
    }
    rowptr[n] = nnz_all;
    orc_scalar *x_out = x;
    b = bb;
    x = xx;
    if (inv_copy) inv_diag = inv_copy;
    orc_scalar rho = 0.0, prev_rho = 1.0, beta = 0.0, norm = 0.0;
    criterion_reset(st);
#pragma omp parallel for schedule(static)
    for (orc_label row = 0; row < n; ++row) z[row] = p[row] = q[row] = r[row] = 0.0;  /* first touch */
    const double t_loop_begin = omp_get_wtime();
    if (t_setup_s) *t_setup_s = t_loop_begin - t_begin;
#pragma omp parallel for schedule(static)
    for (orc_label row = 0; row < n; ++row) {
        orc_scalar sum = b[row];
        for (orc_label k = rowptr[row]; k < rowptr[row + 1]; ++k) sum += -1.0 * vals[k] * x[cols[k]];
        r[row] = sum;
    }
    /* Passes fused the way the GPU kernels fuse them (a competent OpenMP baseline does not stream r three times
     * per turn): [x += t p; r -= t q of the turn before; rho = r.z; sum|r|] | check | [p = z + t1 p, z = M^-1 r on
     * the fly] | [q = A p; beta = p.q].  z is never stored.  Same arithmetic per element as the sequential loop. */
    int have_update = 0;
    orc_scalar t2 = 0.0;
    omp_phase_s[0] = omp_phase_s[1] = omp_phase_s[2] = 0.0;
    for (;;) {
        rho = 0.0;
        norm = 0.0;
        double tp = omp_get_wtime();
#pragma omp parallel for schedule(static) reduction(+ : rho, norm)
        for (orc_label i = 0; i < n; ++i) {
            orc_scalar ri = r[i];
            if (have_update) {
                x[i] += t2 * p[i];
                ri -= t2 * q[i];
                r[i] = ri;
            }
            const orc_scalar zi = inv_diag ? ri * inv_diag[i] : ri;
            rho += ri * zi;
            norm += fabs(ri);
        }
        omp_phase_s[0] += omp_get_wtime() - tp;
        /* criterion (same policy as criterion_check, norm already reduced) */
        int stop = 0;
        if (st->iter > 0 && st->iter < crit->min_iter) {
            st->iter += 1;
        } else if (st->iter % crit->frequency != 0) {
            st->iter += 1;
        } else {
            st->n_evals += 1;
            if (st->iter == 0) {
                /* StoppingCriterion.C:11-69 with parallel loops (q is free at this point) */
                orc_scalar xsum = 0.0, nf = 0.0;
#pragma omp parallel for schedule(static) reduction(+ : xsum)
                for (orc_label i = 0; i < n; ++i) xsum += x[i];
                const orc_scalar xbar = xsum / (orc_scalar)n;
#pragma omp parallel for schedule(static) reduction(+ : nf)
                for (orc_label row = 0; row < n; ++row) {
                    orc_scalar ax = 0.0;
                    for (orc_label k = rowptr[row]; k < rowptr[row + 1]; ++k) ax += vals[k] * xbar;
                    const orc_scalar t = b[row] - ax;
                    nf += fabs(t - r[row]) + fabs(t);
                }
                st->norm_factor = nf + ORC_SMALL;
                st->init_residual = norm / st->norm_factor;
            }
            norm /= st->norm_factor;
            if (crit->export_res && st->history) st->history[st->iter] = norm;
            st->residual = norm;
            if (st->iter >= crit->max_iter) stop = 1;
            if (norm < crit->tolerance) stop = 1;
            if (crit->rel_tol > 0 && norm < crit->rel_tol * st->init_residual) stop = 1;
            st->iter += 1;
        }
        if (stop) break;
        const orc_scalar t1 = (prev_rho == 0.0) ? 0.0 : rho / prev_rho;
        tp = omp_get_wtime();
#pragma omp parallel for schedule(static)
        for (orc_label i = 0; i < n; ++i) p[i] = (inv_diag ? r[i] * inv_diag[i] : r[i]) + t1 * p[i];
        omp_phase_s[1] += omp_get_wtime() - tp;
        tp = omp_get_wtime();
        beta = 0.0;
#pragma omp parallel for schedule(static) reduction(+ : beta)
        for (orc_label row = 0; row < n; ++row) {
            orc_scalar sum = 0.0;
            for (orc_label k = rowptr[row]; k < rowptr[row + 1]; ++k) sum += vals[k] * p[cols[k]];
            q[row] = sum;
            beta += p[row] * sum;
        }
        omp_phase_s[2] += omp_get_wtime() - tp;
        have_update = beta != 0.0;
        if (have_update) t2 = rho / beta;
        prev_rho = rho;
    }
    if (t_loop_s) *t_loop_s = omp_get_wtime() - t_loop_begin;
    memcpy(x_out, x, sizeof(orc_scalar) * (size_t)n);
    for (int i = 0; i < n_owned; ++i) free(owned[i]);
    return st->iter;
#endif
}

orc_label orc_cg_omp(const orc_dist_matrix *A, const orc_scalar *b, orc_scalar *x,
                     const orc_scalar *inv_diag, const orc_criterion *crit,
                     orc_criterion_state *st, int n_threads) {
    return orc_cg_omp_timed(A, b, x, inv_diag, crit, st, n_threads, 0, 0);
}
