// kernels_precond.hip -- coefficient permutation, Jacobi / block-Jacobi / ISAI generate, block-Jacobi apply, renumbering gathers
// (geometry, reduction tree and the -ffp-contract=off rule: device_common.hpp)
#include "device_common.hpp"

namespace ogl {

namespace {

// ------------------------------------------------------------------------------------------
// coefficient permutation (K9) and scalar Jacobi generate (K8)
// ------------------------------------------------------------------------------------------
__global__ __launch_bounds__(BLOCK) void k_gather_coeffs(int nnz, const int *__restrict__ map,
                                                         const double *__restrict__ src,
                                                         double *__restrict__ out)
{
    const int e = (blockIdx.x * BLOCK + threadIdx.x) * 4;
    if (e + 3 < nnz) {
        const int4 m = *reinterpret_cast<const int4 *>(map + e);
        double2 a, c;
        a.x = src[m.x];
        a.y = src[m.y];
        c.x = src[m.z];
        c.y = src[m.w];
        *reinterpret_cast<double2 *>(out + e) = a;
        *reinterpret_cast<double2 *>(out + e + 2) = c;
    } else {
        for (int k = e; k < nnz; ++k) out[k] = src[map[k]];
    }
}

__global__ __launch_bounds__(BLOCK) void k_jacobi_generate(int n_rows,
                                                           const int *__restrict__ row_ptrs,
                                                           const int *__restrict__ cols,
                                                           const double *__restrict__ vals,
                                                           double *__restrict__ inv_diag)
{
    const int row = blockIdx.x * BLOCK + threadIdx.x;
    if (row >= n_rows) return;
    double d = 0.0;
    for (int k = row_ptrs[row]; k < row_ptrs[row + 1]; ++k)
        if (cols[k] == row) {
            d = vals[k];
            break;
        }
    inv_diag[row] = 1.0 / d;
}

// The same from the precomputed position of each row's first diagonal entry (-1: none -> 1/0 as
// above): 12 bytes per row instead of a walk through the row.
__global__ __launch_bounds__(BLOCK) void k_jacobi_generate_pos(int n_rows, const int *__restrict__ diag_pos,
                                                               const double *__restrict__ vals,
                                                               double *__restrict__ inv_diag)
{
    const int row = blockIdx.x * BLOCK + threadIdx.x;
    if (row >= n_rows) return;
    const int k = diag_pos[row];
    const double d = k >= 0 ? vals[k] : 0.0;
    inv_diag[row] = 1.0 / d;
}

// ... and from the diagonal as the lduMatrix holds it (the staged source of the coefficient update, device rows in the
// caller's order, no same-rank interface entries): 16 bytes per row instead of a strided walk over the CSR values.
__global__ __launch_bounds__(BLOCK) void k_jacobi_generate_diag(int n_rows, const double *__restrict__ diag,
                                                                double *__restrict__ inv_diag)
{
    const int row = blockIdx.x * BLOCK + threadIdx.x;
    if (row >= n_rows) return;
    inv_diag[row] = 1.0 / diag[row];
}

// Block Jacobi generate: one thread inverts one diagonal block in place (global memory; runs
// once per preconditioner generation).  Same operation order as oracle/ogl_oracle.c invert_block:
// Gauss-Jordan, partial (row) pivoting, pivot row scaled first, then the other rows eliminated,
// finally the row swaps undone as a column permutation.
// The block lives in a per-thread array of the smallest power-of-two leading dimension LD that
// holds maxBlockSize (scratch of 8*LD*LD bytes per thread: 128 B for LD = 4), not in global memory,
// and is written out once.
template <int LD>
__global__ __launch_bounds__(64) void k_bj_generate(int n_blocks, const int *__restrict__ block_ptrs,
                                                    const int *__restrict__ row_ptrs,
                                                    const int *__restrict__ cols,
                                                    const double *__restrict__ vals,
                                                    double *__restrict__ blocks, int ld,
                                                    const int *__restrict__ rows, const int *__restrict__ pos,
                                                    int by_device_row)
{
    const int b = blockIdx.x * 64 + threadIdx.x;
    if (b >= n_blocks) return;
    const int r0 = block_ptrs[b], bs = block_ptrs[b + 1] - r0;
    double a[LD * LD];
    for (int i = 0; i < LD * LD; ++i) a[i] = 0.0;
    // rows != nullptr (the device copy is renumbered): block members r0 .. r0 + bs are positions in the CALLER's
    // numbering -- member i is device row rows[r0 + i], a device column c sits at position pos[c]
    for (int i = 0; i < bs; ++i) {
        const int r = rows ? rows[r0 + i] : r0 + i;
        for (int k = row_ptrs[r]; k < row_ptrs[r + 1]; ++k) {
            const int c = (rows ? pos[cols[k]] : cols[k]) - r0;
            if (c >= 0 && c < bs) a[i * LD + c] = vals[k];
        }
    }
    int perm[LD];
    for (int k = 0; k < bs; ++k) perm[k] = k;
    for (int k = 0; k < bs; ++k) {
        int piv = k;
        double best = fabs(a[k * LD + k]);
        for (int i = k + 1; i < bs; ++i)
            if (fabs(a[i * LD + k]) > best) {
                best = fabs(a[i * LD + k]);
                piv = i;
            }
        if (piv != k) {
            for (int j = 0; j < bs; ++j) {
                const double t = a[k * LD + j];
                a[k * LD + j] = a[piv * LD + j];
                a[piv * LD + j] = t;
            }
            const int t = perm[k];
            perm[k] = perm[piv];
            perm[piv] = t;
        }
        const double d = a[k * LD + k];
        a[k * LD + k] = 1.0;
        for (int j = 0; j < bs; ++j) a[k * LD + j] /= d;
        for (int i = 0; i < bs; ++i) {
            if (i == k) continue;
            const double f = a[i * LD + k];
            a[i * LD + k] = 0.0;
            for (int j = 0; j < bs; ++j) a[i * LD + j] -= f * a[k * LD + j];
        }
    }
    // block-major, `ld` doubles per block row -- except through a permutation, where member i's row of the inverse
    // is stored at its DEVICE row (rows[r0 + i] * ld): the apply then reads it coalesced instead of from wherever
    // the block sits in the caller's order
    double *out = blocks + (size_t)b * ld * ld;
    const bool dev = rows && by_device_row;
    if (!dev)
        for (int i = 0; i < ld * ld; ++i) out[i] = 0.0;
    double row[LD];
    for (int i = 0; i < bs; ++i) {
        for (int j = 0; j < bs; ++j) row[perm[j]] = a[i * LD + j];
        double *o = dev ? blocks + (size_t)rows[r0 + i] * ld : out + (size_t)i * ld;
        for (int j = 0; j < bs; ++j) o[j] = row[j];
        if (dev)
            for (int j = bs; j < ld; ++j) o[j] = 0.0;
    }
}

// Value of lane `idx` of this lane's group of G (idx: a constant once the loops are unrolled).  Groups of four are the
// hardware's quads: a DPP quad_perm move (full-rate vector ALU) instead of a trip through the LDS crossbar.
__device__ __forceinline__ int quad_get(int v, int idx)
{
    switch (idx & 3) {
    case 0: return __builtin_amdgcn_mov_dpp(v, 0x00, 0xf, 0xf, true);
    case 1: return __builtin_amdgcn_mov_dpp(v, 0x55, 0xf, 0xf, true);
    case 2: return __builtin_amdgcn_mov_dpp(v, 0xaa, 0xf, 0xf, true);
    default: return __builtin_amdgcn_mov_dpp(v, 0xff, 0xf, 0xf, true);
    }
}
template <int G>
__device__ __forceinline__ int grp_get(int v, int base, int idx)
{
    if (G == 4) return quad_get(v, idx);
    return __shfl(v, base + idx, WAVE);
}
template <int G>
__device__ __forceinline__ double grp_get(double v, int base, int idx)
{
    if (G == 4) return __hiloint2double(quad_get(__double2hiint(v), idx), quad_get(__double2loint(v), idx));
    return __shfl(v, base + idx, WAVE);
}

// The same inversion with G lanes per block (blocks of at most G rows, 64 / G blocks per wavefront): lane c owns COLUMN c of
// its block in registers, every loop unrolled over G -- no scratch memory (the thread-per-block kernel above spends 1.4 ms
// per generation on the 2.5 M blocks of BJ(4) at 216^3, every solve).  The pivot column is scanned by its owner, the row
// swap is a swap inside every lane's column, the pivot row is scaled by one division per lane, f = a[i][k] comes from
// lane k: every element sees invert_block's operations in invert_block's order, so the bits agree.
template <int G>
__global__ __launch_bounds__(BLOCK) void k_bj_generate_grp(int n_blocks, const int *__restrict__ block_ptrs,
                                                           const int *__restrict__ row_ptrs,
                                                           const int *__restrict__ cols,
                                                           const double *__restrict__ vals,
                                                           double *__restrict__ blocks, int ld,
                                                           const int *__restrict__ rows, const int *__restrict__ pos,
                                                           int by_device_row)
{
    static_assert(G == 4 || G == 8, "group widths instantiated");
    const int t = blockIdx.x * BLOCK + threadIdx.x;
    const int b = t / G, c = t % G;
    const int lane = threadIdx.x & (WAVE - 1), base = lane - c;
    const bool live = b < n_blocks;
    const int r0 = live ? block_ptrs[b] : 0, bs = live ? block_ptrs[b + 1] - r0 : 0;
    double a[G];
    int perm[G];
#pragma unroll
    for (int i = 0; i < G; ++i) {
        a[i] = 0.0;
        perm[i] = i;
    }
    // member row i of the block: every lane of the group walks it (the loads are shared) and keeps the entry of its column
#pragma unroll
    for (int i = 0; i < G; ++i) {
        if (i < bs) {
            const int r = rows ? rows[r0 + i] : r0 + i;
            for (int k = row_ptrs[r]; k < row_ptrs[r + 1]; ++k) {
                const int cc = (rows ? pos[cols[k]] : cols[k]) - r0;
                if (cc == c && c < bs) a[i] = vals[k];
            }
        }
    }
#pragma unroll
    for (int k = 0; k < G; ++k) {
        const bool step = k < bs;  // (group-uniform)
        int piv = k;
        {
            double best = fabs(a[k]);
#pragma unroll
            for (int i = k + 1; i < G; ++i)
                if (i < bs && fabs(a[i]) > best) {
                    best = fabs(a[i]);
                    piv = i;
                }
        }
        piv = grp_get<G>(piv, base, k);
        if (step && piv != k) {
            const double ak = a[k];
            double ap = 0.0;
            int pp = 0;
#pragma unroll
            for (int i = 0; i < G; ++i)
                if (i == piv) {
                    ap = a[i];
                    pp = perm[i];
                    a[i] = ak;
                    perm[i] = perm[k];
                }
            a[k] = ap;
            perm[k] = pp;
        }
        const double d = grp_get<G>(a[k], base, k);
        if (step) {
            if (c == k) a[k] = 1.0;
            if (c < bs) a[k] /= d;
        }
#pragma unroll
        for (int i = 0; i < G; ++i) {
            if (i == k) continue;
            const double f = grp_get<G>(a[i], base, k);  // a[i][k], before lane k clears it
            if (step && i < bs) {
                if (c == k) a[i] = 0.0;
                if (c < bs) a[i] -= f * a[k];
            }
        }
    }
    // block-major, `ld` doubles per block row -- through a permutation (by_device_row) member i's row of the inverse goes
    // to its DEVICE row; column j of the inverse is the column whose pivot history says perm[j] (the row swaps undone)
    if (!live) return;
    const bool dev = rows && by_device_row;
    int tgt = c;
#pragma unroll
    for (int j = 0; j < G; ++j)
        if (j == c && c < bs) tgt = perm[j];
    if (c < ld) {
#pragma unroll
        for (int i = 0; i < G; ++i) {
            if (i >= (dev ? bs : ld)) continue;
            double *o = dev ? blocks + (size_t)rows[r0 + i] * ld : blocks + (size_t)b * ld * ld + (size_t)i * ld;
            o[tgt] = (i < bs && c < bs) ? a[i] : 0.0;
        }
    }
}

// ISAI generate: same operation order as oracle/ogl_oracle.c (csr_entry, solve_dense)
__device__ double csr_entry(const int *__restrict__ row_ptrs, const int *__restrict__ cols,
                            const double *__restrict__ vals, int r, int c)
{
    for (int k = row_ptrs[r]; k < row_ptrs[r + 1]; ++k)
        if (cols[k] == c) return vals[k];
    return 0.0;
}

template <int LD>
__global__ __launch_bounds__(64) void k_isai_generate(int n_rows, const int *__restrict__ row_ptrs,
                                                      const int *__restrict__ cols,
                                                      const double *__restrict__ vals, int spd,
                                                      const int *__restrict__ w_row_ptrs,
                                                      const int *__restrict__ w_cols,
                                                      double *__restrict__ w_vals)
{
    const int i = blockIdx.x * 64 + threadIdx.x;
    if (i >= n_rows) return;
    const int w0 = w_row_ptrs[i], bs = w_row_ptrs[i + 1] - w0;
    if (bs > LD) return;  // a wide row: k_isai_generate_wide takes it
    int J[LD], pos = -1;
    for (int r = 0; r < bs; ++r) {
        J[r] = w_cols[w0 + r];
        if (J[r] == i) pos = r;
    }
    double a[LD * LD], rhs[LD];
    for (int r = 0; r < bs; ++r) {
        for (int c = 0; c < bs; ++c)
            a[r * LD + c] = spd ? csr_entry(row_ptrs, cols, vals, J[r], J[c])
                                : csr_entry(row_ptrs, cols, vals, J[c], J[r]);
        rhs[r] = (r == pos) ? 1.0 : 0.0;
    }
    for (int k = 0; k < bs; ++k) {
        int piv = k;
        double best = fabs(a[k * LD + k]);
        for (int r = k + 1; r < bs; ++r)
            if (fabs(a[r * LD + k]) > best) {
                best = fabs(a[r * LD + k]);
                piv = r;
            }
        if (piv != k) {
            for (int j = 0; j < bs; ++j) {
                const double t = a[k * LD + j];
                a[k * LD + j] = a[piv * LD + j];
                a[piv * LD + j] = t;
            }
            const double t = rhs[k];
            rhs[k] = rhs[piv];
            rhs[piv] = t;
        }
        for (int r = k + 1; r < bs; ++r) {
            const double f = a[r * LD + k] / a[k * LD + k];
            for (int j = k + 1; j < bs; ++j) a[r * LD + j] -= f * a[k * LD + j];
            rhs[r] -= f * rhs[k];
        }
    }
    for (int r = bs - 1; r >= 0; --r) {
        double t = rhs[r];
        for (int j = r + 1; j < bs; ++j) t -= a[r * LD + j] * rhs[j];
        rhs[r] = t / a[r * LD + r];
    }
    const double scale = spd ? sqrt(rhs[pos]) : 1.0;
    for (int r = 0; r < bs; ++r) w_vals[w0 + r] = spd ? rhs[r] / scale : rhs[r];
}

// The same solve with G lanes per row (rows of at most G entries; 64 / G rows per wavefront): lane c of a group owns COLUMN
// c of the row's dense system in registers (a[r] = A(r, c), every loop unrolled over G) and the right-hand side of ROW c.
// The thread-per-row kernel above keeps its LD x LD system in scratch memory and walks it alone: 1.7 ms per generation on
// the 2.1 M rows of configs[2], every solve.  Here the entries are looked up by G lanes side by side (the lanes of a group
// search the SAME matrix row for their columns), the pivot column is scanned by its owner, the factors a[r][k] / a[k][k]
// are formed once by lane k and handed round, and the back substitution takes the products a[r][j] * x[j] from their
// owners in the serial order of j -- every element sees the operations of solve_dense (oracle/ogl_oracle.c) in the same
// order, so the bits agree.
template <int G>
__global__ __launch_bounds__(BLOCK) void k_isai_generate_grp(int n_rows, const int *__restrict__ row_ptrs,
                                                             const int *__restrict__ cols,
                                                             const double *__restrict__ vals, int spd,
                                                             const int *__restrict__ w_row_ptrs,
                                                             const int *__restrict__ w_cols,
                                                             double *__restrict__ w_vals)
{
    static_assert(G == 4 || G == 8, "group widths instantiated");
    const int t = blockIdx.x * BLOCK + threadIdx.x;
    const int i = t / G, c = t % G;                 // row of W, this lane's column of its dense system
    const int lane = threadIdx.x & (WAVE - 1), base = lane - c;  // first lane of the group within the wavefront
    const bool live = i < n_rows;
    const int w0 = live ? w_row_ptrs[i] : 0, bs = live ? w_row_ptrs[i + 1] - w0 : 0;
    const bool row_ok = bs <= G;                    // (a wider row: k_isai_generate_wide takes it)
    const bool on = live && row_ok && c < bs;
    const int Jc = on ? w_cols[w0 + c] : -1;
    // matrix row Jc, its first ROWE entries in registers (all loads in flight at once; a longer row -- a pattern beyond
    // the stencil -- falls back to the search in memory): general ISAI needs A(Jc, J[r]), this lane's own row; the spd
    // variant A(J[r], Jc), found in the registers of lane r
    constexpr int ROWE = 8;
    int ce[ROWE];
    double ve[ROWE];
    int rp0 = 0, len = 0;
    if (on) {
        rp0 = row_ptrs[Jc];
        len = row_ptrs[Jc + 1] - rp0;
    }
#pragma unroll
    for (int e = 0; e < ROWE; ++e) {
        ce[e] = (on && e < len) ? cols[rp0 + e] : -1;
        ve[e] = (on && e < len) ? vals[rp0 + e] : 0.0;
    }
    const bool longer = __any(on && len > ROWE);  // (wavefront-uniform: the shuffles below stay convergent)
    double a[G];
#pragma unroll
    for (int r = 0; r < G; ++r) {
        const int Jr = grp_get<G>(Jc, base, r);
        a[r] = 0.0;
        if (longer) {
            if (on && r < bs) a[r] = spd ? csr_entry(row_ptrs, cols, vals, Jr, Jc) : csr_entry(row_ptrs, cols, vals, Jc, Jr);
        } else if (spd) {
#pragma unroll
            for (int e = ROWE - 1; e >= 0; --e) {  // (descending: the first match in the row's order is the one that stays)
                const int col = grp_get<G>(ce[e], base, r);
                const double val = grp_get<G>(ve[e], base, r);
                if (on && r < bs && col == Jc) a[r] = val;
            }
        } else {
#pragma unroll
            for (int e = ROWE - 1; e >= 0; --e)
                if (on && r < bs && ce[e] == Jr) a[r] = ve[e];
        }
    }
    double rhs = (on && Jc == i) ? 1.0 : 0.0;  // of row c
    // (pos: which member is the row itself)
    int pos = 0;
#pragma unroll
    for (int r = 0; r < G; ++r)
        if (grp_get<G>(Jc, base, r) == i && r < bs) pos = r;
#pragma unroll
    for (int k = 0; k < G; ++k) {
        // pivot: lane k scans its column from row k down, first largest
        int piv = k;
        {
            double best = fabs(a[k]);
#pragma unroll
            for (int r = k + 1; r < G; ++r)
                if (r < bs && fabs(a[r]) > best) {
                    best = fabs(a[r]);
                    piv = r;
                }
        }
        piv = grp_get<G>(piv, base, k);
        const bool step = k < bs;  // (group-uniform)
        if (step && piv != k) {
            // rows k and piv change places: every lane in its column, lanes k and piv their right-hand sides
            double ak = a[k], ap = 0.0;
#pragma unroll
            for (int r = 0; r < G; ++r)
                if (r == piv) ap = a[r];
#pragma unroll
            for (int r = 0; r < G; ++r)
                if (r == piv) a[r] = ak;
            a[k] = ap;
        }
        {
            const double rk = grp_get<G>(rhs, base, k), rp = __shfl(rhs, base + (piv < G ? piv : k), WAVE);
            if (step && piv != k) {
                if (c == k) rhs = rp;
                else if (c == piv) rhs = rk;
            }
        }
        // eliminate below the pivot: the factors come from lane k (one division each, as in the serial walk)
        const double akk = grp_get<G>(a[k], base, k);
        const double rk = grp_get<G>(rhs, base, k);
        const double akc = a[k];
#pragma unroll
        for (int r = k + 1; r < G; ++r) {
            const double f = grp_get<G>(a[r] / akk, base, k);
            if (step && r < bs) {
                if (c > k) a[r] -= f * akc;
                if (c == r) rhs -= f * rk;
            }
        }
    }
    // back substitution: x[r] = (rhs[r] - sum_{j > r} a[r][j] x[j]) / a[r][r], j ascending; lane j owns a[r][j] and x[j]
#pragma unroll
    for (int r = G - 1; r >= 0; --r) {
        double tv = rhs;  // (meaningful in lane r)
#pragma unroll
        for (int j = r + 1; j < G; ++j) {
            const double pr = grp_get<G>(a[r] * rhs, base, j);  // lane j: a[r][j] * x[j] (x[j] is final by now)
            if (j < bs) tv -= pr;
        }
        if (c == r && r < bs) rhs = tv / a[r];
    }
    const double xpos = __shfl(rhs, base + pos, WAVE);
    if (on) w_vals[w0 + c] = spd ? rhs / sqrt(xpos) : rhs;
}

// The same solve for one WIDE row (ISAI_THREAD_ROW < entries <= MAX_ISAI_ROW) per wavefront: the
// dense system sits in LDS, lane c owns column c.  Every element sees the operations of the
// thread-per-row kernel (and of the oracle's solve_dense) in the same order, so the bits agree.
__global__ __launch_bounds__(WAVE) void k_isai_generate_wide(int n_wide, const int *__restrict__ wide_rows,
                                                           const int *__restrict__ row_ptrs,
                                                           const int *__restrict__ cols,
                                                           const double *__restrict__ vals, int spd,
                                                           const int *__restrict__ w_row_ptrs,
                                                           const int *__restrict__ w_cols,
                                                           double *__restrict__ w_vals)
{
    constexpr int LD = MAX_ISAI_ROW + 1;  // odd leading dimension: column walks hit distinct banks
    __shared__ double a[MAX_ISAI_ROW * LD];
    __shared__ double rhs[MAX_ISAI_ROW];
    __shared__ int Js[MAX_ISAI_ROW];
    __shared__ int piv_s;
    if ((int)blockIdx.x >= n_wide) return;
    const int i = wide_rows[blockIdx.x];
    const int w0 = w_row_ptrs[i], bs = w_row_ptrs[i + 1] - w0;
    const int c = threadIdx.x;  // this lane's column (and, where rows are walked in parallel, its row)
    const bool on = c < bs;
    const int Jc = on ? w_cols[w0 + c] : -1;
    if (on) {
        Js[c] = Jc;
        rhs[c] = Jc == i ? 1.0 : 0.0;
    }
    __syncthreads();
    for (int r = 0; r < bs; ++r)
        if (on) a[r * LD + c] = spd ? csr_entry(row_ptrs, cols, vals, Js[r], Jc) : csr_entry(row_ptrs, cols, vals, Jc, Js[r]);
    __syncthreads();
    for (int k = 0; k < bs; ++k) {
        // pivot: the first row >= k with the largest |a[r][k]| (lane r looks at row r)
        double mine = (on && c >= k) ? fabs(a[c * LD + k]) : -1.0;
        int idx = c;
#pragma unroll
        for (int off = WAVE / 2; off >= 1; off >>= 1) {
            const double ov = __shfl_xor(mine, off, WAVE);
            const int oi = __shfl_xor(idx, off, WAVE);
            if (ov > mine || (ov == mine && oi < idx)) {
                mine = ov;
                idx = oi;
            }
        }
        if (c == 0) piv_s = idx;
        __syncthreads();
        const int piv = piv_s;
        if (piv != k) {
            if (on) {
                const double tv = a[k * LD + c];
                a[k * LD + c] = a[piv * LD + c];
                a[piv * LD + c] = tv;
            }
            if (c == 0) {
                const double tv = rhs[k];
                rhs[k] = rhs[piv];
                rhs[piv] = tv;
            }
        }
        __syncthreads();
        // eliminate below the pivot: lane c updates column c (> k) of every row; lane k the right-hand side
        const double akk = a[k * LD + k];
        const double akc = on ? a[k * LD + c] : 0.0;
        const double rk = rhs[k];
        for (int r = k + 1; r < bs; ++r) {
            const double f = a[r * LD + k] / akk;
            if (on && c > k) a[r * LD + c] -= f * akc;
            if (c == k) rhs[r] -= f * rk;
        }
        __syncthreads();
    }
    if (c == 0) {  // back substitution, left to right like the oracle
        for (int r = bs - 1; r >= 0; --r) {
            double tv = rhs[r];
            for (int j = r + 1; j < bs; ++j) tv -= a[r * LD + j] * rhs[j];
            rhs[r] = tv / a[r * LD + r];
        }
    }
    __syncthreads();
    if (on) {
        double scale = 1.0;
        if (spd) {
            int pos = 0;
            for (int r = 0; r < bs; ++r)
                if (Js[r] == i) pos = r;
            scale = sqrt(rhs[pos]);
        }
        w_vals[w0 + c] = spd ? rhs[c] / scale : rhs[c];
    }
}

// One HUGE row (MAX_ISAI_ROW < entries <= MAX_ISAI_HUGE_ROW) per workgroup: the dense system lives in global
// scratch (row-major, leading dimension bs; it stays in L2), right-hand side and column list in LDS.  Elimination:
// every element of the trailing block sees `a[i][j] -= (a[i][k] / a[k][k]) * a[k][j]` exactly as in the oracle's
// solve_dense_wide (same factor expression, products and differences rounded separately); pivot = the first row
// with the largest |a[r][k]|; back substitution column by column.
__global__ __launch_bounds__(BLOCK) void k_isai_generate_huge(const int *__restrict__ huge_rows,
                                                             const long long *__restrict__ scratch_off,
                                                             double *__restrict__ scratch,
                                                             const int *__restrict__ row_ptrs,
                                                             const int *__restrict__ cols,
                                                             const double *__restrict__ vals, int spd,
                                                             const int *__restrict__ w_row_ptrs,
                                                             const int *__restrict__ w_cols,
                                                             double *__restrict__ w_vals)
{
    __shared__ int Js[MAX_ISAI_HUGE_ROW];
    __shared__ double rhs[MAX_ISAI_HUGE_ROW];
    __shared__ double red_v[N_WAVES];
    __shared__ int red_i[N_WAVES];
    __shared__ int piv_s;
    const int i = huge_rows[blockIdx.x];
    const int w0 = w_row_ptrs[i], bs = w_row_ptrs[i + 1] - w0;
    double *a = scratch + scratch_off[blockIdx.x];
    const int tid = threadIdx.x;
    for (int c = tid; c < bs; c += BLOCK) {
        const int J = w_cols[w0 + c];
        Js[c] = J;
        rhs[c] = J == i ? 1.0 : 0.0;
    }
    __syncthreads();
    for (int r = tid / WAVE; r < bs; r += N_WAVES)
        for (int c = tid & (WAVE - 1); c < bs; c += WAVE)
            a[(long)r * bs + c] =
                spd ? csr_entry(row_ptrs, cols, vals, Js[r], Js[c]) : csr_entry(row_ptrs, cols, vals, Js[c], Js[r]);
    __syncthreads();
    for (int k = 0; k < bs; ++k) {
        // pivot: the first row >= k with the largest |a[r][k]|
        double best = -1.0;
        int bi = 0x7fffffff;
        for (int r = k + tid; r < bs; r += BLOCK) {
            const double v = fabs(a[(long)r * bs + k]);
            if (v > best) {
                best = v;
                bi = r;
            }
        }
#pragma unroll
        for (int off = WAVE / 2; off >= 1; off >>= 1) {
            const double ov = __shfl_xor(best, off, WAVE);
            const int oi = __shfl_xor(bi, off, WAVE);
            if (ov > best || (ov == best && oi < bi)) {
                best = ov;
                bi = oi;
            }
        }
        if ((tid & (WAVE - 1)) == 0) {
            red_v[tid / WAVE] = best;
            red_i[tid / WAVE] = bi;
        }
        __syncthreads();
        if (tid == 0) {
            double bv = red_v[0];
            int bidx = red_i[0];
            for (int w = 1; w < N_WAVES; ++w)
                if (red_v[w] > bv || (red_v[w] == bv && red_i[w] < bidx)) {
                    bv = red_v[w];
                    bidx = red_i[w];
                }
            // (a column of NaNs -- diverged coefficients, a zero pivot earlier -- wins no comparison: keep row k, as the
            //  oracle's search does, and let the NaN propagate into W instead of indexing outside the scratch)
            piv_s = (bidx >= k && bidx < bs) ? bidx : k;
        }
        __syncthreads();
        const int piv = piv_s;
        if (piv != k) {
            for (int j = tid; j < bs; j += BLOCK) {
                const double t = a[(long)k * bs + j];
                a[(long)k * bs + j] = a[(long)piv * bs + j];
                a[(long)piv * bs + j] = t;
            }
            if (tid == 0) {
                const double t = rhs[k];
                rhs[k] = rhs[piv];
                rhs[piv] = t;
            }
            __syncthreads();
        }
        // trailing block + right-hand side: a wavefront per row (lanes along the row: coalesced, no index division),
        // the row's factor formed once per lane from the same expression
        const double akk = a[(long)k * bs + k], rk = rhs[k];
        const int lane = tid & (WAVE - 1);
        for (int r = k + 1 + tid / WAVE; r < bs; r += N_WAVES) {
            const double f = a[(long)r * bs + k] / akk;
            for (int c = k + 1 + lane; c < bs; c += WAVE) a[(long)r * bs + c] -= f * a[(long)k * bs + c];
            if (lane == 0) rhs[r] -= f * rk;
        }
        __syncthreads();
    }
    for (int r = bs - 1; r >= 0; --r) {  // column-wise back substitution
        if (tid == 0) rhs[r] = rhs[r] / a[(long)r * bs + r];
        __syncthreads();
        const double xr = rhs[r];
        for (int q = tid; q < r; q += BLOCK) rhs[q] -= a[(long)q * bs + r] * xr;
        __syncthreads();
    }
    double scale = 1.0;
    if (spd) {
        int pos = 0;
        for (int r = 0; r < bs; ++r)
            if (Js[r] == i) pos = r;
        scale = sqrt(rhs[pos]);
    }
    for (int c = tid; c < bs; c += BLOCK) w_vals[w0 + c] = spd ? rhs[c] / scale : rhs[c];
}

__global__ __launch_bounds__(BLOCK) void k_permute_scatter(int n, const int *__restrict__ new_id,
                                                           const double *__restrict__ in,
                                                           double *__restrict__ out)
{
    const int i = blockIdx.x * BLOCK + threadIdx.x;
    if (i < n) out[new_id[i]] = in[i];
}

__global__ __launch_bounds__(BLOCK) void k_permute_gather(int n, const int *__restrict__ new_id,
                                                          const double *__restrict__ in,
                                                          double *__restrict__ out)
{
    const int i = blockIdx.x * BLOCK + threadIdx.x;
    if (i < n) out[i] = in[new_id[i]];
}

// Block Jacobi through a permutation, staged (launch_bj_apply_staged): the input once into the CALLER's order
// (out[i] = in[idx[i]], one gather per row instead of one per block member), the contiguous apply there, and back:
__global__ __launch_bounds__(BLOCK) void k_gather_gated(int n, const int *__restrict__ idx, const double *__restrict__ in,
                                                        double *__restrict__ out, const DevScalars *gate)
{
    if (gate && gate->stop) return;
    const int i = blockIdx.x * BLOCK + threadIdx.x;
    if (i < n) out[i] = in[idx[i]];
}
// out[row] = src[idx[row]] for this chunk's rows, and (NDOT) the chunk's partial of sum_i w_i * out_i in the
// canonical per-chunk tree (k_partials' bits)
template <int NDOT>
__global__ __launch_bounds__(BLOCK) void k_gather_back_dot(int n, const int *__restrict__ idx,
                                                           const double *__restrict__ src, double *__restrict__ out,
                                                           const double *__restrict__ w,
                                                           double *__restrict__ dot_part, const DevScalars *gate)
{
    __shared__ double slot[N_WAVES];
    if (gate && gate->stop) return;
    const int chunk = blockIdx.x;
    const RowPair rp = my_rows(chunk, n);
    double2 v;
    v.x = rp.n > 0 ? src[idx[rp.row]] : 0.0;
    v.y = rp.n > 1 ? src[idx[rp.row + 1]] : 0.0;
    st2(out, rp, v);
    if (NDOT) {
        const double2 vw = ld2(w, rp);
        double d = 0.0;
        if (rp.n > 0) d += vw.x * v.x;
        if (rp.n > 1) d += vw.y * v.y;
        const double s = block_sum(d, slot);
        if (threadIdx.x == 0) dot_part[chunk] = s;
    }
}

// block-Jacobi apply (DevBlockJacobi): one row per thread, CHUNK_ROWS threads per workgroup (the
// dependent index loads want many rows in flight).  NDOT = 1: also the chunk's partial of
// sum_i in_i*out_i (CG's rho = r . M^-1 r): the products go through LDS to the first BLOCK
// threads, which add their two rows and run the usual per-chunk tree -- same bits as k_partials.
template <int NDOT>
__global__ __launch_bounds__(CHUNK_ROWS) void k_bj_apply(int n_rows,
                                                         const int *__restrict__ block_ptrs,
                                                         const int *__restrict__ row_block,
                                                         const double *__restrict__ blocks, int ld,
                                                         int uniform,
                                                         const double *__restrict__ in,
                                                         double *__restrict__ out,
                                                         double *__restrict__ dot_part,
                                                         const DevScalars *gate, const int *__restrict__ rows,
                                                         const int *__restrict__ pos)
{
    __shared__ double prod[CHUNK_ROWS];
    __shared__ double slot[N_WAVES];
    if (gate && gate->stop) return;
    const int chunk = blockIdx.x;
    const int row = chunk * CHUNK_ROWS + threadIdx.x;
    double sum = 0.0, mine = 0.0;
    if (row < n_rows) {
        // (renumbered device copy: blocks are runs of rows of the CALLER's numbering -- this row is position
        //  pos[row] there, the block's j-th member is device row rows[r0 + j])
        const int at = rows ? pos[row] : row;
        int b, r0, bs;
        if (uniform) {  // blocks of exactly `ld` rows (the last one may be shorter)
            b = at / ld;
            r0 = b * ld;
            bs = min(ld, n_rows - r0);
        } else {
            b = row_block[at];
            r0 = block_ptrs[b];
            bs = block_ptrs[b + 1] - r0;
        }
        const double *a = rows ? blocks + (size_t)row * ld : blocks + (size_t)b * ld * ld + (size_t)(at - r0) * ld;
        if (rows)
            for (int j = 0; j < bs; ++j) sum += a[j] * in[rows[r0 + j]];
        else
            for (int j = 0; j < bs; ++j) sum += a[j] * in[r0 + j];
        out[row] = sum;
        mine = in[row];
    }
    if (NDOT >= 1) {
        prod[threadIdx.x] = mine * sum;
        __syncthreads();
        double d = 0.0;
        if (threadIdx.x < BLOCK) {
            const RowPair rp = my_rows(chunk, n_rows);
            if (rp.n > 0) d += prod[ROWS_PER_THREAD * threadIdx.x];
            if (rp.n > 1) d += prod[ROWS_PER_THREAD * threadIdx.x + 1];
            d = wave_sum(d);
            if ((threadIdx.x & (WAVE - 1)) == 0) slot[threadIdx.x / WAVE] = d;
        }
        __syncthreads();
        if (threadIdx.x == 0) {
            double s = slot[0];
            for (int w = 1; w < N_WAVES; ++w) s += slot[w];
            dot_part[chunk] = s;
        }
    }
}

// Block Jacobi through a permutation in ONE pass over the blocks (launch_bj_apply_staged, fused form): a workgroup takes
// CHUNK_ROWS consecutive positions of the CALLER's order, every thread gathers the input of its own position once
// (in[rows[position]]; the blocks' members are far apart on the device, so this is the one scattered read per row
// the staged form also pays) into LDS together with the MAX_JACOBI_BLOCK - 1 positions either side that a block
// straddling the workgroup's range may need, applies its block row from there (same products, same left-to-right
// sum as k_bj_apply) and stores the result at its device row.  The dot partials of the device order follow in a
// pass of their own (k_partials): 2 launches and 52 + 16 bytes per row (block size 4) instead of 3 launches and 100.
__global__ __launch_bounds__(CHUNK_ROWS) void k_bj_apply_perm(int n_rows, const int *__restrict__ block_ptrs,
                                                              const int *__restrict__ row_block,
                                                              const double *__restrict__ blocks, int ld, int uniform,
                                                              const double *__restrict__ in, double *__restrict__ out,
                                                              const DevScalars *gate, const int *__restrict__ rows,
                                                              int xgroup)
{
    constexpr int HALO = MAX_JACOBI_BLOCK - 1;
    __shared__ double v[CHUNK_ROWS + 2 * HALO];
    if (gate && gate->stop) return;
    // slabs of `xgroup` consecutive ranges per XCD: the scattered 8-byte reads of `in` and writes of `out` of neighbouring
    // positions land in the same 64-byte sectors (positions close in the caller's order are close in the device's), and
    // only one L2 fetches / merges a sector when its positions run on ONE XCD -- with the ranges dealt round-robin every
    // sector crossed the fabric once per XCD (2.1 M rows shuffled in windows of 65,536, BJ(4): 53 -> 31 us per apply, 137 -> 114 us per turn)
    const int range = xcd_chunk(blockIdx.x, xgroup);
    if ((long)range * CHUNK_ROWS >= n_rows) return;
    const int c0 = range * CHUNK_ROWS, at = c0 + (int)threadIdx.x;
    const int dev = at < n_rows ? rows[at] : -1;
    v[HALO + threadIdx.x] = dev >= 0 ? in[dev] : 0.0;
    if (threadIdx.x < 2 * HALO) {  // the positions before and behind the range
        const int h = threadIdx.x < HALO ? c0 - HALO + (int)threadIdx.x : c0 + CHUNK_ROWS + (int)threadIdx.x - HALO;
        v[threadIdx.x < HALO ? threadIdx.x : CHUNK_ROWS + threadIdx.x] = (h >= 0 && h < n_rows) ? in[rows[h]] : 0.0;
    }
    __syncthreads();
    if (dev < 0) return;
    int b, r0, bs;
    if (uniform) {  // blocks of exactly `ld` rows (the last one may be shorter)
        b = at / ld;
        r0 = b * ld;
        bs = min(ld, n_rows - r0);
    } else {
        b = row_block[at];
        r0 = block_ptrs[b];
        bs = block_ptrs[b + 1] - r0;
    }
    const double *a = blocks + (size_t)b * ld * ld + (size_t)(at - r0) * ld;
    const double *x = v + HALO + (r0 - c0);
    double sum = 0.0;
    for (int j = 0; j < bs; ++j) sum += a[j] * x[j];
    out[dev] = sum;
}

}  // namespace

// ------------------------------------------------------------------------------------------
// launchers
// ------------------------------------------------------------------------------------------
void launch_gather_coeffs(hipStream_t st, int32_t nnz, const int32_t *ldu_mapping,
                          const double *source, double *coeffs)
{
    if (nnz == 0) return;
    hipLaunchKernelGGL(k_gather_coeffs, dim3(blocks_for(((int64_t)nnz + 3) / 4)), dim3(BLOCK), 0,
                       st, nnz, ldu_mapping, source, coeffs);
}

void launch_jacobi_generate_pos(hipStream_t st, const DevCsr &A, const int32_t *diag_pos, double *inv_diag)
{
    if (A.n_rows == 0) return;
    hipLaunchKernelGGL(k_jacobi_generate_pos, dim3(blocks_for(A.n_rows)), dim3(BLOCK), 0, st, A.n_rows,
                       diag_pos, A.vals, inv_diag);
}

void launch_jacobi_generate_diag(hipStream_t st, int32_t n_rows, const double *diag, double *inv_diag)
{
    if (n_rows == 0) return;
    hipLaunchKernelGGL(k_jacobi_generate_diag, dim3(blocks_for(n_rows)), dim3(BLOCK), 0, st, n_rows, diag, inv_diag);
}

void launch_jacobi_generate(hipStream_t st, const DevCsr &A, double *inv_diag)
{
    if (A.n_rows == 0) return;
    hipLaunchKernelGGL(k_jacobi_generate, dim3(blocks_for(A.n_rows)), dim3(BLOCK), 0, st, A.n_rows,
                       A.row_ptrs, A.cols, A.vals, inv_diag);
}

void launch_bj_generate(hipStream_t st, const DevCsr &A, const DevBlockJacobi &J, bool group_lanes)
{
    if (J.n_blocks == 0) return;
    const dim3 grid((J.n_blocks + 63) / 64), block(64);
#define OGL_BJ(LD)                                                                              \
    hipLaunchKernelGGL((k_bj_generate<LD>), grid, block, 0, st, J.n_blocks, J.block_ptrs,        \
                       A.row_ptrs, A.cols, A.vals, J.blocks, J.stride, J.rows, J.pos, J.by_device_row)
#define OGL_BJ_GRP(G)                                                                                                   \
    hipLaunchKernelGGL((k_bj_generate_grp<G>), dim3((unsigned)(((int64_t)J.n_blocks * G + BLOCK - 1) / BLOCK)), dim3(BLOCK), 0,  \
                       st, J.n_blocks, J.block_ptrs, A.row_ptrs, A.cols, A.vals, J.blocks, J.stride, J.rows, J.pos,      \
                       J.by_device_row)
    if (J.stride <= 4 && J.stride > 1 && group_lanes)
        OGL_BJ_GRP(4);
    else if (J.stride <= 8 && J.stride > 1 && group_lanes)
        OGL_BJ_GRP(8);
    else if (J.stride <= 2)
        OGL_BJ(2);
    else if (J.stride <= 4)
        OGL_BJ(4);
    else if (J.stride <= 8)
        OGL_BJ(8);
    else if (J.stride <= 16)
        OGL_BJ(16);
    else
        OGL_BJ(32);
#undef OGL_BJ
#undef OGL_BJ_GRP
}

void launch_bj_apply(hipStream_t st, const DevBlockJacobi &J, const double *in, double *out,
                     double *dot_part, const DevScalars *gate)
{
    if (J.n_rows == 0) return;
    const dim3 grid((unsigned)n_chunks(J.n_rows)), block(CHUNK_ROWS);
    if (dot_part)
        hipLaunchKernelGGL((k_bj_apply<1>), grid, block, 0, st, J.n_rows, J.block_ptrs, J.row_block,
                           J.blocks, J.stride, J.uniform, in, out, dot_part, gate, J.rows, J.pos);
    else
        hipLaunchKernelGGL((k_bj_apply<0>), grid, block, 0, st, J.n_rows, J.block_ptrs, J.row_block,
                           J.blocks, J.stride, J.uniform, in, out, dot_part, gate, J.rows, J.pos);
}

void launch_bj_apply_staged(hipStream_t st, const DevBlockJacobi &J, const double *in, double *out, double *dot_part,
                            const DevScalars *gate, double *tmp_in, double *tmp_out)
{
    if (J.n_rows == 0) return;
    if (!tmp_in) {  // the fused form: one pass over the blocks in the caller's order, then the dot partials
        const int nc = (int)n_chunks(J.n_rows);
        const int xg = J.perm_xcd_group > 0 ? J.perm_xcd_group : std::max(4, std::min(256, nc / 32));
        hipLaunchKernelGGL(k_bj_apply_perm, dim3((unsigned)xcd_grid(nc, xg)), dim3(CHUNK_ROWS), 0, st, J.n_rows,
                           J.block_ptrs, J.row_block, J.blocks, J.stride, J.uniform, in, out, gate, J.rows, xg);
        if (dot_part) launch_partials_dot(st, J.n_rows, in, out, dot_part, gate);
        return;
    }
    hipLaunchKernelGGL(k_gather_gated, dim3(blocks_for(J.n_rows)), dim3(BLOCK), 0, st, J.n_rows, J.rows, in, tmp_in, gate);
    DevBlockJacobi C = J;  // the blocks as they lie in the caller's order
    C.rows = C.pos = nullptr;
    launch_bj_apply(st, C, tmp_in, tmp_out, nullptr, gate);
    const dim3 grid((unsigned)n_chunks(J.n_rows)), block(BLOCK);
    if (dot_part)
        hipLaunchKernelGGL((k_gather_back_dot<1>), grid, block, 0, st, J.n_rows, J.pos, tmp_out, out, in, dot_part, gate);
    else
        hipLaunchKernelGGL((k_gather_back_dot<0>), grid, block, 0, st, J.n_rows, J.pos, tmp_out, out, in, dot_part, gate);
}

void launch_isai_generate(hipStream_t st, const DevCsr &A, int spd, const int32_t *w_row_ptrs,
                          const int32_t *w_cols, double *w_vals, int32_t max_row,
                          const int32_t *wide_rows, int32_t n_wide, bool group_lanes)
{
    if (A.n_rows == 0) return;
    const dim3 grid((A.n_rows + 63) / 64), block(64);
#define OGL_ISAI(LD)                                                                             \
    hipLaunchKernelGGL((k_isai_generate<LD>), grid, block, 0, st, A.n_rows, A.row_ptrs, A.cols,   \
                       A.vals, spd, w_row_ptrs, w_cols, w_vals)
    if (max_row <= 4 && group_lanes)
        hipLaunchKernelGGL((k_isai_generate_grp<4>), dim3((unsigned)(((int64_t)A.n_rows * 4 + BLOCK - 1) / BLOCK)), dim3(BLOCK), 0,
                           st, A.n_rows, A.row_ptrs, A.cols, A.vals, spd, w_row_ptrs, w_cols, w_vals);
    else if (max_row <= 8 && group_lanes)
        hipLaunchKernelGGL((k_isai_generate_grp<8>), dim3((unsigned)(((int64_t)A.n_rows * 8 + BLOCK - 1) / BLOCK)), dim3(BLOCK), 0,
                           st, A.n_rows, A.row_ptrs, A.cols, A.vals, spd, w_row_ptrs, w_cols, w_vals);
    else if (max_row <= 8)
        OGL_ISAI(8);
    else if (max_row <= 16)
        OGL_ISAI(16);
    else
        OGL_ISAI(32);
#undef OGL_ISAI
    if (n_wide > 0)
        hipLaunchKernelGGL(k_isai_generate_wide, dim3(n_wide), dim3(WAVE), 0, st, n_wide, wide_rows,
                           A.row_ptrs, A.cols, A.vals, spd, w_row_ptrs, w_cols, w_vals);
}

void launch_isai_generate_huge(hipStream_t st, const DevCsr &A, int spd, const int32_t *w_row_ptrs,
                               const int32_t *w_cols, double *w_vals, const int32_t *huge_rows,
                               const int64_t *scratch_off, int32_t first, int32_t count, double *scratch)
{
    if (count <= 0) return;
    static_assert(sizeof(long long) == sizeof(int64_t), "scratch offsets");
    hipLaunchKernelGGL(k_isai_generate_huge, dim3(count), dim3(BLOCK), 0, st, huge_rows + first,
                       reinterpret_cast<const long long *>(scratch_off) + first, scratch, A.row_ptrs, A.cols, A.vals,
                       spd, w_row_ptrs, w_cols, w_vals);
}

void launch_permute_scatter(hipStream_t st, int32_t n, const int32_t *new_id, const double *in, double *out)
{
    if (n == 0) return;
    hipLaunchKernelGGL(k_permute_scatter, dim3(blocks_for(n)), dim3(BLOCK), 0, st, n, new_id, in, out);
}

void launch_permute_gather(hipStream_t st, int32_t n, const int32_t *new_id, const double *in, double *out)
{
    if (n == 0) return;
    hipLaunchKernelGGL(k_permute_gather, dim3(blocks_for(n)), dim3(BLOCK), 0, st, n, new_id, in, out);
}

}  // namespace ogl
