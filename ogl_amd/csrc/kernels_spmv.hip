// kernels_spmv.hip -- CSR-stream (32-bit and packed 21-bit columns), ELL and boundary SpMV, send-buffer pack
// (geometry, reduction tree and the -ffp-contract=off rule: device_common.hpp)
#include "device_common.hpp"

namespace ogl {

namespace {

// ------------------------------------------------------------------------------------------
// CSR-stream SpMV (K2).  A workgroup streams its chunk's non-zeros with 16-byte loads
// (2 x double2 values + 1 x int4 columns per lane and step, 1 KiB per wave instruction),
// gathers x (served by L2 / Infinity Cache), parks the products in LDS, and then every thread
// adds up its own rows left to right -- the same order as the reference executor's row loop,
// so y is bit-identical to it.
// ------------------------------------------------------------------------------------------
// NDOT = 1: partials of sum_i w_i*y_i (w = x for CG's p.q, w = rr or s for BiCGStab);
// NDOT = 2: additionally partials of sum_i y_i*y_i (BiCGStab's t.t).
// STREAM: the matrix is larger than the Infinity Cache -- values and columns are streamed past the caches
// (non-temporal), which then hold the vectors; a matrix that fits keeps the default policy and is served from
// the cache turn after turn.
// ROUNDS: the 4096 entries a workgroup has in registers per pass go through LDS in this many rounds (products of round h
// stored, rows summed over that range, next round): 2 rounds = 16 KiB of LDS instead of 32, six resident workgroups per CU
// instead of four (84 VGPRs), same order of additions.
template <int MODE, int NDOT, bool STREAM, int ROUNDS>
__global__ __launch_bounds__(BLOCK) void k_spmv_stream(
    int n_rows, int n_chunks, const int *__restrict__ row_ptrs, const int *__restrict__ cols,
    const double *__restrict__ vals, const double *__restrict__ x, const double *__restrict__ b,
    double *__restrict__ y, const double *__restrict__ w, double *__restrict__ dot_partials,
    double *__restrict__ dot2_partials, const DevScalars *gate, int xgroup, HaloFused hf,
    const int *__restrict__ block_order)
{
    constexpr int LDS_TILE = SPMV_TILE / ROUNDS;
    static_assert(LDS_TILE >= CHUNK_ROWS && SPMV_TILE % (ROUNDS * BLOCK * 2) == 0, "rounds of whole groups; halo_fused_add needs CHUNK_ROWS doubles");
    __shared__ __attribute__((aligned(16))) double prod[LDS_TILE];
    __shared__ double slot[N_WAVES];
    if (gate && gate->stop) return;
    // (banded patterns: the chunks of rows r and r +- band on one XCD, band_block_order -- as the half-storage kernels)
    const int chunk = block_order ? block_order[blockIdx.x] : xcd_chunk(blockIdx.x, xgroup);
    if (chunk < 0 || chunk >= n_chunks) return;
    const int tid = threadIdx.x;
    const int r0 = chunk * CHUNK_ROWS;
    const int r1 = min(r0 + CHUNK_ROWS, n_rows);
    const int nz0 = row_ptrs[r0];
    const int nz1 = row_ptrs[r1];

    // this thread's rows
    const int row = r0 + tid * ROWS_PER_THREAD;
    int rs[ROWS_PER_THREAD + 1];
#pragma unroll
    for (int j = 0; j <= ROWS_PER_THREAD; ++j) rs[j] = row_ptrs[min(row + j, r1)];
    double acc[ROWS_PER_THREAD];
#pragma unroll
    for (int j = 0; j < ROWS_PER_THREAD; ++j)
        acc[j] = (MODE == SPMV_RESIDUAL && row + j < r1) ? b[row + j] : 0.0;

    // Two consecutive entries per lane and load (16 B of values, 8 B of columns): every load instruction of a
    // wavefront covers whole, disjoint cache lines, so values and columns -- read exactly once per launch -- can
    // be streamed past the caches (non-temporal), which then hold the vectors.  (With four entries per lane as
    // two 16-byte loads the two instructions share their lines and a non-temporal hint fetches them twice:
    // 208 us instead of 186, profiles/spmv_tune_r02.txt.)
    constexpr int GROUPS = SPMV_TILE / (BLOCK * 2);
    constexpr int RGROUPS = GROUPS / ROUNDS;  // groups per LDS round
    typedef double d2v __attribute__((ext_vector_type(2)));
    typedef int i2v __attribute__((ext_vector_type(2)));
    for (int t0 = nz0 & ~3; t0 < nz1; t0 += SPMV_TILE) {
        d2v va[GROUPS];
        i2v cc[GROUPS];
#pragma unroll
        for (int g = 0; g < GROUPS; ++g) {
            const int e = t0 + (g * BLOCK + tid) * 2;
            const int ec = e < nz1 ? e : t0;  // clamp: stay inside the (padded) arrays
            if (STREAM) {
                va[g] = __builtin_nontemporal_load(reinterpret_cast<const d2v *>(vals + ec));
                cc[g] = __builtin_nontemporal_load(reinterpret_cast<const i2v *>(cols + ec));
            } else {
                va[g] = *reinterpret_cast<const d2v *>(vals + ec);
                cc[g] = *reinterpret_cast<const i2v *>(cols + ec);
            }
        }
        double2 pr[GROUPS];
#pragma unroll
        for (int g = 0; g < GROUPS; ++g) {
            const double x0 = x[cc[g].x], x1 = x[cc[g].y];
            pr[g].x = va[g].x * x0;
            pr[g].y = va[g].y * x1;
        }
#pragma unroll
        for (int h = 0; h < ROUNDS; ++h) {
            const int h0 = t0 + h * LDS_TILE;
            if (ROUNDS > 1 && h0 >= nz1) break;  // (workgroup-uniform)
#pragma unroll
            for (int g = 0; g < RGROUPS; ++g)
                *reinterpret_cast<double2 *>(prod + (g * BLOCK + tid) * 2) = pr[h * RGROUPS + g];
            __syncthreads();
            const int h1 = h0 + LDS_TILE;
#pragma unroll
            for (int j = 0; j < ROWS_PER_THREAD; ++j) {
                const int kb = max(rs[j], h0), ke = min(rs[j + 1], h1);
                for (int k = kb; k < ke; ++k) {
                    if (MODE == SPMV_RESIDUAL)
                        acc[j] -= prod[k - h0];
                    else
                        acc[j] += prod[k - h0];
                }
            }
            __syncthreads();
        }
    }

    if (hf.chunk_bptr) halo_fused_add<MODE>(hf, chunk, acc[0], acc[1], prod);
    double d = 0.0, d2 = 0.0;
#pragma unroll
    for (int j = 0; j < ROWS_PER_THREAD; ++j) {
        if (row + j < r1) {
            y[row + j] = acc[j];
            if (NDOT >= 1) d += w[row + j] * acc[j];
            if (NDOT >= 2) d2 += acc[j] * acc[j];
        }
    }
    if (NDOT >= 1) {
        const double s = block_sum(d, slot);
        if (tid == 0) dot_partials[chunk] = s;
    }
    if (NDOT >= 2) {
        const double s = block_sum(d2, slot);
        if (tid == 0) dot2_partials[chunk] = s;
    }
}

// The same with the columns read from the packed stream (Stream21Chunk, common.hpp): one 16-byte word brings the
// six columns a lane needs for a group -- entries (k * 512 + 2 * lane, + 1), k = 0..2, of the group's 1536 -- as
// 21-bit offsets from the chunk's smallest column.  Values, row phase and sums as above: same bits.
template <int MODE, int NDOT, bool STREAM>
__global__ __launch_bounds__(BLOCK) void k_spmv_stream21(
    int n_rows, int n_chunks, const int *__restrict__ row_ptrs, const Stream21Chunk *__restrict__ chunks21,
    const uint4 *__restrict__ codes, const double *__restrict__ vals, const double *__restrict__ x,
    const double *__restrict__ b, double *__restrict__ y, const double *__restrict__ w,
    double *__restrict__ dot_partials, double *__restrict__ dot2_partials, const DevScalars *gate, int xgroup,
    HaloFused hf, const int *__restrict__ far_idx, const int *__restrict__ far_col)
{
    __shared__ __attribute__((aligned(16))) double prod[STREAM21_TILE];
    __shared__ double slot[N_WAVES];
    if (gate && gate->stop) return;
    const int chunk = xcd_chunk(blockIdx.x, xgroup);
    if (chunk >= n_chunks) return;
    const int tid = threadIdx.x;
    const int r0 = chunk * CHUNK_ROWS;
    const int r1 = min(r0 + CHUNK_ROWS, n_rows);
    const int nz0 = row_ptrs[r0];
    const int nz1 = row_ptrs[r1];
    const Stream21Chunk ck = chunks21[chunk];
    const int row = r0 + tid * ROWS_PER_THREAD;
    int rs[ROWS_PER_THREAD + 1];
#pragma unroll
    for (int j = 0; j <= ROWS_PER_THREAD; ++j) rs[j] = row_ptrs[min(row + j, r1)];
    double acc[ROWS_PER_THREAD];
#pragma unroll
    for (int j = 0; j < ROWS_PER_THREAD; ++j)
        acc[j] = (MODE == SPMV_RESIDUAL && row + j < r1) ? b[row + j] : 0.0;
    typedef double d2v __attribute__((ext_vector_type(2)));
    typedef unsigned u4v __attribute__((ext_vector_type(4)));
    const uint4 *cw = codes + ck.word_off + tid;
    constexpr unsigned long long M = (1ull << STREAM21_BITS) - 1;
    int tile = 0;
    for (int t0 = nz0 & ~3; t0 < nz1; t0 += STREAM21_TILE, ++tile) {
        d2v va[STREAM21_GROUPS][3];
        u4v cc[STREAM21_GROUPS];
#pragma unroll
        for (int g = 0; g < STREAM21_GROUPS; ++g) {
            const u4v *cp = reinterpret_cast<const u4v *>(cw + (long)(tile * STREAM21_GROUPS + g) * BLOCK);
            cc[g] = STREAM ? __builtin_nontemporal_load(cp) : *cp;
#pragma unroll
            for (int k = 0; k < 3; ++k) {
                const int e = t0 + ((g * 3 + k) * BLOCK + tid) * 2;
                const int ec = e < nz1 ? e : t0;  // clamp: stay inside the (padded) array
                va[g][k] = STREAM ? __builtin_nontemporal_load(reinterpret_cast<const d2v *>(vals + ec))
                                  : *reinterpret_cast<const d2v *>(vals + ec);
            }
        }
#pragma unroll
        for (int g = 0; g < STREAM21_GROUPS; ++g) {
            const unsigned long long lo = (unsigned long long)cc[g].x | ((unsigned long long)cc[g].y << 32);
            const unsigned long long hi = (unsigned long long)cc[g].z | ((unsigned long long)cc[g].w << 32);
            int c[6];
            c[0] = ck.base + (int)(lo & M);
            c[1] = ck.base + (int)((lo >> 21) & M);
            c[2] = ck.base + (int)((lo >> 42) & M);
            c[3] = ck.base + (int)(((lo >> 63) | (hi << 1)) & M);
            c[4] = ck.base + (int)((hi >> 20) & M);
            c[5] = ck.base + (int)((hi >> 41) & M);
            double xv[6];
#pragma unroll
            for (int i = 0; i < 6; ++i) xv[i] = x[c[i]];
#pragma unroll
            for (int k = 0; k < 3; ++k) {
                double2 p0;
                p0.x = va[g][k].x * xv[2 * k];
                p0.y = va[g][k].y * xv[2 * k + 1];
                *reinterpret_cast<double2 *>(prod + ((g * 3 + k) * BLOCK + tid) * 2) = p0;
            }
        }
        __syncthreads();
        const int t1 = t0 + STREAM21_TILE;
        if (ck.far_n) {  // the chunk's far entries (coded as offset 0 above): their products put right (workgroup-uniform)
            for (int i = tid; i < ck.far_n; i += BLOCK) {
                const int e = far_idx[ck.far_off + i];
                if (e >= t0 && e < t1) prod[e - t0] = vals[e] * x[far_col[ck.far_off + i]];
            }
            __syncthreads();
        }
#pragma unroll
        for (int j = 0; j < ROWS_PER_THREAD; ++j) {
            const int kb = max(rs[j], t0), ke = min(rs[j + 1], t1);
            for (int k = kb; k < ke; ++k) {
                if (MODE == SPMV_RESIDUAL)
                    acc[j] -= prod[k - t0];
                else
                    acc[j] += prod[k - t0];
            }
        }
        __syncthreads();
    }
    if (hf.chunk_bptr) halo_fused_add<MODE>(hf, chunk, acc[0], acc[1], prod);
    double d = 0.0, d2 = 0.0;
#pragma unroll
    for (int j = 0; j < ROWS_PER_THREAD; ++j) {
        if (row + j < r1) {
            y[row + j] = acc[j];
            if (NDOT >= 1) d += w[row + j] * acc[j];
            if (NDOT >= 2) d2 += acc[j] * acc[j];
        }
    }
    if (NDOT >= 1) {
        const double s = block_sum(d, slot);
        if (tid == 0) dot_partials[chunk] = s;
    }
    if (NDOT >= 2) {
        const double s = block_sum(d2, slot);
        if (tid == 0) dot2_partials[chunk] = s;
    }
}

// y[row] (+/-)= A_non_local(row,:) * recv, continuing the accumulator the local kernel stored.
template <int MODE>
__global__ __launch_bounds__(BLOCK) void k_spmv_non_local(int n_boundary,
                                                          const int *__restrict__ boundary_rows,
                                                          const int *__restrict__ entry_ptrs,
                                                          const int *__restrict__ cols,
                                                          const double *__restrict__ vals,
                                                          const double *__restrict__ recv,
                                                          double *__restrict__ y,
                                                          const DevScalars *gate)
{
    if (gate && gate->stop) return;
    const int i = blockIdx.x * BLOCK + threadIdx.x;
    if (i >= n_boundary) return;
    const int row = boundary_rows[i];
    double acc = y[row];
    for (int k = entry_ptrs[i]; k < entry_ptrs[i + 1]; ++k) {
        const double t = vals[k] * recv[cols[k]];
        acc = (MODE == SPMV_RESIDUAL) ? acc - t : acc + t;
    }
    y[row] = acc;
}

__global__ __launch_bounds__(BLOCK) void k_pack(int n_send, const int *__restrict__ send_idxs,
                                                const double *__restrict__ x,
                                                double *__restrict__ send, const DevScalars *gate)
{
    if (gate && gate->stop) return;
    const int i = blockIdx.x * BLOCK + threadIdx.x;
    if (i < n_send) send[i] = x[send_idxs[i]];
}

// ------------------------------------------------------------------------------------------
// ELL SpMV (matrixFormat Ell).  Slot-major planes: every load is a 16-byte (values) / 8-byte
// (columns) coalesced access over the chunk's rows, no LDS, no row pointers; a thread owns rows
// 2t, 2t+1 and adds the slots in order (= stored column order; padding slots are skipped), so the
// result and the fused dot partials are bit-identical to the CSR kernel's.
// ------------------------------------------------------------------------------------------
template <int MODE, int NDOT, bool STREAM>
__global__ __launch_bounds__(BLOCK) void k_spmv_ell(int n_rows, int n_chunks, int width, long stride,
                                                    const int *__restrict__ cols,
                                                    const double *__restrict__ vals,
                                                    const double *__restrict__ x,
                                                    const double *__restrict__ b,
                                                    double *__restrict__ y,
                                                    const double *__restrict__ w,
                                                    double *__restrict__ dot_partials,
                                                    double *__restrict__ dot2_partials,
                                                    const DevScalars *gate, HaloFused hf)
{
    __shared__ double slot[N_WAVES];
    __shared__ double ys[CHUNK_ROWS];
    if (gate && gate->stop) return;
    const int chunk = xcd_chunk(blockIdx.x);
    if (chunk >= n_chunks) return;
    const RowPair rp = my_rows(chunk, n_rows);
    double2 acc;
    acc.x = acc.y = 0.0;
    if (MODE == SPMV_RESIDUAL) acc = ld2(b, rp);
    // the planes are padded to an even stride (+2), so the pair load of the last odd row is in bounds
    const long r = rp.n > 0 ? rp.row : 0;
    constexpr int BATCH = 8;
    for (int i0 = 0; i0 < width; i0 += BATCH) {
        double2 v[BATCH];
        int2 c[BATCH];
#pragma unroll
        for (int k = 0; k < BATCH; ++k) {
            const int i = min(i0 + k, width - 1);  // clamp: always a valid plane
            if (STREAM) {  // planes larger than the Infinity Cache: read once, streamed past the caches
                typedef double d2v __attribute__((ext_vector_type(2)));
                typedef int i2v __attribute__((ext_vector_type(2)));
                v[k].x = v[k].y = 0.0;
                c[k].x = c[k].y = -1;
                if (i0 + k < width) {  // (no second, clamped read of the last plane: it would be fetched again)
                    const d2v tv = __builtin_nontemporal_load(reinterpret_cast<const d2v *>(vals + (long)i * stride + r));
                    const i2v tc = __builtin_nontemporal_load(reinterpret_cast<const i2v *>(cols + (long)i * stride + r));
                    v[k].x = tv.x;
                    v[k].y = tv.y;
                    c[k].x = tc.x;
                    c[k].y = tc.y;
                }
            } else {
                v[k] = *reinterpret_cast<const double2 *>(vals + (long)i * stride + r);
                c[k] = *reinterpret_cast<const int2 *>(cols + (long)i * stride + r);
                if (i0 + k >= width) c[k].x = c[k].y = -1;
            }
        }
        double xv0[BATCH], xv1[BATCH];
#pragma unroll
        for (int k = 0; k < BATCH; ++k) {
            xv0[k] = c[k].x >= 0 ? x[c[k].x] : 0.0;
            xv1[k] = c[k].y >= 0 ? x[c[k].y] : 0.0;
        }
#pragma unroll
        for (int k = 0; k < BATCH; ++k) {
            if (c[k].x >= 0) {
                const double t = v[k].x * xv0[k];
                acc.x = (MODE == SPMV_RESIDUAL) ? acc.x - t : acc.x + t;
            }
            if (c[k].y >= 0) {
                const double t = v[k].y * xv1[k];
                acc.y = (MODE == SPMV_RESIDUAL) ? acc.y - t : acc.y + t;
            }
        }
    }
    if (hf.chunk_bptr) halo_fused_add<MODE>(hf, chunk, acc.x, acc.y, ys);
    st2(y, rp, acc);
    if (NDOT >= 1) {
        const double2 vw = ld2(w, rp);
        double d = 0.0, d2 = 0.0;
        if (rp.n > 0) {
            d += vw.x * acc.x;
            d2 += acc.x * acc.x;
        }
        if (rp.n > 1) {
            d += vw.y * acc.y;
            d2 += acc.y * acc.y;
        }
        const double s = block_sum(d, slot);
        if (threadIdx.x == 0) dot_partials[chunk] = s;
        if (NDOT >= 2) {
            const double s2 = block_sum(d2, slot);
            if (threadIdx.x == 0) dot2_partials[chunk] = s2;
        }
    }
}

}  // namespace

// ------------------------------------------------------------------------------------------
// launchers
// ------------------------------------------------------------------------------------------
void launch_spmv(hipStream_t st, const DevCsr &A, int mode, const double *x, const double *b,
                 double *y, const SpmvDots &dots, const DevScalars *gate, const HaloFused &hf)
{
    if (A.n_rows == 0) return;
    const int nc = (int)n_chunks(A.n_rows);
    const int xg = A.xcd_group > 0 ? A.xcd_group : XCD_GROUP;
    const bool ordered = A.block_order && !A.codes21;
    const dim3 grid(ordered ? A.n_blocks : xcd_grid(nc, xg)), block(BLOCK);
#define OGL_SPMV_K(MODE, NDOT, STREAM)                                                                              \
    do {                                                                                                            \
        if (A.lds_rounds == 1)                                                                                      \
            hipLaunchKernelGGL((k_spmv_stream<MODE, NDOT, STREAM, 1>), grid, block, 0, st, A.n_rows, nc, A.row_ptrs, \
                               A.cols, A.vals, x, b, y, dots.with, dots.part, dots.part_yy, gate, xg, hf,           \
                               A.block_order);                                                                      \
        else                                                                                                        \
            hipLaunchKernelGGL((k_spmv_stream<MODE, NDOT, STREAM, 2>), grid, block, 0, st, A.n_rows, nc, A.row_ptrs, \
                               A.cols, A.vals, x, b, y, dots.with, dots.part, dots.part_yy, gate, xg, hf,           \
                               A.block_order);                                                                      \
    } while (0)
#define OGL_SPMV21_K(MODE, NDOT, STREAM)                                                                   \
    hipLaunchKernelGGL((k_spmv_stream21<MODE, NDOT, STREAM>), grid, block, 0, st, A.n_rows, nc, A.row_ptrs, \
                       A.chunks21, A.codes21, A.vals, x, b, y, dots.with, dots.part, dots.part_yy, gate, xg, hf, \
                       A.far_idx21, A.far_col21)
#define OGL_SPMV(MODE, NDOT)                 \
    do {                                     \
        if (A.codes21 && A.stream)           \
            OGL_SPMV21_K(MODE, NDOT, true);  \
        else if (A.codes21)                  \
            OGL_SPMV21_K(MODE, NDOT, false); \
        else if (A.stream)                   \
            OGL_SPMV_K(MODE, NDOT, true);    \
        else                                 \
            OGL_SPMV_K(MODE, NDOT, false);   \
    } while (0)
    if (mode == SPMV_RESIDUAL) {
        OGL_SPMV(SPMV_RESIDUAL, 0);
    } else if (dots.part && dots.part_yy) {
        OGL_SPMV(SPMV_PLAIN, 2);
    } else if (dots.part) {
        OGL_SPMV(SPMV_PLAIN, 1);
    } else {
        OGL_SPMV(SPMV_PLAIN, 0);
    }
#undef OGL_SPMV
#undef OGL_SPMV_K
#undef OGL_SPMV21_K
}

void launch_spmv_ell(hipStream_t st, const DevEll &A, int mode, const double *x, const double *b,
                     double *y, const SpmvDots &dots, const DevScalars *gate, const HaloFused &hf)
{
    if (A.n_rows == 0) return;
    const int nc = (int)n_chunks(A.n_rows);
    const dim3 grid(xcd_grid(nc)), block(BLOCK);
#define OGL_ELL_K(MODE, NDOT, STREAM)                                                                   \
    hipLaunchKernelGGL((k_spmv_ell<MODE, NDOT, STREAM>), grid, block, 0, st, A.n_rows, nc, A.width,      \
                       (long)A.stride, A.cols, A.vals, x, b, y, dots.with, dots.part, dots.part_yy, gate, hf)
#define OGL_ELL(MODE, NDOT)               \
    do {                                  \
        if (A.stream)                     \
            OGL_ELL_K(MODE, NDOT, true);  \
        else                              \
            OGL_ELL_K(MODE, NDOT, false); \
    } while (0)
    if (mode == SPMV_RESIDUAL) {
        OGL_ELL(SPMV_RESIDUAL, 0);
    } else if (dots.part && dots.part_yy) {
        OGL_ELL(SPMV_PLAIN, 2);
    } else if (dots.part) {
        OGL_ELL(SPMV_PLAIN, 1);
    } else {
        OGL_ELL(SPMV_PLAIN, 0);
    }
#undef OGL_ELL
#undef OGL_ELL_K
}

void launch_spmv_non_local(hipStream_t st, const DevHalo &H, int mode, const double *recv,
                           double *y, const DevScalars *gate)
{
    if (H.n_boundary_rows == 0) return;
    const dim3 grid(blocks_for(H.n_boundary_rows)), block(BLOCK);
    if (mode == SPMV_RESIDUAL)
        hipLaunchKernelGGL((k_spmv_non_local<SPMV_RESIDUAL>), grid, block, 0, st,
                           H.n_boundary_rows, H.boundary_rows, H.entry_ptrs, H.cols, H.vals, recv,
                           y, gate);
    else
        hipLaunchKernelGGL((k_spmv_non_local<SPMV_PLAIN>), grid, block, 0, st, H.n_boundary_rows,
                           H.boundary_rows, H.entry_ptrs, H.cols, H.vals, recv, y, gate);
}

void launch_pack(hipStream_t st, const DevHalo &H, const double *x, double *send,
                 const DevScalars *gate)
{
    if (H.n_send == 0) return;
    hipLaunchKernelGGL(k_pack, dim3(blocks_for(H.n_send)), dim3(BLOCK), 0, st, H.n_send,
                       H.send_idxs, x, send, gate);
}

}  // namespace ogl
