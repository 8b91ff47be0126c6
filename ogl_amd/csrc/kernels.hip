// kernels.hip -- hand-written gfx950 (CDNA4) kernels of the Krylov inner loop.
//
// Everything here is HBM-bandwidth bound (SpMV arithmetic intensity 0.134 flop/B), so there is
// no MFMA: the design rules are coalesced 16-byte-per-lane streams, LDS staging of the SpMV
// products, 64-lane shuffle reductions and an XCD-aware block -> row-chunk map.
//
// Geometry (common.hpp): one 256-thread workgroup (4 wavefronts of 64) owns one chunk of
// CHUNK_ROWS = 512 consecutive rows; thread t owns rows chunk*512 + 2t, +1.
//
// Reductions are deterministic (no atomics): a chunk partial is
//     thread sums (rows in order)  ->  64-lane xor tree (32,16,8,4,2,1)  ->  wave0+wave1+wave2+wave3
// and the finaliser (one 1024-thread workgroup) sums the partials: thread t takes partials
// t, t+1024, ... in order, xor tree per wave, then the 16 wave sums left to right.
// oracle/ogl_oracle.c mirrors this tree in its BLOCKED mode so tests can compare bit for bit.
//
// Compiled with -ffp-contract=off: every product and every sum rounds once, like the reference
// executor of Ginkgo on a baseline x86-64 build (no FMA contraction).
#include "kernels.hpp"

namespace ogl {

namespace {

constexpr int N_WAVES = BLOCK / WAVE;

// ------------------------------------------------------------------------------------------
// reduction tree
// ------------------------------------------------------------------------------------------
__device__ __forceinline__ double wave_sum(double v)
{
#pragma unroll
    for (int off = WAVE / 2; off >= 1; off >>= 1) v += __shfl_xor(v, off, WAVE);
    return v;
}

// Every thread returns the block total.  `slot` = N_WAVES doubles of LDS.
__device__ __forceinline__ double block_sum(double v, double *slot)
{
    v = wave_sum(v);
    const int lane = threadIdx.x & (WAVE - 1), wave = threadIdx.x / WAVE;
    if (lane == 0) slot[wave] = v;
    __syncthreads();
    double s = slot[0];
#pragma unroll
    for (int w = 1; w < N_WAVES; ++w) s += slot[w];
    __syncthreads();
    return s;
}

// Two block totals in one pass (same tree for each, one barrier pair instead of two).  `slot` = 2 N_WAVES doubles.
__device__ __forceinline__ void block_sum2(double &a, double &b, double *slot)
{
    a = wave_sum(a);
    b = wave_sum(b);
    const int lane = threadIdx.x & (WAVE - 1), wave = threadIdx.x / WAVE;
    if (lane == 0) {
        slot[wave] = a;
        slot[N_WAVES + wave] = b;
    }
    __syncthreads();
    double sa = slot[0], sb = slot[N_WAVES];
#pragma unroll
    for (int w = 1; w < N_WAVES; ++w) {
        sa += slot[w];
        sb += slot[N_WAVES + w];
    }
    __syncthreads();
    a = sa;
    b = sb;
}

// Finaliser tree (one workgroup of FIN_BLOCK = 1024 threads = 16 wavefronts): thread t adds
// partials t, t+1024, ... in that order, then the 64-lane xor tree, then the 16 wave sums left to
// right.  The loads of a batch are issued together (they are independent) and only the adds stay
// ordered, so a 10M-row vector (19,683 partials) costs about one memory latency.
constexpr int FIN_BLOCK = 1024;
constexpr int FIN_WAVES = FIN_BLOCK / WAVE;
constexpr int FIN_BATCH = 8;  // (20 = one batch for 10M rows measured no faster)

__device__ __forceinline__ double fin_block_sum(double v, double *slot)
{
    v = wave_sum(v);
    const int lane = threadIdx.x & (WAVE - 1), wave = threadIdx.x / WAVE;
    if (lane == 0) slot[wave] = v;
    __syncthreads();
    double s = slot[0];
#pragma unroll
    for (int w = 1; w < FIN_WAVES; ++w) s += slot[w];
    __syncthreads();
    return s;
}

// Reduces one or two partial arrays at once (loads of both in flight together).
template <int K>
__device__ __forceinline__ void reduce_partials(const double *const (&part)[2], int m, double *slot,
                                                double (&out)[2])
{
    double s[2] = {0.0, 0.0};
    for (int i0 = threadIdx.x; i0 < m; i0 += FIN_BLOCK * FIN_BATCH) {
        double v[2][FIN_BATCH];
#pragma unroll
        for (int k = 0; k < FIN_BATCH; ++k) {
            const int i = i0 + k * FIN_BLOCK;
#pragma unroll
            for (int a = 0; a < K; ++a) v[a][k] = i < m ? part[a][i] : 0.0;
        }
#pragma unroll
        for (int k = 0; k < FIN_BATCH; ++k)
            if (i0 + k * FIN_BLOCK < m) {
#pragma unroll
                for (int a = 0; a < K; ++a) s[a] += v[a][k];
            }
    }
    out[0] = fin_block_sum(s[0], slot);
    out[1] = K > 1 ? fin_block_sum(s[1], slot) : 0.0;
}

// Non-local part of a chunk's rows inside the local SpMV kernel (HaloFused, kernels.hpp).  acc0 / acc1: the
// accumulators of this thread's two rows after the local entries; ys: CHUNK_ROWS doubles of LDS that the
// kernel does not need any more.  Workgroup-uniform: chunks without boundary rows return at once.
// TURN (the merged step_1x + SpMV kernel of a multi-rank GKOCG turn): what the neighbours have put is z, not p -- the
// new p of a halo column is recomputed here from it and the OLD p of that column, which this rank keeps
// (p_new = z + tmp p_old: the expression, scalars and operands of the owner's own update, hence its bits), and left
// behind in the other of two halo-p buffers for the next turn.
// wait accounting (DevScalars::halo_wait_ticks): the waiting lanes leave their wait in *longest (LDS, zeroed before),
// one thread adds the workgroup's figure to the solve's counters afterwards
__device__ __forceinline__ void note_wait(unsigned *longest, long long t0)
{
    atomicMax(longest, (unsigned)min((long long)0xffffffffll, wall_clock64() - t0));
}
__device__ __forceinline__ void add_halo_wait(DevScalars *s, unsigned longest)
{
    atomicAdd(&s->halo_wait_ticks, (unsigned long long)longest);
    atomicAdd(&s->halo_waits, 1u);
}

template <int MODE, bool TURN = false>
__device__ __forceinline__ void halo_fused_add(const HaloFused &H, int chunk, double &acc0, double &acc1, double *ys,
                                               double tmp = 0.0, const double *__restrict__ ph_in = nullptr,
                                               double *__restrict__ ph_out = nullptr)
{
    const int b0 = H.chunk_bptr[chunk], b1 = H.chunk_bptr[chunk + 1];
    if (b0 == b1) return;
    __shared__ int halo_timed_out;
    __shared__ unsigned halo_waited;
    if (threadIdx.x == 0) {
        halo_timed_out = 0;
        halo_waited = 0;
    }
    __syncthreads();
    if ((int)threadIdx.x < H.n_neigh) {
        const long long t0 = wall_clock64();
        for (;;) {
            const unsigned long long f =
                __hip_atomic_load(H.local_flag + threadIdx.x, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_SYSTEM);
            if ((uint32_t)f == H.seq) break;
            if (wall_clock64() - t0 > H.timeout_ticks) {
                halo_timed_out = 1;
                break;
            }
            __builtin_amdgcn_s_sleep(2);
        }
        note_wait(&halo_waited, t0);
    }
    ys[ROWS_PER_THREAD * threadIdx.x] = acc0;
    ys[ROWS_PER_THREAD * threadIdx.x + 1] = acc1;
    __syncthreads();
    if (threadIdx.x == 0) add_halo_wait(H.s, halo_waited);
    if (halo_timed_out) {  // a neighbour is gone: end the solve (y stays the local product)
        if (threadIdx.x == 0) {
            H.s->comm_error = 1;
            H.s->stop = 1;
        }
        return;
    }
    for (int i = b0 + threadIdx.x; i < b1; i += BLOCK) {
        const int li = H.boundary_rows[i] - chunk * CHUNK_ROWS;
        double a = ys[li];
        for (int k = H.entry_ptrs[i]; k < H.entry_ptrs[i + 1]; ++k) {
            const int c = H.cols[k];
            double v = H.recv[c];
            if (TURN) {
                v = v + tmp * ph_in[c];
                ph_out[c] = v;
            }
            const double t = H.vals[k] * v;
            a = (MODE == SPMV_RESIDUAL) ? a - t : a + t;
        }
        ys[li] = a;
    }
    __syncthreads();
    acc0 = ys[ROWS_PER_THREAD * threadIdx.x];
    acc1 = ys[ROWS_PER_THREAD * threadIdx.x + 1];
}

// XCD-aware chunk map.  The dispatcher places block b on XCD b % 8 (MI355X_MICROARCH.md,
// "Workgroup dispatch"); each XCD has a private 4 MiB L2.  Measured on the 216^3 case
// (tools/spmv_tune.hip, profiles/spmv_tune_r01.txt):
//   * one contiguous eighth of the rows per XCD  -> 208 us  (8 separate DRAM fronts)
//   * plain chunk = block                        -> 193 us
//   * groups of 4 consecutive chunks per XCD, all XCDs advancing on ONE front -> 190 us
// so neighbouring rows (the +-1 / +-nx stencil legs) share an L2 while HBM still sees a single
// streaming front.  Purely a speed choice: results do not depend on placement.
// On an irregular pattern (unstructured mesh in RCM order) the group is a launch parameter: with slabs of
// tens of thousands of rows per XCD the window of x a slab gathers from is fetched into ONE L2 instead of
// all eight (DevCsr::xcd_group, profiles/r03_xcd_group.txt).
constexpr int XCD_GROUP = 4;
__device__ __forceinline__ int xcd_chunk(int block, int group = XCD_GROUP)
{
    const int slot = block / N_XCD, xcd = block % N_XCD;
    return (slot / group) * (N_XCD * group) + xcd * group + slot % group;
}
inline int xcd_grid(int n_chunks, int group = XCD_GROUP)
{
    const int q = N_XCD * group;
    return ((n_chunks + q - 1) / q) * q;
}

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
template <int MODE, int NDOT, bool STREAM>
__global__ __launch_bounds__(BLOCK) void k_spmv_stream(
    int n_rows, int n_chunks, const int *__restrict__ row_ptrs, const int *__restrict__ cols,
    const double *__restrict__ vals, const double *__restrict__ x, const double *__restrict__ b,
    double *__restrict__ y, const double *__restrict__ w, double *__restrict__ dot_partials,
    double *__restrict__ dot2_partials, const DevScalars *gate, int xgroup, HaloFused hf,
    const int *__restrict__ block_order)
{
    __shared__ __attribute__((aligned(16))) double prod[SPMV_TILE];
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
#pragma unroll
        for (int g = 0; g < GROUPS; ++g) {
            const double x0 = x[cc[g].x], x1 = x[cc[g].y];
            double2 p0;
            p0.x = va[g].x * x0;
            p0.y = va[g].y * x1;
            *reinterpret_cast<double2 *>(prod + (g * BLOCK + tid) * 2) = p0;
        }
        __syncthreads();
        const int t1 = t0 + SPMV_TILE;
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

__global__ __launch_bounds__(BLOCK) void k_scale(int n, double *__restrict__ v, double f)
{
    const int i = blockIdx.x * BLOCK + threadIdx.x;
    if (i < n) v[i] = v[i] * f;
}

__global__ __launch_bounds__(BLOCK) void k_fill_xbar(int n, double *__restrict__ v,
                                                     const DevScalars *s)
{
    const int i = blockIdx.x * BLOCK + threadIdx.x;
    if (i < n) v[i] = s->xbar;
}

// ------------------------------------------------------------------------------------------
// chunk-shaped vector kernels: thread t of block `chunk` owns rows chunk*512 + 2t, +1
// ------------------------------------------------------------------------------------------
struct RowPair {
    int row;  // first row
    int n;    // valid rows (0, 1 or 2)
};
__device__ __forceinline__ RowPair my_rows(int chunk, int n_rows)
{
    RowPair r;
    r.row = chunk * CHUNK_ROWS + threadIdx.x * ROWS_PER_THREAD;
    const int left = n_rows - r.row;
    r.n = left >= ROWS_PER_THREAD ? ROWS_PER_THREAD : (left > 0 ? left : 0);
    return r;
}
__device__ __forceinline__ double2 ld2(const double *__restrict__ p, const RowPair &r)
{
    double2 v;
    if (r.n == 2) {
        v = *reinterpret_cast<const double2 *>(p + r.row);
    } else {
        v.x = r.n == 1 ? p[r.row] : 0.0;
        v.y = 0.0;
    }
    return v;
}
__device__ __forceinline__ void st2(double *__restrict__ p, const RowPair &r, double2 v)
{
    if (r.n == 2)
        *reinterpret_cast<double2 *>(p + r.row) = v;
    else if (r.n == 1)
        p[r.row] = v.x;
}
// the same for data that is not touched again this turn (streamed past the caches, so that the vectors the
// next kernel needs stay in the Infinity Cache; +2.5 % turn rate at 216^3 for the x update alone)
__device__ __forceinline__ double2 ld2_stream(const double *__restrict__ p, const RowPair &r)
{
    typedef double d2v __attribute__((ext_vector_type(2)));
    double2 v;
    if (r.n == 2) {
        const d2v t = __builtin_nontemporal_load(reinterpret_cast<const d2v *>(p + r.row));
        v.x = t.x;
        v.y = t.y;
    } else {
        v.x = r.n == 1 ? p[r.row] : 0.0;
        v.y = 0.0;
    }
    return v;
}
// a pair of values of a matrix plane that is read exactly once per launch
__device__ __forceinline__ double2 ld_pair_stream(const double *__restrict__ p)
{
    typedef double d2v __attribute__((ext_vector_type(2)));
    const d2v t = __builtin_nontemporal_load(reinterpret_cast<const d2v *>(p));
    double2 v;
    v.x = t.x;
    v.y = t.y;
    return v;
}
__device__ __forceinline__ void st2_stream(double *__restrict__ p, const RowPair &r, double2 v)
{
    typedef double d2v __attribute__((ext_vector_type(2)));
    if (r.n == 2) {
        d2v t;
        t.x = v.x;
        t.y = v.y;
        __builtin_nontemporal_store(t, reinterpret_cast<d2v *>(p + r.row));
    } else if (r.n == 1) {
        p[r.row] = v.x;
    }
}
static_assert(ROWS_PER_THREAD == 2, "vector kernels are written for two rows per thread");

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
                                                              const DevScalars *gate, const int *__restrict__ rows)
{
    constexpr int HALO = MAX_JACOBI_BLOCK - 1;
    __shared__ double v[CHUNK_ROWS + 2 * HALO];
    if (gate && gate->stop) return;
    const int c0 = blockIdx.x * CHUNK_ROWS, at = c0 + (int)threadIdx.x;
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

enum PartialOp { P_SUM = 0, P_DOT = 1, P_NORM1 = 2 };
template <int OP>
__global__ __launch_bounds__(BLOCK) void k_partials(int n, int n_chunks,
                                                    const double *__restrict__ a,
                                                    const double *__restrict__ b,
                                                    double *__restrict__ part,
                                                    const DevScalars *gate,
                                                    const int *__restrict__ chunk_list)
{
    __shared__ double slot[N_WAVES];
    if (gate && gate->stop) return;
    // chunk_list: only these chunks (the ones holding boundary rows, whose fused partials of the
    // local SpMV are stale once the non-local part has been added)
    const int chunk = chunk_list ? chunk_list[blockIdx.x] : (int)blockIdx.x;
    const RowPair r = my_rows(chunk, n);
    const double2 va = ld2(a, r);
    double d = 0.0;
    if (OP == P_SUM) {
        if (r.n > 0) d += va.x;
        if (r.n > 1) d += va.y;
    } else if (OP == P_NORM1) {
        if (r.n > 0) d += fabs(va.x);
        if (r.n > 1) d += fabs(va.y);
    } else {
        const double2 vb = ld2(b, r);
        if (r.n > 0) d += va.x * vb.x;
        if (r.n > 1) d += va.y * vb.y;
    }
    const double s = block_sum(d, slot);
    if (threadIdx.x == 0) part[chunk] = s;
    (void)n_chunks;
}

// StoppingCriterion.C:53-61: t = b - Axref ; e = |t - r| + |t|
__global__ __launch_bounds__(BLOCK) void k_partials_normfactor(int n, const double *__restrict__ b,
                                                               const double *__restrict__ w,
                                                               const double *__restrict__ r,
                                                               double *__restrict__ part)
{
    __shared__ double slot[N_WAVES];
    const int chunk = blockIdx.x;
    const RowPair rp = my_rows(chunk, n);
    const double2 vb = ld2(b, rp), vw = ld2(w, rp), vr = ld2(r, rp);
    double d = 0.0;
    if (rp.n > 0) {
        const double t = vb.x - 1.0 * vw.x;
        const double p2 = fabs(t);
        d += fabs(fabs(t - 1.0 * vr.x) + 1.0 * p2);
    }
    if (rp.n > 1) {
        const double t = vb.y - 1.0 * vw.y;
        const double p2 = fabs(t);
        d += fabs(fabs(t - 1.0 * vr.y) + 1.0 * p2);
    }
    const double s = block_sum(d, slot);
    if (threadIdx.x == 0) part[chunk] = s;
}

// z = M^-1 r (scalar Jacobi: r * inv_diag; identity: r), rho partial = sum r z, norm partial = sum |r|
__global__ __launch_bounds__(BLOCK) void k_cg_rho_norm(int n, const double *__restrict__ r,
                                                       const double *__restrict__ inv_diag,
                                                       double *__restrict__ part_rho,
                                                       double *__restrict__ part_norm,
                                                       const DevScalars *gate)
{
    __shared__ double slot[N_WAVES];
    if (gate && gate->stop) return;
    const int chunk = blockIdx.x;
    const RowPair rp = my_rows(chunk, n);
    const double2 vr = ld2(r, rp);
    double2 vz = vr;
    if (inv_diag) {
        const double2 vi = ld2(inv_diag, rp);
        vz.x = vr.x * vi.x;
        vz.y = vr.y * vi.y;
    }
    double d = 0.0, a = 0.0;
    if (rp.n > 0) {
        d += vr.x * vz.x;
        a += fabs(vr.x);
    }
    if (rp.n > 1) {
        d += vr.y * vz.y;
        a += fabs(vr.y);
    }
    const double s0 = block_sum(d, slot);
    const double s1 = block_sum(a, slot);
    if (threadIdx.x == 0) {
        part_rho[chunk] = s0;
        part_norm[chunk] = s1;
    }
}

// step_1: p = z + (rho / prev_rho) * p   (tmp = 0 when prev_rho == 0)
__global__ __launch_bounds__(BLOCK) void k_cg_step1(int n, double *__restrict__ p,
                                                    const double *__restrict__ r,
                                                    const double *__restrict__ inv_diag,
                                                    const DevScalars *s)
{
    if (s->stop) return;
    const double rho = s->rho, prev = s->prev_rho;
    const double tmp = (prev == 0.0) ? 0.0 : rho / prev;
    const RowPair rp = my_rows(blockIdx.x, n);
    double2 vz = ld2(r, rp);
    if (inv_diag) {
        const double2 vi = ld2(inv_diag, rp);
        vz.x = vz.x * vi.x;
        vz.y = vz.y * vi.y;
    }
    double2 vp = ld2(p, rp);
    vp.x = vz.x + tmp * vp.x;
    vp.y = vz.y + tmp * vp.y;
    st2(p, rp, vp);
}

// step_2: if (beta != 0) { t = rho / beta ; x += t p ; r -= t q } -- then the next turn's
// z = M^-1 r, rho = r.z and sum|r| partials, fused so r is not re-read (K5+K6+K8).
__global__ __launch_bounds__(BLOCK) void k_cg_step2(int n, double *__restrict__ x,
                                                    double *__restrict__ r,
                                                    const double *__restrict__ p,
                                                    const double *__restrict__ q,
                                                    const double *__restrict__ inv_diag,
                                                    double *__restrict__ part_rho,
                                                    double *__restrict__ part_norm,
                                                    const DevScalars *s)
{
    __shared__ double slot[N_WAVES];
    if (s->stop) return;
    const double rho = s->rho, beta = s->beta;
    const int chunk = blockIdx.x;
    const RowPair rp = my_rows(chunk, n);
    double2 vr = ld2(r, rp);
    if (beta != 0.0) {
        const double t = rho / beta;
        double2 vx = ld2_stream(x, rp);  // x: once per turn; q: last use of the turn
        const double2 vp = ld2(p, rp), vq = ld2_stream(q, rp);
        vx.x += t * vp.x;
        vx.y += t * vp.y;
        vr.x -= t * vq.x;
        vr.y -= t * vq.y;
        st2_stream(x, rp, vx);
        st2(r, rp, vr);
    }
    double2 vz = vr;
    if (inv_diag) {
        const double2 vi = ld2(inv_diag, rp);
        vz.x = vr.x * vi.x;
        vz.y = vr.y * vi.y;
    }
    double d = 0.0, a = 0.0;
    if (rp.n > 0) {
        d += vr.x * vz.x;
        a += fabs(vr.x);
    }
    if (rp.n > 1) {
        d += vr.y * vz.y;
        a += fabs(vr.y);
    }
    const double s0 = block_sum(d, slot);
    const double s1 = block_sum(a, slot);
    if (threadIdx.x == 0) {
        part_rho[chunk] = s0;
        part_norm[chunk] = s1;
    }
}

// The same two steps with the x update DEFERRED by one turn, so that p is read once per turn instead
// of twice (80 N instead of 88 N bytes per turn with scalar Jacobi):
//   step_2r (turn j)  : r -= t_j q ; partials of the next rho and sum|r|        (x, p untouched)
//   step_1x (turn j+1): x += t_j p  with the OLD p, then p = z + (rho/prev_rho) p
// t_j = rho_j / beta_j is formed from the same two scalars as in step_2 (after the check that closed
// turn j they sit in prev_rho and beta), so x receives the same bits, one kernel later.  When that
// check stops the solve, the step_1x of turn j+1 still applies the pending update (it recognises
// its turn by iter == turn + 1: no check runs after the stop) and leaves p alone; the host flushes
// with an extra step_1x when the stop came after the last enqueued turn.
// Halo values of the SpMV that follows, put by the kernel that produces them: this chunk's send rows (v0, v1 = the
// two rows of the thread) go straight into the neighbours' receive blocks -- through LDS, the row's owner holds it in
// registers -- and the last such workgroup of the launch raises the flags.  Workgroup-uniform.
__device__ __forceinline__ void halo_put_chunk(const HaloPutFused &put, int chunk, double v0, double v1, double *ps,
                                               int *last)
{
    const int s0 = put.chunk_sptr[chunk], s1 = put.chunk_sptr[chunk + 1];
    if (s0 == s1) return;
    ps[ROWS_PER_THREAD * threadIdx.x] = v0;
    ps[ROWS_PER_THREAD * threadIdx.x + 1] = v1;
    __syncthreads();
    for (int k = s0 + threadIdx.x; k < s1; k += BLOCK) {
        const int j = put.send_pos[k];
        int i = 0;
        while (i + 1 < put.P.n_neigh && j >= put.P.send_off[i + 1]) ++i;
        put.P.remote_recv[i][j - put.P.send_off[i]] = ps[put.send_idxs[j] - chunk * CHUNK_ROWS];
    }
    __threadfence_system();
    __syncthreads();
    if (threadIdx.x == 0) {
        *last = atomicAdd(put.ticket, 1u) == (unsigned)put.n_put_chunks - 1;
        if (*last) *put.ticket = 0;
    }
    __syncthreads();
    if (*last && (int)threadIdx.x < put.P.n_neigh) {
        __threadfence_system();
        __hip_atomic_store(put.P.remote_flag[threadIdx.x], (unsigned long long)put.P.seq, __ATOMIC_RELEASE,
                           __HIP_MEMORY_SCOPE_SYSTEM);
    }
}

template <bool PUT>
__global__ __launch_bounds__(BLOCK) void k_cg_step1x(int n, double *__restrict__ p,
                                                     double *__restrict__ x,
                                                     const double *__restrict__ r,
                                                     const double *__restrict__ inv_diag,
                                                     const DevScalars *s, HaloPutFused put)
{
    const int stop = s->stop;
    const bool pending = s->x_pending != 0;
    if (stop && !pending) return;
    const RowPair rp = my_rows(blockIdx.x, n);
    double2 vp = ld2(p, rp);
    if (pending) {
        const double beta = s->beta;
        if (beta != 0.0) {
            const double t = s->prev_rho / beta;
            double2 vx = ld2_stream(x, rp);  // x is touched once per turn
            vx.x += t * vp.x;
            vx.y += t * vp.y;
            st2_stream(x, rp, vx);
        }
    }
    if (stop) return;
    const double rho = s->rho, prev = s->prev_rho;
    const double tmp = (prev == 0.0) ? 0.0 : rho / prev;
    double2 vz = ld2_stream(r, rp);  // r and inv_diag: last use of this turn
    if (inv_diag) {
        const double2 vi = ld2_stream(inv_diag, rp);
        vz.x = vz.x * vi.x;
        vz.y = vz.y * vi.y;
    }
    vp.x = vz.x + tmp * vp.x;
    vp.y = vz.y + tmp * vp.y;
    st2(p, rp, vp);
    if (PUT) {  // the halo values of the SpMV that follows
        __shared__ double ps[CHUNK_ROWS];
        __shared__ int last;
        halo_put_chunk(put, blockIdx.x, vp.x, vp.y, ps, &last);
    }
}

// PUT (multi-rank merged turn): the z of this chunk's send rows goes to the neighbours, whose next merged kernel
// forms p_new = z + tmp p_old at its halo columns itself (halo_fused_add<.., TURN>)
template <bool PUT>
__global__ __launch_bounds__(BLOCK) void k_cg_step2r(int n, double *__restrict__ r,
                                                     const double *__restrict__ q,
                                                     const double *__restrict__ inv_diag,
                                                     double *__restrict__ part_rho,
                                                     double *__restrict__ part_norm,
                                                     const DevScalars *s, double *__restrict__ z_out,
                                                     HaloPutFused put)
{
    __shared__ double slot[N_WAVES];
    if (s->stop) return;
    const double rho = s->rho, beta = s->beta;
    const int chunk = blockIdx.x;
    const RowPair rp = my_rows(chunk, n);
    double2 vr = ld2(r, rp);
    if (beta != 0.0) {
        const double t = rho / beta;
        const double2 vq = ld2_stream(q, rp);  // q: last use of this turn
        vr.x -= t * vq.x;
        vr.y -= t * vq.y;
        st2(r, rp, vr);
    }
    double2 vz = vr;
    if (inv_diag) {
        const double2 vi = ld2(inv_diag, rp);
        vz.x = vr.x * vi.x;
        vz.y = vr.y * vi.y;
    }
    if (z_out) st2(z_out, rp, vz);  // (kept for the gathers of k_cg_turn_sym_big)
    double d = 0.0, a = 0.0;
    if (rp.n > 0) {
        d += vr.x * vz.x;
        a += fabs(vr.x);
    }
    if (rp.n > 1) {
        d += vr.y * vz.y;
        a += fabs(vr.y);
    }
    const double s0 = block_sum(d, slot);
    const double s1 = block_sum(a, slot);
    if (threadIdx.x == 0) {
        part_rho[chunk] = s0;
        part_norm[chunk] = s1;
    }
    if (PUT) {
        __shared__ double ps[CHUNK_ROWS];
        __shared__ int last;
        halo_put_chunk(put, chunk, vz.x, vz.y, ps, &last);
    }
}

__device__ void criterion_check(DevScalars *s, const DevCriterion &c, double norm, double *history);

// ------------------------------------------------------------------------------------------
// Small systems (a turn of a 64^3 case is 5 dependent launches of ~4.5 us for ~6 us of memory time): the two
// single-workgroup finalisers of a GKOCG turn are folded into the kernels that consume their results.  Every
// workgroup of step_1x / step_2r first reduces the (few hundred) per-chunk partials ITSELF -- with 256 threads
// walking the 1024-thread tree of k_finalize, so the sums have the same bits -- and runs the scalar logic on its
// own copy of the solver scalars; workgroup 0 stores the new scalars.  The scalars ping-pong between two slots
// (a kernel reads `sin`, writes `sout`), so that no workgroup can see them half-way.  Turn = 3 launches:
//   [check of the previous turn + pending x update + step_1]  ->  SpMV  ->  [beta + step_2r]
// ------------------------------------------------------------------------------------------
constexpr int FIN_VT = FIN_BLOCK / BLOCK;  // virtual finaliser threads per thread
// the partials this thread's virtual threads own (at most one each: n_part <= FIN_BLOCK), asked for early
template <int K>
__device__ __forceinline__ void load_partials_as_finaliser(const double *__restrict__ p0,
                                                           const double *__restrict__ p1, int m,
                                                           double (&pv)[2][FIN_VT])
{
#pragma unroll
    for (int j = 0; j < FIN_VT; ++j) {
        const int i = threadIdx.x + BLOCK * j;
        pv[0][j] = i < m ? p0[i] : 0.0;
        pv[1][j] = (K > 1 && i < m) ? p1[i] : 0.0;
    }
}
template <int K>
__device__ __forceinline__ void reduce_partials_as_finaliser(const double (&pv)[2][FIN_VT], int m, double *lds,
                                                             double (&out)[2])
{
    // virtual thread v = t + 256 j of the finaliser's 1024 sums partials v, v + 1024, ... (here: just v) from 0.0;
    // its wavefront (v / 64 = t / 64 + 4 j) is summed by the xor tree; then the 16 wavefront sums left to right
    const int t = threadIdx.x;
    double s[2][FIN_VT];
#pragma unroll
    for (int j = 0; j < FIN_VT; ++j) {
        const bool has = t + BLOCK * j < m;
        s[0][j] = has ? 0.0 + pv[0][j] : 0.0;
        s[1][j] = (K > 1 && has) ? 0.0 + pv[1][j] : 0.0;
    }
#pragma unroll
    for (int j = 0; j < FIN_VT; ++j) {
        if (BLOCK * j >= m) continue;  // (nothing but 0.0 in these virtual wavefronts: their tree gives 0.0)
        s[0][j] = wave_sum(s[0][j]);
        if (K > 1) s[1][j] = wave_sum(s[1][j]);
    }
    if ((t & (WAVE - 1)) == 0) {
#pragma unroll
        for (int j = 0; j < FIN_VT; ++j) {
            lds[t / WAVE + N_WAVES * j] = s[0][j];
            if (K > 1) lds[FIN_WAVES + t / WAVE + N_WAVES * j] = s[1][j];
        }
    }
    __syncthreads();
    double a = lds[0], b = K > 1 ? lds[FIN_WAVES] : 0.0;
#pragma unroll
    for (int w = 1; w < FIN_WAVES; ++w) {
        a += lds[w];
        if (K > 1) b += lds[FIN_WAVES + w];
    }
    out[0] = a;
    out[1] = b;
    __syncthreads();
}
static_assert(FIN_BLOCK % BLOCK == 0 && FIN_WAVES == N_WAVES * (FIN_BLOCK / BLOCK), "virtual finaliser threads");

__global__ __launch_bounds__(BLOCK) void k_cg_step1x_fin(int n, double *__restrict__ p, double *__restrict__ x,
                                                         const double *__restrict__ r,
                                                         const double *__restrict__ inv_diag,
                                                         const DevScalars *sin, DevScalars *sout,
                                                         const double *__restrict__ part_rho,
                                                         const double *__restrict__ part_norm, int n_part,
                                                         double *history, int first)
{
    __shared__ double red[2 * FIN_WAVES];
    __shared__ double sh[4];
    __shared__ int sh_stop;
    // everything this workgroup will need is asked for at once -- the scalars, the partials and its own rows of
    // p, x, r, 1/d: one memory round trip instead of three in a row (scalars -> partials -> vectors).  The
    // scalars are read field by field into registers (a private copy of the struct would live in scratch memory,
    // and a kernel with scratch costs more to dispatch than the finaliser launch this is meant to save).
    const int stopped = sin->stop;
    const double s_rho = sin->rho, s_beta = sin->beta, s_nf = sin->norm_factor, s_init = sin->init_res;
    const int s_iter = sin->iter, s_evals = sin->n_evals;
    const double c_tol = sin->crit.tolerance, c_rel = sin->crit.rel_tol;
    const int c_min = sin->crit.min_iter, c_max = sin->crit.max_iter, c_freq = sin->crit.frequency,
              c_exp = sin->crit.export_res;
    static_assert(sizeof(DevScalars) % 8 == 0, "copied as 8-byte words");
    if (blockIdx.x == 0 && threadIdx.x < sizeof(DevScalars) / 8)  // fields this kernel leaves alone
        reinterpret_cast<unsigned long long *>(sout)[threadIdx.x] =
            reinterpret_cast<const unsigned long long *>(sin)[threadIdx.x];
    const RowPair rp = my_rows(blockIdx.x, n);
    double2 vp = ld2(p, rp);
    double2 vx = ld2_stream(x, rp);
    double2 vz = ld2_stream(r, rp);
    double2 vi;
    vi.x = vi.y = 1.0;
    if (inv_diag) vi = ld2_stream(inv_diag, rp);
    double pv[2][FIN_VT];
    load_partials_as_finaliser<2>(part_rho, part_norm, n_part, pv);
    if (stopped) return;  // (the solve has ended: workgroup 0 has handed the scalars on, nothing else to do)
    double v[2];
    reduce_partials_as_finaliser<2>(pv, n_part, red, v);  // (its barriers order the copy above before the stores below)
    if (threadIdx.x == 0) {
        // FIN_CG_CHECK: swap(prev_rho, rho) of the previous turn, then criterion_check (StoppingCriterion.C:71-151)
        const double prev_rho = s_rho, rho = v[0];
        int iter = s_iter, n_evals = s_evals, stop = 0;
        double init_res = s_init, res = 0.0;
        bool evaluated = false;
        if (iter > 0 && iter < c_min) {           // :77-81
            iter += 1;
        } else if (iter % c_freq != 0) {          // :84-87
            iter += 1;
        } else {
            evaluated = true;
            n_evals += 1;
            res = v[1];
            if (iter == 0) init_res = res / s_nf;  // :102-111
            res /= s_nf;                           // :113
            if (c_exp && history && blockIdx.x == 0) history[iter] = res;  // :115-117
            if (iter >= c_max) stop = 1;                                   // :124
            if (res < c_tol) stop = 1;                                     // :128
            if (c_rel > 0 && res < c_rel * init_res) stop = 1;             // :132-136
            iter += 1;                                                     // :143
        }
        sh[0] = s_beta;
        sh[1] = prev_rho;
        sh[2] = rho;
        sh_stop = stop;
        if (blockIdx.x == 0) {
            sout->prev_rho = prev_rho;
            sout->rho = rho;
            sout->iter = iter;
            sout->x_pending = 0;
            if (evaluated) {
                sout->n_evals = n_evals;
                sout->init_res = init_res;
                sout->res = res;
            }
            if (stop) sout->stop = 1;
        }
    }
    __syncthreads();
    const double beta = sh[0], prev = sh[1], rho = sh[2];
    const int stop = sh_stop;
    if (!first && beta != 0.0) {  // x += t_j p of the turn this check closed (same scalars, same bits as step_2)
        const double t = prev / beta;
        vx.x += t * vp.x;
        vx.y += t * vp.y;
        st2_stream(x, rp, vx);
    }
    if (stop) return;
    const double tmp = (prev == 0.0) ? 0.0 : rho / prev;
    if (inv_diag) {
        vz.x = vz.x * vi.x;
        vz.y = vz.y * vi.y;
    }
    vp.x = vz.x + tmp * vp.x;
    vp.y = vz.y + tmp * vp.y;
    st2(p, rp, vp);
}

__global__ __launch_bounds__(BLOCK) void k_cg_step2r_fin(int n, double *__restrict__ r,
                                                         const double *__restrict__ q,
                                                         const double *__restrict__ inv_diag,
                                                         double *__restrict__ part_rho,
                                                         double *__restrict__ part_norm, const DevScalars *sin,
                                                         DevScalars *sout, const double *__restrict__ part_beta,
                                                         int n_part, double *__restrict__ z_out)
{
    __shared__ double red[2 * FIN_WAVES];
    __shared__ double sh[2];
    __shared__ double slot[2 * N_WAVES];
    // (all loads up front and the scalars field by field, as in step_1x_fin)
    const int stopped = sin->stop;
    const double s_rho = sin->rho;
    if (blockIdx.x == 0 && threadIdx.x < sizeof(DevScalars) / 8)
        reinterpret_cast<unsigned long long *>(sout)[threadIdx.x] =
            reinterpret_cast<const unsigned long long *>(sin)[threadIdx.x];
    const int chunk = blockIdx.x;
    const RowPair rp = my_rows(chunk, n);
    double2 vr = ld2(r, rp);
    const double2 vq = ld2_stream(q, rp);  // q: last use of this turn
    double2 vi;
    vi.x = vi.y = 1.0;
    if (inv_diag) vi = ld2(inv_diag, rp);
    double pv[2][FIN_VT];
    load_partials_as_finaliser<1>(part_beta, nullptr, n_part, pv);
    if (stopped) return;
    double v[2];
    reduce_partials_as_finaliser<1>(pv, n_part, red, v);
    if (threadIdx.x == 0) {
        sh[0] = s_rho;
        sh[1] = v[0];
        if (blockIdx.x == 0) sout->beta = v[0];  // FIN_BETA
    }
    __syncthreads();
    const double rho = sh[0], beta = sh[1];
    if (beta != 0.0) {
        const double t = rho / beta;
        vr.x -= t * vq.x;
        vr.y -= t * vq.y;
        st2(r, rp, vr);
    }
    double2 vz = vr;
    if (inv_diag) {
        vz.x = vr.x * vi.x;
        vz.y = vr.y * vi.y;
    }
    if (z_out) st2(z_out, rp, vz);  // (the 2-launch turn gathers z at the columns of its rows)
    double d = 0.0, a = 0.0;
    if (rp.n > 0) {
        d += vr.x * vz.x;
        a += fabs(vr.x);
    }
    if (rp.n > 1) {
        d += vr.y * vz.y;
        a += fabs(vr.y);
    }
    block_sum2(d, a, slot);
    if (threadIdx.x == 0) {
        part_rho[chunk] = d;
        part_norm[chunk] = a;
    }
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

// ------------------------------------------------------------------------------------------
// Half storage of a symmetric matrix on a banded pattern (SymLayout, host_matrix.hpp).  ND planes per
// chunk: plane j holds A(r, r + d[j]), d[0] = 0.  A lower entry A(r, r - d[j]) is read where its twin
// lives: plane j of row r - d[j] -- a coalesced strip of values that some workgroup reads (or has read) as
// upper entries, so DRAM delivers every coefficient once; the second reader finds it in L2 / the Infinity
// Cache.  8 ND + 1 bytes per row instead of 8.1 per stored entry: measured 115 us against 139 us for the
// pattern-coded full storage on the 216^3 matrix (tools/sym_tune.hip, profiles/spmv_tune_r02.txt).  Rows are
// summed in ascending column order -- furthest lower entry first, diagonal, upper entries -- so y and the
// fused dot partials have the same bits as k_spmv_sell / k_spmv_stream.
// ------------------------------------------------------------------------------------------
struct SymOffsets {
    int d[SYM_MAX_OFFSETS];
};
// FAST: d[1] == 1 and every further distance even (a box with an even line length) -- known at compile
// time, so the kernel stays straight-line code (a run-time test of the parity splits the loads into basic
// blocks that wait for each other: 130 us instead of 113, tools/sym_tune.hip var1/var2).  Then the two rows
// of a lane are an aligned pair in every strip: x and the lower values of the even distances come as one
// 16-byte load per pair, the d = 1 neighbours from the diagonal pair and the lane's own plane-1 value.
// STREAM (planes + vectors larger than the Infinity Cache): the planes that are read exactly once per launch are
// streamed past the caches -- the diagonal always, and with FAST also plane 1, whose second reader (the d = 1
// lower entry of the next row) is the neighbouring lane: a lane shuffle instead of a load.
template <int MODE, int NDOT, int ND, bool FAST, bool STREAM>
__global__ __launch_bounds__(BLOCK) void k_spmv_sym(int n_rows, int n_chunks, SymOffsets off,
                                                    const uint8_t *__restrict__ mask,
                                                    const double *__restrict__ planes,
                                                    const double *__restrict__ x, const double *__restrict__ b,
                                                    double *__restrict__ y, const double *__restrict__ w,
                                                    double *__restrict__ dot_partials,
                                                    double *__restrict__ dot2_partials, const DevScalars *gate,
                                                    const int *__restrict__ block_order, HaloFused hf)
{
    __shared__ double slot[N_WAVES];
    __shared__ double ys[CHUNK_ROWS];
    if (gate && gate->stop) return;
    // banded patterns: the host's order puts the chunks of rows r and r +- d[ND-1] on one XCD (band_block_order)
    const int chunk = block_order ? block_order[blockIdx.x] : xcd_chunk(blockIdx.x);
    if (chunk < 0 || chunk >= n_chunks) return;
    const int t = threadIdx.x;
    const RowPair rp = my_rows(chunk, n_rows);
    const int row = rp.row;
    // which of the 2 ND - 1 entries the two rows have: bit (ND-1-j) = the one at -d[j], bit (ND-1+j) = at +d[j]
    const unsigned mm = *reinterpret_cast<const unsigned short *>(mask + row);
    const unsigned m0 = mm & 0xffu, m1 = mm >> 8;
    double2 acc;
    acc.x = acc.y = 0.0;
    if (MODE == SPMV_RESIDUAL) acc = ld2(b, rp);
    // own planes: diagonal and upper entries of the two rows
    double2 up[ND];
    const double *own = planes + (long)chunk * (ND * CHUNK_ROWS) + t * ROWS_PER_THREAD;
#pragma unroll
    for (int j = 0; j < ND; ++j)
        up[j] = (STREAM && (j == 0 || (FAST && j == 1))) ? ld_pair_stream(own + (long)j * CHUNK_ROWS)
                                                         : *reinterpret_cast<const double2 *>(own + (long)j * CHUNK_ROWS);
    const double2 xd = ld2(x, rp);
    // lower entries: plane j at rows row - d[j], row + 1 - d[j].  Every load is issued whatever the mask says, at an
    // index clamped into its array (the mask decides below what is used): the loads do not wait for the mask
    const int last = n_rows - 1, last_pair = last & ~1;  // (a pair load at the last even row: the vectors are allocated two past n_rows)
    double2 lo[ND];
#pragma unroll
    for (int j = 1; j < ND; ++j) {
        const int r0 = max(row - off.d[j], 0), r1 = max(row + 1 - off.d[j], 0);
        const long a0 = (long)(r0 >> 9) * (ND * CHUNK_ROWS) + (long)j * CHUNK_ROWS + (r0 & (CHUNK_ROWS - 1));
        const long a1 = (long)(r1 >> 9) * (ND * CHUNK_ROWS) + (long)j * CHUNK_ROWS + (r1 & (CHUNK_ROWS - 1));
        if (FAST && j >= 2) {  // even distance: rows r0, r0 + 1 are an aligned pair of one chunk's plane
            lo[j] = *reinterpret_cast<const double2 *>(planes + a0);
        } else if (FAST) {     // d = 1: A(row + 1, row) is this lane's own upper entry of row
            if (STREAM) {      // A(row, row - 1) is the previous lane's second plane-1 value (lane 0: from memory)
                lo[j].x = __shfl_up(up[1].y, 1, WAVE);
                if ((t & (WAVE - 1)) == 0) lo[j].x = planes[a0];
            } else {
                lo[j].x = planes[a0];
            }
            lo[j].y = up[1].x;
        } else {
            lo[j].x = planes[a0];
            lo[j].y = planes[a1];
        }
    }
    static_assert(CHUNK_ROWS == 512, "row >> 9 above");
    double2 xl[ND], xu[ND];
#pragma unroll
    for (int j = 1; j < ND; ++j) {
        if (FAST && j >= 2) {
            xl[j] = *reinterpret_cast<const double2 *>(x + min(max(row - off.d[j], 0), last_pair));
            xu[j] = *reinterpret_cast<const double2 *>(x + min(row + off.d[j], last_pair));
        } else if (FAST) {     // the neighbours of a pair at distance 1: the pair itself + one on each side
            xl[j].x = x[min(max(row - 1, 0), last)];
            xl[j].y = xd.x;
            xu[j].x = xd.y;
            xu[j].y = x[min(row + 2, last)];
        } else {
            xl[j].x = x[min(max(row - off.d[j], 0), last)];
            xl[j].y = x[min(max(row + 1 - off.d[j], 0), last)];
            xu[j].x = x[min(row + off.d[j], last)];
            xu[j].y = x[min(row + 1 + off.d[j], last)];
        }
    }
#pragma unroll
    for (int j = ND - 1; j >= 1; --j) {  // ascending columns: the furthest lower entry first
        if ((m0 >> (ND - 1 - j)) & 1u) {
            const double p = lo[j].x * xl[j].x;
            acc.x = (MODE == SPMV_RESIDUAL) ? acc.x - p : acc.x + p;
        }
        if ((m1 >> (ND - 1 - j)) & 1u) {
            const double p = lo[j].y * xl[j].y;
            acc.y = (MODE == SPMV_RESIDUAL) ? acc.y - p : acc.y + p;
        }
    }
    if ((m0 >> (ND - 1)) & 1u) {
        const double p = up[0].x * xd.x;
        acc.x = (MODE == SPMV_RESIDUAL) ? acc.x - p : acc.x + p;
    }
    if ((m1 >> (ND - 1)) & 1u) {
        const double p = up[0].y * xd.y;
        acc.y = (MODE == SPMV_RESIDUAL) ? acc.y - p : acc.y + p;
    }
#pragma unroll
    for (int j = 1; j < ND; ++j) {
        if ((m0 >> (ND - 1 + j)) & 1u) {
            const double p = up[j].x * xu[j].x;
            acc.x = (MODE == SPMV_RESIDUAL) ? acc.x - p : acc.x + p;
        }
        if ((m1 >> (ND - 1 + j)) & 1u) {
            const double p = up[j].y * xu[j].y;
            acc.y = (MODE == SPMV_RESIDUAL) ? acc.y - p : acc.y + p;
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

// ------------------------------------------------------------------------------------------
// Small systems on half storage, two launches per GKOCG turn: step_1x_fin and the SpMV in one kernel.  The SpMV
// needs p_new = z + (rho/rho') p at the columns of its rows, which other workgroups own -- but p_new is an
// elementwise function of z (which step_2r_fin leaves behind for this; r itself without a preconditioner) and the
// old p, so every workgroup recomputes it for the columns it gathers (same expression, same rounding:
// -ffp-contract=off) instead of waiting for a kernel boundary.  The new p of the
// own rows goes to the other of two p buffers (neighbours still read the old one).  Everything else -- check of
// the previous turn, pending x update, row sums in ascending column order, partial of p.q -- is what
// k_cg_step1x_fin followed by k_spmv_sym<SPMV_PLAIN, 1> does, bit for bit.
//   turn = [this kernel] -> k_cg_step2r_fin
// ------------------------------------------------------------------------------------------
// what a thread of the merged kernels holds of its two rows: [0] p, [1] z at the rows and at the gathered columns,
// own planes (diagonal, upper entries) and the twins of the lower entries
template <int ND>
struct TurnSymRegs {
    double2 vd[2], vl[2][ND], vu[2][ND], up[ND], lo[ND];
};
// Every load is issued without waiting for the mask (mask -> gathers would be two round trips, and a small system
// is all latency), at an index clamped into the vector; the mask decides later what is used.
template <int ND, bool FAST, bool STREAM>
__device__ __forceinline__ void turn_sym_load(TurnSymRegs<ND> &R, int chunk, const RowPair &rp, int n_rows,
                                              const SymOffsets &off, const double *__restrict__ planes,
                                              const double *__restrict__ p_in, const double *__restrict__ z)
{
    const int row = rp.row;
    const double *src[2] = {p_in, z};
    const int last = n_rows - 1, last_pair = last & ~1;  // (a pair load at the last even row: the vectors are allocated two past n_rows)
#pragma unroll
    for (int a = 0; a < 2; ++a) {
        const double *__restrict__ v = src[a];
        R.vd[a] = ld2(v, rp);
#pragma unroll
        for (int j = 1; j < ND; ++j) {
            if (FAST && j >= 2) {
                R.vl[a][j] = *reinterpret_cast<const double2 *>(v + min(max(row - off.d[j], 0), last_pair));
                R.vu[a][j] = *reinterpret_cast<const double2 *>(v + min(row + off.d[j], last_pair));
            } else if (FAST) {
                R.vl[a][j].x = v[min(max(row - 1, 0), last)];
                R.vl[a][j].y = R.vd[a].x;
                R.vu[a][j].x = R.vd[a].y;
                R.vu[a][j].y = v[min(row + 2, last)];
            } else {
                R.vl[a][j].x = v[min(max(row - off.d[j], 0), last)];
                R.vl[a][j].y = v[min(max(row + 1 - off.d[j], 0), last)];
                R.vu[a][j].x = v[min(row + off.d[j], last)];
                R.vu[a][j].y = v[min(row + 1 + off.d[j], last)];
            }
        }
    }
    // the matrix, as k_spmv_sym reads it (STREAM: the planes read once per launch go past the caches)
    const double *own = planes + (long)chunk * (ND * CHUNK_ROWS) + threadIdx.x * ROWS_PER_THREAD;
#pragma unroll
    for (int j = 0; j < ND; ++j)
        R.up[j] = (STREAM && (j == 0 || (FAST && j == 1))) ? ld_pair_stream(own + (long)j * CHUNK_ROWS)
                                                           : *reinterpret_cast<const double2 *>(own + (long)j * CHUNK_ROWS);
#pragma unroll
    for (int j = 1; j < ND; ++j) {
        const int r0 = max(row - off.d[j], 0), r1 = max(row + 1 - off.d[j], 0);
        const long a0 = (long)(r0 >> 9) * (ND * CHUNK_ROWS) + (long)j * CHUNK_ROWS + (r0 & (CHUNK_ROWS - 1));
        const long a1 = (long)(r1 >> 9) * (ND * CHUNK_ROWS) + (long)j * CHUNK_ROWS + (r1 & (CHUNK_ROWS - 1));
        if (FAST && j >= 2) {
            R.lo[j] = *reinterpret_cast<const double2 *>(planes + a0);
        } else if (FAST) {
            if (STREAM) {  // A(row, row - 1) is the previous lane's second plane-1 value (lane 0: from memory)
                const double prev = __shfl_up(R.up[1].y, 1, WAVE);
                R.lo[j].x = prev;
                if ((threadIdx.x & (WAVE - 1)) == 0) R.lo[j].x = planes[a0];
            } else {
                R.lo[j].x = planes[a0];
            }
            R.lo[j].y = R.up[1].x;
        } else {
            R.lo[j].x = planes[a0];
            R.lo[j].y = planes[a1];
        }
    }
}
// p_new = z + tmp p for the own rows (-> xd) and for every gathered column, then the two row sums in ascending
// column order (k_spmv_sym's)
template <int ND>
__device__ __forceinline__ double2 turn_sym_rows(const TurnSymRegs<ND> &R, unsigned m0, unsigned m1, double tmp,
                                                 double2 &xd)
{
    double2 xl[ND], xu[ND];
    xd.x = R.vd[1].x + tmp * R.vd[0].x;
    xd.y = R.vd[1].y + tmp * R.vd[0].y;
#pragma unroll
    for (int j = 1; j < ND; ++j) {
        xl[j].x = R.vl[1][j].x + tmp * R.vl[0][j].x;
        xl[j].y = R.vl[1][j].y + tmp * R.vl[0][j].y;
        xu[j].x = R.vu[1][j].x + tmp * R.vu[0][j].x;
        xu[j].y = R.vu[1][j].y + tmp * R.vu[0][j].y;
    }
    double2 acc;
    acc.x = acc.y = 0.0;
#pragma unroll
    for (int j = ND - 1; j >= 1; --j) {  // ascending columns: the furthest lower entry first
        if ((m0 >> (ND - 1 - j)) & 1u) acc.x = acc.x + R.lo[j].x * xl[j].x;
        if ((m1 >> (ND - 1 - j)) & 1u) acc.y = acc.y + R.lo[j].y * xl[j].y;
    }
    if ((m0 >> (ND - 1)) & 1u) acc.x = acc.x + R.up[0].x * xd.x;
    if ((m1 >> (ND - 1)) & 1u) acc.y = acc.y + R.up[0].y * xd.y;
#pragma unroll
    for (int j = 1; j < ND; ++j) {
        if ((m0 >> (ND - 1 + j)) & 1u) acc.x = acc.x + R.up[j].x * xu[j].x;
        if ((m1 >> (ND - 1 + j)) & 1u) acc.y = acc.y + R.up[j].y * xu[j].y;
    }
    return acc;
}

template <int ND, bool FAST>
__global__ __launch_bounds__(BLOCK) void k_cg_turn_sym(int n_rows, int n_chunks, SymOffsets off,
                                                       const uint8_t *__restrict__ mask,
                                                       const double *__restrict__ planes,
                                                       const double *__restrict__ p_in, double *__restrict__ p_out,
                                                       double *__restrict__ x, const double *__restrict__ z,
                                                       double *__restrict__ q,
                                                       double *__restrict__ part_beta, const DevScalars *sin,
                                                       DevScalars *sout, const double *__restrict__ part_rho,
                                                       const double *__restrict__ part_norm, int n_part,
                                                       double *history, int first,
                                                       const int *__restrict__ block_order)
{
    __shared__ double red[2 * FIN_WAVES];
    __shared__ double sh[4];
    __shared__ int sh_stop;
    __shared__ double slot[N_WAVES];
    const int chunk = block_order ? block_order[blockIdx.x] : xcd_chunk(blockIdx.x);
    if (chunk < 0 || chunk >= n_chunks) return;
    const bool lead = chunk == 0;  // the workgroup that stores the scalars and the history entry
    // scalars field by field (k_cg_step1x_fin), and every load of the kernel asked for before the first wait
    const int stopped = sin->stop;
    const double s_rho = sin->rho, s_beta = sin->beta, s_nf = sin->norm_factor, s_init = sin->init_res;
    const int s_iter = sin->iter, s_evals = sin->n_evals;
    const double c_tol = sin->crit.tolerance, c_rel = sin->crit.rel_tol;
    const int c_min = sin->crit.min_iter, c_max = sin->crit.max_iter, c_freq = sin->crit.frequency,
              c_exp = sin->crit.export_res;
    if (lead && threadIdx.x < sizeof(DevScalars) / 8)
        reinterpret_cast<unsigned long long *>(sout)[threadIdx.x] =
            reinterpret_cast<const unsigned long long *>(sin)[threadIdx.x];
    const RowPair rp = my_rows(chunk, n_rows);
    const unsigned mm = *reinterpret_cast<const unsigned short *>(mask + rp.row);
    const unsigned m0 = mm & 0xffu, m1 = mm >> 8;
    double pv[2][FIN_VT];
    load_partials_as_finaliser<2>(part_rho, part_norm, n_part, pv);
    double2 vx = ld2(x, rp);
    TurnSymRegs<ND> R;
    turn_sym_load<ND, FAST, false>(R, chunk, rp, n_rows, off, planes, p_in, z);
    if (stopped) return;  // (the solve has ended: the lead workgroup has handed the scalars on)
    double v[2];
    reduce_partials_as_finaliser<2>(pv, n_part, red, v);
    if (threadIdx.x == 0) {
        // FIN_CG_CHECK as in k_cg_step1x_fin (StoppingCriterion.C:71-151)
        const double prev_rho = s_rho, rho = v[0];
        int iter = s_iter, n_evals = s_evals, stop = 0;
        double init_res = s_init, res = 0.0;
        bool evaluated = false;
        if (iter > 0 && iter < c_min) {           // :77-81
            iter += 1;
        } else if (iter % c_freq != 0) {          // :84-87
            iter += 1;
        } else {
            evaluated = true;
            n_evals += 1;
            res = v[1];
            if (iter == 0) init_res = res / s_nf;  // :102-111
            res /= s_nf;                           // :113
            if (c_exp && history && lead) history[iter] = res;  // :115-117
            if (iter >= c_max) stop = 1;                        // :124
            if (res < c_tol) stop = 1;                          // :128
            if (c_rel > 0 && res < c_rel * init_res) stop = 1;  // :132-136
            iter += 1;                                          // :143
        }
        sh[0] = s_beta;
        sh[1] = prev_rho;
        sh[2] = rho;
        sh_stop = stop;
        if (lead) {
            sout->prev_rho = prev_rho;
            sout->rho = rho;
            sout->iter = iter;
            sout->x_pending = 0;
            if (evaluated) {
                sout->n_evals = n_evals;
                sout->init_res = init_res;
                sout->res = res;
            }
            if (stop) sout->stop = 1;
        }
    }
    __syncthreads();
    const double beta = sh[0], prev = sh[1], rho = sh[2];
    const int stop = sh_stop;
    if (!first && beta != 0.0) {  // x += t_j p of the turn this check closed
        const double t = prev / beta;
        vx.x += t * R.vd[0].x;
        vx.y += t * R.vd[0].y;
        st2(x, rp, vx);
    }
    if (stop) return;
    const double tmp = (prev == 0.0) ? 0.0 : rho / prev;
    double2 xd;
    const double2 acc = turn_sym_rows<ND>(R, m0, m1, tmp, xd);
    st2(p_out, rp, xd);
    st2(q, rp, acc);
    double d = 0.0;
    if (rp.n > 0) d += xd.x * acc.x;
    if (rp.n > 1) d += xd.y * acc.y;
    const double sd = block_sum(d, slot);
    if (threadIdx.x == 0) part_beta[chunk] = sd;
}

// The same merge for systems of any size, between the single-workgroup finalisers of the five-launch turn (which
// becomes four): the scalars are read where k_cg_step1x reads them, the pending x update included.  Per turn the
// vectors cost 8 N bytes less than step_1x + SpMV + step_2r (p is read once, z written once and read once instead of
// r and 1/d read twice), and one kernel boundary goes.
// HALO (several ranks, peer-put transport): the neighbours' step_2r has put z of the halo columns; a workgroup whose
// chunk holds boundary rows waits for it, forms p_new at those columns from the old halo p it keeps (ph_in -> ph_out)
// and continues its boundary rows over their non-local entries -- no put and no wait for a put of THIS launch here.
template <int ND, bool FAST, bool STREAM, bool HALO>
__global__ __launch_bounds__(BLOCK) void k_cg_turn_sym_big(int n_rows, int n_chunks, SymOffsets off,
                                                           const uint8_t *__restrict__ mask,
                                                           const double *__restrict__ planes,
                                                           const double *__restrict__ p_in,
                                                           double *__restrict__ p_out, double *__restrict__ x,
                                                           const double *__restrict__ z, double *__restrict__ q,
                                                           double *__restrict__ part_beta, const DevScalars *s,
                                                           const int *__restrict__ block_order, HaloFused hf,
                                                           const double *__restrict__ ph_in,
                                                           double *__restrict__ ph_out)
{
    __shared__ double slot[N_WAVES];
    const int stop = s->stop;
    const bool pending = s->x_pending != 0;
    if (stop && !pending) return;
    const int chunk = block_order ? block_order[blockIdx.x] : xcd_chunk(blockIdx.x);
    if (chunk < 0 || chunk >= n_chunks) return;
    const RowPair rp = my_rows(chunk, n_rows);
    if (stop) {  // the solve has ended with an update still to apply: that only
        const double beta = s->beta;
        if (beta != 0.0) {
            const double t = s->prev_rho / beta;
            const double2 vp = ld2(p_in, rp);
            double2 vx = ld2_stream(x, rp);
            vx.x += t * vp.x;
            vx.y += t * vp.y;
            st2_stream(x, rp, vx);
        }
        return;
    }
    const unsigned mm = *reinterpret_cast<const unsigned short *>(mask + rp.row);
    const unsigned m0 = mm & 0xffu, m1 = mm >> 8;
    TurnSymRegs<ND> R;
    turn_sym_load<ND, FAST, STREAM>(R, chunk, rp, n_rows, off, planes, p_in, z);
    if (pending) {
        const double beta = s->beta;
        if (beta != 0.0) {
            const double t = s->prev_rho / beta;
            double2 vx = ld2_stream(x, rp);  // x is touched once per turn
            vx.x += t * R.vd[0].x;
            vx.y += t * R.vd[0].y;
            st2_stream(x, rp, vx);
        }
    }
    const double rho = s->rho, prev = s->prev_rho;
    const double tmp = (prev == 0.0) ? 0.0 : rho / prev;
    double2 xd;
    double2 acc = turn_sym_rows<ND>(R, m0, m1, tmp, xd);
    st2(p_out, rp, xd);
    if (HALO) {
        __shared__ double ys[CHUNK_ROWS];
        if (hf.chunk_bptr) halo_fused_add<SPMV_PLAIN, true>(hf, chunk, acc.x, acc.y, ys, tmp, ph_in, ph_out);
    }
    st2(q, rp, acc);
    double d = 0.0;
    if (rp.n > 0) d += xd.x * acc.x;
    if (rp.n > 1) d += xd.y * acc.y;
    const double sd = block_sum(d, slot);
    if (threadIdx.x == 0) part_beta[chunk] = sd;
}

// ------------------------------------------------------------------------------------------
// Half storage with per-chunk distances and explicit exceptions (SymxChunk, common.hpp; build_symx_layout).
// As k_spmv_sym, but the distances are the chunk's own (run-time values from its header, which also says where
// the twins of its lower entries live: in the planes of the chunk of row r - d or of the next one), and rows
// flagged in their mask byte carry explicit entries -- column + value lists -- that are merged into the row sum by
// column, so that every row is still summed in ascending column order and y keeps the bits of the other kernels.
// ------------------------------------------------------------------------------------------
// The explicit entries of a chunk (few: the couplings across block faces) are staged by its workgroup -- thread i
// takes entry i: column and value * x[column] into LDS -- while the plane loads are in flight; a row then walks its
// own entries in LDS.  Per row that costs two registers (next entry, end) where prefetching the first entries into
// registers cost 40 for the kernel, i.e. three of eight wavefronts per SIMD.  A chunk with more than
// SYMX_LDS_ENTRIES of them reads them from memory as it goes.
template <int MODE, bool IN_LDS>
__device__ __forceinline__ void symx_explicit(double &acc, int &k, int end, int limit, const int *lds_cols,
                                              const double *lds_prod, const int *__restrict__ ex_cols,
                                              const double *__restrict__ ex_vals, const double *__restrict__ x)
{
    while (k < end) {
        const int c = IN_LDS ? lds_cols[k] : ex_cols[k];
        if (c >= limit) break;
        const double p = IN_LDS ? lds_prod[k] : ex_vals[k] * x[c];
        acc = (MODE == SPMV_RESIDUAL) ? acc - p : acc + p;
        ++k;
    }
}

// FAST (as in k_spmv_sym; known per layout: every chunk's first distance is 1 and its further ones are even --
// blocks with even line lengths): the two rows of a lane are an aligned pair in every strip, so x and the lower
// values of the even distances come as one 16-byte load per pair and the d = 1 neighbours from the lane's own
// diagonal pair and plane-1 value.
template <int MODE, int NDOT, bool STREAM, bool FAST, bool GENERAL>
__global__ __launch_bounds__(BLOCK) void k_spmv_symx(int n_rows, int n_chunks, const SymxChunk *__restrict__ hdr,
                                                     const uint8_t *__restrict__ mask,
                                                     const double *__restrict__ planes,
                                                     const int *__restrict__ ex_rowptr,
                                                     const int *__restrict__ ex_cols,
                                                     const double *__restrict__ ex_vals,
                                                     const int *__restrict__ ex_lrow,
                                                     const double *__restrict__ x, const double *__restrict__ b,
                                                     double *__restrict__ y, const double *__restrict__ w,
                                                     double *__restrict__ dot_partials,
                                                     double *__restrict__ dot2_partials, const DevScalars *gate,
                                                     HaloFused hf)
{
    __shared__ double slot[N_WAVES];
    __shared__ double ys[CHUNK_ROWS];
    // GENERAL: the chunk's explicit entries as a list (column, product), walked by their rows; lean kernel: at most
    // one ahead of and one behind the planar entries of a row -- a slot per row for each, and a flag that says it is taken
    __shared__ int ex_c[GENERAL ? SYMX_LDS_ENTRIES : 1];
    __shared__ double ex_p[GENERAL ? SYMX_LDS_ENTRIES : 1];
    __shared__ double ex_a[GENERAL ? 1 : CHUNK_ROWS], ex_b[GENERAL ? 1 : CHUNK_ROWS];
    __shared__ unsigned char ex_fa[GENERAL ? 2 : CHUNK_ROWS], ex_fb[GENERAL ? 2 : CHUNK_ROWS];
    if (gate && gate->stop) return;
    // the headers are stored in dispatch order and name their chunk: one round trip (header -> data) instead of
    // two (order -> header -> data) in front of the loads.  The header is only ever indexed with compile-time
    // constants (every loop below is fully unrolled): it stays in scalar registers; a run-time index would push all
    // 96 bytes into scratch memory
    const SymxChunk h = hdr[blockIdx.x];
    const int chunk = h.chunk;
    if (chunk < 0 || chunk >= n_chunks) return;
    const int t = threadIdx.x;
    const RowPair rp = my_rows(chunk, n_rows);
    const int row = rp.row, r0 = chunk * CHUNK_ROWS;
    const unsigned mm = *reinterpret_cast<const unsigned short *>(mask + row);
    const unsigned m0 = mm & 0xffu, m1 = mm >> 8;
    const bool chunk_explicit = h.ex_rp_off >= 0;                                   // (workgroup-uniform)
    const bool has_explicit = GENERAL && chunk_explicit && ((m0 | m1) & SYMX_EXTRAS_BIT);   // this lane's rows
    const bool in_lds = h.ex_count <= SYMX_LDS_ENTRIES;
    // GENERAL: next explicit entry / end of the two rows, as indices into the staged entries (or into the arrays)
    int k0 = 0, e0 = 0, k1 = 0, e1 = 0;
    if (has_explicit) {
        const int *rpx = ex_rowptr + h.ex_rp_off + t * ROWS_PER_THREAD;
        const int shift = in_lds ? h.ex_begin : 0;
        k0 = rpx[0] - shift;
        e0 = k1 = rpx[1] - shift;
        e1 = rpx[2] - shift;
    }
    // the entry this thread stages (entry t of the chunk; the few chunks with more than 256 loop)
    int my_col = 0, my_lrow = 0;
    double my_val = 0.0;
    const bool stage = chunk_explicit && (in_lds || !GENERAL) && t < h.ex_count;
    if (stage) {
        my_col = ex_cols[h.ex_begin + t];
        my_val = ex_vals[h.ex_begin + t];
        if (!GENERAL) my_lrow = ex_lrow[h.ex_begin + t];
    }
    if (!GENERAL && chunk_explicit) {  // (own rows' flags down; the barrier below orders this before the staging)
        *reinterpret_cast<unsigned short *>(ex_fa + ROWS_PER_THREAD * t) = 0;
        *reinterpret_cast<unsigned short *>(ex_fb + ROWS_PER_THREAD * t) = 0;
    }
    double2 acc;
    acc.x = acc.y = 0.0;
    if (MODE == SPMV_RESIDUAL) acc = ld2(b, rp);
    // Distances, plane positions and twin bases padded to three with dummies (distance 0, plane 0) that no mask bit
    // refers to, and every load issued whatever the mask says, at an index clamped into its array: straight-line
    // code -- a run-time "does this plane exist" in front of each group of loads splits them into basic blocks that
    // wait for each other (tools/sym_tune.hip var1/var2)
    const int last = n_rows - 1, last_pair = last & ~1;  // (a pair load at the last even row: the vectors are allocated two past n_rows)
    int dj[4] = {0, 0, 0, 0};
    long pj[4] = {0, 0, 0, 0}, b0[4], b1[4];
#pragma unroll
    for (int j = 1; j < 4; ++j) {
        const bool have = j < h.nd;
        dj[j] = have ? h.d[j - 1] : 0;
        pj[j] = have ? (long)j * CHUNK_ROWS : 0;
        b0[j] = (have && h.lo_base[j - 1][0] >= 0) ? h.lo_base[j - 1][0] : h.val_off;
        b1[j] = (have && h.lo_base[j - 1][1] >= 0) ? h.lo_base[j - 1][1] : h.val_off;
    }
    // own planes: diagonal and upper entries of the two rows
    double2 up[4];
    const double *own = planes + h.val_off + t * ROWS_PER_THREAD;
#pragma unroll
    for (int j = 0; j < 4; ++j)
        up[j] = (STREAM && (j == 0 || (FAST && j == 1))) ? ld_pair_stream(own + pj[j])
                                                         : *reinterpret_cast<const double2 *>(own + pj[j]);
    const double2 xd = ld2(x, rp);
    double2 lo[4], xl[4], xu[4];
#pragma unroll
    for (int j = 1; j < 4; ++j) {
        const int d = dj[j];
        // first row of the chunk minus d: its chunk (floor division) is lo_base[.][0], the next one [.][1]
        const int cs0 = (r0 - d) >> 9;
        if (FAST && j >= 2) {  // even distance: rows (row - d, row + 1 - d) are an aligned pair of one chunk's plane
            const int rs = row - d;
            lo[j] = *reinterpret_cast<const double2 *>(planes + ((rs >> 9) == cs0 ? b0[j] : b1[j]) +
                                                       (rs & (CHUNK_ROWS - 1)));
            xl[j] = *reinterpret_cast<const double2 *>(x + min(max(rs, 0), last_pair));
            xu[j] = *reinterpret_cast<const double2 *>(x + min(row + d, last_pair));
        } else if (FAST) {     // d = 1: A(row + 1, row) is this lane's own plane-1 value of row
            const int rs = row - 1;
            const long at = ((rs >> 9) == cs0 ? b0[j] : b1[j]) + (rs & (CHUNK_ROWS - 1));
            if (STREAM) {      // A(row, row - 1) is the previous lane's second plane-1 value (lane 0: from memory)
                lo[j].x = __shfl_up(up[1].y, 1, WAVE);
                if ((t & (WAVE - 1)) == 0) lo[j].x = planes[at];
            } else {
                lo[j].x = planes[at];
            }
            xl[j].x = x[min(max(rs, 0), last)];
            lo[j].y = up[1].x;
            xl[j].y = xd.x;
            xu[j].x = xd.y;
            xu[j].y = x[min(row + 2, last)];
        } else {
            const int ra = row - d, rb = row + 1 - d;
            lo[j].x = planes[((ra >> 9) == cs0 ? b0[j] : b1[j]) + (ra & (CHUNK_ROWS - 1))];
            lo[j].y = planes[((rb >> 9) == cs0 ? b0[j] : b1[j]) + (rb & (CHUNK_ROWS - 1))];
            xl[j].x = x[min(max(ra, 0), last)];
            xl[j].y = x[min(max(rb, 0), last)];
            xu[j].x = x[min(row + d, last)];
            xu[j].y = x[min(row + 1 + d, last)];
        }
    }
    static_assert(CHUNK_ROWS == 512, "row >> 9 above");
    // staging of the chunk's explicit entries (their x gather is the last link of the header -> entry -> x chain; the
    // plane and x loads above are in flight meanwhile)
    if (GENERAL && chunk_explicit && in_lds) {
        for (int i = t; i < h.ex_count; i += BLOCK) {
            const int c = i == t ? my_col : ex_cols[h.ex_begin + i];
            const double v = i == t ? my_val : ex_vals[h.ex_begin + i];
            ex_c[i] = c;
            ex_p[i] = v * x[c];
        }
        __syncthreads();
    }
    if (!GENERAL && chunk_explicit) {
        __syncthreads();
        for (int i = t; i < h.ex_count; i += BLOCK) {
            const int c = i == t ? my_col : ex_cols[h.ex_begin + i];
            const double v = i == t ? my_val : ex_vals[h.ex_begin + i];
            const int lr = i == t ? my_lrow : ex_lrow[h.ex_begin + i];
            const double p = v * x[c];
            if (lr & SYMX_BEHIND_BIT) {
                ex_b[lr & (CHUNK_ROWS - 1)] = p;
                ex_fb[lr & (CHUNK_ROWS - 1)] = 1;
            } else {
                ex_a[lr] = p;
                ex_fa[lr] = 1;
            }
        }
        __syncthreads();
        const int l0 = ROWS_PER_THREAD * t;
        if (ex_fa[l0]) acc.x = (MODE == SPMV_RESIDUAL) ? acc.x - ex_a[l0] : acc.x + ex_a[l0];
        if (ex_fa[l0 + 1]) acc.y = (MODE == SPMV_RESIDUAL) ? acc.y - ex_a[l0 + 1] : acc.y + ex_a[l0 + 1];
    }
    // the row walk in ascending column order: the furthest lower entry first.  Explicit entries: the lean kernel has
    // added the one ahead of the planar entries above and adds the one behind them below; the general kernel merges
    // a row's list by column (an entry that repeats a column a plane holds comes after the plane's entry: `<` in
    // symx_explicit)
    auto ex_row0 = [&](int limit) {
        if (in_lds)
            symx_explicit<MODE, true>(acc.x, k0, e0, limit, ex_c, ex_p, ex_cols, ex_vals, x);
        else
            symx_explicit<MODE, false>(acc.x, k0, e0, limit, ex_c, ex_p, ex_cols, ex_vals, x);
    };
    auto ex_row1 = [&](int limit) {
        if (in_lds)
            symx_explicit<MODE, true>(acc.y, k1, e1, limit, ex_c, ex_p, ex_cols, ex_vals, x);
        else
            symx_explicit<MODE, false>(acc.y, k1, e1, limit, ex_c, ex_p, ex_cols, ex_vals, x);
    };
    // (the general kernel only sees chunks with SymxChunk::merge set: its rows' explicit entries are merged by column)
    const bool merge = has_explicit;
#pragma unroll
    for (int j = 3; j >= 1; --j) {
        if ((m0 >> (3 - j)) & 1u) {
            if (merge) ex_row0(row - dj[j]);
            const double p = lo[j].x * xl[j].x;
            acc.x = (MODE == SPMV_RESIDUAL) ? acc.x - p : acc.x + p;
        }
        if ((m1 >> (3 - j)) & 1u) {
            if (merge) ex_row1(row + 1 - dj[j]);
            const double p = lo[j].y * xl[j].y;
            acc.y = (MODE == SPMV_RESIDUAL) ? acc.y - p : acc.y + p;
        }
    }
    if ((m0 >> 3) & 1u) {
        if (merge) ex_row0(row);
        const double p = up[0].x * xd.x;
        acc.x = (MODE == SPMV_RESIDUAL) ? acc.x - p : acc.x + p;
    }
    if ((m1 >> 3) & 1u) {
        if (merge) ex_row1(row + 1);
        const double p = up[0].y * xd.y;
        acc.y = (MODE == SPMV_RESIDUAL) ? acc.y - p : acc.y + p;
    }
#pragma unroll
    for (int j = 1; j < 4; ++j) {
        if ((m0 >> (3 + j)) & 1u) {
            if (merge) ex_row0(row + dj[j]);
            const double p = up[j].x * xu[j].x;
            acc.x = (MODE == SPMV_RESIDUAL) ? acc.x - p : acc.x + p;
        }
        if ((m1 >> (3 + j)) & 1u) {
            if (merge) ex_row1(row + 1 + dj[j]);
            const double p = up[j].y * xu[j].y;
            acc.y = (MODE == SPMV_RESIDUAL) ? acc.y - p : acc.y + p;
        }
    }
    if (has_explicit) {
        ex_row0(INT32_MAX);
        ex_row1(INT32_MAX);
    }
    if (!GENERAL && chunk_explicit) {
        const int l0 = ROWS_PER_THREAD * t;
        if (ex_fb[l0]) acc.x = (MODE == SPMV_RESIDUAL) ? acc.x - ex_b[l0] : acc.x + ex_b[l0];
        if (ex_fb[l0 + 1]) acc.y = (MODE == SPMV_RESIDUAL) ? acc.y - ex_b[l0 + 1] : acc.y + ex_b[l0 + 1];
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

// ------------------------------------------------------------------------------------------
// Index-compressed chunked ELL SpMV (SellChunk, common.hpp).  Values: eight 16-byte loads per lane
// and group of 8 slots.  Columns, per chunk: pattern mode -- one 2-byte load brings the pattern ids
// of the lane's two rows, the pattern table in LDS gives the offsets; offset mode -- one 16-byte
// load brings 16 column codes, the offset dictionary in LDS turns a code into column = row +
// offset.  8.1 / 9 bytes per stored entry instead of 12 -- measured 137 / 146 us against 187 us
// for the CSR-stream kernel on the 216^3 matrix (profiles/spmv_tune_r01.txt).  Same per-row order
// as k_spmv_stream.
// ------------------------------------------------------------------------------------------
// STREAM: as in k_spmv_stream -- value planes and 16 / 32-bit code words of a matrix larger than the Infinity
// Cache are streamed past the caches.
template <int MODE, int NDOT, bool STREAM>
__global__ __launch_bounds__(BLOCK) void k_spmv_sell(int n_rows, int n_chunks,
                                                     const SellChunk *__restrict__ chunks,
                                                     const int *__restrict__ dict,
                                                     const uint8_t *__restrict__ codes,
                                                     const double *__restrict__ vals,
                                                     const int *__restrict__ spill_chunk_ptr,
                                                     const int *__restrict__ spill_rows,
                                                     const int *__restrict__ spill_ptrs,
                                                     const int *__restrict__ spill_cols,
                                                     const double *__restrict__ spill_vals,
                                                     const double *__restrict__ x,
                                                     const double *__restrict__ b,
                                                     double *__restrict__ y,
                                                     const double *__restrict__ w,
                                                     double *__restrict__ dot_partials,
                                                     double *__restrict__ dot2_partials,
                                                     const DevScalars *gate, int xgroup, HaloFused hf,
                                                     const uint16_t *__restrict__ rmap,
                                                     const int *__restrict__ block_order)
{
    __shared__ double slot[N_WAVES];
    __shared__ int stab[SELL_TABLE_INTS];
    static_assert(SELL_TABLE_INTS * sizeof(int) >= CHUNK_ROWS * sizeof(double), "the table doubles as the row-sum exchange");
    if (gate && gate->stop) return;
    const int chunk = block_order ? block_order[blockIdx.x] : xcd_chunk(blockIdx.x, xgroup);
    if (chunk < 0 || chunk >= n_chunks) return;
    const SellChunk h = chunks[chunk];
    const int t = threadIdx.x;
    // (DevSell::rmap: whose rows this thread's two slot rows hold -- asked for now, needed at the very end)
    unsigned own = 0;
    if (rmap) own = *reinterpret_cast<const unsigned *>(rmap + (long)chunk * CHUNK_ROWS + t * ROWS_PER_THREAD);
    // spill (the tails of this chunk's long rows): which row this thread will finish and where its
    // tail sits -- asked for now, so that the answers arrive while the planes are being worked on
    int sp0 = 0, sp1 = 0, s_row = 0, s_k0 = 0, s_k1 = 0;
    if (spill_chunk_ptr) {
        sp0 = spill_chunk_ptr[chunk];
        sp1 = spill_chunk_ptr[chunk + 1];
        if (sp0 + t < sp1) {
            s_row = spill_rows[sp0 + t];
            s_k0 = spill_ptrs[sp0 + t];
            s_k1 = spill_ptrs[sp0 + t + 1];
        }
    }
    if (h.mode() <= SELL_MODE_OFFSET8)  // (no table in delta / column mode)
        for (int i = t; i < h.dict_len(); i += BLOCK) stab[i] = dict[h.dict_off + i];
    __syncthreads();
    const RowPair rp = my_rows(chunk, n_rows);
    const int row = chunk * CHUNK_ROWS + t * ROWS_PER_THREAD;
    double2 acc;
    acc.x = acc.y = 0.0;
    if (MODE == SPMV_RESIDUAL) {
        if (rmap) {  // (b of the rows the slot rows hold)
            const int ra = chunk * CHUNK_ROWS + (int)(own & 0xffffu), rb = chunk * CHUNK_ROWS + (int)(own >> 16);
            acc.x = ra < n_rows ? b[ra] : 0.0;
            acc.y = rb < n_rows ? b[rb] : 0.0;
        } else {
            acc = ld2(b, rp);
        }
    }
    const double *v = vals + h.val_off + t * ROWS_PER_THREAD;
    // this wavefront runs to the longest of ITS rows: the planes beyond (padding up to the chunk's
    // longest row) are never touched.  Wave-uniform, so the loops below do not diverge.
    const int ww = h.wave_width(__builtin_amdgcn_readfirstlane(t / WAVE)), width = h.width();
    constexpr int BATCH = 8;
    // ... and every lane loads up to the longer of ITS two rows only (lengths: one byte per row in front
    // of the codes; not in pattern mode).  Padding that shares no 128-byte line with a slot in use costs
    // no memory traffic -- with the rows of a wavefront sorted by length that is nearly all of it.
    int ml = ww;
    if (h.mode() != SELL_MODE_PATTERN) {
        const unsigned ll =
            *reinterpret_cast<const unsigned short *>(codes + h.code_off - SELL_LEN_BYTES + t * ROWS_PER_THREAD);
        ml = (int)max(ll & 0xffu, ll >> 8);
    }
    if (h.mode() == SELL_MODE_DELTA16) {
        // delta mode: 16 bits per (row, slot), group-major 16-byte words of 4 slots x 2 rows; the
        // column of a slot is the running sum of the row's codes (first code relative to
        // row + dict_off).  The eight columns of a batch are formed first, then the eight gathers.
        static_assert(SELL_D16_GROUP * 2 == BATCH, "two code words per batch");
        const uint4 *cw = reinterpret_cast<const uint4 *>(codes + h.code_off) + t;
        int c0 = row + h.dict_off, c1 = row + 1 + h.dict_off;
        for (int s0 = 0; s0 < ww; s0 += BATCH) {
            const int g = s0 / SELL_D16_GROUP;
            uint4 wa, wb;
            wa.x = wa.y = wa.z = wa.w = 0xffffffffu;
            wb = wa;
            // (code words: read once per launch, whole lines per instruction -> streamed like the values)
            typedef unsigned u4v __attribute__((ext_vector_type(4)));
            if (s0 < ml) {
                const u4v tw = STREAM ? __builtin_nontemporal_load(reinterpret_cast<const u4v *>(cw + (long)g * BLOCK))
                                      : *reinterpret_cast<const u4v *>(cw + (long)g * BLOCK);
                wa.x = tw.x;
                wa.y = tw.y;
                wa.z = tw.z;
                wa.w = tw.w;
            }
            if (s0 + SELL_D16_GROUP < ml) {
                const u4v tw = STREAM ? __builtin_nontemporal_load(reinterpret_cast<const u4v *>(cw + (long)(g + 1) * BLOCK))
                                      : *reinterpret_cast<const u4v *>(cw + (long)(g + 1) * BLOCK);
                wb.x = tw.x;
                wb.y = tw.y;
                wb.z = tw.z;
                wb.w = tw.w;
            }
            const unsigned w8[BATCH] = {wa.x, wa.y, wa.z, wa.w, wb.x, wb.y, wb.z, wb.w};
            double2 vv[BATCH];
#pragma unroll
            for (int k = 0; k < BATCH; ++k) {
                vv[k].x = vv[k].y = 0.0;
                if (s0 + k < ml) vv[k] = STREAM ? ld_pair_stream(v + (long)(s0 + k) * CHUNK_ROWS) : *reinterpret_cast<const double2 *>(v + (long)(s0 + k) * CHUNK_ROWS);
            }
            int a0[BATCH], a1[BATCH];
            bool ok0[BATCH], ok1[BATCH];
#pragma unroll
            for (int k = 0; k < BATCH; ++k) {
                const unsigned d0 = w8[k] & 0xffffu, d1 = w8[k] >> 16;
                ok0[k] = (s0 + k < ml) && d0 != 0xffffu;
                ok1[k] = (s0 + k < ml) && d1 != 0xffffu;
                if (ok0[k]) c0 += (int)d0;
                if (ok1[k]) c1 += (int)d1;
                a0[k] = c0;
                a1[k] = c1;
            }
            double x0[BATCH], x1[BATCH];
#pragma unroll
            for (int k = 0; k < BATCH; ++k) {
                x0[k] = ok0[k] ? x[a0[k]] : 0.0;
                x1[k] = ok1[k] ? x[a1[k]] : 0.0;
            }
#pragma unroll
            for (int k = 0; k < BATCH; ++k) {
                if (ok0[k]) {
                    const double p = vv[k].x * x0[k];
                    acc.x = (MODE == SPMV_RESIDUAL) ? acc.x - p : acc.x + p;
                }
                if (ok1[k]) {
                    const double p = vv[k].y * x1[k];
                    acc.y = (MODE == SPMV_RESIDUAL) ? acc.y - p : acc.y + p;
                }
            }
        }
    } else if (h.mode() == SELL_MODE_COL32) {
        // column mode: plain 32-bit columns (-1 = padding), 16-byte words of 2 slots x 2 rows
        static_assert(SELL_C32_GROUP * 4 == BATCH, "four code words per batch");
        const int4 *cw = reinterpret_cast<const int4 *>(codes + h.code_off) + t;
        for (int s0 = 0; s0 < ww; s0 += BATCH) {
            const int g = s0 / SELL_C32_GROUP;
            int a0[BATCH], a1[BATCH];
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                int4 w;
                w.x = w.y = w.z = w.w = -1;
                if (s0 + SELL_C32_GROUP * q < ml) {
                    typedef int i4v __attribute__((ext_vector_type(4)));
                    const i4v tw = STREAM ? __builtin_nontemporal_load(reinterpret_cast<const i4v *>(cw + (long)(g + q) * BLOCK))
                                          : *reinterpret_cast<const i4v *>(cw + (long)(g + q) * BLOCK);
                    w.x = tw.x;
                    w.y = tw.y;
                    w.z = tw.z;
                    w.w = tw.w;
                }
                a0[2 * q] = w.x;
                a1[2 * q] = w.y;
                a0[2 * q + 1] = (s0 + 2 * q + 1 < ml) ? w.z : -1;
                a1[2 * q + 1] = (s0 + 2 * q + 1 < ml) ? w.w : -1;
            }
            double2 vv[BATCH];
#pragma unroll
            for (int k = 0; k < BATCH; ++k) {
                vv[k].x = vv[k].y = 0.0;
                if (s0 + k < ml) vv[k] = STREAM ? ld_pair_stream(v + (long)(s0 + k) * CHUNK_ROWS) : *reinterpret_cast<const double2 *>(v + (long)(s0 + k) * CHUNK_ROWS);
            }
            double x0[BATCH], x1[BATCH];
#pragma unroll
            for (int k = 0; k < BATCH; ++k) {
                x0[k] = a0[k] >= 0 ? x[a0[k]] : 0.0;
                x1[k] = a1[k] >= 0 ? x[a1[k]] : 0.0;
            }
#pragma unroll
            for (int k = 0; k < BATCH; ++k) {
                if (a0[k] >= 0) {
                    const double p = vv[k].x * x0[k];
                    acc.x = (MODE == SPMV_RESIDUAL) ? acc.x - p : acc.x + p;
                }
                if (a1[k] >= 0) {
                    const double p = vv[k].y * x1[k];
                    acc.y = (MODE == SPMV_RESIDUAL) ? acc.y - p : acc.y + p;
                }
            }
        }
    } else if (h.mode() == SELL_MODE_PATTERN) {
        // pattern mode: one byte per row -> `width` offsets of the row in the LDS table
        const unsigned short pp =
            *reinterpret_cast<const unsigned short *>(codes + h.code_off + t * ROWS_PER_THREAD);
        const int p0 = (int)(pp & 0xffu) * width, p1 = (int)(pp >> 8) * width;
        for (int s0 = 0; s0 < ww; s0 += BATCH) {
            double2 vv[BATCH];
            int d0[BATCH], d1[BATCH];
#pragma unroll
            for (int k = 0; k < BATCH; ++k) {
                const int s = min(s0 + k, ww - 1);  // clamp: always a valid plane
                vv[k] = STREAM ? ld_pair_stream(v + (long)s * CHUNK_ROWS) : *reinterpret_cast<const double2 *>(v + (long)s * CHUNK_ROWS);
                d0[k] = (s0 + k < ww) ? stab[p0 + s] : SELL_PAD_OFFSET;
                d1[k] = (s0 + k < ww) ? stab[p1 + s] : SELL_PAD_OFFSET;
            }
            double x0[BATCH], x1[BATCH];
#pragma unroll
            for (int k = 0; k < BATCH; ++k) {
                x0[k] = d0[k] != SELL_PAD_OFFSET ? x[row + d0[k]] : 0.0;
                x1[k] = d1[k] != SELL_PAD_OFFSET ? x[row + 1 + d1[k]] : 0.0;
            }
#pragma unroll
            for (int k = 0; k < BATCH; ++k) {
                if (d0[k] != SELL_PAD_OFFSET) {
                    const double p = vv[k].x * x0[k];
                    acc.x = (MODE == SPMV_RESIDUAL) ? acc.x - p : acc.x + p;
                }
                if (d1[k] != SELL_PAD_OFFSET) {
                    const double p = vv[k].y * x1[k];
                    acc.y = (MODE == SPMV_RESIDUAL) ? acc.y - p : acc.y + p;
                }
            }
        }
    } else {
        // offset mode: one byte per (row, slot) -> entry of the chunk's offset dictionary
        const uint8_t *c = codes + h.code_off + (long)t * h.code_stride();
        for (int s0 = 0; s0 < ww; s0 += BATCH) {
            uint4 cw;
            cw.x = cw.y = cw.z = cw.w = 0xffffffffu;
            if (s0 < ml) cw = *reinterpret_cast<const uint4 *>(c + 2 * s0);
            const unsigned w4[4] = {cw.x, cw.y, cw.z, cw.w};
            double2 vv[BATCH];
#pragma unroll
            for (int k = 0; k < BATCH; ++k) {
                vv[k].x = vv[k].y = 0.0;
                if (s0 + k < ml) vv[k] = STREAM ? ld_pair_stream(v + (long)(s0 + k) * CHUNK_ROWS) : *reinterpret_cast<const double2 *>(v + (long)(s0 + k) * CHUNK_ROWS);
            }
            double x0[BATCH], x1[BATCH];
            bool ok0[BATCH], ok1[BATCH];
#pragma unroll
            for (int k = 0; k < BATCH; ++k) {
                const unsigned pair = (w4[k / 2] >> (16 * (k & 1))) & 0xffffu;
                const unsigned c0 = pair & 0xffu, c1 = pair >> 8;
                // padding slots carry code 255; rows past n_rows only have padding slots
                ok0[k] = (s0 + k < ml) && c0 != 255u;
                ok1[k] = (s0 + k < ml) && c1 != 255u;
                x0[k] = ok0[k] ? x[row + stab[c0]] : 0.0;
                x1[k] = ok1[k] ? x[row + 1 + stab[c1]] : 0.0;
            }
#pragma unroll
            for (int k = 0; k < BATCH; ++k) {
                if (ok0[k]) {
                    const double p = vv[k].x * x0[k];
                    acc.x = (MODE == SPMV_RESIDUAL) ? acc.x - p : acc.x + p;
                }
                if (ok1[k]) {
                    const double p = vv[k].y * x1[k];
                    acc.y = (MODE == SPMV_RESIDUAL) ? acc.y - p : acc.y + p;
                }
            }
        }
    }
    // spill: the tails of this chunk's long rows (beyond the chunk's cap).  The row sums go through LDS
    // to the threads that walk the tails -- one row each, entries in stored order, so every row is still
    // summed left to right -- and back.  Workgroup-uniform branch; chunks without long rows skip it.
    if (sp1 > sp0) {
        double *ys = reinterpret_cast<double *>(stab);  // the table is not needed any more
        __syncthreads();
        ys[ROWS_PER_THREAD * t] = acc.x;
        ys[ROWS_PER_THREAD * t + 1] = acc.y;
        __syncthreads();
        for (int j = sp0 + t; j < sp1; j += BLOCK) {
            if (j != sp0 + t) {  // (more than BLOCK long rows in one chunk: the later ones were not prefetched)
                s_row = spill_rows[j];
                s_k0 = spill_ptrs[j];
                s_k1 = spill_ptrs[j + 1];
            }
            const int li = s_row - chunk * CHUNK_ROWS;
            double a = ys[li];
            constexpr int SB = 4;  // values, columns and x of SB entries in flight; the adds stay in order
            for (int k0 = s_k0; k0 < s_k1; k0 += SB) {
                double sv[SB], sx[SB];
                int sc[SB];
#pragma unroll
                for (int i = 0; i < SB; ++i) {
                    const int k = min(k0 + i, s_k1 - 1);
                    sv[i] = spill_vals[k];
                    sc[i] = spill_cols[k];
                }
#pragma unroll
                for (int i = 0; i < SB; ++i) sx[i] = x[sc[i]];
#pragma unroll
                for (int i = 0; i < SB; ++i)
                    if (k0 + i < s_k1) {
                        const double p = sv[i] * sx[i];
                        a = (MODE == SPMV_RESIDUAL) ? a - p : a + p;
                    }
            }
            ys[li] = a;
        }
        __syncthreads();
        acc.x = ys[ROWS_PER_THREAD * t];
        acc.y = ys[ROWS_PER_THREAD * t + 1];
    }
    if (rmap) {  // the sums go to the rows they belong to (workgroup-uniform branch)
        double *ys = reinterpret_cast<double *>(stab);  // the table is not needed any more
        __syncthreads();
        ys[own & 0xffffu] = acc.x;
        ys[own >> 16] = acc.y;
        __syncthreads();
        acc.x = ys[ROWS_PER_THREAD * t];
        acc.y = ys[ROWS_PER_THREAD * t + 1];
    }
    if (hf.chunk_bptr) halo_fused_add<MODE>(hf, chunk, acc.x, acc.y, reinterpret_cast<double *>(stab));
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

__global__ __launch_bounds__(BLOCK) void k_gather_coeffs_masked(long n, const int *__restrict__ map,
                                                                const double *__restrict__ src,
                                                                double *__restrict__ out)
{
    const long i = ((long)blockIdx.x * BLOCK + threadIdx.x) * 2;
    if (i + 1 < n) {
        const int2 m = *reinterpret_cast<const int2 *>(map + i);
        double2 v;
        v.x = m.x >= 0 ? src[m.x] : 0.0;
        v.y = m.y >= 0 ? src[m.y] : 0.0;
        *reinterpret_cast<double2 *>(out + i) = v;
    } else if (i < n) {
        out[i] = map[i] >= 0 ? src[map[i]] : 0.0;
    }
}

// The same for the chunked layout (SellChunk): one workgroup fills all planes of its chunk, so the
// chunk's CSR value range (a few tens of KB) is fetched once into one XCD's L2 instead of once per
// plane (the flat kernel above amplified the reads 7x on the 7-point matrix).
__global__ __launch_bounds__(BLOCK) void k_gather_sell(int n_chunks, const SellChunk *__restrict__ chunks,
                                                       const int *__restrict__ map,
                                                       const double *__restrict__ src,
                                                       double *__restrict__ out)
{
    const int chunk = blockIdx.x;
    if (chunk >= n_chunks) return;
    const SellChunk h = chunks[chunk];
    const int width = h.width();
    for (int s = 0; s < width; ++s) {
        const long i = h.val_off + (long)s * CHUNK_ROWS + threadIdx.x * ROWS_PER_THREAD;
        const int2 m = *reinterpret_cast<const int2 *>(map + i);
        double2 v;
        v.x = m.x >= 0 ? src[m.x] : 0.0;
        v.y = m.y >= 0 ? src[m.y] : 0.0;
        *reinterpret_cast<double2 *>(out + i) = v;
    }
}

// ------------------------------------------------------------------------------------------
// BiCGStab steps ([UPSTREAM] bicgstab::step_1 / step_2 / step_3 / finalize)
// ------------------------------------------------------------------------------------------
// step_1: p = r + (rho/prev_rho * alpha/omega) (p - omega v)   [p = r when prev_rho*omega == 0];
// then y = M^-1 p (scalar Jacobi; with the identity y aliases p and is not written)
__global__ __launch_bounds__(BLOCK) void k_bicg_step1(int n, double *__restrict__ p,
                                                      const double *__restrict__ r,
                                                      const double *__restrict__ v,
                                                      const double *__restrict__ inv_diag,
                                                      double *__restrict__ y, const DevScalars *s)
{
    if (s->stop) return;
    const double rho = s->rho, prev = s->prev_rho, alpha = s->alpha, omega = s->omega;
    const RowPair rp = my_rows(blockIdx.x, n);
    // r, p, v, inv_diag: a whole SpMV passes before any of them is touched again -> streamed past the caches;
    // y is gathered by the SpMV that follows and stays cached
    const double2 vr = ld2_stream(r, rp);
    double2 vp = vr;
    if (prev * omega != 0.0) {
        const double tmp = rho / prev * alpha / omega;
        const double2 po = ld2_stream(p, rp), vv = ld2_stream(v, rp);
        vp.x = vr.x + tmp * (po.x - omega * vv.x);
        vp.y = vr.y + tmp * (po.y - omega * vv.y);
    }
    if (inv_diag) st2_stream(p, rp, vp); else st2(p, rp, vp);  // (without a preconditioner p itself is the SpMV's input)
    if (inv_diag) {
        const double2 vi = ld2_stream(inv_diag, rp);
        double2 vy;
        vy.x = vp.x * vi.x;
        vy.y = vp.y * vi.y;
        st2(y, rp, vy);
    }
}

// step_2: s = r - alpha v (alpha = rho/beta from the finaliser; s = r when beta == 0); z = M^-1 s;
// partial of sum|s| for the mid-turn criterion check
__global__ __launch_bounds__(BLOCK) void k_bicg_step2(int n, const double *__restrict__ r,
                                                      const double *__restrict__ v,
                                                      double *__restrict__ sv,
                                                      const double *__restrict__ inv_diag,
                                                      double *__restrict__ z,
                                                      double *__restrict__ part_norm,
                                                      const DevScalars *s)
{
    __shared__ double slot[N_WAVES];
    if (s->stop) return;
    const double alpha = s->alpha, beta = s->beta;
    const int chunk = blockIdx.x;
    const RowPair rp = my_rows(chunk, n);
    double2 vs = ld2_stream(r, rp);  // (as in step_1: r, v and inv_diag are not touched again before an SpMV has passed)
    if (beta != 0.0) {
        const double2 vv = ld2_stream(v, rp);
        vs.x = vs.x - alpha * vv.x;
        vs.y = vs.y - alpha * vv.y;
    }
    st2(sv, rp, vs);  // (s is read by the SpMV that follows, for the fused t.s: stays cached)
    if (inv_diag) {
        const double2 vi = ld2_stream(inv_diag, rp);
        double2 vz;
        vz.x = vs.x * vi.x;
        vz.y = vs.y * vi.y;
        st2(z, rp, vz);
    }
    double a = 0.0;
    if (rp.n > 0) a += fabs(vs.x);
    if (rp.n > 1) a += fabs(vs.y);
    const double s1 = block_sum(a, slot);
    if (threadIdx.x == 0) part_norm[chunk] = s1;
}

// step_3: x += alpha y + omega z ; r = s - omega t ; then the partials of the next turn's
// rho = rr.r and of sum|r|
__global__ __launch_bounds__(BLOCK) void k_bicg_step3(int n, double *__restrict__ x,
                                                      double *__restrict__ r,
                                                      const double *__restrict__ sv,
                                                      const double *__restrict__ t,
                                                      const double *__restrict__ y,
                                                      const double *__restrict__ z,
                                                      const double *__restrict__ rr,
                                                      double *__restrict__ part_rho,
                                                      double *__restrict__ part_norm,
                                                      const DevScalars *s, int turn)
{
    __shared__ double slot[N_WAVES];
    if (s->stop) {
        // bicgstab::finalize: x += alpha y, only on the turn whose mid-step check stopped the solver
        if (s->stop_phase == 1 && s->stop_turn == turn) {
            const double alpha = s->alpha;
            const RowPair rp = my_rows(blockIdx.x, n);
            double2 vx = ld2(x, rp);
            const double2 vy = ld2(y, rp);
            vx.x += alpha * vy.x;
            vx.y += alpha * vy.y;
            st2(x, rp, vx);
        }
        return;
    }
    const double alpha = s->alpha, omega = s->omega;
    const int chunk = blockIdx.x;
    const RowPair rp = my_rows(chunk, n);
    // x, y, z, s, t, rr: touched here for the last (x, rr: only) time of the turn -> streamed past the caches
    double2 vx = ld2_stream(x, rp);
    const double2 vy = ld2_stream(y, rp), vz = ld2_stream(z, rp), vs = ld2_stream(sv, rp), vt = ld2_stream(t, rp),
                  vrr = ld2_stream(rr, rp);
    vx.x += alpha * vy.x + omega * vz.x;
    vx.y += alpha * vy.y + omega * vz.y;
    double2 vr;
    vr.x = vs.x - omega * vt.x;
    vr.y = vs.y - omega * vt.y;
    st2_stream(x, rp, vx);
    st2(r, rp, vr);
    double d = 0.0, a = 0.0;
    if (rp.n > 0) {
        d += vrr.x * vr.x;
        a += fabs(vr.x);
    }
    if (rp.n > 1) {
        d += vrr.y * vr.y;
        a += fabs(vr.y);
    }
    const double s0 = block_sum(d, slot);
    const double s1 = block_sum(a, slot);
    if (threadIdx.x == 0) {
        part_rho[chunk] = s0;
        part_norm[chunk] = s1;
    }
}

// ------------------------------------------------------------------------------------------
// GKOBiCGStab on small systems (<= FUSED_FIN_MAX_CHUNKS chunks, one rank): the three single-workgroup finalisers of
// a turn folded into the step kernels that consume their results, as for GKOCG (k_cg_step1x_fin): every workgroup
// reduces the per-chunk partials itself -- 256 threads walking the 1024-thread tree of k_finalize, same bits -- and runs
// the scalar logic on its own copy of the scalars; workgroup 0 stores them.  The scalars ping-pong between two slots
// (a kernel reads `sin`, writes `sout`).  Turn:
//   [check of the previous turn + step_1] -> (M^-1) -> SpMV -> [alpha + step_2] -> (M^-1) -> SpMV
//   -> [mid-turn check + omega + step_3 (or bicgstab::finalize when that check stops the solve)]
// 5 launches instead of 8 (+ the preconditioner's own).  A kernel that reads partials never writes the arrays it
// reads -- another workgroup may still be reducing them -- so the turn uses six partial arrays.
// ------------------------------------------------------------------------------------------
// FIN_CG_CHECK + step_1.  The closing check of a solve is one more launch of this kernel (the step it then takes on
// p is harmless: the solve has stopped, or fails with "did not stop").
__global__ __launch_bounds__(BLOCK) void k_bicg_fold1(int n, double *__restrict__ p, const double *__restrict__ r,
                                                      const double *__restrict__ v,
                                                      const double *__restrict__ inv_diag, double *__restrict__ y,
                                                      const DevScalars *sin, DevScalars *sout,
                                                      const double *__restrict__ part_rho,
                                                      const double *__restrict__ part_norm, int n_part,
                                                      double *history)
{
    __shared__ double red[2 * FIN_WAVES];
    __shared__ double sh[2];
    __shared__ int sh_stop;
    // (everything asked for at once, the scalars field by field: see k_cg_step1x_fin)
    const int stopped = sin->stop;
    const double s_rho = sin->rho, alpha = sin->alpha, omega = sin->omega, s_nf = sin->norm_factor,
                 s_init = sin->init_res;
    const int s_iter = sin->iter, s_evals = sin->n_evals;
    const double c_tol = sin->crit.tolerance, c_rel = sin->crit.rel_tol;
    const int c_min = sin->crit.min_iter, c_max = sin->crit.max_iter, c_freq = sin->crit.frequency,
              c_exp = sin->crit.export_res;
    if (blockIdx.x == 0 && threadIdx.x < sizeof(DevScalars) / 8)  // fields this kernel leaves alone
        reinterpret_cast<unsigned long long *>(sout)[threadIdx.x] =
            reinterpret_cast<const unsigned long long *>(sin)[threadIdx.x];
    const RowPair rp = my_rows(blockIdx.x, n);
    const double2 vr = ld2_stream(r, rp);
    const double2 po = ld2_stream(p, rp), vv = ld2_stream(v, rp);
    double2 vi;
    vi.x = vi.y = 1.0;
    if (inv_diag) vi = ld2_stream(inv_diag, rp);
    double pv[2][FIN_VT];
    load_partials_as_finaliser<2>(part_rho, part_norm, n_part, pv);
    if (stopped) return;
    double vsum[2];
    reduce_partials_as_finaliser<2>(pv, n_part, red, vsum);
    if (threadIdx.x == 0) {
        // FIN_CG_CHECK: swap(prev_rho, rho) of the previous turn, then criterion_check (StoppingCriterion.C:71-151)
        const double prev_rho = s_rho, rho = vsum[0];
        int iter = s_iter, n_evals = s_evals, stop = 0;
        double init_res = s_init, res = 0.0;
        bool evaluated = false;
        if (iter > 0 && iter < c_min) {           // :77-81
            iter += 1;
        } else if (iter % c_freq != 0) {          // :84-87
            iter += 1;
        } else {
            evaluated = true;
            n_evals += 1;
            res = vsum[1];
            if (iter == 0) init_res = res / s_nf;  // :102-111
            res /= s_nf;                           // :113
            if (c_exp && history && blockIdx.x == 0) history[iter] = res;  // :115-117
            if (iter >= c_max) stop = 1;                                   // :124
            if (res < c_tol) stop = 1;                                     // :128
            if (c_rel > 0 && res < c_rel * init_res) stop = 1;             // :132-136
            iter += 1;                                                     // :143
        }
        sh[0] = prev_rho;
        sh[1] = rho;
        sh_stop = stop;
        if (blockIdx.x == 0) {
            sout->prev_rho = prev_rho;
            sout->rho = rho;
            sout->iter = iter;
            sout->x_pending = 0;
            if (evaluated) {
                sout->n_evals = n_evals;
                sout->init_res = init_res;
                sout->res = res;
            }
            if (stop) sout->stop = 1;
        }
    }
    __syncthreads();
    if (sh_stop) return;
    const double prev = sh[0], rho = sh[1];
    double2 vp = vr;
    if (prev * omega != 0.0) {  // step_1 (k_bicg_step1)
        const double tmp = rho / prev * alpha / omega;
        vp.x = vr.x + tmp * (po.x - omega * vv.x);
        vp.y = vr.y + tmp * (po.y - omega * vv.y);
    }
    if (inv_diag) st2_stream(p, rp, vp); else st2(p, rp, vp);
    if (inv_diag) {
        double2 vy;
        vy.x = vp.x * vi.x;
        vy.y = vp.y * vi.y;
        st2(y, rp, vy);
    }
}

// FIN_BICG_ALPHA + step_2
__global__ __launch_bounds__(BLOCK) void k_bicg_fold2(int n, const double *__restrict__ r,
                                                      const double *__restrict__ v, double *__restrict__ sv,
                                                      const double *__restrict__ inv_diag, double *__restrict__ z,
                                                      double *__restrict__ part_norm_out, const DevScalars *sin,
                                                      DevScalars *sout, const double *__restrict__ part_beta,
                                                      int n_part)
{
    __shared__ double red[2 * FIN_WAVES];
    __shared__ double sh[2];
    __shared__ double slot[N_WAVES];
    const int stopped = sin->stop;
    const double s_rho = sin->rho;
    if (blockIdx.x == 0 && threadIdx.x < sizeof(DevScalars) / 8)
        reinterpret_cast<unsigned long long *>(sout)[threadIdx.x] =
            reinterpret_cast<const unsigned long long *>(sin)[threadIdx.x];
    const int chunk = blockIdx.x;
    const RowPair rp = my_rows(chunk, n);
    double2 vs = ld2_stream(r, rp);
    const double2 vv = ld2_stream(v, rp);
    double2 vi;
    vi.x = vi.y = 1.0;
    if (inv_diag) vi = ld2_stream(inv_diag, rp);
    double pv[2][FIN_VT];
    load_partials_as_finaliser<1>(part_beta, nullptr, n_part, pv);
    if (stopped) return;
    double vsum[2];
    reduce_partials_as_finaliser<1>(pv, n_part, red, vsum);
    if (threadIdx.x == 0) {  // beta = rr.v ; alpha = rho / beta (0 when beta == 0)
        const double beta = vsum[0], alpha = (beta != 0.0) ? s_rho / beta : 0.0;
        sh[0] = alpha;
        sh[1] = beta;
        if (blockIdx.x == 0) {
            sout->beta = beta;
            sout->alpha = alpha;
        }
    }
    __syncthreads();
    const double alpha = sh[0], beta = sh[1];
    if (beta != 0.0) {  // step_2 (k_bicg_step2)
        vs.x = vs.x - alpha * vv.x;
        vs.y = vs.y - alpha * vv.y;
    }
    st2(sv, rp, vs);
    if (inv_diag) {
        double2 vz;
        vz.x = vs.x * vi.x;
        vz.y = vs.y * vi.y;
        st2(z, rp, vz);
    }
    double a = 0.0;
    if (rp.n > 0) a += fabs(vs.x);
    if (rp.n > 1) a += fabs(vs.y);
    const double s1 = block_sum(a, slot);
    if (threadIdx.x == 0) part_norm_out[chunk] = s1;
}

// FIN_BICG_CHECK2_OMEGA + step_3 (bicgstab::finalize, x += alpha y, when the mid-turn check stops the solve)
__global__ __launch_bounds__(BLOCK) void k_bicg_fold3(int n, double *__restrict__ x, double *__restrict__ r,
                                                      const double *__restrict__ sv, const double *__restrict__ t,
                                                      const double *__restrict__ y, const double *__restrict__ z,
                                                      const double *__restrict__ rr,
                                                      double *__restrict__ part_rho_out,
                                                      double *__restrict__ part_norm_out, const DevScalars *sin,
                                                      DevScalars *sout, const double *__restrict__ part_gamma,
                                                      const double *__restrict__ part_tt,
                                                      const double *__restrict__ part_snorm, int n_part,
                                                      double *history, int turn)
{
    __shared__ double red[2 * FIN_WAVES];
    __shared__ double sh[1];
    __shared__ int sh_stop;
    __shared__ double slot[2 * N_WAVES];
    const int stopped = sin->stop;
    const double alpha = sin->alpha, s_nf = sin->norm_factor, s_init = sin->init_res;
    const int s_iter = sin->iter, s_evals = sin->n_evals;
    const double c_tol = sin->crit.tolerance, c_rel = sin->crit.rel_tol;
    const int c_min = sin->crit.min_iter, c_max = sin->crit.max_iter, c_freq = sin->crit.frequency,
              c_exp = sin->crit.export_res;
    if (blockIdx.x == 0 && threadIdx.x < sizeof(DevScalars) / 8)
        reinterpret_cast<unsigned long long *>(sout)[threadIdx.x] =
            reinterpret_cast<const unsigned long long *>(sin)[threadIdx.x];
    const int chunk = blockIdx.x;
    const RowPair rp = my_rows(chunk, n);
    double2 vx = ld2_stream(x, rp);
    const double2 vy = ld2_stream(y, rp), vz = ld2_stream(z, rp), vs = ld2_stream(sv, rp), vt = ld2_stream(t, rp),
                  vrr = ld2_stream(rr, rp);
    double pv[2][FIN_VT], pn[2][FIN_VT];
    load_partials_as_finaliser<2>(part_gamma, part_tt, n_part, pv);
    load_partials_as_finaliser<1>(part_snorm, nullptr, n_part, pn);
    if (stopped) return;
    double vsum[2], vnorm[2];
    reduce_partials_as_finaliser<2>(pv, n_part, red, vsum);
    reduce_partials_as_finaliser<1>(pn, n_part, red, vnorm);
    if (threadIdx.x == 0) {
        // the mid-turn check on s (criterion_check, StoppingCriterion.C:71-151), then gamma = s.t, beta = t.t,
        // omega = gamma / beta unless it stopped
        int iter = s_iter, n_evals = s_evals, stop = 0;
        double init_res = s_init, res = 0.0;
        bool evaluated = false;
        if (iter > 0 && iter < c_min) {
            iter += 1;
        } else if (iter % c_freq != 0) {
            iter += 1;
        } else {
            evaluated = true;
            n_evals += 1;
            res = vnorm[0];
            if (iter == 0) init_res = res / s_nf;
            res /= s_nf;
            if (c_exp && history && blockIdx.x == 0) history[iter] = res;
            if (iter >= c_max) stop = 1;
            if (res < c_tol) stop = 1;
            if (c_rel > 0 && res < c_rel * init_res) stop = 1;
            iter += 1;
        }
        const double omega = (vsum[1] != 0.0) ? vsum[0] / vsum[1] : 0.0;
        sh[0] = omega;
        sh_stop = stop;
        if (blockIdx.x == 0) {
            sout->iter = iter;
            if (evaluated) {
                sout->n_evals = n_evals;
                sout->init_res = init_res;
                sout->res = res;
            }
            if (stop) {
                sout->stop = 1;
                sout->stop_phase = 1;
                sout->stop_turn = turn;
            } else {
                sout->gamma = vsum[0];
                sout->beta = vsum[1];
                sout->omega = omega;
            }
        }
    }
    __syncthreads();
    if (sh_stop) {  // bicgstab::finalize: x += alpha y
        vx.x += alpha * vy.x;
        vx.y += alpha * vy.y;
        st2(x, rp, vx);
        return;
    }
    const double omega = sh[0];
    vx.x += alpha * vy.x + omega * vz.x;  // step_3 (k_bicg_step3)
    vx.y += alpha * vy.y + omega * vz.y;
    double2 vr;
    vr.x = vs.x - omega * vt.x;
    vr.y = vs.y - omega * vt.y;
    st2_stream(x, rp, vx);
    st2(r, rp, vr);
    double d = 0.0, a = 0.0;
    if (rp.n > 0) {
        d += vrr.x * vr.x;
        a += fabs(vr.x);
    }
    if (rp.n > 1) {
        d += vrr.y * vr.y;
        a += fabs(vr.y);
    }
    block_sum2(d, a, slot);
    if (threadIdx.x == 0) {
        part_rho_out[chunk] = d;
        part_norm_out[chunk] = a;
    }
}

// ------------------------------------------------------------------------------------------
// GMRES vector kernels
// ------------------------------------------------------------------------------------------
__global__ __launch_bounds__(BLOCK) void k_gmres_scale(int n, double *__restrict__ out,
                                                       const double *__restrict__ in,
                                                       const double *__restrict__ denom,
                                                       const DevScalars *gate)
{
    if (gate && gate->stop) return;
    const double d = *denom;
    const RowPair rp = my_rows(blockIdx.x, n);
    double2 v = ld2(in, rp);
    v.x = v.x / d;
    v.y = v.y / d;
    st2(out, rp, v);
}

__global__ __launch_bounds__(BLOCK) void k_gmres_mgs(int n, double *__restrict__ w,
                                                     const double *__restrict__ vprev,
                                                     const double *__restrict__ hprev,
                                                     const double *__restrict__ vdot,
                                                     double *__restrict__ part,
                                                     const DevScalars *gate)
{
    __shared__ double slot[N_WAVES];
    if (gate && gate->stop) return;
    const int chunk = blockIdx.x;
    const RowPair rp = my_rows(chunk, n);
    double2 vw = ld2(w, rp);
    if (vprev) {
        const double h = *hprev;
        const double2 vp = ld2(vprev, rp);
        vw.x -= h * vp.x;
        vw.y -= h * vp.y;
        st2(w, rp, vw);
    }
    const double2 vd = vdot ? ld2(vdot, rp) : vw;
    double d = 0.0;
    if (rp.n > 0) d += vw.x * vd.x;
    if (rp.n > 1) d += vw.y * vd.y;
    const double s0 = block_sum(d, slot);
    if (threadIdx.x == 0) part[chunk] = s0;
}

// Small single-rank systems (<= FUSED_FIN_MAX_CHUNKS chunks): the finaliser between two Gram-Schmidt links (FIN_GMRES_H:
// H(k, it) = sum of the link's partials) folded into the next link's kernel -- every workgroup reduces the partials
// itself in the finaliser's order (same bits), workgroup 0 stores H(k, it).  One launch per link instead of two; the
// link reads `part_in` and writes `part_out` (never the same array: another workgroup may still be reducing).
__global__ __launch_bounds__(BLOCK) void k_gmres_mgs_fold(int n, double *__restrict__ w,
                                                          const double *__restrict__ vprev,
                                                          double *__restrict__ h_out,
                                                          const double *__restrict__ vdot,
                                                          const double *__restrict__ part_in, int n_part,
                                                          double *__restrict__ part_out, const DevScalars *gate)
{
    __shared__ double red[2 * FIN_WAVES];
    __shared__ double sh_h;
    __shared__ double slot[N_WAVES];
    if (gate && gate->stop) return;
    const int chunk = blockIdx.x;
    const RowPair rp = my_rows(chunk, n);
    double2 vw = ld2(w, rp);
    if (vprev) {
        const double2 vp = ld2(vprev, rp);
        double pv[2][FIN_VT], v[2];
        load_partials_as_finaliser<1>(part_in, nullptr, n_part, pv);
        reduce_partials_as_finaliser<1>(pv, n_part, red, v);
        if (threadIdx.x == 0) {
            sh_h = v[0];
            if (blockIdx.x == 0) *h_out = v[0];  // FIN_GMRES_H
        }
        __syncthreads();
        const double h = sh_h;
        vw.x -= h * vp.x;
        vw.y -= h * vp.y;
        st2(w, rp, vw);
    }
    const double2 vd = vdot ? ld2(vdot, rp) : vw;
    double d = 0.0;
    if (rp.n > 0) d += vw.x * vd.x;
    if (rp.n > 1) d += vw.y * vd.y;
    const double s0 = block_sum(d, slot);
    if (threadIdx.x == 0) part_out[chunk] = s0;
}

__global__ __launch_bounds__(BLOCK) void k_gmres_update_x(int n, const double *__restrict__ V,
                                                          long ld, const double *__restrict__ y,
                                                          int it, const double *__restrict__ inv_diag,
                                                          double *__restrict__ x,
                                                          double *__restrict__ before,
                                                          const DevScalars *gate)
{
    if (gate && gate->stop) return;
    const RowPair rp = my_rows(blockIdx.x, n);
    double2 sum;
    sum.x = 0.0;
    sum.y = 0.0;
    for (int j = 0; j < it; ++j) {
        const double yj = y[j];
        const double2 v = ld2(V + (size_t)j * ld, rp);
        sum.x += v.x * yj;
        sum.y += v.y * yj;
    }
    if (before) {
        st2(before, rp, sum);
        return;
    }
    if (inv_diag) {
        const double2 vi = ld2(inv_diag, rp);
        sum.x = sum.x * vi.x;
        sum.y = sum.y * vi.y;
    }
    double2 vx = ld2(x, rp);
    vx.x += sum.x;
    vx.y += sum.y;
    st2(x, rp, vx);
}

__global__ __launch_bounds__(BLOCK) void k_mul(int n, double *__restrict__ out,
                                               const double *__restrict__ in,
                                               const double *__restrict__ inv_diag,
                                               const DevScalars *gate)
{
    if (gate && gate->stop) return;
    const RowPair rp = my_rows(blockIdx.x, n);
    double2 v = ld2(in, rp);
    const double2 vi = ld2(inv_diag, rp);
    v.x = v.x * vi.x;
    v.y = v.y * vi.y;
    st2(out, rp, v);
}

__global__ __launch_bounds__(BLOCK) void k_add(int n, double *__restrict__ x,
                                               const double *__restrict__ a, const DevScalars *gate)
{
    if (gate && gate->stop) return;
    const RowPair rp = my_rows(blockIdx.x, n);
    double2 vx = ld2(x, rp);
    const double2 va = ld2(a, rp);
    vx.x += va.x;
    vx.y += va.y;
    st2(x, rp, vx);
}

// GMRES dense-state accessors (layout: kernels.hpp gmres_state_len)
struct GmresState {
    double *H, *gs, *gc, *rnc, *y;
    int m;
    __device__ GmresState(double *base, int m_) : m(m_)
    {
        H = base;
        gs = H + (size_t)(m + 1) * m;
        gc = gs + m;
        rnc = gc + m;
        y = rnc + (m + 1);
    }
    __device__ double &h(int i, int j) const { return H[(size_t)j * (m + 1) + i]; }
};

// ------------------------------------------------------------------------------------------
// finalisers (one workgroup): reduce the per-chunk partials, then the scalar logic
// ------------------------------------------------------------------------------------------

// StoppingCriterion.C:71-151 on the device.  `norm` is sum|r| over all ranks.
__device__ void criterion_check(DevScalars *s, const DevCriterion &c, double norm, double *history)
{
    const int iter = s->iter;
    if (iter > 0 && iter < c.min_iter) {  // :77-81
        s->iter = iter + 1;
        return;
    }
    if (iter % c.frequency != 0) {  // :84-87
        s->iter = iter + 1;
        return;
    }
    s->n_evals += 1;
    double res = norm;
    if (iter == 0) s->init_res = res / s->norm_factor;  // :102-111 (norm_factor set before)
    res /= s->norm_factor;                              // :113
    if (c.export_res && history) history[iter] = res;   // :115-117
    s->res = res;                                       // :119
    bool stop = false;
    if (iter >= c.max_iter) stop = true;                                  // :124
    if (res < c.tolerance) stop = true;                                   // :128
    if (c.rel_tol > 0 && res < c.rel_tol * s->init_res) stop = true;      // :132-136
    s->iter = iter + 1;                                                   // :143
    if (stop) s->stop = 1;
}

// Peer-write all-reduce (PeerArgs, kernels.hpp).  Called by every thread of a workgroup of >= 64
// threads; v0, v1 are thread 0's local sums on entry and the rank-ordered global sums on return
// (thread 0 only).  Lane (q, e) sends half-word e to rank q and waits for rank q's half-word e.
__device__ __forceinline__ size_t peer_word(int slot, int src, int e)
{
    return ((size_t)slot * PEER_MAX_RANKS + src) * PEER_ELEMS + e;
}
__device__ bool peer_allreduce2(const PeerArgs &pa, double &v0, double &v1, unsigned *waited_out = nullptr)
{
    __shared__ unsigned halves[PEER_MAX_RANKS * PEER_ELEMS];
    __shared__ double mine[2];
    __shared__ int timed_out;
    __shared__ unsigned waited;  // the longest mailbox wait of this all-reduce (DevScalars::reduce_wait_ticks)
    if (threadIdx.x == 0) {
        mine[0] = v0;
        mine[1] = v1;
        timed_out = 0;
        waited = 0;
    }
    __syncthreads();
    const int t = threadIdx.x;
    if (t < pa.world * PEER_ELEMS) {
        const int q = t / PEER_ELEMS, e = t % PEER_ELEMS;
        const unsigned long long bits = (unsigned long long)__double_as_longlong(mine[e >> 1]);
        const unsigned half = (e & 1) ? (unsigned)(bits >> 32) : (unsigned)bits;
        const unsigned long long word = ((unsigned long long)pa.seq << 32) | half;
        const int slot = (int)(pa.seq % PEER_SLOTS);
        __hip_atomic_store(pa.box[q] + peer_word(slot, pa.rank, e), word, __ATOMIC_RELAXED,
                           __HIP_MEMORY_SCOPE_SYSTEM);
        const unsigned long long *src = pa.box[pa.rank] + peer_word(slot, q, e);
        const long long t0 = wall_clock64();
        unsigned long long w;
        for (;;) {
            w = __hip_atomic_load(src, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
            if ((unsigned)(w >> 32) == pa.seq) break;
            if (wall_clock64() - t0 > pa.timeout_ticks) {
                timed_out = 1;
                break;
            }
            __builtin_amdgcn_s_sleep(2);
        }
        note_wait(&waited, t0);
        halves[q * PEER_ELEMS + e] = (unsigned)w;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        if (waited_out) *waited_out = waited;
        double s0 = 0.0, s1 = 0.0;
        for (int q = 0; q < pa.world; ++q) {
            const unsigned *h = halves + q * PEER_ELEMS;
            s0 += __longlong_as_double((long long)(((unsigned long long)h[1] << 32) | h[0]));
            s1 += __longlong_as_double((long long)(((unsigned long long)h[3] << 32) | h[2]));
        }
        v0 = s0;
        v1 = s1;
    }
    return timed_out == 0;
}

__global__ __launch_bounds__(64) void k_peer_allreduce(PeerArgs pa, double *vals, int n, int32_t *error)
{
    double v0 = 0.0, v1 = 0.0;
    if (threadIdx.x == 0) {
        v0 = vals[0];
        if (n > 1) v1 = vals[1];
    }
    const bool ok = peer_allreduce2(pa, v0, v1);
    if (threadIdx.x == 0) {
        vals[0] = v0;
        if (n > 1) vals[1] = v1;
        if (!ok && error) *error = 1;
    }
}

// ---- peer-put halo exchange (PeerHalo, kernels.hpp) ----
__global__ __launch_bounds__(BLOCK) void k_pack_put(int n_send, const int *__restrict__ send_idxs,
                                                    PeerHalo P, const double *__restrict__ x,
                                                    const DevScalars *gate)
{
    if (gate && gate->stop) return;
    const int j = blockIdx.x * BLOCK + threadIdx.x;
    if (j >= n_send) return;
    int i = 0;
    while (i + 1 < P.n_neigh && j >= P.send_off[i + 1]) ++i;
    P.remote_recv[i][j - P.send_off[i]] = x[send_idxs[j]];
    __threadfence_system();  // the put has left this GPU before the kernel (and its signal) completes
}

// pack + signal in one launch: the last workgroup to finish (atomic ticket) stores the flags, after
// every workgroup's puts have been fenced at system scope.
__global__ __launch_bounds__(BLOCK) void k_pack_put_signal(int n_send,
                                                           const int *__restrict__ send_idxs,
                                                           PeerHalo P, const double *__restrict__ x,
                                                           const DevScalars *gate, unsigned *ticket)
{
    if (gate && gate->stop) return;
    const int j = blockIdx.x * BLOCK + threadIdx.x;
    if (j < n_send) {
        int i = 0;
        while (i + 1 < P.n_neigh && j >= P.send_off[i + 1]) ++i;
        P.remote_recv[i][j - P.send_off[i]] = x[send_idxs[j]];
    }
    __threadfence_system();
    __syncthreads();
    __shared__ int last;
    if (threadIdx.x == 0) {
        last = atomicAdd(ticket, 1u) == gridDim.x - 1;
        if (last) *ticket = 0;
    }
    __syncthreads();
    if (last && (int)threadIdx.x < P.n_neigh) {
        __threadfence_system();
        __hip_atomic_store(P.remote_flag[threadIdx.x], (unsigned long long)P.seq, __ATOMIC_RELEASE,
                           __HIP_MEMORY_SCOPE_SYSTEM);
    }
}

// wait + "y += A_non_local recv" + the dot partials of the touched chunks in one launch: one
// workgroup per chunk that holds boundary rows.  Same accumulation order as k_spmv_non_local and
// the same per-chunk tree as k_partials, so nothing changes in the bits.
template <int MODE, int NDOT>
__global__ __launch_bounds__(BLOCK) void k_halo_finish(int n_rows,
                                                       const int *__restrict__ chunk_list,
                                                       const int *__restrict__ chunk_row_ptr,
                                                       const int *__restrict__ boundary_rows,
                                                       const int *__restrict__ entry_ptrs,
                                                       const int *__restrict__ cols,
                                                       const double *__restrict__ vals,
                                                       const double *recv, double *y,
                                                       const double *w, double *part,
                                                       double *part_yy, PeerHalo P,
                                                       const DevScalars *gate, DevScalars *s)
{
    __shared__ double slot[N_WAVES];
    __shared__ int timed_out;
    __shared__ unsigned waited;
    if (gate && gate->stop) return;
    if (threadIdx.x == 0) {
        timed_out = 0;
        waited = 0;
    }
    __syncthreads();
    if ((int)threadIdx.x < P.n_neigh) {
        const long long t0 = wall_clock64();
        for (;;) {
            const unsigned long long f = __hip_atomic_load(P.local_flag + threadIdx.x, __ATOMIC_ACQUIRE,
                                                           __HIP_MEMORY_SCOPE_SYSTEM);
            if ((uint32_t)f == P.seq) break;
            if (wall_clock64() - t0 > P.timeout_ticks) {
                timed_out = 1;
                break;
            }
            __builtin_amdgcn_s_sleep(2);
        }
        note_wait(&waited, t0);
    }
    __syncthreads();
    if (threadIdx.x == 0) add_halo_wait(s, waited);
    if (timed_out) {  // a neighbour is gone: end the solve (y stays the local product)
        if (threadIdx.x == 0) {
            s->comm_error = 1;
            s->stop = 1;
        }
        return;
    }
    const int chunk = chunk_list[blockIdx.x];
    for (int i = chunk_row_ptr[blockIdx.x] + threadIdx.x; i < chunk_row_ptr[blockIdx.x + 1]; i += BLOCK) {
        const int row = boundary_rows[i];
        double acc = y[row];
        for (int k = entry_ptrs[i]; k < entry_ptrs[i + 1]; ++k) {
            const double t = vals[k] * recv[cols[k]];
            acc = (MODE == SPMV_RESIDUAL) ? acc - t : acc + t;
        }
        y[row] = acc;
    }
    if (NDOT >= 1) {
        __threadfence_block();
        __syncthreads();
        const RowPair rp = my_rows(chunk, n_rows);
        const double2 vy = ld2(y, rp), vw = ld2(w, rp);
        double d = 0.0, d2 = 0.0;
        if (rp.n > 0) {
            d += vw.x * vy.x;
            d2 += vy.x * vy.x;
        }
        if (rp.n > 1) {
            d += vw.y * vy.y;
            d2 += vy.y * vy.y;
        }
        const double sm = block_sum(d, slot);
        if (threadIdx.x == 0) part[chunk] = sm;
        if (NDOT >= 2) {
            const double s2 = block_sum(d2, slot);
            if (threadIdx.x == 0) part_yy[chunk] = s2;
        }
    }
}

__global__ __launch_bounds__(64) void k_halo_signal(PeerHalo P, const DevScalars *gate)
{
    if (gate && gate->stop) return;
    const int i = threadIdx.x;
    if (i < P.n_neigh)
        __hip_atomic_store(P.remote_flag[i], (unsigned long long)P.seq, __ATOMIC_RELEASE,
                           __HIP_MEMORY_SCOPE_SYSTEM);
}

__global__ __launch_bounds__(64) void k_halo_wait(PeerHalo P, const DevScalars *gate, DevScalars *s)
{
    __shared__ int timed_out;
    __shared__ unsigned waited;
    if (gate && gate->stop) return;
    if (threadIdx.x == 0) {
        timed_out = 0;
        waited = 0;
    }
    __syncthreads();
    const int i = threadIdx.x;
    if (i < P.n_neigh) {
        const long long t0 = wall_clock64();
        for (;;) {
            const unsigned long long w =
                __hip_atomic_load(P.local_flag + i, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_SYSTEM);
            if ((uint32_t)w == P.seq) break;
            if (wall_clock64() - t0 > P.timeout_ticks) {
                timed_out = 1;
                break;
            }
            __builtin_amdgcn_s_sleep(2);
        }
        note_wait(&waited, t0);
    }
    __syncthreads();
    if (threadIdx.x == 0) add_halo_wait(s, waited);
    if (threadIdx.x == 0 && timed_out) {  // a neighbour is gone: end the solve
        s->comm_error = 1;
        s->stop = 1;
    }
}

__global__ void k_peer_post(unsigned long long *dst, unsigned long long w0, unsigned long long w1,
                            unsigned long long w2, unsigned long long w3)
{
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    __hip_atomic_store(dst + 1, w1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    __hip_atomic_store(dst + 2, w2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    __hip_atomic_store(dst + 3, w3, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    __threadfence_system();
    __hip_atomic_store(dst, w0, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
}

template <int PHASE>
__global__ __launch_bounds__(FIN_BLOCK) void k_finalize(DevScalars *s, FinArgs a)
{
    __shared__ double slot[FIN_WAVES];
    if (PHASE != FIN_MEAN && PHASE != FIN_NORMFACTOR && PHASE != FIN_RAW &&
        PHASE != FIN_GMRES_SOLVE && s->stop) {
        // after a stop the one step_1x that followed has applied the pending x update
        if (PHASE == FIN_BETA && threadIdx.x == 0 && s->x_pending) s->x_pending = 0;
        return;
    }
    // thread 0 fetches the scalar block up front (its latency hides behind the partial loads),
    // does the logic in registers and stores the block once: the criterion's dependent global
    // round trips would otherwise cost more than the reduction itself
    DevScalars L;
    if (threadIdx.x == 0 && a.do_logic) L = *s;
    double v0 = 0.0, v1 = 0.0, v2 = 0.0;
    if (a.do_reduce) {
        const double *const parts[2] = {a.part[0], a.part[1]};
        double r[2];
        if (a.n_sums > 1)
            reduce_partials<2>(parts, a.n_part, slot, r);
        else
            reduce_partials<1>(parts, a.n_part, slot, r);
        v0 = r[0];
        v1 = r[1];
        if (PHASE == FIN_BICG_CHECK2_OMEGA) {  // (single rank: nothing of this goes through an all-reduce)
            const double *const extra[2] = {a.part_extra, nullptr};
            reduce_partials<1>(extra, a.n_part, slot, r);
            v2 = r[0];
        }
        if (PHASE == FIN_MEAN) {
            // distributed compute_mean [UPSTREAM]: local mean, weighted by n_local / n_global
            v0 /= a.n_local;
            v0 *= a.n_local / a.n_global;
        }
        if (threadIdx.x == 0 && !a.do_logic) {
            s->sums[0] = v0;
            s->sums[1] = v1;
        }
    }
    bool comm_ok = true;
    unsigned reduce_waited = 0;
    const bool peer_reduce = a.peer.world > 1 && a.do_reduce && a.do_logic;
    if (peer_reduce) comm_ok = peer_allreduce2(a.peer, v0, v1, &reduce_waited);
    if (!a.do_logic || threadIdx.x != 0) return;
    if (peer_reduce) {
        L.reduce_wait_ticks += reduce_waited;
        L.reduce_waits += 1;
    }
    if (!comm_ok) {  // a rank is gone: end the solve, the host reports OGL_ERR_COMM
        L.comm_error = 1;
        L.stop = 1;
    }
    if (!a.do_reduce) {
        v0 = L.sums[0];
        v1 = L.sums[1];
    } else {
        L.sums[0] = v0;
        L.sums[1] = v1;
    }
    if (PHASE == FIN_MEAN) {
        L.xbar = v0;
    } else if (PHASE == FIN_NORMFACTOR) {
        L.norm_factor = v0 + 1.0e-15;  // + SMALL, StoppingCriterion.C:68
    } else if (PHASE == FIN_CG_CHECK) {
        L.prev_rho = L.rho;  // swap(prev_rho, rho) of the previous turn
        L.rho = v0;
        criterion_check(&L, L.crit, v1, a.history);
        L.x_pending = a.turn ? 1 : 0;  // deferred-x path: step_2r's update waits for the next step_1x
    } else if (PHASE == FIN_BETA) {
        L.beta = v0;
        L.x_pending = 0;  // the step_1x before this SpMV has applied it
    } else if (PHASE == FIN_BICG_ALPHA) {  // beta = rr.v ; alpha = rho / beta (0 when beta == 0)
        L.beta = v0;
        L.alpha = (v0 != 0.0) ? L.rho / v0 : 0.0;
    } else if (PHASE == FIN_BICG_CHECK2) {  // mid-turn check on s
        criterion_check(&L, L.crit, v0, a.history);
        if (L.stop) {
            L.stop_phase = 1;
            L.stop_turn = a.turn;
        }
    } else if (PHASE == FIN_BICG_CHECK2_OMEGA) {  // the two phases around the second SpMV in one
        criterion_check(&L, L.crit, v2, a.history);
        if (L.stop) {
            L.stop_phase = 1;
            L.stop_turn = a.turn;
        } else {
            L.gamma = v0;
            L.beta = v1;
            L.omega = (v1 != 0.0) ? v0 / v1 : 0.0;
        }
    } else if (PHASE == FIN_BICG_OMEGA) {  // gamma = s.t ; beta = t.t ; omega = gamma / beta
        L.gamma = v0;
        L.beta = v1;
        L.omega = (v1 != 0.0) ? v0 / v1 : 0.0;
    } else if (PHASE == FIN_GMRES_RESTART) {  // gmres::restart
        GmresState g(a.gm, a.m);
        const double rn = sqrt(v0);
        g.rnc[0] = rn;
        L.beta = rn;  // V_0 = r / rn
        L.stale_norm = v1;
    } else if (PHASE == FIN_GMRES_H) {  // finish_arnoldi: H(k, it)
        GmresState g(a.gm, a.m);
        g.h(a.k, a.turn) = v0;
    } else if (PHASE == FIN_GMRES_COL) {  // norm of the new basis vector, then givens_rotation
        GmresState g(a.gm, a.m);
        const int it = a.turn;
        const double hn = sqrt(v0);
        g.h(it + 1, it) = hn;
        L.beta = hn;  // V_{it+1} /= hn
        for (int j = 0; j < it; ++j) {
            const double t = g.gc[j] * g.h(j, it) + g.gs[j] * g.h(j + 1, it);
            g.h(j + 1, it) = -g.gs[j] * g.h(j, it) + g.gc[j] * g.h(j + 1, it);
            g.h(j, it) = t;
        }
        if (g.h(it, it) == 0.0) {
            g.gc[it] = 0.0;
            g.gs[it] = 1.0;
        } else {
            const double scale = fabs(g.h(it, it)) + fabs(g.h(it + 1, it));
            const double a0 = g.h(it, it) / scale, a1 = g.h(it + 1, it) / scale;
            const double hyp = scale * sqrt(a0 * a0 + a1 * a1);
            g.gc[it] = g.h(it, it) / hyp;
            g.gs[it] = g.h(it + 1, it) / hyp;
        }
        g.h(it, it) = g.gc[it] * g.h(it, it) + g.gs[it] * g.h(it + 1, it);
        g.h(it + 1, it) = 0.0;
        g.rnc[it + 1] = -g.gs[it] * g.rnc[it];
        g.rnc[it] = g.gc[it] * g.rnc[it];
    } else if (PHASE == FIN_GMRES_CHECK) {
        criterion_check(&L, L.crit, L.stale_norm, a.history);
    } else if (PHASE == FIN_GMRES_SOLVE) {  // solve_upper_triangular over `turn` columns
        GmresState g(a.gm, a.m);
        for (int i = a.turn - 1; i >= 0; --i) {
            double t = g.rnc[i];
            for (int j = i + 1; j < a.turn; ++j) t -= g.h(i, j) * g.y[j];
            g.y[i] = t / g.h(i, i);
        }
    }
    *s = L;
}

__global__ void k_reset_scalars(DevScalars *s, DevCriterion crit)
{
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    DevScalars z{};
    z.crit = crit;
    z.rho = 1.0;  // becomes prev_rho = 1 at the first check ([UPSTREAM] cg::initialize)
    z.prev_rho = 1.0;
    z.alpha = z.omega = z.gamma = z.beta = 1.0;
    z.norm_factor = 1.0;  // StoppingCriterion.H:136
    *s = z;
}

inline int blocks_for(int64_t n) { return (int)((n + BLOCK - 1) / BLOCK); }

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
#define OGL_SPMV_K(MODE, NDOT, STREAM)                                                                     \
    hipLaunchKernelGGL((k_spmv_stream<MODE, NDOT, STREAM>), grid, block, 0, st, A.n_rows, nc, A.row_ptrs,  \
                       A.cols, A.vals, x, b, y, dots.with, dots.part, dots.part_yy, gate, xg, hf, A.block_order)
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

void launch_spmv_sell(hipStream_t st, const DevSell &A, int mode, const double *x, const double *b,
                      double *y, const SpmvDots &dots, const DevScalars *gate, const HaloFused &hf)
{
    if (A.n_rows == 0) return;
    const int nc = (int)n_chunks(A.n_rows);
    const int xg = A.xcd_group > 0 ? A.xcd_group : XCD_GROUP;
    const dim3 grid(A.block_order ? A.n_blocks : xcd_grid(nc, xg)), block(BLOCK);
#define OGL_SELL_K(MODE, NDOT, STREAM)                                                                   \
    hipLaunchKernelGGL((k_spmv_sell<MODE, NDOT, STREAM>), grid, block, 0, st, A.n_rows, nc, A.chunks,    \
                       A.dict, A.codes, A.vals, A.spill_chunk_ptr, A.spill_rows, A.spill_ptrs,          \
                       A.spill_cols, A.spill_vals, x, b, y, dots.with, dots.part, dots.part_yy, gate, xg, hf, \
                       A.rmap, A.block_order)
#define OGL_SELL(MODE, NDOT)               \
    do {                                   \
        if (A.stream)                      \
            OGL_SELL_K(MODE, NDOT, true);  \
        else                               \
            OGL_SELL_K(MODE, NDOT, false); \
    } while (0)
    if (mode == SPMV_RESIDUAL) {
        OGL_SELL(SPMV_RESIDUAL, 0);
    } else if (dots.part && dots.part_yy) {
        OGL_SELL(SPMV_PLAIN, 2);
    } else if (dots.part) {
        OGL_SELL(SPMV_PLAIN, 1);
    } else {
        OGL_SELL(SPMV_PLAIN, 0);
    }
#undef OGL_SELL
#undef OGL_SELL_K
}

void launch_spmv_sym(hipStream_t st, const DevSym &A, int mode, const double *x, const double *b, double *y,
                     const SpmvDots &dots, const DevScalars *gate, const HaloFused &hf)
{
    if (A.n_rows == 0) return;
    const int nc = (int)n_chunks(A.n_rows);
    const dim3 grid(A.block_order ? A.n_blocks : xcd_grid(nc)), block(BLOCK);
    SymOffsets off;
    for (int j = 0; j < SYM_MAX_OFFSETS; ++j) off.d[j] = A.d[j];
    // d[1] == 1 and the further distances even: the straight-line pair-load instantiation
    bool fast = A.nd >= 2 && A.d[1] == 1;
    for (int j = 2; j < A.nd; ++j) fast = fast && (A.d[j] % 2 == 0);
#define OGL_SYM_K(MODE, NDOT, ND, FAST, STREAM)                                                                          \
    hipLaunchKernelGGL((k_spmv_sym<MODE, NDOT, ND, FAST, STREAM>), grid, block, 0, st, A.n_rows, nc, off, A.mask, A.planes, \
                       x, b, y, dots.with, dots.part, dots.part_yy, gate, A.block_order, hf)
#define OGL_SYM_ND(MODE, NDOT, ND)                   \
    do {                                             \
        if (fast && A.stream)                        \
            OGL_SYM_K(MODE, NDOT, ND, true, true);   \
        else if (fast)                               \
            OGL_SYM_K(MODE, NDOT, ND, true, false);  \
        else if (A.stream)                           \
            OGL_SYM_K(MODE, NDOT, ND, false, true);  \
        else                                         \
            OGL_SYM_K(MODE, NDOT, ND, false, false); \
    } while (0)
#define OGL_SYM(MODE, NDOT)                    \
    do {                                       \
        if (A.nd == 2)                         \
            OGL_SYM_ND(MODE, NDOT, 2);         \
        else if (A.nd == 3)                    \
            OGL_SYM_ND(MODE, NDOT, 3);         \
        else                                   \
            OGL_SYM_ND(MODE, NDOT, 4);         \
    } while (0)
    static_assert(SYM_MAX_OFFSETS == 4, "instantiations above");
    if (mode == SPMV_RESIDUAL) {
        OGL_SYM(SPMV_RESIDUAL, 0);
    } else if (dots.part && dots.part_yy) {
        OGL_SYM(SPMV_PLAIN, 2);
    } else if (dots.part) {
        OGL_SYM(SPMV_PLAIN, 1);
    } else {
        OGL_SYM(SPMV_PLAIN, 0);
    }
#undef OGL_SYM
#undef OGL_SYM_ND
#undef OGL_SYM_K
}

void launch_spmv_symx(hipStream_t st, const DevSymx &A, int mode, const double *x, const double *b, double *y,
                      const SpmvDots &dots, const DevScalars *gate, const HaloFused &hf)
{
    if (A.n_rows == 0) return;
    const int nc = (int)n_chunks(A.n_rows);
    const dim3 block(BLOCK);
    // two launches at most: the chunks without explicit entries or with simple ones (lean kernel), then the others
#define OGL_SYMX_K(MODE, NDOT, STREAM, FAST, GENERAL, HDR, NB)                                                       \
    hipLaunchKernelGGL((k_spmv_symx<MODE, NDOT, STREAM, FAST, GENERAL>), dim3(NB), block, 0, st, A.n_rows, nc, HDR,    \
                       A.mask, A.planes, A.ex_rowptr, A.ex_cols, A.ex_vals, A.ex_lrow, x, b, y, dots.with, dots.part, \
                       dots.part_yy, gate, hf)
#define OGL_SYMX_G(MODE, NDOT, GENERAL, HDR, NB)               \
    do {                                                       \
        if (A.stream && A.fast)                                \
            OGL_SYMX_K(MODE, NDOT, true, true, GENERAL, HDR, NB);   \
        else if (A.stream)                                     \
            OGL_SYMX_K(MODE, NDOT, true, false, GENERAL, HDR, NB);  \
        else if (A.fast)                                       \
            OGL_SYMX_K(MODE, NDOT, false, true, GENERAL, HDR, NB);  \
        else                                                   \
            OGL_SYMX_K(MODE, NDOT, false, false, GENERAL, HDR, NB); \
    } while (0)
#define OGL_SYMX(MODE, NDOT)                                                             \
    do {                                                                                 \
        if (A.n_blocks > 0) OGL_SYMX_G(MODE, NDOT, false, A.chunks, A.n_blocks);         \
        if (A.n_blocks_general > 0) OGL_SYMX_G(MODE, NDOT, true, A.chunks_general, A.n_blocks_general); \
    } while (0)
    if (mode == SPMV_RESIDUAL) {
        OGL_SYMX(SPMV_RESIDUAL, 0);
    } else if (dots.part && dots.part_yy) {
        OGL_SYMX(SPMV_PLAIN, 2);
    } else if (dots.part) {
        OGL_SYMX(SPMV_PLAIN, 1);
    } else {
        OGL_SYMX(SPMV_PLAIN, 0);
    }
#undef OGL_SYMX
#undef OGL_SYMX_G
#undef OGL_SYMX_K
}

void launch_gather_coeffs_masked(hipStream_t st, int64_t n, const int32_t *map, const double *source,
                                 double *out)
{
    if (n == 0) return;
    hipLaunchKernelGGL(k_gather_coeffs_masked, dim3(blocks_for((n + 1) / 2)), dim3(BLOCK), 0, st,
                       (long)n, map, source, out);
}

void launch_gather_sell(hipStream_t st, int32_t n_chunks_, const SellChunk *chunks, const int32_t *map,
                        const double *source, double *out)
{
    if (n_chunks_ == 0) return;
    hipLaunchKernelGGL(k_gather_sell, dim3(n_chunks_), dim3(BLOCK), 0, st, n_chunks_, chunks, map,
                       source, out);
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

void launch_jacobi_generate(hipStream_t st, const DevCsr &A, double *inv_diag)
{
    if (A.n_rows == 0) return;
    hipLaunchKernelGGL(k_jacobi_generate, dim3(blocks_for(A.n_rows)), dim3(BLOCK), 0, st, A.n_rows,
                       A.row_ptrs, A.cols, A.vals, inv_diag);
}

void launch_bj_generate(hipStream_t st, const DevCsr &A, const DevBlockJacobi &J)
{
    if (J.n_blocks == 0) return;
    const dim3 grid((J.n_blocks + 63) / 64), block(64);
#define OGL_BJ(LD)                                                                              \
    hipLaunchKernelGGL((k_bj_generate<LD>), grid, block, 0, st, J.n_blocks, J.block_ptrs,        \
                       A.row_ptrs, A.cols, A.vals, J.blocks, J.stride, J.rows, J.pos, J.by_device_row)
    if (J.stride <= 2)
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
        hipLaunchKernelGGL(k_bj_apply_perm, dim3((unsigned)n_chunks(J.n_rows)), dim3(CHUNK_ROWS), 0, st, J.n_rows,
                           J.block_ptrs, J.row_block, J.blocks, J.stride, J.uniform, in, out, gate, J.rows);
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
                          const int32_t *wide_rows, int32_t n_wide)
{
    if (A.n_rows == 0) return;
    const dim3 grid((A.n_rows + 63) / 64), block(64);
#define OGL_ISAI(LD)                                                                             \
    hipLaunchKernelGGL((k_isai_generate<LD>), grid, block, 0, st, A.n_rows, A.row_ptrs, A.cols,   \
                       A.vals, spd, w_row_ptrs, w_cols, w_vals)
    if (max_row <= 8)
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

void launch_scale(hipStream_t st, int32_t n, double *v, double factor)
{
    if (n == 0) return;
    hipLaunchKernelGGL(k_scale, dim3(blocks_for(n)), dim3(BLOCK), 0, st, n, v, factor);
}

void launch_fill_xbar(hipStream_t st, int32_t n, double *v, const DevScalars *s)
{
    if (n == 0) return;
    hipLaunchKernelGGL(k_fill_xbar, dim3(blocks_for(n)), dim3(BLOCK), 0, st, n, v, s);
}

void launch_partials_sum(hipStream_t st, int32_t n, const double *a, double *part)
{
    const int nc = (int)n_chunks(n);
    if (nc == 0) return;
    hipLaunchKernelGGL((k_partials<P_SUM>), dim3(nc), dim3(BLOCK), 0, st, n, nc, a, nullptr, part,
                       nullptr, nullptr);
}

void launch_partials_dot(hipStream_t st, int32_t n, const double *a, const double *b, double *part,
                         const DevScalars *gate)
{
    const int nc = (int)n_chunks(n);
    if (nc == 0) return;
    hipLaunchKernelGGL((k_partials<P_DOT>), dim3(nc), dim3(BLOCK), 0, st, n, nc, a, b, part, gate,
                       nullptr);
}

void launch_partials_dot_chunks(hipStream_t st, int32_t n, const double *a, const double *b,
                                double *part, const DevScalars *gate, const int32_t *chunk_list,
                                int32_t count)
{
    if (count == 0) return;
    hipLaunchKernelGGL((k_partials<P_DOT>), dim3(count), dim3(BLOCK), 0, st, n, (int)n_chunks(n), a, b,
                       part, gate, chunk_list);
}

void launch_partials_norm1(hipStream_t st, int32_t n, const double *a, double *part)
{
    const int nc = (int)n_chunks(n);
    if (nc == 0) return;
    hipLaunchKernelGGL((k_partials<P_NORM1>), dim3(nc), dim3(BLOCK), 0, st, n, nc, a, nullptr,
                       part, nullptr, nullptr);
}

void launch_partials_normfactor(hipStream_t st, int32_t n, const double *b, const double *w,
                                const double *r, double *part)
{
    const int nc = (int)n_chunks(n);
    if (nc == 0) return;
    hipLaunchKernelGGL(k_partials_normfactor, dim3(nc), dim3(BLOCK), 0, st, n, b, w, r, part);
}

void launch_cg_rho_norm(hipStream_t st, int32_t n, const double *r, const double *inv_diag,
                        double *part_rho, double *part_norm, const DevScalars *gate)
{
    const int nc = (int)n_chunks(n);
    if (nc == 0) return;
    hipLaunchKernelGGL(k_cg_rho_norm, dim3(nc), dim3(BLOCK), 0, st, n, r, inv_diag, part_rho,
                       part_norm, gate);
}

void launch_cg_step1(hipStream_t st, int32_t n, double *p, const double *r, const double *inv_diag,
                     const DevScalars *s)
{
    const int nc = (int)n_chunks(n);
    if (nc == 0) return;
    hipLaunchKernelGGL(k_cg_step1, dim3(nc), dim3(BLOCK), 0, st, n, p, r, inv_diag, s);
}

void launch_cg_step1x(hipStream_t st, int32_t n, double *p, double *x, const double *r,
                      const double *inv_diag, const DevScalars *s, const HaloPutFused *put)
{
    const int nc = (int)n_chunks(n);
    if (nc == 0) return;
    if (put && put->chunk_sptr)
        hipLaunchKernelGGL((k_cg_step1x<true>), dim3(nc), dim3(BLOCK), 0, st, n, p, x, r, inv_diag, s, *put);
    else
        hipLaunchKernelGGL((k_cg_step1x<false>), dim3(nc), dim3(BLOCK), 0, st, n, p, x, r, inv_diag, s,
                           HaloPutFused{});
}

void launch_cg_step2r(hipStream_t st, int32_t n, double *r, const double *q, const double *inv_diag,
                      double *part_rho, double *part_norm, const DevScalars *s, double *z_out,
                      const HaloPutFused *put)
{
    const int nc = (int)n_chunks(n);
    if (nc == 0) return;
    if (put && put->chunk_sptr)
        hipLaunchKernelGGL((k_cg_step2r<true>), dim3(nc), dim3(BLOCK), 0, st, n, r, q, inv_diag, part_rho,
                           part_norm, s, z_out, *put);
    else
        hipLaunchKernelGGL((k_cg_step2r<false>), dim3(nc), dim3(BLOCK), 0, st, n, r, q, inv_diag, part_rho,
                           part_norm, s, z_out, HaloPutFused{});
}

void launch_cg_step1x_fin(hipStream_t st, int32_t n, double *p, double *x, const double *r, const double *inv_diag,
                          const DevScalars *sin, DevScalars *sout, const double *part_rho,
                          const double *part_norm, double *history, int first)
{
    const int nc = (int)n_chunks(n);
    if (nc == 0) return;
    hipLaunchKernelGGL(k_cg_step1x_fin, dim3(nc), dim3(BLOCK), 0, st, n, p, x, r, inv_diag, sin, sout, part_rho,
                       part_norm, nc, history, first);
}

void launch_cg_turn_sym(hipStream_t st, const DevSym &A, const double *p_in, double *p_out, double *x, const double *z,
                        double *q, double *part_beta, const DevScalars *sin, DevScalars *sout,
                        const double *part_rho, const double *part_norm, double *history, int first)
{
    if (A.n_rows == 0) return;
    const int nc = (int)n_chunks(A.n_rows);
    const dim3 grid(A.block_order ? A.n_blocks : xcd_grid(nc)), block(BLOCK);
    SymOffsets off;
    for (int j = 0; j < SYM_MAX_OFFSETS; ++j) off.d[j] = A.d[j];
    bool fast = A.nd >= 2 && A.d[1] == 1;
    for (int j = 2; j < A.nd; ++j) fast = fast && (A.d[j] % 2 == 0);
#define OGL_TURN_K(ND, FAST)                                                                                        \
    hipLaunchKernelGGL((k_cg_turn_sym<ND, FAST>), grid, block, 0, st, A.n_rows, nc, off, A.mask, A.planes, p_in, p_out, \
                       x, z, q, part_beta, sin, sout, part_rho, part_norm, nc, history, first, A.block_order)
#define OGL_TURN_ND(ND)              \
    do {                             \
        if (fast)                    \
            OGL_TURN_K(ND, true);    \
        else                         \
            OGL_TURN_K(ND, false);   \
    } while (0)
    if (A.nd == 2)
        OGL_TURN_ND(2);
    else if (A.nd == 3)
        OGL_TURN_ND(3);
    else
        OGL_TURN_ND(4);
#undef OGL_TURN_ND
#undef OGL_TURN_K
}

void launch_cg_turn_sym_big(hipStream_t st, const DevSym &A, const double *p_in, double *p_out, double *x,
                            const double *z, double *q, double *part_beta, const DevScalars *s,
                            const HaloFused &hf, const double *p_halo_in, double *p_halo_out)
{
    if (A.n_rows == 0) return;
    const int nc = (int)n_chunks(A.n_rows);
    const dim3 grid(A.block_order ? A.n_blocks : xcd_grid(nc)), block(BLOCK);
    SymOffsets off;
    for (int j = 0; j < SYM_MAX_OFFSETS; ++j) off.d[j] = A.d[j];
    bool fast = A.nd >= 2 && A.d[1] == 1;
    for (int j = 2; j < A.nd; ++j) fast = fast && (A.d[j] % 2 == 0);
#define OGL_TURN_K(ND, FAST, STREAM)                                                                                \
    do {                                                                                                            \
        if (hf.chunk_bptr)                                                                                          \
            hipLaunchKernelGGL((k_cg_turn_sym_big<ND, FAST, STREAM, true>), grid, block, 0, st, A.n_rows, nc, off,  \
                               A.mask, A.planes, p_in, p_out, x, z, q, part_beta, s, A.block_order, hf, p_halo_in,  \
                               p_halo_out);                                                                         \
        else                                                                                                        \
            hipLaunchKernelGGL((k_cg_turn_sym_big<ND, FAST, STREAM, false>), grid, block, 0, st, A.n_rows, nc, off, \
                               A.mask, A.planes, p_in, p_out, x, z, q, part_beta, s, A.block_order, HaloFused{},    \
                               nullptr, nullptr);                                                                   \
    } while (0)
#define OGL_TURN_ND(ND)                     \
    do {                                    \
        if (fast && A.stream)               \
            OGL_TURN_K(ND, true, true);     \
        else if (fast)                      \
            OGL_TURN_K(ND, true, false);    \
        else if (A.stream)                  \
            OGL_TURN_K(ND, false, true);    \
        else                                \
            OGL_TURN_K(ND, false, false);   \
    } while (0)
    if (A.nd == 2)
        OGL_TURN_ND(2);
    else if (A.nd == 3)
        OGL_TURN_ND(3);
    else
        OGL_TURN_ND(4);
#undef OGL_TURN_ND
#undef OGL_TURN_K
}

void launch_cg_step2r_fin(hipStream_t st, int32_t n, double *r, const double *q, const double *inv_diag,
                          double *part_rho, double *part_norm, const DevScalars *sin, DevScalars *sout,
                          const double *part_beta, double *z_out)
{
    const int nc = (int)n_chunks(n);
    if (nc == 0) return;
    hipLaunchKernelGGL(k_cg_step2r_fin, dim3(nc), dim3(BLOCK), 0, st, n, r, q, inv_diag, part_rho, part_norm, sin,
                       sout, part_beta, nc, z_out);
}

void launch_cg_step2(hipStream_t st, int32_t n, double *x, double *r, const double *p,
                     const double *q, const double *inv_diag, double *part_rho, double *part_norm,
                     const DevScalars *s)
{
    const int nc = (int)n_chunks(n);
    if (nc == 0) return;
    hipLaunchKernelGGL(k_cg_step2, dim3(nc), dim3(BLOCK), 0, st, n, x, r, p, q, inv_diag, part_rho,
                       part_norm, s);
}

void launch_bicg_step1(hipStream_t st, int32_t n, double *p, const double *r, const double *v,
                       const double *inv_diag, double *y, const DevScalars *s)
{
    const int nc = (int)n_chunks(n);
    if (nc == 0) return;
    hipLaunchKernelGGL(k_bicg_step1, dim3(nc), dim3(BLOCK), 0, st, n, p, r, v, inv_diag, y, s);
}

void launch_bicg_step2(hipStream_t st, int32_t n, const double *r, const double *v, double *sv,
                       const double *inv_diag, double *z, double *part_norm, const DevScalars *s)
{
    const int nc = (int)n_chunks(n);
    if (nc == 0) return;
    hipLaunchKernelGGL(k_bicg_step2, dim3(nc), dim3(BLOCK), 0, st, n, r, v, sv, inv_diag, z,
                       part_norm, s);
}

void launch_bicg_step3(hipStream_t st, int32_t n, double *x, double *r, const double *sv,
                       const double *t, const double *y, const double *z, const double *rr,
                       double *part_rho, double *part_norm, const DevScalars *s, int turn)
{
    const int nc = (int)n_chunks(n);
    if (nc == 0) return;
    hipLaunchKernelGGL(k_bicg_step3, dim3(nc), dim3(BLOCK), 0, st, n, x, r, sv, t, y, z, rr,
                       part_rho, part_norm, s, turn);
}

void launch_gmres_mgs_fold(hipStream_t st, int32_t n, double *w, const double *vprev, double *h_out, const double *vdot,
                           const double *part_in, double *part_out, const DevScalars *gate)
{
    const int nc = (int)n_chunks(n);
    if (nc == 0) return;
    hipLaunchKernelGGL(k_gmres_mgs_fold, dim3(nc), dim3(BLOCK), 0, st, n, w, vprev, h_out, vdot, part_in, nc, part_out,
                       gate);
}

void launch_bicg_fold1(hipStream_t st, int32_t n, double *p, const double *r, const double *v, const double *inv_diag,
                       double *y, const DevScalars *sin, DevScalars *sout, const double *part_rho,
                       const double *part_norm, double *history)
{
    const int nc = (int)n_chunks(n);
    if (nc == 0) return;
    hipLaunchKernelGGL(k_bicg_fold1, dim3(nc), dim3(BLOCK), 0, st, n, p, r, v, inv_diag, y, sin, sout, part_rho,
                       part_norm, nc, history);
}

void launch_bicg_fold2(hipStream_t st, int32_t n, const double *r, const double *v, double *sv, const double *inv_diag,
                       double *z, double *part_norm_out, const DevScalars *sin, DevScalars *sout,
                       const double *part_beta)
{
    const int nc = (int)n_chunks(n);
    if (nc == 0) return;
    hipLaunchKernelGGL(k_bicg_fold2, dim3(nc), dim3(BLOCK), 0, st, n, r, v, sv, inv_diag, z, part_norm_out, sin, sout,
                       part_beta, nc);
}

void launch_bicg_fold3(hipStream_t st, int32_t n, double *x, double *r, const double *sv, const double *t,
                       const double *y, const double *z, const double *rr, double *part_rho_out, double *part_norm_out,
                       const DevScalars *sin, DevScalars *sout, const double *part_gamma, const double *part_tt,
                       const double *part_snorm, double *history, int turn)
{
    const int nc = (int)n_chunks(n);
    if (nc == 0) return;
    hipLaunchKernelGGL(k_bicg_fold3, dim3(nc), dim3(BLOCK), 0, st, n, x, r, sv, t, y, z, rr, part_rho_out,
                       part_norm_out, sin, sout, part_gamma, part_tt, part_snorm, nc, history, turn);
}

void launch_gmres_scale(hipStream_t st, int32_t n, double *out, const double *in,
                        const double *denom, const DevScalars *gate)
{
    const int nc = (int)n_chunks(n);
    if (nc == 0) return;
    hipLaunchKernelGGL(k_gmres_scale, dim3(nc), dim3(BLOCK), 0, st, n, out, in, denom, gate);
}

void launch_gmres_mgs(hipStream_t st, int32_t n, double *w, const double *vprev,
                      const double *hprev, const double *vdot, double *part,
                      const DevScalars *gate)
{
    const int nc = (int)n_chunks(n);
    if (nc == 0) return;
    hipLaunchKernelGGL(k_gmres_mgs, dim3(nc), dim3(BLOCK), 0, st, n, w, vprev, hprev, vdot, part, gate);
}

void launch_gmres_update_x(hipStream_t st, int32_t n, const double *V, int64_t ld, const double *y,
                           int32_t it, const double *inv_diag, double *x, double *before,
                           const DevScalars *gate)
{
    const int nc = (int)n_chunks(n);
    if (nc == 0) return;
    hipLaunchKernelGGL(k_gmres_update_x, dim3(nc), dim3(BLOCK), 0, st, n, V, (long)ld, y, it,
                       inv_diag, x, before, gate);
}

void launch_mul(hipStream_t st, int32_t n, double *out, const double *in, const double *inv_diag,
                const DevScalars *gate)
{
    const int nc = (int)n_chunks(n);
    if (nc == 0) return;
    hipLaunchKernelGGL(k_mul, dim3(nc), dim3(BLOCK), 0, st, n, out, in, inv_diag, gate);
}

void launch_add(hipStream_t st, int32_t n, double *x, const double *a, const DevScalars *gate)
{
    const int nc = (int)n_chunks(n);
    if (nc == 0) return;
    hipLaunchKernelGGL(k_add, dim3(nc), dim3(BLOCK), 0, st, n, x, a, gate);
}

void launch_finalize(hipStream_t st, int phase, DevScalars *s, const FinArgs &a)
{
    const dim3 grid(1), block(FIN_BLOCK);
    switch (phase) {
    case FIN_MEAN:
        hipLaunchKernelGGL((k_finalize<FIN_MEAN>), grid, block, 0, st, s, a);
        break;
    case FIN_NORMFACTOR:
        hipLaunchKernelGGL((k_finalize<FIN_NORMFACTOR>), grid, block, 0, st, s, a);
        break;
    case FIN_CG_CHECK:
        hipLaunchKernelGGL((k_finalize<FIN_CG_CHECK>), grid, block, 0, st, s, a);
        break;
    case FIN_BETA:
        hipLaunchKernelGGL((k_finalize<FIN_BETA>), grid, block, 0, st, s, a);
        break;
    case FIN_BICG_ALPHA:
        hipLaunchKernelGGL((k_finalize<FIN_BICG_ALPHA>), grid, block, 0, st, s, a);
        break;
    case FIN_BICG_CHECK2:
        hipLaunchKernelGGL((k_finalize<FIN_BICG_CHECK2>), grid, block, 0, st, s, a);
        break;
    case FIN_BICG_OMEGA:
        hipLaunchKernelGGL((k_finalize<FIN_BICG_OMEGA>), grid, block, 0, st, s, a);
        break;
    case FIN_BICG_CHECK2_OMEGA:
        hipLaunchKernelGGL((k_finalize<FIN_BICG_CHECK2_OMEGA>), grid, block, 0, st, s, a);
        break;
    case FIN_GMRES_RESTART:
        hipLaunchKernelGGL((k_finalize<FIN_GMRES_RESTART>), grid, block, 0, st, s, a);
        break;
    case FIN_GMRES_H:
        hipLaunchKernelGGL((k_finalize<FIN_GMRES_H>), grid, block, 0, st, s, a);
        break;
    case FIN_GMRES_COL:
        hipLaunchKernelGGL((k_finalize<FIN_GMRES_COL>), grid, block, 0, st, s, a);
        break;
    case FIN_GMRES_CHECK:
        hipLaunchKernelGGL((k_finalize<FIN_GMRES_CHECK>), grid, block, 0, st, s, a);
        break;
    case FIN_GMRES_SOLVE:
        hipLaunchKernelGGL((k_finalize<FIN_GMRES_SOLVE>), grid, block, 0, st, s, a);
        break;
    default:
        hipLaunchKernelGGL((k_finalize<FIN_RAW>), grid, block, 0, st, s, a);
        break;
    }
}

void launch_peer_allreduce(hipStream_t st, const PeerArgs &pa, double *vals, int n, int32_t *error)
{
    hipLaunchKernelGGL(k_peer_allreduce, dim3(1), dim3(64), 0, st, pa, vals, n, error);
}

void launch_pack_put(hipStream_t st, const DevHalo &H, const PeerHalo &P, const double *x,
                     const DevScalars *gate)
{
    if (H.n_send == 0) return;
    hipLaunchKernelGGL(k_pack_put, dim3(blocks_for(H.n_send)), dim3(BLOCK), 0, st, H.n_send,
                       H.send_idxs, P, x, gate);
}

void launch_pack_put_signal(hipStream_t st, const DevHalo &H, const PeerHalo &P, const double *x,
                            const DevScalars *gate, unsigned *ticket)
{
    if (H.n_send == 0) return;
    hipLaunchKernelGGL(k_pack_put_signal, dim3(blocks_for(H.n_send)), dim3(BLOCK), 0, st, H.n_send,
                       H.send_idxs, P, x, gate, ticket);
}

void launch_halo_finish(hipStream_t st, const DevHalo &H, int mode, int32_t n_rows,
                        const int32_t *chunk_list, const int32_t *chunk_row_ptr, int32_t n_chunks_b,
                        const double *recv, double *y, const SpmvDots &dots, const PeerHalo &P,
                        const DevScalars *gate, DevScalars *s)
{
    if (n_chunks_b == 0) return;
    const dim3 grid(n_chunks_b), block(BLOCK);
#define OGL_HF(MODE, NDOT)                                                                        \
    hipLaunchKernelGGL((k_halo_finish<MODE, NDOT>), grid, block, 0, st, n_rows, chunk_list,       \
                       chunk_row_ptr, H.boundary_rows, H.entry_ptrs, H.cols, H.vals, recv, y,     \
                       dots.with, dots.part, dots.part_yy, P, gate, s)
    if (mode == SPMV_RESIDUAL) {
        OGL_HF(SPMV_RESIDUAL, 0);
    } else if (dots.part && dots.part_yy) {
        OGL_HF(SPMV_PLAIN, 2);
    } else if (dots.part) {
        OGL_HF(SPMV_PLAIN, 1);
    } else {
        OGL_HF(SPMV_PLAIN, 0);
    }
#undef OGL_HF
}

void launch_halo_signal(hipStream_t st, const PeerHalo &P, const DevScalars *gate)
{
    if (P.n_neigh == 0) return;
    hipLaunchKernelGGL(k_halo_signal, dim3(1), dim3(64), 0, st, P, gate);
}

void launch_halo_wait(hipStream_t st, const PeerHalo &P, const DevScalars *gate, DevScalars *s)
{
    if (P.n_neigh == 0) return;
    hipLaunchKernelGGL(k_halo_wait, dim3(1), dim3(64), 0, st, P, gate, s);
}

void launch_peer_post(hipStream_t st, unsigned long long *dst, unsigned long long w0,
                      unsigned long long w1, unsigned long long w2, unsigned long long w3)
{
    hipLaunchKernelGGL(k_peer_post, dim3(1), dim3(64), 0, st, dst, w0, w1, w2, w3);
}

void launch_reset_scalars(hipStream_t st, DevScalars *s, const DevCriterion &crit)
{
    hipLaunchKernelGGL(k_reset_scalars, dim3(1), dim3(64), 0, st, s, crit);
}

}  // namespace ogl
