// win_tune.hip -- development harness (not product): the CSR-stream SpMV with the chunk's x WINDOW staged in LDS,
// against the product's packed-column kernel (k_spmv_stream21), on a pattern read from a file (tools/dump_pattern.py:
// int32 n, int32 nnz, row_ptrs[n+1], cols[nnz]), e.g. the Voronoi proxy in the library's RCM numbering.
//
// Why: a polyhedral mesh in RCM order gathers x at ~8,400 entries per chunk of 512 rows, but they fall into only
// ~2,100 distinct columns on ~450 distinct 64-byte lines.  The product kernel issues every one of the 8,400 as an
// 8-byte lane gather to the vector cache; here the chunk's distinct LINES are loaded once (coalesced 64-byte pieces)
// into LDS and the entries address that window with 16-bit local indices (line slot << 3 | column & 7):
// 8 + 2 bytes per entry + 4 bytes per distinct line instead of 8 + 2.67.
//
//   hipcc --offload-arch=gfx950 -O3 -ffp-contract=off tools/win_tune.hip -o tools/bin/win_tune
//   tools/bin/win_tune pattern.bin [reps]
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CK(x)                                                                            \
    do {                                                                                 \
        hipError_t e = (x);                                                              \
        if (e != hipSuccess) {                                                           \
            printf("HIP error %s at %s:%d\n", hipGetErrorString(e), __FILE__, __LINE__); \
            exit(1);                                                                     \
        }                                                                                \
    } while (0)

typedef double d2 __attribute__((ext_vector_type(2)));
typedef unsigned u4 __attribute__((ext_vector_type(4)));
constexpr int BLOCK = 256, N_WAVES = 4, CHUNK_ROWS = 512, RPT = 2;

__device__ __forceinline__ int xcd_chunk(int block, int G)
{
    const int slot = block / 8, xcd = block % 8;
    return (slot / G) * (8 * G) + xcd * G + slot % G;
}
__device__ __forceinline__ double wave_sum(double v)
{
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) v += __shfl_xor(v, off, 64);
    return v;
}
__device__ __forceinline__ double block_sum(double v, double *slot)
{
    v = wave_sum(v);
    const int lane = threadIdx.x & 63, wave = threadIdx.x / 64;
    if (lane == 0) slot[wave] = v;
    __syncthreads();
    double s = slot[0];
    for (int w = 1; w < N_WAVES; ++w) s += slot[w];
    __syncthreads();
    return s;
}

// ---- the product kernel (kernels.hip k_spmv_stream21<PLAIN, 1, true>), restated for the same-box baseline ----
constexpr int S21_TILE = 3072, S21_GROUPS = 2, S21_BITS = 21;
struct S21Chunk {
    int base, word_off;
};
// FAKE: the gathers of one instruction fall on 4 consecutive 64-byte lines (wrong result on purpose): what the
// kernel would run at if the vector cache saw few lines per gather instruction
// ALIGNED: the values come from a copy in which every chunk's first tile starts on a 128-byte line (val_shift):
// every non-temporal wave load then covers whole lines (the CSR array starts a chunk wherever the previous one ended,
// so a wave's 1 KiB piece straddles 9 lines instead of 8 and the two shared ones are fetched by both neighbours)
template <bool FAKE, bool ALIGNED = false>
__global__ __launch_bounds__(BLOCK) void k_stream21(int n_rows, int n_chunks, const int *__restrict__ row_ptrs,
                                                    const S21Chunk *__restrict__ chunks21,
                                                    const uint4 *__restrict__ codes, const double *__restrict__ vals,
                                                    const double *__restrict__ x, double *__restrict__ y,
                                                    double *__restrict__ dot_partials,
                                                    const long *__restrict__ val_shift = nullptr)
{
    __shared__ __attribute__((aligned(16))) double prod[S21_TILE];
    __shared__ double slot[N_WAVES];
    const int chunk = xcd_chunk(blockIdx.x, 4);
    if (chunk >= n_chunks) return;
    const int tid = threadIdx.x;
    const int r0 = chunk * CHUNK_ROWS, r1 = min(r0 + CHUNK_ROWS, n_rows);
    const int nz0 = row_ptrs[r0], nz1 = row_ptrs[r1];
    const S21Chunk ck = chunks21[chunk];
    const int row = r0 + tid * RPT;
    int rs[RPT + 1];
#pragma unroll
    for (int j = 0; j <= RPT; ++j) rs[j] = row_ptrs[min(row + j, r1)];
    double acc[RPT] = {0.0, 0.0};
    const uint4 *cw = codes + ck.word_off + tid;
    constexpr unsigned long long M = (1ull << S21_BITS) - 1;
    int tile = 0;
    for (int t0 = nz0 & ~3; t0 < nz1; t0 += S21_TILE, ++tile) {
        d2 va[S21_GROUPS][3];
        u4 cc[S21_GROUPS];
#pragma unroll
        for (int g = 0; g < S21_GROUPS; ++g) {
            cc[g] = __builtin_nontemporal_load(reinterpret_cast<const u4 *>(cw + (long)(tile * S21_GROUPS + g) * BLOCK));
#pragma unroll
            for (int k = 0; k < 3; ++k) {
                const int e = t0 + ((g * 3 + k) * BLOCK + tid) * 2;
                const int ec = e < nz1 ? e : t0;
                va[g][k] = __builtin_nontemporal_load(
                    reinterpret_cast<const d2 *>(vals + (ALIGNED ? val_shift[chunk] : 0) + ec));
            }
        }
#pragma unroll
        for (int g = 0; g < S21_GROUPS; ++g) {
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
            for (int i = 0; i < 6; ++i)
                xv[i] = FAKE ? x[min(r0 + ((tid & 63) >> 1) + 32 * i + (c[i] & 1), n_rows - 1)] : x[c[i]];
#pragma unroll
            for (int k = 0; k < 3; ++k) {
                d2 p0;
                p0.x = va[g][k].x * xv[2 * k];
                p0.y = va[g][k].y * xv[2 * k + 1];
                *reinterpret_cast<d2 *>(prod + ((g * 3 + k) * BLOCK + tid) * 2) = p0;
            }
        }
        __syncthreads();
        const int t1 = t0 + S21_TILE;
#pragma unroll
        for (int j = 0; j < RPT; ++j) {
            const int kb = max(rs[j], t0), ke = min(rs[j + 1], t1);
            for (int k = kb; k < ke; ++k) acc[j] += prod[k - t0];
        }
        __syncthreads();
    }
    double d = 0.0;
#pragma unroll
    for (int j = 0; j < RPT; ++j)
        if (row + j < r1) {
            y[row + j] = acc[j];
            d += x[row + j] * acc[j];
        }
    const double s = block_sum(d, slot);
    if (tid == 0) dot_partials[chunk] = s;
}

// ---- the product kernel with the loop rotated: the value / code loads of tile t + 1 are issued as soon as the
// products of tile t sit in LDS, i.e. BEFORE the barrier and the row sums of tile t, so that their round trip to
// HBM overlaps the LDS phase instead of following it (same registers: they are free once the products are written)
template <int RB, int GROUPS = S21_GROUPS, bool ROT = true, int MINW = 1>
__global__ __launch_bounds__(BLOCK, MINW) void k_stream21_rot(int n_rows, int n_chunks, const int *__restrict__ row_ptrs,
                                                        const S21Chunk *__restrict__ chunks21,
                                                        const uint4 *__restrict__ codes,
                                                        const double *__restrict__ vals, const double *__restrict__ x,
                                                        double *__restrict__ y, double *__restrict__ dot_partials)
{
    constexpr int TILE = GROUPS * 1536;
    __shared__ __attribute__((aligned(16))) double prod[TILE];
    __shared__ double slot[N_WAVES];
    const int chunk = xcd_chunk(blockIdx.x, 4);
    if (chunk >= n_chunks) return;
    const int tid = threadIdx.x;
    const int r0 = chunk * CHUNK_ROWS, r1 = min(r0 + CHUNK_ROWS, n_rows);
    const int nz0 = row_ptrs[r0], nz1 = row_ptrs[r1];
    const S21Chunk ck = chunks21[chunk];
    const int row = r0 + tid * RPT;
    int rs[RPT + 1];
#pragma unroll
    for (int j = 0; j <= RPT; ++j) rs[j] = row_ptrs[min(row + j, r1)];
    double acc[RPT] = {0.0, 0.0};
    const uint4 *cw = codes + ck.word_off + tid;
    constexpr unsigned long long M = (1ull << S21_BITS) - 1;
    d2 va[GROUPS][3];
    u4 cc[GROUPS];
    auto load_tile = [&](int t0, int tile) {
#pragma unroll
        for (int g = 0; g < GROUPS; ++g) {
            cc[g] = __builtin_nontemporal_load(reinterpret_cast<const u4 *>(cw + (long)(tile * GROUPS + g) * BLOCK));
#pragma unroll
            for (int k = 0; k < 3; ++k) {
                const int e = t0 + ((g * 3 + k) * BLOCK + tid) * 2;
                const int ec = e < nz1 ? e : t0;
                va[g][k] = __builtin_nontemporal_load(reinterpret_cast<const d2 *>(vals + ec));
            }
        }
    };
    int tile = 0;
    const int tfirst = nz0 & ~3;
    if (ROT && tfirst < nz1) load_tile(tfirst, 0);
    for (int t0 = tfirst; t0 < nz1; t0 += TILE, ++tile) {
        if (!ROT) load_tile(t0, tile);
#pragma unroll
        for (int g = 0; g < GROUPS; ++g) {
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
                d2 p0;
                p0.x = va[g][k].x * xv[2 * k];
                p0.y = va[g][k].y * xv[2 * k + 1];
                *reinterpret_cast<d2 *>(prod + ((g * 3 + k) * BLOCK + tid) * 2) = p0;
            }
        }
        if (ROT && t0 + TILE < nz1) load_tile(t0 + TILE, tile + 1);  // in flight during the row sums below
        __syncthreads();
        const int t1 = t0 + TILE;
#pragma unroll
        for (int j = 0; j < RPT; ++j) {
            const int kb = max(rs[j], t0), ke = min(rs[j + 1], t1);
            if (RB == 1) {
                for (int k = kb; k < ke; ++k) acc[j] += prod[k - t0];
            } else {
                for (int k = kb; k < ke; k += RB) {
                    double pv[RB];
#pragma unroll
                    for (int i = 0; i < RB; ++i) pv[i] = prod[min(k + i, ke - 1) - t0];
#pragma unroll
                    for (int i = 0; i < RB; ++i)
                        if (k + i < ke) acc[j] += pv[i];
                }
            }
        }
        __syncthreads();
    }
    double d = 0.0;
#pragma unroll
    for (int j = 0; j < RPT; ++j)
        if (row + j < r1) {
            y[row + j] = acc[j];
            d += x[row + j] * acc[j];
        }
    const double s = block_sum(d, slot);
    if (tid == 0) dot_partials[chunk] = s;
}

// ---- the window kernel ----
// per chunk: {first line of its line list, number of lines, first 16-byte code word}
struct WinChunk {
    int line_off, n_lines, word_off, pad;
};
// GROUPS groups of 2048 entries per pass (8 entries per lane and group: 4 value pairs + one 16-byte word of 8 codes);
// ABL: 0 full kernel, 1 = window not loaded (x taken as the code: wrong on purpose), 2 = no row phase (wrong),
// 3 = window loaded but entries gather x from global memory through the line list (same result: what the LDS
// window itself buys); RB: row-phase reads issued RB at a time
template <int GROUPS, int ABL, int RB, int MINW>
__global__ __launch_bounds__(BLOCK, MINW) void k_win(int n_rows, int n_chunks, const int *__restrict__ row_ptrs,
                                                     const WinChunk *__restrict__ wc, const int *__restrict__ lines,
                                                     const uint4 *__restrict__ codes, const double *__restrict__ vals,
                                                     const double *__restrict__ x, double *__restrict__ y,
                                                     double *__restrict__ dot_partials, int xg, int win_doubles)
{
    extern __shared__ __attribute__((aligned(16))) double lds[];
    constexpr int TILE = GROUPS * 2048;
    double *prod = lds;          // [TILE]
    double *xs = lds + TILE;     // [win_doubles]
    __shared__ double slot[N_WAVES];
    const int chunk = xcd_chunk(blockIdx.x, xg);
    if (chunk >= n_chunks) return;
    const int tid = threadIdx.x;
    const int r0 = chunk * CHUNK_ROWS, r1 = min(r0 + CHUNK_ROWS, n_rows);
    const int nz0 = row_ptrs[r0], nz1 = row_ptrs[r1];
    const WinChunk ck = wc[chunk];
    const int row = r0 + tid * RPT;
    int rs[RPT + 1];
#pragma unroll
    for (int j = 0; j <= RPT; ++j) rs[j] = row_ptrs[min(row + j, r1)];
    // window: the chunk's distinct 64-byte lines of x, 8 lanes per line
    if (ABL != 1) {
        const int nw = ck.n_lines * 8;
        const int last = n_rows - 1;
        for (int i = tid; i < nw; i += BLOCK) {
            const int ln = lines[ck.line_off + (i >> 3)];
            xs[i] = x[min(ln * 8 + (i & 7), last)];
        }
    }
    double acc[RPT] = {0.0, 0.0};
    double tsum = 0.0;
    const uint4 *cw = codes + ck.word_off + tid;
    int tile = 0;
    bool first = true;
    for (int t0 = nz0 & ~3; t0 < nz1; t0 += TILE, ++tile) {
        d2 va[GROUPS][4];
        u4 cc[GROUPS];
#pragma unroll
        for (int g = 0; g < GROUPS; ++g) {
            if (t0 + g * 2048 >= nz1) break;
            cc[g] = __builtin_nontemporal_load(reinterpret_cast<const u4 *>(cw + (long)(tile * GROUPS + g) * BLOCK));
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const int e = t0 + ((g * 4 + k) * BLOCK + tid) * 2;
                const int ec = e < nz1 ? e : t0;
                va[g][k] = __builtin_nontemporal_load(reinterpret_cast<const d2 *>(vals + ec));
            }
        }
        if (first) {
            __syncthreads();  // the window is in LDS (its loads were issued before the first tile's)
            first = false;
        }
#pragma unroll
        for (int g = 0; g < GROUPS; ++g) {
            if (t0 + g * 2048 >= nz1) break;
            const unsigned w[4] = {cc[g].x, cc[g].y, cc[g].z, cc[g].w};
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const int c0 = w[k] & 0xffffu, c1 = w[k] >> 16;
                double x0, x1;
                if (ABL == 1) {
                    x0 = (double)c0;
                    x1 = (double)c1;
                } else if (ABL == 3) {
                    x0 = x[min(lines[ck.line_off + (c0 >> 3)] * 8 + (c0 & 7), n_rows - 1)];
                    x1 = x[min(lines[ck.line_off + (c1 >> 3)] * 8 + (c1 & 7), n_rows - 1)];
                } else {
                    x0 = xs[c0];
                    x1 = xs[c1];
                }
                d2 p0;
                p0.x = va[g][k].x * x0;
                p0.y = va[g][k].y * x1;
                if (ABL == 2)
                    tsum += p0.x + p0.y;
                else
                    *reinterpret_cast<d2 *>(prod + ((g * 4 + k) * BLOCK + tid) * 2) = p0;
            }
        }
        if (ABL != 2) {
            __syncthreads();
            const int t1 = t0 + TILE;
#pragma unroll
            for (int j = 0; j < RPT; ++j) {
                const int kb = max(rs[j], t0), ke = min(rs[j + 1], t1);
                if (RB == 1) {
                    for (int k = kb; k < ke; ++k) acc[j] += prod[k - t0];
                } else {
                    for (int k = kb; k < ke; k += RB) {
                        double pv[RB];
#pragma unroll
                        for (int i = 0; i < RB; ++i) pv[i] = prod[min(k + i, ke - 1) - t0];
#pragma unroll
                        for (int i = 0; i < RB; ++i)
                            if (k + i < ke) acc[j] += pv[i];
                    }
                }
            }
            __syncthreads();
        }
    }
    double d = 0.0;
#pragma unroll
    for (int j = 0; j < RPT; ++j)
        if (row + j < r1) {
            const double v = ABL == 2 ? tsum : acc[j];
            y[row + j] = v;
            d += x[row + j] * v;
        }
    const double s = block_sum(d, slot);
    if (tid == 0) dot_partials[chunk] = s;
}

// ---- column-sorted tiles ----
// The entries of a tile (CS_TILE consecutive CSR entries of a chunk) are STORED in ascending column order, so that
// the 64 lanes of a gather instruction fall on a handful of lines instead of ~50; every entry carries where its
// product belongs in the tile (12 bits) next to its column (20-bit offset from the chunk's smallest column): the
// products are scattered into LDS at their CSR positions and the row phase is the product kernel's.  12 bytes per
// entry like a plain CSR.  Lane `tid`, entry j of group g = sorted index (g * 4 + j) * 256 + tid of the tile; the
// values of (j = 2p, 2p + 1) are adjacent in memory (one 16-byte load), the four codes are one 16-byte word.
constexpr int CS_GROUPS = 3, CS_TILE = CS_GROUPS * 1024;
struct CsChunk {
    int base, first_tile;
};
template <int ABL, int RB>
__global__ __launch_bounds__(BLOCK) void k_csort(int n_rows, int n_chunks, const int *__restrict__ row_ptrs,
                                                 const CsChunk *__restrict__ cs, const uint4 *__restrict__ codes,
                                                 const double *__restrict__ svals, const double *__restrict__ x,
                                                 double *__restrict__ y, double *__restrict__ dot_partials, int xg)
{
    __shared__ __attribute__((aligned(16))) double prod[CS_TILE];
    __shared__ double slot[N_WAVES];
    const int chunk = xcd_chunk(blockIdx.x, xg);
    if (chunk >= n_chunks) return;
    const int tid = threadIdx.x;
    const int r0 = chunk * CHUNK_ROWS, r1 = min(r0 + CHUNK_ROWS, n_rows);
    const int nz0 = row_ptrs[r0], nz1 = row_ptrs[r1];
    const CsChunk ck = cs[chunk];
    const int row = r0 + tid * RPT;
    int rs[RPT + 1];
#pragma unroll
    for (int j = 0; j <= RPT; ++j) rs[j] = row_ptrs[min(row + j, r1)];
    double acc[RPT] = {0.0, 0.0};
    int tile = ck.first_tile;
    for (int t0 = nz0; t0 < nz1; t0 += CS_TILE, ++tile) {
        const int in_tile = min(CS_TILE, nz1 - t0);
        d2 va[CS_GROUPS][2];
        u4 cc[CS_GROUPS];
        const long tbase = (long)tile * CS_TILE;
#pragma unroll
        for (int g = 0; g < CS_GROUPS; ++g) {
            if (g * 1024 >= in_tile) break;
            cc[g] = __builtin_nontemporal_load(reinterpret_cast<const u4 *>(codes + (tbase >> 2) + g * BLOCK + tid));
#pragma unroll
            for (int p = 0; p < 2; ++p)
                va[g][p] = __builtin_nontemporal_load(
                    reinterpret_cast<const d2 *>(svals + tbase + ((g * 2 + p) * BLOCK + tid) * 2));
        }
#pragma unroll
        for (int g = 0; g < CS_GROUPS; ++g) {
            if (g * 1024 >= in_tile) break;
            const unsigned w[4] = {cc[g].x, cc[g].y, cc[g].z, cc[g].w};
            double xv[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) xv[j] = ABL == 1 ? (double)(w[j] >> 12) : x[ck.base + (int)(w[j] >> 12)];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const double v = (j & 1) ? va[g][j >> 1].y : va[g][j >> 1].x;
                // (padding entries of the last tile carry value 0 and a position past the tile's entries)
                prod[w[j] & 0xfffu] = v * xv[j];
            }
        }
        __syncthreads();
        const int t1 = t0 + CS_TILE;
#pragma unroll
        for (int j = 0; j < RPT; ++j) {
            const int kb = max(rs[j], t0), ke = min(rs[j + 1], t1);
            if (RB == 1) {
                for (int k = kb; k < ke; ++k) acc[j] += prod[k - t0];
            } else {
                for (int k = kb; k < ke; k += RB) {
                    double pv[RB];
#pragma unroll
                    for (int i = 0; i < RB; ++i) pv[i] = prod[min(k + i, ke - 1) - t0];
#pragma unroll
                    for (int i = 0; i < RB; ++i)
                        if (k + i < ke) acc[j] += pv[i];
                }
            }
        }
        __syncthreads();
    }
    double d = 0.0;
#pragma unroll
    for (int j = 0; j < RPT; ++j)
        if (row + j < r1) {
            y[row + j] = acc[j];
            d += x[row + j] * acc[j];
        }
    const double s = block_sum(d, slot);
    if (tid == 0) dot_partials[chunk] = s;
}

struct Dev {
    int n, nnz, nc;
    long *val_shift;
    double *vals_al;
    CsChunk *cs;
    uint4 *cs_codes;
    double *cs_vals;
    int *rp, *lines;
    S21Chunk *c21;
    uint4 *codes21, *codes16;
    WinChunk *wc;
    double *vals, *x0, *x1, *y, *part;
};

template <class L>
static void timeit(const char *name, const Dev &D, const std::vector<double> &yref, int reps, bool check, double bytes,
                   L launch)
{
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    launch(D.x0);
    CK(hipDeviceSynchronize());
    CK(hipGetLastError());
    long bad = -1;
    if (check) {
        std::vector<double> y(D.n);
        CK(hipMemcpy(y.data(), D.y, sizeof(double) * D.n, hipMemcpyDeviceToHost));
        bad = 0;
        for (int i = 0; i < D.n; ++i) bad += (y[i] != yref[i]);
    }
    for (int i = 0; i < 5; ++i) launch(i & 1 ? D.x1 : D.x0);
    CK(hipEventRecord(e0));
    for (int i = 0; i < reps; ++i) launch(i & 1 ? D.x1 : D.x0);
    CK(hipEventRecord(e1));
    CK(hipEventSynchronize(e1));
    float ms;
    CK(hipEventElapsedTime(&ms, e0, e1));
    ms /= reps;
    const double csr = 12.0 * D.nnz + 20.0 * D.n + 4;
    printf("%-52s b2b %7.1f us  moved %6.1f MB = %5.1f%% of 8 TB/s   CSR-equivalent %5.1f%%   mismatches %ld\n", name,
           1e3 * ms, bytes / 1e6, 100.0 * bytes / (ms * 1e-3) / 8e12, 100.0 * csr / (ms * 1e-3) / 8e12, bad);
    fflush(stdout);
}

int main(int argc, char **argv)
{
    if (argc < 2) {
        printf("usage: win_tune pattern.bin [reps]\n");
        return 2;
    }
    const int reps = argc > 2 ? atoi(argv[2]) : 50;
    FILE *f = fopen(argv[1], "rb");
    if (!f) {
        printf("cannot open %s\n", argv[1]);
        return 2;
    }
    int hdr[2];
    if (fread(hdr, 4, 2, f) != 2) return 2;
    Dev D{};
    D.n = hdr[0];
    D.nnz = hdr[1];
    std::vector<int> rp(D.n + 1), cols(D.nnz);
    if (fread(rp.data(), 4, rp.size(), f) != rp.size() || fread(cols.data(), 4, cols.size(), f) != cols.size()) return 2;
    fclose(f);
    D.nc = (D.n + CHUNK_ROWS - 1) / CHUNK_ROWS;
    std::vector<double> vals((size_t)D.nnz + 8192, 0.0), x0(D.n + 2), x1(D.n + 2), yref(D.n);
    for (int i = 0; i < D.n; ++i) {
        x0[i] = sin(0.001 * i) + 0.5;
        x1[i] = cos(0.002 * i) - 0.25;
    }
    for (int r = 0; r < D.n; ++r) {
        double s = 0;
        for (int k = rp[r]; k < rp[r + 1]; ++k) {
            vals[k] = cols[k] == r ? (rp[r + 1] - rp[r]) + 1e-3 * (1 + r % 7) : -1.0 - 1e-3 * ((k * 7) % 13);
            s += vals[k] * x0[cols[k]];
        }
        yref[r] = s;
    }
    // ---- layouts ----
    std::vector<S21Chunk> c21(D.nc);
    std::vector<WinChunk> wc(D.nc);
    std::vector<uint64_t> w21;   // pairs of 64-bit halves
    std::vector<uint32_t> w16;   // 4 per word
    std::vector<int> lines;
    long max_lines = 0, sum_lines = 0, sum_distinct = 0, span21 = 0;
    std::vector<int> tmp, slot_of;
    for (int c = 0; c < D.nc; ++c) {
        const int a = rp[c * CHUNK_ROWS], b = rp[std::min(D.n, (c + 1) * CHUNK_ROWS)];
        const int t_first = a & ~3;
        int lo = INT32_MAX, hi = 0;
        for (int k = a; k < b; ++k) {
            lo = std::min(lo, cols[k]);
            hi = std::max(hi, cols[k]);
        }
        if (a == b) lo = hi = 0;
        span21 = std::max<long>(span21, hi - lo);
        // 21-bit layout: tiles of 3072 = 2 groups x (3 pairs x 256 lanes x 2)
        c21[c].base = lo;
        c21[c].word_off = (int)(w21.size() / 2);
        const int n_tiles21 = (b - t_first + S21_TILE - 1) / S21_TILE;
        for (int t = 0; t < n_tiles21; ++t)
            for (int g = 0; g < S21_GROUPS; ++g)
                for (int lane = 0; lane < BLOCK; ++lane) {
                    unsigned long long v[6];
                    for (int k = 0; k < 3; ++k)
                        for (int h = 0; h < 2; ++h) {
                            const long e = (long)t_first + (long)t * S21_TILE + ((g * 3 + k) * BLOCK + lane) * 2 + h;
                            v[2 * k + h] = (e >= a && e < b) ? (unsigned long long)(cols[e] - lo) : 0ull;
                        }
                    const unsigned long long l64 = v[0] | (v[1] << 21) | (v[2] << 42) | (v[3] << 63);
                    const unsigned long long h64 = (v[3] >> 1) | (v[4] << 20) | (v[5] << 41);
                    w21.push_back(l64);
                    w21.push_back(h64);
                }
        // window layout: distinct lines, 16-bit local codes, groups of 2048 = 4 pairs x 256 lanes x 2
        tmp.clear();
        for (int k = a; k < b; ++k) tmp.push_back(cols[k] >> 3);
        std::sort(tmp.begin(), tmp.end());
        tmp.erase(std::unique(tmp.begin(), tmp.end()), tmp.end());
        {
            std::vector<int> dc(cols.begin() + a, cols.begin() + b);
            std::sort(dc.begin(), dc.end());
            sum_distinct += std::unique(dc.begin(), dc.end()) - dc.begin();
        }
        wc[c].line_off = (int)lines.size();
        wc[c].n_lines = (int)tmp.size();
        wc[c].word_off = (int)(w16.size() / 4);
        wc[c].pad = 0;
        lines.insert(lines.end(), tmp.begin(), tmp.end());
        max_lines = std::max<long>(max_lines, tmp.size());
        sum_lines += tmp.size();
        const int n_groups = (b - t_first + 2047) / 2048;
        for (int g = 0; g < n_groups; ++g)
            for (int lane = 0; lane < BLOCK; ++lane)
                for (int k = 0; k < 4; ++k) {
                    uint32_t word = 0;
                    for (int h = 0; h < 2; ++h) {
                        const long e = (long)t_first + (long)g * 2048 + (k * BLOCK + lane) * 2 + h;
                        uint32_t code = 0;
                        if (e >= a && e < b) {
                            const int ln = cols[e] >> 3;
                            const int s = (int)(std::lower_bound(tmp.begin(), tmp.end(), ln) - tmp.begin());
                            code = (uint32_t)(s * 8 + (cols[e] & 7));
                        }
                        word |= code << (16 * h);
                    }
                    w16.push_back(word);
                }
    }
    const bool fits16 = max_lines * 8 <= 65536;
    printf("%s: rows %d nnz %d (%.2f per row); per chunk: %.0f entries, %.0f distinct columns on %.0f lines (max %ld lines = "
           "%ld KB of LDS); widest span %ld\n",
           argv[1], D.n, D.nnz, (double)D.nnz / D.n, (double)D.nnz / D.nc, (double)sum_distinct / D.nc,
           (double)sum_lines / D.nc, max_lines, max_lines * 64 / 1024, span21);
    if (!fits16 || span21 >= (1 << 21)) {
        printf("layout limits exceeded\n");
        return 1;
    }
    const double bytes21 = 8.0 * D.nnz + 8.0 * w21.size() + 4.0 * D.n + 8.0 * D.nc + 16.0 * D.n;
    const double bytes16 = 8.0 * D.nnz + 4.0 * w16.size() + 4.0 * lines.size() + 64.0 * lines.size() + 4.0 * D.n +
                           16.0 * D.nc + 8.0 * D.n;  // (x through the lines of the windows instead of 8 N)
    CK(hipMalloc(&D.rp, 4 * (D.n + 1)));
    CK(hipMalloc(&D.lines, 4 * (lines.size() + 64)));
    CK(hipMalloc(&D.c21, sizeof(S21Chunk) * D.nc));
    CK(hipMalloc(&D.wc, sizeof(WinChunk) * D.nc));
    CK(hipMalloc(&D.codes21, 8 * (w21.size() + 4096)));
    CK(hipMalloc(&D.codes16, 4 * (w16.size() + 4096)));
    CK(hipMalloc(&D.vals, 8 * vals.size()));
    CK(hipMalloc(&D.x0, 8 * (D.n + 2)));
    CK(hipMalloc(&D.x1, 8 * (D.n + 2)));
    CK(hipMalloc(&D.y, 8 * (D.n + 2)));
    CK(hipMalloc(&D.part, 8 * (D.nc + 64)));
    CK(hipMemset(D.codes21, 0, 8 * (w21.size() + 4096)));
    CK(hipMemset(D.codes16, 0, 4 * (w16.size() + 4096)));
    CK(hipMemcpy(D.rp, rp.data(), 4 * rp.size(), hipMemcpyHostToDevice));
    CK(hipMemcpy(D.lines, lines.data(), 4 * lines.size(), hipMemcpyHostToDevice));
    CK(hipMemcpy(D.c21, c21.data(), sizeof(S21Chunk) * D.nc, hipMemcpyHostToDevice));
    CK(hipMemcpy(D.wc, wc.data(), sizeof(WinChunk) * D.nc, hipMemcpyHostToDevice));
    CK(hipMemcpy(D.codes21, w21.data(), 8 * w21.size(), hipMemcpyHostToDevice));
    CK(hipMemcpy(D.codes16, w16.data(), 4 * w16.size(), hipMemcpyHostToDevice));
    CK(hipMemcpy(D.vals, vals.data(), 8 * vals.size(), hipMemcpyHostToDevice));
    CK(hipMemcpy(D.x0, x0.data(), 8 * x0.size(), hipMemcpyHostToDevice));
    CK(hipMemcpy(D.x1, x1.data(), 8 * x1.size(), hipMemcpyHostToDevice));
    // ---- chunk-aligned copy of the values (every chunk's first tile on a 128-byte line) ----
    {
        std::vector<long> shift(D.nc);
        std::vector<double> val((size_t)D.nnz + 16L * D.nc + 8192, 0.0);
        long at = 0;
        for (int c = 0; c < D.nc; ++c) {
            const int a = rp[c * CHUNK_ROWS], b = rp[std::min(D.n, (c + 1) * CHUNK_ROWS)];
            const int t_first = a & ~3;
            at = (at + 15) / 16 * 16;
            shift[c] = at - t_first;
            for (int k = a; k < b; ++k) val[(size_t)(k + shift[c])] = vals[k];
            at += b - t_first;
        }
        CK(hipMalloc(&D.val_shift, 8 * D.nc));
        CK(hipMalloc(&D.vals_al, 8 * val.size()));
        CK(hipMemcpy(D.val_shift, shift.data(), 8 * D.nc, hipMemcpyHostToDevice));
        CK(hipMemcpy(D.vals_al, val.data(), 8 * val.size(), hipMemcpyHostToDevice));
    }
    // ---- column-sorted tiles ----
    std::vector<CsChunk> cs(D.nc);
    std::vector<uint32_t> cs_codes;
    std::vector<double> cs_vals;
    long cs_lines = 0, cs_instr = 0;
    bool cs_ok = true;
    {
        std::vector<int> ord;
        for (int c = 0; c < D.nc; ++c) {
            const int a = rp[c * CHUNK_ROWS], b = rp[std::min(D.n, (c + 1) * CHUNK_ROWS)];
            int lo = INT32_MAX, hi = 0;
            for (int k = a; k < b; ++k) {
                lo = std::min(lo, cols[k]);
                hi = std::max(hi, cols[k]);
            }
            if (a == b) lo = hi = 0;
            if (hi - lo >= (1 << 20)) cs_ok = false;
            cs[c].base = lo;
            cs[c].first_tile = (int)(cs_vals.size() / CS_TILE);
            for (int t0 = a; t0 < b; t0 += CS_TILE) {
                const int m = std::min(CS_TILE, b - t0);
                ord.resize(m);
                for (int i = 0; i < m; ++i) ord[i] = i;
                std::stable_sort(ord.begin(), ord.end(), [&](int p, int q) { return cols[t0 + p] < cols[t0 + q]; });
                const size_t vb = cs_vals.size(), cb = cs_codes.size();
                cs_vals.resize(vb + CS_TILE, 0.0);
                cs_codes.resize(cb + CS_TILE, 0u);
                for (int g = 0; g < CS_GROUPS; ++g)
                    for (int j = 0; j < 4; ++j) {
                        for (int wv = 0; wv < 4; ++wv) {  // lines one gather instruction touches
                            int prev = -1;
                            bool any = false;
                            for (int l = 0; l < 64; ++l) {
                                const int si = (g * 4 + j) * BLOCK + wv * 64 + l;
                                if (si >= m) continue;
                                any = true;
                                const int ln = cols[t0 + ord[si]] >> 3;
                                if (ln != prev) ++cs_lines;
                                prev = ln;
                            }
                            cs_instr += any;
                        }
                        for (int lane = 0; lane < BLOCK; ++lane) {
                            const int si = (g * 4 + j) * BLOCK + lane;  // sorted index
                            uint32_t code;
                            double v = 0.0;
                            if (si < m) {
                                code = ((uint32_t)(cols[t0 + ord[si]] - lo) << 12) | (uint32_t)ord[si];
                                v = vals[t0 + ord[si]];
                            } else {
                                code = (uint32_t)std::min(CS_TILE - 1, m + (si - m) % std::max(1, CS_TILE - m));
                                if (m == CS_TILE) code = 0;  // (never reached: a full tile has no padding)
                            }
                            cs_codes[cb + (size_t)(g * BLOCK + lane) * 4 + j] = code;
                            cs_vals[vb + (size_t)((g * 2 + (j >> 1)) * BLOCK + lane) * 2 + (j & 1)] = v;
                        }
                    }
            }
        }
    }
    printf("column-sorted tiles of %d: %.1f lines per gather instruction (product layout: see above), eligible %d\n",
           CS_TILE, (double)cs_lines / std::max(1L, cs_instr), (int)cs_ok);
    const double bytes_cs = 12.0 * cs_vals.size() + 4.0 * D.n + 8.0 * D.nc + 16.0 * D.n;
    CK(hipMalloc(&D.cs, sizeof(CsChunk) * D.nc));
    CK(hipMalloc(&D.cs_codes, 4 * (cs_codes.size() + 4096)));
    CK(hipMalloc(&D.cs_vals, 8 * (cs_vals.size() + 4096)));
    CK(hipMemcpy(D.cs, cs.data(), sizeof(CsChunk) * D.nc, hipMemcpyHostToDevice));
    CK(hipMemcpy(D.cs_codes, cs_codes.data(), 4 * cs_codes.size(), hipMemcpyHostToDevice));
    CK(hipMemcpy(D.cs_vals, cs_vals.data(), 8 * cs_vals.size(), hipMemcpyHostToDevice));
    const int win_doubles = (int)max_lines * 8;
    auto grid_for = [&](int xg) {
        const int q = 8 * xg;
        return (D.nc + q - 1) / q * q;
    };
#define RUNW(GROUPS, ABL, RB, MINW, XG)                                                                               \
    do {                                                                                                              \
        const size_t shm = sizeof(double) * ((size_t)GROUPS * 2048 + win_doubles);                                    \
        CK(hipFuncSetAttribute(reinterpret_cast<const void *>(&k_win<GROUPS, ABL, RB, MINW>),                         \
                               hipFuncAttributeMaxDynamicSharedMemorySize, (int)shm));                                \
        char nm[160];                                                                                                 \
        snprintf(nm, sizeof(nm), "window: groups%d abl%d rowbatch%d minw%d xcdgroup%d lds %zu KB", GROUPS, ABL, RB,   \
                 MINW, XG, shm / 1024);                                                                               \
        timeit(nm, D, yref, reps, ABL == 0 || ABL == 3, bytes16, [&](const double *x) {                               \
            hipLaunchKernelGGL((k_win<GROUPS, ABL, RB, MINW>), dim3(grid_for(XG)), dim3(BLOCK), shm, 0, D.n, D.nc,    \
                               D.rp, D.wc, D.lines, D.codes16, D.vals, x, D.y, D.part, XG, win_doubles);              \
        });                                                                                                           \
    } while (0)
#define RUNCS(ABL, RB, XG)                                                                                             \
    do {                                                                                                              \
        char nm[160];                                                                                                 \
        snprintf(nm, sizeof(nm), "column-sorted tiles: abl%d rowbatch%d xcdgroup%d", ABL, RB, XG);                    \
        if (cs_ok)                                                                                                    \
            timeit(nm, D, yref, reps, ABL == 0, bytes_cs, [&](const double *x) {                                      \
                hipLaunchKernelGGL((k_csort<ABL, RB>), dim3(grid_for(XG)), dim3(BLOCK), 0, 0, D.n, D.nc, D.rp, D.cs,   \
                                   D.cs_codes, D.cs_vals, x, D.y, D.part, XG);                                        \
            });                                                                                                       \
    } while (0)
    const bool run_window = getenv("WIN_TUNE_WINDOW") != nullptr;
    for (int pass = 0; pass < 2; ++pass) {
        timeit("product: k_spmv_stream21 (21-bit columns, global gathers)", D, yref, reps, true, bytes21,
               [&](const double *x) {
                   hipLaunchKernelGGL((k_stream21<false, false>), dim3(grid_for(4)), dim3(BLOCK), 0, 0, D.n, D.nc, D.rp,
                                      D.c21, D.codes21, D.vals, x, D.y, D.part, nullptr);
               });
        timeit("product with 4 lines per gather instruction (wrong on purpose)", D, yref, reps, false, bytes21,
               [&](const double *x) {
                   hipLaunchKernelGGL((k_stream21<true, false>), dim3(grid_for(4)), dim3(BLOCK), 0, 0, D.n, D.nc, D.rp,
                                      D.c21, D.codes21, D.vals, x, D.y, D.part, nullptr);
               });
        timeit("product on chunk-aligned values", D, yref, reps, true, bytes21, [&](const double *x) {
            hipLaunchKernelGGL((k_stream21<false, true>), dim3(grid_for(4)), dim3(BLOCK), 0, 0, D.n, D.nc, D.rp, D.c21,
                               D.codes21, D.vals_al, x, D.y, D.part, D.val_shift);
        });
        timeit("product on chunk-aligned values, 4 lines per gather (wrong)", D, yref, reps, false, bytes21,
               [&](const double *x) {
                   hipLaunchKernelGGL((k_stream21<true, true>), dim3(grid_for(4)), dim3(BLOCK), 0, 0, D.n, D.nc, D.rp,
                                      D.c21, D.codes21, D.vals_al, x, D.y, D.part, D.val_shift);
               });
        timeit("product, loop rotated (next tile's loads before the row sums)", D, yref, reps, true, bytes21,
               [&](const double *x) {
                   hipLaunchKernelGGL((k_stream21_rot<1>), dim3(grid_for(4)), dim3(BLOCK), 0, 0, D.n, D.nc, D.rp, D.c21,
                                      D.codes21, D.vals, x, D.y, D.part);
               });
#define RUNR(RB, G, ROT, MINW)                                                                                        \
    timeit("product variant: rowbatch" #RB " groups" #G " rotated" #ROT " minw" #MINW, D, yref, reps, true, bytes21,  \
           [&](const double *x) {                                                                                     \
               hipLaunchKernelGGL((k_stream21_rot<RB, G, ROT, MINW>), dim3(grid_for(4)), dim3(BLOCK), 0, 0, D.n, D.nc, \
                                  D.rp, D.c21, D.codes21, D.vals, x, D.y, D.part);                                    \
           })
        RUNR(1, 1, false, 1);
        RUNR(1, 1, true, 1);
        RUNR(1, 1, false, 2);
        RUNR(1, 1, true, 2);
        RUNR(4, 1, false, 2);
        RUNR(1, 2, false, 2);
        RUNR(1, 3, false, 1);
        RUNR(1, 3, true, 1);
        RUNR(1, 4, false, 1);
        RUNR(1, 4, true, 1);
        RUNR(1, 5, false, 1);
        RUNR(1, 6, false, 1);
        RUNR(4, 4, false, 1);
        timeit("product, loop rotated, rowbatch4", D, yref, reps, true, bytes21, [&](const double *x) {
            hipLaunchKernelGGL((k_stream21_rot<4>), dim3(grid_for(4)), dim3(BLOCK), 0, 0, D.n, D.nc, D.rp, D.c21,
                               D.codes21, D.vals, x, D.y, D.part);
        });
        RUNCS(0, 1, 4);
        RUNCS(0, 4, 4);
        RUNCS(0, 4, 1);
        RUNCS(0, 4, 16);
        RUNCS(1, 4, 4);
        if (run_window) {
            RUNW(1, 0, 1, 1, 4);
            RUNW(2, 0, 1, 1, 4);
            RUNW(1, 0, 4, 1, 4);
            RUNW(2, 0, 4, 1, 4);
            RUNW(1, 0, 4, 1, 1);
            RUNW(1, 0, 4, 1, 16);
            RUNW(1, 0, 4, 2, 4);
            RUNW(1, 3, 4, 1, 4);
            RUNW(1, 1, 4, 1, 4);
            RUNW(1, 2, 4, 1, 4);
        }
    }
    return 0;
}
