// csr_tune.hip -- development harness (not product): variants of the CSR-stream SpMV on a pattern read
// from a file (tools/dump_pattern.py writes it: int32 n, int32 nnz, row_ptrs[n+1], cols[nnz]), e.g. the
// Voronoi (polyhedral) proxy in its RCM numbering.  Values are synthetic; every variant is checked against
// the host's left-to-right row sums.
//   hipcc --offload-arch=gfx950 -O3 -ffp-contract=off tools/csr_tune.hip -o tools/bin/csr_tune
//   tools/bin/csr_tune pattern.bin [reps]
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
typedef int i4 __attribute__((ext_vector_type(4)));
typedef unsigned short us4 __attribute__((ext_vector_type(4)));
constexpr int BLOCK = 256, N_WAVES = 4, CHUNK_ROWS = 512, RPT = 2;

template <int G = 4>
__device__ __forceinline__ int xcd_chunk(int block)
{
    if (G == 0) return block;
    const int slot = block / 8, xcd = block % 8;
    return (slot / G) * (8 * G) + xcd * G + slot % G;
}
template <bool NT, class T>
__device__ __forceinline__ T ldg(const T *p)
{
    if (NT) return __builtin_nontemporal_load(p);
    return *p;
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

// COLS: 0 = int32 columns, 1 = 16-bit codes relative to the chunk's smallest column
// SKIP: wave-uniform skip of the load groups that lie entirely beyond the chunk's last entry
// ABL:  0 = full kernel, 1 = no gather (x := column as double; wrong result on purpose),
//       2 = no LDS / row phase (per-thread sum of its products; wrong result on purpose)
// NT: non-temporal loads of values and columns; RB: LDS reads of the row phase issued RB at a time (the
// adds stay in order); XG: consecutive chunks per XCD (0 = chunk = block)
template <int TILE, int COLS, bool SKIP, int ABL, int MINW, bool NT = false, int RB = 1, int XG = 4>
__global__ __launch_bounds__(BLOCK, MINW) void k_spmv(int n_rows, int n_chunks, const int *__restrict__ row_ptrs,
                                                      const int *__restrict__ cols,
                                                      const unsigned short *__restrict__ cols16,
                                                      const int *__restrict__ chunk_base,
                                                      const double *__restrict__ vals, const double *__restrict__ x,
                                                      double *__restrict__ y, double *__restrict__ dot_partials)
{
    __shared__ __attribute__((aligned(16))) double prod[TILE];
    __shared__ double slot[N_WAVES];
    const int chunk = xcd_chunk<XG>(blockIdx.x);
    if (chunk >= n_chunks) return;
    const int tid = threadIdx.x;
    const int r0 = chunk * CHUNK_ROWS;
    const int r1 = min(r0 + CHUNK_ROWS, n_rows);
    const int nz0 = row_ptrs[r0], nz1 = row_ptrs[r1];
    const int base = COLS == 1 ? chunk_base[chunk] : 0;
    const int row = r0 + tid * RPT;
    int rs[RPT + 1];
#pragma unroll
    for (int j = 0; j <= RPT; ++j) rs[j] = row_ptrs[min(row + j, r1)];
    double acc[RPT] = {0.0, 0.0};
    double tsum = 0.0;
    constexpr int GROUPS = TILE / (BLOCK * 4);
    for (int t0 = nz0 & ~3; t0 < nz1; t0 += TILE) {
        d2 va[GROUPS], vb[GROUPS];
        i4 cc[GROUPS];
#pragma unroll
        for (int g = 0; g < GROUPS; ++g) {
            if (SKIP && t0 + g * BLOCK * 4 >= nz1) break;
            const int e = t0 + (g * BLOCK + tid) * 4;
            const int ec = e < nz1 ? e : t0;
            va[g] = ldg<NT>(reinterpret_cast<const d2 *>(vals + ec));
            vb[g] = ldg<NT>(reinterpret_cast<const d2 *>(vals + ec + 2));
            if (COLS == 1) {
                const us4 c = ldg<NT>(reinterpret_cast<const us4 *>(cols16 + ec));
                // (the up to 3 entries before nz0 carry the previous chunk's codes: keep them in range)
                cc[g].x = min(base + c.x, n_rows - 1);
                cc[g].y = min(base + c.y, n_rows - 1);
                cc[g].z = min(base + c.z, n_rows - 1);
                cc[g].w = min(base + c.w, n_rows - 1);
            } else {
                cc[g] = ldg<NT>(reinterpret_cast<const i4 *>(cols + ec));
            }
        }
#pragma unroll
        for (int g = 0; g < GROUPS; ++g) {
            if (SKIP && t0 + g * BLOCK * 4 >= nz1) break;
            double x0, x1, x2, x3;
            if (ABL == 1) {
                x0 = (double)cc[g].x; x1 = (double)cc[g].y; x2 = (double)cc[g].z; x3 = (double)cc[g].w;
            } else {
                x0 = x[cc[g].x]; x1 = x[cc[g].y]; x2 = x[cc[g].z]; x3 = x[cc[g].w];
            }
            d2 p0, p1;
            p0.x = va[g].x * x0;
            p0.y = va[g].y * x1;
            p1.x = vb[g].x * x2;
            p1.y = vb[g].y * x3;
            if (ABL == 2) {
                tsum += p0.x + p0.y + p1.x + p1.y;
            } else {
                const int le = (g * BLOCK + tid) * 4;
                *reinterpret_cast<d2 *>(prod + le) = p0;
                *reinterpret_cast<d2 *>(prod + le + 2) = p1;
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

// One wavefront per 128 rows, no workgroup barrier: a wave streams ITS rows' entries (contiguous in CSR)
// through its own quarter of the LDS tile; waves of a workgroup run independently (the loads of one wave
// overlap the row sums of another without waiting for the slowest).  WTILE entries per wave and pass.
template <int WTILE, int COLS>
__global__ __launch_bounds__(BLOCK) void k_spmv_wave(int n_rows, int n_chunks, const int *__restrict__ row_ptrs,
                                                     const int *__restrict__ cols,
                                                     const unsigned short *__restrict__ cols16,
                                                     const int *__restrict__ chunk_base,
                                                     const double *__restrict__ vals, const double *__restrict__ x,
                                                     double *__restrict__ y, double *__restrict__ dot_partials)
{
    __shared__ __attribute__((aligned(16))) double prod[N_WAVES][WTILE];
    __shared__ double slot[N_WAVES];
    const int chunk = xcd_chunk<4>(blockIdx.x);
    if (chunk >= n_chunks) return;
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int r0 = chunk * CHUNK_ROWS;
    const int r1 = min(r0 + CHUNK_ROWS, n_rows);
    const int base = COLS == 1 ? chunk_base[chunk] : 0;
    const int wr0 = min(r0 + wv * 128, r1), wr1 = min(wr0 + 128, r1);
    const int nz0 = row_ptrs[wr0], nz1 = row_ptrs[wr1];
    const int row = r0 + tid * RPT;
    int rs[RPT + 1];
#pragma unroll
    for (int j = 0; j <= RPT; ++j) rs[j] = row_ptrs[min(row + j, r1)];
    double acc[RPT] = {0.0, 0.0};
    double *my = prod[wv];
    constexpr int GROUPS = WTILE / (64 * 4);
    for (int t0 = nz0 & ~3; t0 < nz1; t0 += WTILE) {
        d2 va[GROUPS], vb[GROUPS];
        i4 cc[GROUPS];
#pragma unroll
        for (int g = 0; g < GROUPS; ++g) {
            if (t0 + g * 64 * 4 >= nz1) break;
            const int e = t0 + (g * 64 + lane) * 4;
            const int ec = e < nz1 ? e : t0;
            va[g] = *reinterpret_cast<const d2 *>(vals + ec);
            vb[g] = *reinterpret_cast<const d2 *>(vals + ec + 2);
            if (COLS == 1) {
                const us4 c = *reinterpret_cast<const us4 *>(cols16 + ec);
                // (the up to 3 entries before nz0 carry the previous chunk's codes: keep them in range)
                cc[g].x = min(base + c.x, n_rows - 1);
                cc[g].y = min(base + c.y, n_rows - 1);
                cc[g].z = min(base + c.z, n_rows - 1);
                cc[g].w = min(base + c.w, n_rows - 1);
            } else {
                cc[g] = *reinterpret_cast<const i4 *>(cols + ec);
            }
        }
#pragma unroll
        for (int g = 0; g < GROUPS; ++g) {
            if (t0 + g * 64 * 4 >= nz1) break;
            const double x0 = x[cc[g].x], x1 = x[cc[g].y], x2 = x[cc[g].z], x3 = x[cc[g].w];
            d2 p0, p1;
            p0.x = va[g].x * x0;
            p0.y = va[g].y * x1;
            p1.x = vb[g].x * x2;
            p1.y = vb[g].y * x3;
            const int le = (g * 64 + lane) * 4;
            *reinterpret_cast<d2 *>(my + le) = p0;
            *reinterpret_cast<d2 *>(my + le + 2) = p1;
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        const int t1 = t0 + WTILE;
#pragma unroll
        for (int j = 0; j < RPT; ++j) {
            const int kb = max(rs[j], t0), ke = min(rs[j + 1], t1);
            for (int k = kb; k < ke; ++k) acc[j] += my[k - t0];
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
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

// Row-group SpMV: LPR consecutive lanes work on the consecutive entries of ONE row (64 / LPR rows per wavefront
// and step), values and columns read straight from the CSR arrays (coalesced within the row, no padding), the
// row sum formed strictly left to right by passing the running sum from lane to lane -- no LDS staging of the
// products, no workgroup barrier in the main loop.  The gather of one instruction covers all entries of 64 / LPR
// neighbouring rows (row-major locality, as in the CSR-stream kernel).  y goes through LDS once per chunk for the
// fused dot's canonical tree.
template <int LPR>
__global__ __launch_bounds__(BLOCK) void k_spmv_rg(int n_rows, int n_chunks, const int *__restrict__ row_ptrs,
                                                   const int *__restrict__ cols, const double *__restrict__ vals,
                                                   const double *__restrict__ x, double *__restrict__ y,
                                                   double *__restrict__ dot_partials)
{
    __shared__ double ys[CHUNK_ROWS];
    __shared__ double slot[N_WAVES];
    const int chunk = xcd_chunk<4>(blockIdx.x);
    if (chunk >= n_chunks) return;
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    constexpr int G = 64 / LPR;  // rows per wavefront and step
    const int g = lane / LPR, l = lane % LPR;
    const int r0 = chunk * CHUNK_ROWS, r1 = min(r0 + CHUNK_ROWS, n_rows);
    const int w0 = r0 + wv * 128, w1 = min(w0 + 128, r1);
    for (int base = w0; base < w1; base += G) {
        const int row = base + g;
        const bool valid = row < w1;
        const int k0 = valid ? row_ptrs[row] : 0, k1 = valid ? row_ptrs[row + 1] : 0;
        double carry = 0.0;
        for (int k = k0; __any(k < k1); k += LPR) {
            const int e = k + l;
            const bool has = e < k1;
            double p = 0.0;
            if (has) p = vals[e] * x[cols[e]];
            // running sum, strictly left to right: lane i takes lane i-1's sum and adds its own product
            double s = l == 0 ? (has ? carry + p : carry) : 0.0;
#pragma unroll
            for (int i = 1; i < LPR; ++i) {
                const double t = __shfl_up(s, 1, LPR);
                if (l == i) s = has ? t + p : t;
            }
            carry = __shfl(s, LPR - 1, LPR);
        }
        if (valid && l == 0) {
            y[row] = carry;
            ys[row - r0] = carry;
        }
    }
    __syncthreads();
    const int row = r0 + tid * RPT;
    double d = 0.0;
#pragma unroll
    for (int j = 0; j < RPT; ++j)
        if (row + j < r1) d += x[row + j] * ys[tid * RPT + j];
    const double sm = block_sum(d, slot);
    if (tid == 0) dot_partials[chunk] = sm;
}

struct Dev {
    int n, nnz, nc, grid;
    int *rp, *cols, *base;
    unsigned short *cols16;
    double *vals, *x0, *x1, *y, *part;
    bool has16;
};

template <class L>
static void timeit(const char *name, const Dev &D, const std::vector<double> &yref, int reps, bool check, L launch)
{
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    launch(D.x0);
    CK(hipDeviceSynchronize());
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
    const double bytes = 12.0 * D.nnz + 20.0 * D.n + 4;
    printf("%-56s b2b %7.1f us  %5.1f%% of 8 TB/s (CSR bytes)  mismatches %ld\n", name, 1e3 * ms,
           100.0 * bytes / (ms * 1e-3) / 8e12, bad);
    fflush(stdout);
}

int main(int argc, char **argv)
{
    if (argc < 2) {
        printf("usage: csr_tune pattern.bin [reps]\n");
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
    D.grid = ((D.nc + 31) / 32) * 32;
    std::vector<double> vals(D.nnz), x0(D.n), x1(D.n), yref(D.n);
    for (int i = 0; i < D.n; ++i) {
        x0[i] = sin(0.001 * i) + 0.5;
        x1[i] = cos(0.002 * i) - 0.25;
    }
    int lmax = 0;
    for (int r = 0; r < D.n; ++r) {
        lmax = std::max(lmax, rp[r + 1] - rp[r]);
        double s = 0;
        for (int k = rp[r]; k < rp[r + 1]; ++k) {
            vals[k] = cols[k] == r ? (rp[r + 1] - rp[r]) + 1e-3 * (1 + r % 7) : -1.0 - 1e-3 * ((k * 7) % 13);
            s += vals[k] * x0[cols[k]];
        }
        yref[r] = s;
    }
    // 16-bit codes relative to the chunk's smallest column
    std::vector<int> base(D.nc);
    std::vector<unsigned short> c16(D.nnz);
    D.has16 = true;
    int worst = 0;
    for (int c = 0; c < D.nc; ++c) {
        const int a = rp[c * CHUNK_ROWS], b = rp[std::min(D.n, (c + 1) * CHUNK_ROWS)];
        int lo = INT32_MAX, hi = 0;
        for (int k = a; k < b; ++k) {
            lo = std::min(lo, cols[k]);
            hi = std::max(hi, cols[k]);
        }
        if (a == b) lo = 0;
        base[c] = lo;
        worst = std::max(worst, hi - lo);
        if (hi - lo > 65535) D.has16 = false;
        for (int k = a; k < b; ++k) c16[k] = (unsigned short)(cols[k] - lo);
    }
    printf("%s: rows %d nnz %d (%.2f per row, longest %d); widest column span of a chunk %d -> 16-bit codes %s\n",
           argv[1], D.n, D.nnz, (double)D.nnz / D.n, lmax, worst, D.has16 ? "possible" : "NOT possible");
    CK(hipMalloc(&D.rp, 4 * (D.n + 1)));
    CK(hipMalloc(&D.cols, 4 * ((size_t)D.nnz + 8192)));
    CK(hipMalloc(&D.cols16, 2 * ((size_t)D.nnz + 8192)));
    CK(hipMalloc(&D.base, 4 * D.nc));
    CK(hipMalloc(&D.vals, 8 * ((size_t)D.nnz + 8192)));
    CK(hipMalloc(&D.x0, 8 * (D.n + 2)));
    CK(hipMalloc(&D.x1, 8 * (D.n + 2)));
    CK(hipMalloc(&D.y, 8 * (D.n + 2)));
    CK(hipMalloc(&D.part, 8 * (D.nc + 64)));
    CK(hipMemset(D.cols, 0, 4 * ((size_t)D.nnz + 8192)));
    CK(hipMemset(D.cols16, 0, 2 * ((size_t)D.nnz + 8192)));
    CK(hipMemset(D.vals, 0, 8 * ((size_t)D.nnz + 8192)));
    CK(hipMemcpy(D.rp, rp.data(), 4 * rp.size(), hipMemcpyHostToDevice));
    CK(hipMemcpy(D.cols, cols.data(), 4 * cols.size(), hipMemcpyHostToDevice));
    CK(hipMemcpy(D.cols16, c16.data(), 2 * c16.size(), hipMemcpyHostToDevice));
    CK(hipMemcpy(D.base, base.data(), 4 * base.size(), hipMemcpyHostToDevice));
    CK(hipMemcpy(D.vals, vals.data(), 8 * vals.size(), hipMemcpyHostToDevice));
    CK(hipMemcpy(D.x0, x0.data(), 8 * x0.size(), hipMemcpyHostToDevice));
    CK(hipMemcpy(D.x1, x1.data(), 8 * x1.size(), hipMemcpyHostToDevice));

#define RUN(TILE, COLS, SKIP, ABL, MINW)                                                                             \
    if (COLS == 0 || D.has16)                                                                                        \
        timeit("tile" #TILE " cols" #COLS " skip" #SKIP " abl" #ABL " minw" #MINW, D, yref, reps, ABL == 0,          \
               [&](const double *x) {                                                                                \
                   hipLaunchKernelGGL((k_spmv<TILE, COLS, SKIP, ABL, MINW>), dim3(D.grid), dim3(BLOCK), 0, 0, D.n,    \
                                      D.nc, D.rp, D.cols, D.cols16, D.base, D.vals, x, D.y, D.part);                 \
               })
#define RUNX(TILE, NT, RB, XG)                                                                                      \
    timeit("tile" #TILE " nt" #NT " rowbatch" #RB " xcdgroup" #XG, D, yref, reps, true, [&](const double *x) {       \
        const int q = XG ? 8 * XG : 1;                                                                               \
        hipLaunchKernelGGL((k_spmv<TILE, 0, true, 0, 1, NT, RB, XG>), dim3((D.nc + q - 1) / q * q), dim3(BLOCK), 0,  \
                           0, D.n, D.nc,                                                                             \
                           D.rp, D.cols, D.cols16, D.base, D.vals, x, D.y, D.part);                                  \
    })
#define RUNW(WTILE, COLS)                                                                                            \
    if (COLS == 0 || D.has16)                                                                                        \
        timeit("per-wave tile" #WTILE " cols" #COLS, D, yref, reps, true, [&](const double *x) {                     \
            hipLaunchKernelGGL((k_spmv_wave<WTILE, COLS>), dim3(D.grid), dim3(BLOCK), 0, 0, D.n, D.nc, D.rp, D.cols, \
                               D.cols16, D.base, D.vals, x, D.y, D.part);                                            \
        })
#define RUNRG(LPR)                                                                                                   \
    timeit("row groups, " #LPR " lanes per row", D, yref, reps, true, [&](const double *x) {                          \
        hipLaunchKernelGGL((k_spmv_rg<LPR>), dim3(D.grid), dim3(BLOCK), 0, 0, D.n, D.nc, D.rp, D.cols, D.vals, x, D.y, \
                           D.part);                                                                                  \
    })
    for (int pass = 0; pass < 2; ++pass) {
        RUNRG(16);
        RUNRG(8);
        RUNRG(32);
        RUN(4096, 0, false, 0, 1);  // the product kernel
        RUN(2048, 0, true, 0, 1);
        RUNX(4096, false, 4, 4);
        RUNX(2048, false, 4, 4);
        RUNX(2048, false, 2, 4);
        RUNX(2048, false, 8, 4);
        RUNX(2048, true, 1, 4);
        RUNX(2048, true, 4, 4);
        RUNX(2048, false, 4, 0);
        RUNX(2048, false, 4, 1);
        RUNX(2048, false, 4, 2);
        RUNX(2048, false, 4, 8);
        RUNX(2048, false, 4, 16);
        RUN(4096, 1, true, 0, 1);
        RUN(2048, 1, true, 0, 1);
        RUN(4096, 0, true, 1, 1);
        RUN(4096, 0, true, 2, 1);
    }
    return 0;
}
