// sell_tune.hip -- development harness (not product): prototype of the index-compressed chunked
// ELL SpMV ("SELL-512 with 1-byte diagonal codes").  Per chunk of 512 rows: width w = longest row,
// values slot-major [w][512], one byte per (row, slot) naming an entry of the chunk's dictionary of
// (col - row) offsets, 255 = padding.  Compares with the plain slot-major ELL (int32 columns).
//   hipcc --offload-arch=gfx950 -O3 -ffp-contract=off tools/sell_tune.hip -o /tmp/sell_tune
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <climits>
#include <cstring>
#include <vector>

#define CK(x)                                                                            \
    do {                                                                                 \
        hipError_t e = (x);                                                              \
        if (e != hipSuccess) {                                                           \
            printf("HIP error %s at %s:%d\n", hipGetErrorString(e), __FILE__, __LINE__); \
            exit(1);                                                                     \
        }                                                                                \
    } while (0)

constexpr int BLOCK = 256;
constexpr int N_WAVES = 4;
constexpr int CHUNK = 512;
constexpr int N_XCD = 8;
constexpr int XCD_GROUP = 4;

__device__ __forceinline__ int xcd_chunk(int block)
{
    const int slot = block / N_XCD, xcd = block % N_XCD;
    return (slot / XCD_GROUP) * (N_XCD * XCD_GROUP) + xcd * XCD_GROUP + slot % XCD_GROUP;
}
static int xcd_grid(int n_chunks)
{
    constexpr int Q = N_XCD * XCD_GROUP;
    return ((n_chunks + Q - 1) / Q) * Q;
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

struct ChunkHdr {
    long val_off;   // doubles
    long code_off;  // bytes
    int dict_off;
    int dict_len;
    int width;
    int code_stride;  // bytes per thread (multiple of 16)
};

// VARIANT 0: dictionary in LDS.  VARIANT 1: dictionary read through global memory (L1/L2).
template <int VARIANT, int XCD>
__global__ __launch_bounds__(BLOCK) void k_sell(int n_rows, int n_chunks, const ChunkHdr *__restrict__ hdr,
                                                const int *__restrict__ dict,
                                                const uint8_t *__restrict__ codes,
                                                const double *__restrict__ vals,
                                                const double *__restrict__ x, double *__restrict__ y,
                                                double *__restrict__ part)
{
    __shared__ double slot[N_WAVES];
    __shared__ int sdict[256];
    const int chunk = XCD ? xcd_chunk(blockIdx.x) : (int)blockIdx.x;
    if (chunk >= n_chunks) return;
    const ChunkHdr h = hdr[chunk];
    const int t = threadIdx.x;
    if (VARIANT == 0) {
        if (t < h.dict_len) sdict[t] = dict[h.dict_off + t];
        __syncthreads();
    }
    const int *gd = dict + h.dict_off;
    const int row = chunk * CHUNK + 2 * t;
    const int nv = min(2, max(0, n_rows - row));
    double a0 = 0.0, a1 = 0.0;
    const double *v = vals + h.val_off + 2 * t;
    const uint8_t *c = codes + h.code_off + (long)t * h.code_stride;
    for (int s0 = 0; s0 < h.width; s0 += 8) {
        const uint4 cw = *reinterpret_cast<const uint4 *>(c + 2 * s0);
        const unsigned w4[4] = {cw.x, cw.y, cw.z, cw.w};
        double2 vv[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            const int s = min(s0 + k, h.width - 1);
            vv[k] = *reinterpret_cast<const double2 *>(v + (long)s * CHUNK);
        }
        double x0[8], x1[8];
        bool ok0[8], ok1[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            const unsigned pair = (w4[k / 2] >> (16 * (k & 1))) & 0xffffu;
            const unsigned c0 = pair & 0xffu, c1 = pair >> 8;
            ok0[k] = (s0 + k < h.width) && c0 != 255u;
            ok1[k] = (s0 + k < h.width) && c1 != 255u;
            const int d0 = VARIANT == 0 ? sdict[c0] : gd[c0 == 255u ? 0 : c0];
            const int d1 = VARIANT == 0 ? sdict[c1] : gd[c1 == 255u ? 0 : c1];
            x0[k] = ok0[k] ? x[row + d0] : 0.0;
            x1[k] = ok1[k] ? x[row + 1 + d1] : 0.0;
        }
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            if (ok0[k]) a0 = a0 + vv[k].x * x0[k];
            if (ok1[k]) a1 = a1 + vv[k].y * x1[k];
        }
    }
    if (nv == 2)
        *reinterpret_cast<double2 *>(y + row) = make_double2(a0, a1);
    else if (nv == 1)
        y[row] = a0;
    double d = 0.0;
    if (nv > 0) d += x[row] * a0;
    if (nv > 1) d += x[row + 1] * a1;
    const double sm = block_sum(d, slot);
    if (t == 0) part[chunk] = sm;
}

// plain slot-major ELL, int32 columns, same chunk-local planes (for an apples-to-apples comparison)
template <int XCD>
__global__ __launch_bounds__(BLOCK) void k_ell(int n_rows, int n_chunks, int width, const int *__restrict__ cols,
                                               const double *__restrict__ vals, const double *__restrict__ x,
                                               double *__restrict__ y, double *__restrict__ part)
{
    __shared__ double slot[N_WAVES];
    const int chunk = XCD ? xcd_chunk(blockIdx.x) : (int)blockIdx.x;
    if (chunk >= n_chunks) return;
    const int t = threadIdx.x;
    const int row = chunk * CHUNK + 2 * t;
    const int nv = min(2, max(0, n_rows - row));
    const long base = (long)chunk * CHUNK * width + 2 * t;
    double a0 = 0.0, a1 = 0.0;
    double2 vv[8];
    int2 cc[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) {
        const int s = min(k, width - 1);
        vv[k] = *reinterpret_cast<const double2 *>(vals + base + (long)s * CHUNK);
        cc[k] = *reinterpret_cast<const int2 *>(cols + base + (long)s * CHUNK);
        if (k >= width) cc[k].x = cc[k].y = -1;
    }
    double x0[8], x1[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) {
        x0[k] = cc[k].x >= 0 ? x[cc[k].x] : 0.0;
        x1[k] = cc[k].y >= 0 ? x[cc[k].y] : 0.0;
    }
#pragma unroll
    for (int k = 0; k < 8; ++k) {
        if (cc[k].x >= 0) a0 = a0 + vv[k].x * x0[k];
        if (cc[k].y >= 0) a1 = a1 + vv[k].y * x1[k];
    }
    if (nv == 2)
        *reinterpret_cast<double2 *>(y + row) = make_double2(a0, a1);
    else if (nv == 1)
        y[row] = a0;
    double d = 0.0;
    if (nv > 0) d += x[row] * a0;
    if (nv > 1) d += x[row + 1] * a1;
    const double sm = block_sum(d, slot);
    if (t == 0) part[chunk] = sm;
}

// VARIANT pid: one byte per ROW naming a row pattern (list of `width` offsets, INT_MIN = padding) in the
// chunk's pattern table (LDS).
constexpr int PID_TABLE = 2048;
template <int XCD>
__global__ __launch_bounds__(BLOCK) void k_sell_pid(int n_rows, int n_chunks, const ChunkHdr *__restrict__ hdr,
                                                    const int *__restrict__ dict,
                                                    const uint8_t *__restrict__ codes,
                                                    const double *__restrict__ vals,
                                                    const double *__restrict__ x, double *__restrict__ y,
                                                    double *__restrict__ part, const int *__restrict__ order = nullptr)
{
    __shared__ double slot[N_WAVES];
    __shared__ int stab[PID_TABLE];
    const int chunk = XCD == 2 ? order[blockIdx.x] : (XCD ? xcd_chunk(blockIdx.x) : (int)blockIdx.x);
    if (chunk < 0 || chunk >= n_chunks) return;
    const ChunkHdr h = hdr[chunk];
    const int t = threadIdx.x;
    for (int i = t; i < h.dict_len; i += BLOCK) stab[i] = dict[h.dict_off + i];
    __syncthreads();
    const int row = chunk * CHUNK + 2 * t;
    const int nv = min(2, max(0, n_rows - row));
    const unsigned short pp = *reinterpret_cast<const unsigned short *>(codes + h.code_off + 2 * t);
    const int p0 = (pp & 0xff) * h.width, p1 = (pp >> 8) * h.width;
    double a0 = 0.0, a1 = 0.0;
    const double *v = vals + h.val_off + 2 * t;
    for (int s0 = 0; s0 < h.width; s0 += 8) {
        double2 vv[8];
        int d0[8], d1[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            const int s = min(s0 + k, h.width - 1);
            vv[k] = *reinterpret_cast<const double2 *>(v + (long)s * CHUNK);
            d0[k] = (s0 + k < h.width) ? stab[p0 + s] : INT_MIN;
            d1[k] = (s0 + k < h.width) ? stab[p1 + s] : INT_MIN;
        }
        double x0[8], x1[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            x0[k] = d0[k] != INT_MIN ? x[row + d0[k]] : 0.0;
            x1[k] = d1[k] != INT_MIN ? x[row + 1 + d1[k]] : 0.0;
        }
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            if (d0[k] != INT_MIN) a0 = a0 + vv[k].x * x0[k];
            if (d1[k] != INT_MIN) a1 = a1 + vv[k].y * x1[k];
        }
    }
    if (nv == 2)
        *reinterpret_cast<double2 *>(y + row) = make_double2(a0, a1);
    else if (nv == 1)
        y[row] = a0;
    double d = 0.0;
    if (nv > 0) d += x[row] * a0;
    if (nv > 1) d += x[row + 1] * a1;
    const double sm = block_sum(d, slot);
    if (t == 0) part[chunk] = sm;
}

struct Csr {
    int n = 0, nnz = 0;
    std::vector<int> rp, cols;
    std::vector<double> vals;
};

static Csr poisson(int n)
{
    Csr A;
    const long N = (long)n * n * n;
    A.n = (int)N;
    A.rp.resize(N + 1);
    A.cols.reserve(7 * N);
    A.vals.reserve(7 * N);
    long e = 0;
    for (long c = 0; c < N; ++c) {
        const int i = c % n, j = (c / n) % n, k = c / ((long)n * n);
        A.rp[c] = (int)e;
        int nb = 0;
        auto add = [&](long col, double v) {
            A.cols.push_back((int)col);
            A.vals.push_back(v);
            ++e;
        };
        if (k > 0) { add(c - (long)n * n, -1.0); ++nb; }
        if (j > 0) { add(c - n, -1.0); ++nb; }
        if (i > 0) { add(c - 1, -1.0); ++nb; }
        const long dpos = e;
        add(c, 0.0);
        if (i < n - 1) { add(c + 1, -1.0); ++nb; }
        if (j < n - 1) { add(c + n, -1.0); ++nb; }
        if (k < n - 1) { add(c + (long)n * n, -1.0); ++nb; }
        A.vals[dpos] = nb + 1e-3 * (1.0 + (c % 7) / 7.0);
    }
    A.rp[N] = (int)e;
    A.nnz = (int)e;
    return A;
}

template <class F>
static void time_it(const char *name, const Csr &A, F launch, double *d_x0, double *d_x1, double *d_y,
                    const std::vector<double> &yref, int reps, double moved_bytes)
{
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    CK(hipMemset(d_y, 0, sizeof(double) * A.n));
    launch(d_x0);
    CK(hipDeviceSynchronize());
    std::vector<double> y(A.n);
    CK(hipMemcpy(y.data(), d_y, sizeof(double) * A.n, hipMemcpyDeviceToHost));
    long bad = 0;
    for (int i = 0; i < A.n; ++i) bad += (y[i] != yref[i]);
    for (int i = 0; i < 5; ++i) launch(i & 1 ? d_x1 : d_x0);
    CK(hipEventRecord(e0));
    for (int i = 0; i < reps; ++i) launch(i & 1 ? d_x1 : d_x0);
    CK(hipEventRecord(e1));
    CK(hipEventSynchronize(e1));
    float b2b;
    CK(hipEventElapsedTime(&b2b, e0, e1));
    b2b /= reps;
    const double bytes = 12.0 * A.nnz + 20.0 * A.n + 4;
    printf("%-40s b2b %7.1f us -> CSR-algorithmic %6.0f GB/s (%5.1f%% of 8 TB/s), moved %6.0f GB/s, mismatches %ld\n",
           name, 1e3 * b2b, bytes / (b2b * 1e-3) / 1e9, 100.0 * bytes / (b2b * 1e-3) / 8e12,
           moved_bytes / (b2b * 1e-3) / 1e9, bad);
    fflush(stdout);
}

int main(int argc, char **argv)
{
    const int n = argc > 1 ? atoi(argv[1]) : 216;
    const int reps = argc > 2 ? atoi(argv[2]) : 50;
    Csr A = poisson(n);
    printf("poisson %d^3: rows %d nnz %d\n", n, A.n, A.nnz);
    std::vector<double> x0(A.n), x1(A.n), yref(A.n);
    for (int i = 0; i < A.n; ++i) {
        x0[i] = sin(0.001 * i) + 0.5;
        x1[i] = cos(0.002 * i) - 0.25;
    }
    for (int r = 0; r < A.n; ++r) {
        double s = 0;
        for (int k = A.rp[r]; k < A.rp[r + 1]; ++k) s += A.vals[k] * x0[A.cols[k]];
        yref[r] = s;
    }
    const int nc = (A.n + CHUNK - 1) / CHUNK;
    // ---- build the compressed layout ----
    std::vector<ChunkHdr> hdr(nc);
    std::vector<int> dict;
    long val_len = 0, code_len = 0;
    for (int c = 0; c < nc; ++c) {
        int w = 0;
        std::vector<int> ds;
        for (int r = c * CHUNK; r < std::min(A.n, (c + 1) * CHUNK); ++r) {
            w = std::max(w, A.rp[r + 1] - A.rp[r]);
            for (int k = A.rp[r]; k < A.rp[r + 1]; ++k) ds.push_back(A.cols[k] - r);
        }
        std::sort(ds.begin(), ds.end());
        ds.erase(std::unique(ds.begin(), ds.end()), ds.end());
        if (ds.size() > 255) {
            printf("chunk %d has %zu distinct offsets: not compressible\n", c, ds.size());
            return 1;
        }
        ChunkHdr &h = hdr[c];
        h.val_off = val_len;
        h.code_off = code_len;
        h.dict_off = (int)dict.size();
        h.dict_len = (int)ds.size();
        h.width = w;
        h.code_stride = ((2 * w + 15) / 16) * 16;
        dict.insert(dict.end(), ds.begin(), ds.end());
        val_len += (long)w * CHUNK;
        code_len += (long)h.code_stride * BLOCK;
    }
    dict.resize(dict.size() + 256, 0);
    std::vector<double> svals(val_len + 2, 0.0);
    std::vector<uint8_t> scodes(code_len + 16, 255);
    int wmax = 0;
    for (int c = 0; c < nc; ++c) {
        const ChunkHdr &h = hdr[c];
        wmax = std::max(wmax, h.width);
        for (int r = c * CHUNK; r < std::min(A.n, (c + 1) * CHUNK); ++r) {
            const int lr = r - c * CHUNK, t = lr / 2, which = lr & 1;
            for (int k = A.rp[r], s = 0; k < A.rp[r + 1]; ++k, ++s) {
                svals[h.val_off + (long)s * CHUNK + lr] = A.vals[k];
                const int d = A.cols[k] - r;
                const int *b = dict.data() + h.dict_off;
                const int code = (int)(std::lower_bound(b, b + h.dict_len, d) - b);
                scodes[h.code_off + (long)t * h.code_stride + 2 * s + which] = (uint8_t)code;
            }
        }
    }
    printf("chunks %d, max width %d, padded slots %ld (%.3f x nnz), dict ints %zu, codes %ld B\n", nc, wmax,
           val_len, (double)val_len / A.nnz, dict.size(), code_len);
    // plain chunk-local ELL with the global max width
    std::vector<int> ecols((size_t)nc * CHUNK * wmax + 2, -1);
    std::vector<double> evals((size_t)nc * CHUNK * wmax + 2, 0.0);
    for (int r = 0; r < A.n; ++r) {
        const int c = r / CHUNK, lr = r % CHUNK;
        for (int k = A.rp[r], s = 0; k < A.rp[r + 1]; ++k, ++s) {
            ecols[(size_t)c * CHUNK * wmax + (size_t)s * CHUNK + lr] = A.cols[k];
            evals[(size_t)c * CHUNK * wmax + (size_t)s * CHUNK + lr] = A.vals[k];
        }
    }

    // ---- pattern-id layout: per chunk a table of distinct row patterns ----
    std::vector<ChunkHdr> phdr(nc);
    std::vector<int> ptab;
    std::vector<uint8_t> pcodes((size_t)nc * 2 * BLOCK + 16, 0);
    for (int c = 0; c < nc; ++c) {
        ChunkHdr &h = phdr[c];
        h = hdr[c];
        h.code_off = (long)c * 2 * BLOCK;
        h.code_stride = 2;
        h.dict_off = (int)ptab.size();
        std::vector<std::vector<int>> pats;
        const int w = h.width;
        for (int lr = 0; lr < CHUNK; ++lr) {
            const int r = c * CHUNK + lr;
            std::vector<int> pat(w, INT_MIN);
            if (r < A.n)
                for (int k = A.rp[r], s = 0; k < A.rp[r + 1]; ++k, ++s) pat[s] = A.cols[k] - r;
            size_t id = 0;
            while (id < pats.size() && pats[id] != pat) ++id;
            if (id == pats.size()) pats.push_back(pat);
            if (id > 255 || pats.size() * w > PID_TABLE) {
                printf("chunk %d: too many row patterns\n", c);
                return 1;
            }
            pcodes[h.code_off + lr] = (uint8_t)id;   // rows 2t, 2t+1 are adjacent bytes
        }
        for (auto &p : pats) ptab.insert(ptab.end(), p.begin(), p.end());
        h.dict_len = (int)(pats.size() * w);
    }
    printf("pattern-id layout: table ints %zu (%.1f per chunk)\n", ptab.size(), (double)ptab.size() / nc);
    ChunkHdr *d_phdr;
    int *d_ptab;
    uint8_t *d_pcodes;
    CK(hipMalloc(&d_phdr, sizeof(ChunkHdr) * nc));
    CK(hipMalloc(&d_ptab, sizeof(int) * (ptab.size() + 1)));
    CK(hipMalloc(&d_pcodes, pcodes.size()));
    CK(hipMemcpy(d_phdr, phdr.data(), sizeof(ChunkHdr) * nc, hipMemcpyHostToDevice));
    CK(hipMemcpy(d_ptab, ptab.data(), sizeof(int) * ptab.size(), hipMemcpyHostToDevice));
    CK(hipMemcpy(d_pcodes, pcodes.data(), pcodes.size(), hipMemcpyHostToDevice));

    ChunkHdr *d_hdr;
    int *d_dict, *d_ecols;
    uint8_t *d_codes;
    double *d_svals, *d_evals, *d_x0, *d_x1, *d_y, *d_part;
    CK(hipMalloc(&d_hdr, sizeof(ChunkHdr) * nc));
    CK(hipMalloc(&d_dict, sizeof(int) * dict.size()));
    CK(hipMalloc(&d_codes, scodes.size()));
    CK(hipMalloc(&d_svals, sizeof(double) * svals.size()));
    CK(hipMalloc(&d_ecols, sizeof(int) * ecols.size()));
    CK(hipMalloc(&d_evals, sizeof(double) * evals.size()));
    CK(hipMalloc(&d_x0, sizeof(double) * (A.n + 2)));
    CK(hipMalloc(&d_x1, sizeof(double) * (A.n + 2)));
    CK(hipMalloc(&d_y, sizeof(double) * (A.n + 2)));
    CK(hipMalloc(&d_part, sizeof(double) * (nc + 16)));
    CK(hipMemcpy(d_hdr, hdr.data(), sizeof(ChunkHdr) * nc, hipMemcpyHostToDevice));
    CK(hipMemcpy(d_dict, dict.data(), sizeof(int) * dict.size(), hipMemcpyHostToDevice));
    CK(hipMemcpy(d_codes, scodes.data(), scodes.size(), hipMemcpyHostToDevice));
    CK(hipMemcpy(d_svals, svals.data(), sizeof(double) * svals.size(), hipMemcpyHostToDevice));
    CK(hipMemcpy(d_ecols, ecols.data(), sizeof(int) * ecols.size(), hipMemcpyHostToDevice));
    CK(hipMemcpy(d_evals, evals.data(), sizeof(double) * evals.size(), hipMemcpyHostToDevice));
    CK(hipMemcpy(d_x0, x0.data(), sizeof(double) * A.n, hipMemcpyHostToDevice));
    CK(hipMemcpy(d_x1, x1.data(), sizeof(double) * A.n, hipMemcpyHostToDevice));

    const double moved_sell = 8.0 * val_len + (double)code_len + 16.0 * A.n + 32.0 * nc;
    const double moved_ell = 12.0 * (double)nc * CHUNK * wmax + 16.0 * A.n;
    for (int rep = 0; rep < 2; ++rep) {
        time_it("ell int32 cols, xcd-grouped", A,
                [&](const double *x) {
                    hipLaunchKernelGGL((k_ell<1>), dim3(xcd_grid(nc)), dim3(BLOCK), 0, 0, A.n, nc, wmax, d_ecols,
                                       d_evals, x, d_y, d_part);
                },
                d_x0, d_x1, d_y, yref, reps, moved_ell);
        time_it("sell 1-byte codes, LDS dict, xcd-grouped", A,
                [&](const double *x) {
                    hipLaunchKernelGGL((k_sell<0, 1>), dim3(xcd_grid(nc)), dim3(BLOCK), 0, 0, A.n, nc, d_hdr,
                                       d_dict, d_codes, d_svals, x, d_y, d_part);
                },
                d_x0, d_x1, d_y, yref, reps, moved_sell);
        time_it("sell 1-byte ROW pattern ids, xcd-grouped", A,
                [&](const double *x) {
                    hipLaunchKernelGGL((k_sell_pid<1>), dim3(xcd_grid(nc)), dim3(BLOCK), 0, 0, A.n, nc, d_phdr,
                                       d_ptab, d_pcodes, d_svals, x, d_y, d_part, (const int *)nullptr);
                },
                d_x0, d_x1, d_y, yref, reps, 8.0 * val_len + 2.0 * BLOCK * nc + 16.0 * A.n + 32.0 * nc);
        for (int G : {1, 2, 8, 16, 32}) {
            // groups of G consecutive chunks per XCD on one common front, as an order table
            const int Q = 8 * G, grid = ((nc + Q - 1) / Q) * Q;
            std::vector<int> order(grid, -1);
            for (int b = 0; b < grid; ++b) {
                const int sl = b / 8, xq = b % 8;
                const int c = (sl / G) * Q + xq * G + sl % G;
                order[b] = c < nc ? c : -1;
            }
            int *d_order;
            CK(hipMalloc(&d_order, sizeof(int) * order.size()));
            CK(hipMemcpy(d_order, order.data(), sizeof(int) * order.size(), hipMemcpyHostToDevice));
            char name[96];
            snprintf(name, sizeof(name), "sell ROW pattern ids, xcd groups of %d", G);
            time_it(name, A,
                    [&](const double *x) {
                        hipLaunchKernelGGL((k_sell_pid<2>), dim3(grid), dim3(BLOCK), 0, 0, A.n, nc, d_phdr,
                                           d_ptab, d_pcodes, d_svals, x, d_y, d_part, d_order);
                    },
                    d_x0, d_x1, d_y, yref, reps, 8.0 * val_len + 2.0 * BLOCK * nc + 16.0 * A.n + 32.0 * nc);
            CK(hipFree(d_order));
        }
        for (double frac : {1.0}) {
            // band-aware order: XCD of chunk c = floor(frac(c * CHUNK / band) * 8); block b (XCD b % 8) takes
            // entry b / 8 of its XCD's ascending list
            const double band = frac * (double)n * n;
            std::vector<std::vector<int>> lists(8);
            for (int c = 0; c < nc; ++c) {
                const double ph = (double)c * CHUNK / band;
                int xq = (int)((ph - floor(ph)) * 8.0);
                lists[xq > 7 ? 7 : xq].push_back(c);
            }
            size_t mx = 0;
            for (auto &l : lists) mx = std::max(mx, l.size());
            std::vector<int> order(mx * 8, -1);
            for (int xq = 0; xq < 8; ++xq)
                for (size_t i = 0; i < lists[xq].size(); ++i) order[i * 8 + xq] = lists[xq][i];
            int *d_order;
            CK(hipMalloc(&d_order, sizeof(int) * order.size()));
            CK(hipMemcpy(d_order, order.data(), sizeof(int) * order.size(), hipMemcpyHostToDevice));
            char name[96];
            snprintf(name, sizeof(name), "sell ROW pattern ids, band-aware x%.1f", frac);
            const int grid = (int)order.size();
            time_it(name, A,
                    [&](const double *x) {
                        hipLaunchKernelGGL((k_sell_pid<2>), dim3(grid), dim3(BLOCK), 0, 0, A.n, nc, d_phdr,
                                           d_ptab, d_pcodes, d_svals, x, d_y, d_part, d_order);
                    },
                    d_x0, d_x1, d_y, yref, reps, 8.0 * val_len + 2.0 * BLOCK * nc + 16.0 * A.n + 32.0 * nc);
            CK(hipFree(d_order));
        }
        time_it("sell 1-byte codes, LDS dict, plain", A,
                [&](const double *x) {
                    hipLaunchKernelGGL((k_sell<0, 0>), dim3(nc), dim3(BLOCK), 0, 0, A.n, nc, d_hdr, d_dict,
                                       d_codes, d_svals, x, d_y, d_part);
                },
                d_x0, d_x1, d_y, yref, reps, moved_sell);
        time_it("sell 1-byte codes, global dict, xcd", A,
                [&](const double *x) {
                    hipLaunchKernelGGL((k_sell<1, 1>), dim3(xcd_grid(nc)), dim3(BLOCK), 0, 0, A.n, nc, d_hdr,
                                       d_dict, d_codes, d_svals, x, d_y, d_part);
                },
                d_x0, d_x1, d_y, yref, reps, moved_sell);
    }
    return 0;
}
