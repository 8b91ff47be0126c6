// sym_tune.hip -- development harness (not product): prototype of HALF storage for symmetric matrices on
// structured meshes (see the comment at k_sym), next to the product's pattern-id kernel.  Derived from sell_tune.hip:
// prototype of the index-compressed chunked
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


// ---- half storage for a symmetric matrix on a banded pattern --------------------------------------------
// Offsets D[0..ND) ascending, D[0] = 0 (diagonal).  A chunk stores ND planes of CHUNK values: plane j holds
// A(r, r + D[j]) for its rows (0 where the neighbour does not exist).  The lower entry A(r, r - D[j]) is not
// stored: by symmetry it is A(r - D[j], r) = plane j of row r - D[j], a coalesced read of values that were
// (or will be) read as upper entries anyway -- DRAM sees every value once.  One byte per row says which of
// the 2 ND - 1 entries exist (bit ND-1-j... see below); rows are summed in ascending column order, so y has
// the same bits as with full storage.
typedef double d2u __attribute__((ext_vector_type(2), aligned(8)));  // a pair at an 8-byte aligned address
constexpr int ND = 4;
struct SymArgs {
    int d[ND];
};
template <int XCD, int VAR>
__global__ __launch_bounds__(BLOCK) void k_sym(int n_rows, int n_chunks, SymArgs S, const uint8_t *__restrict__ mask,
                                               const double *__restrict__ planes, const double *__restrict__ x,
                                               double *__restrict__ y, double *__restrict__ part)
{
    __shared__ double slot[N_WAVES];
    const int chunk = XCD ? xcd_chunk(blockIdx.x) : (int)blockIdx.x;
    if (chunk >= n_chunks) return;
    const int t = threadIdx.x;
    const int row = chunk * CHUNK + 2 * t;
    const int nv = min(2, max(0, n_rows - row));
    const unsigned mm = *reinterpret_cast<const unsigned short *>(mask + row);
    const unsigned m0 = mm & 0xffu, m1 = mm >> 8;  // bit (ND-1-j): lower entry -D[j] (j >= 1); bit (ND-1+j): upper +D[j]
    // own planes
    double2 up[ND];
    const double *own = planes + (long)chunk * ND * CHUNK + 2 * t;
#pragma unroll
    for (int j = 0; j < ND; ++j) up[j] = *reinterpret_cast<const double2 *>(own + (long)j * CHUNK);
    // lower entries: plane j at rows row - D[j], row + 1 - D[j]
    double lo0[ND], lo1[ND];
#pragma unroll
    for (int j = 1; j < ND; ++j) {
        const int r0 = row - S.d[j], r1 = r0 + 1;
        // VAR 5: the loads do not wait for the mask (which only gates the arithmetic): predicates from the row index
        const bool ok0 = VAR == 5 ? (r0 >= 0) : ((m0 >> (ND - 1 - j)) & 1u), ok1 = VAR == 5 ? (r1 >= 0) : ((m1 >> (ND - 1 - j)) & 1u);
        const long a0 = (long)(r0 >> 9) * (ND * CHUNK) + (long)j * CHUNK + (r0 & (CHUNK - 1));
        const long a1 = (long)(r1 >> 9) * (ND * CHUNK) + (long)j * CHUNK + (r1 & (CHUNK - 1));
        const bool even_j = VAR >= 3 ? (j >= 2) : ((S.d[j] & 1) == 0);
        const bool one_j = VAR >= 3 ? (j == 1) : (S.d[j] == 1);
        if (VAR == 6 && j == 1) {
            lo0[j] = ok0 ? planes[a0] : 0.0;
            lo1[j] = up[j].x;
        } else if (VAR == 6) {
            lo0[j] = ok0 ? planes[a0] : 0.0;
            lo1[j] = ok1 ? planes[a1] : 0.0;
        } else if ((VAR == 2 || VAR >= 4) && even_j) {
            double2 pl = make_double2(0.0, 0.0);
            if (ok0 || ok1) pl = *reinterpret_cast<const double2 *>(planes + a0);
            lo0[j] = pl.x;
            lo1[j] = pl.y;
        } else if ((VAR == 2 || VAR >= 4) && one_j) {
            lo0[j] = ok0 ? planes[a0] : 0.0;  // plane 1 of row - 1 (the previous lane's second row)
            lo1[j] = up[j].x;                 // plane 1 of row = this lane's own upper entry
        } else {
            lo0[j] = ok0 ? planes[a0] : 0.0;
            lo1[j] = ok1 ? planes[a1] : 0.0;
        }
    }
    double xl0[ND], xl1[ND], xu0[ND], xu1[ND];
    double xd0 = 0.0, xd1 = 0.0;
    if (nv > 0) xd0 = x[row];
    if (nv > 1) xd1 = x[row + 1];
#pragma unroll
    for (int j = 1; j < ND; ++j) {
        const bool l0 = VAR == 5 ? (row - S.d[j] >= 0) : ((m0 >> (ND - 1 - j)) & 1u), l1 = VAR == 5 ? (row + 1 - S.d[j] >= 0) : ((m1 >> (ND - 1 - j)) & 1u);
        const bool u0 = VAR == 5 ? (row + S.d[j] < n_rows) : ((m0 >> (ND - 1 + j)) & 1u), u1 = VAR == 5 ? (row + 1 + S.d[j] < n_rows) : ((m1 >> (ND - 1 + j)) & 1u);
        const bool even_j = VAR >= 3 ? (j >= 2) : ((S.d[j] & 1) == 0);  // VAR 3+: known at compile time (d = 1, even, even)
        const bool one_j = VAR >= 3 ? (j == 1) : (S.d[j] == 1);
        if (VAR == 6 && j >= 2) {  // any parity: the pair as one 16-byte load from an 8-byte aligned address
            // (NOT RUN: with an odd distance the pair of row = d - 1 starts at x[-1]; needs a scalar fallback there)
            d2u pl = {0.0, 0.0}, pu = {0.0, 0.0};
            if (l0 || l1) pl = *reinterpret_cast<const d2u *>(x + row - S.d[j]);
            if (u0 || u1) pu = *reinterpret_cast<const d2u *>(x + row + S.d[j]);
            xl0[j] = pl.x; xl1[j] = pl.y; xu0[j] = pu.x; xu1[j] = pu.y;
        } else if (VAR == 6) {
            xl0[j] = l0 ? x[row - 1] : 0.0;
            xl1[j] = xd0;
            xu0[j] = xd1;
            xu1[j] = u1 ? x[row + 2] : 0.0;
        } else if (VAR >= 1 && even_j) {  // even distance: the two rows' x values are an aligned pair
            double2 pl = make_double2(0.0, 0.0), pu = make_double2(0.0, 0.0);
            if (l0 || l1) pl = *reinterpret_cast<const double2 *>(x + row - S.d[j]);
            if (u0 || u1) pu = *reinterpret_cast<const double2 *>(x + row + S.d[j]);
            xl0[j] = pl.x; xl1[j] = pl.y; xu0[j] = pu.x; xu1[j] = pu.y;
        } else if (VAR >= 1 && one_j) {  // the neighbours of a pair are the pair itself + one on each side
            xl0[j] = l0 ? x[row - 1] : 0.0;
            xl1[j] = xd0;
            xu0[j] = xd1;
            xu1[j] = u1 ? x[row + 2] : 0.0;
        } else {
            xl0[j] = l0 ? x[row - S.d[j]] : 0.0;
            xl1[j] = l1 ? x[row + 1 - S.d[j]] : 0.0;
            xu0[j] = u0 ? x[row + S.d[j]] : 0.0;
            xu1[j] = u1 ? x[row + 1 + S.d[j]] : 0.0;
        }
    }
    double a0 = 0.0, a1 = 0.0;
#pragma unroll
    for (int j = ND - 1; j >= 1; --j) {  // ascending columns: the furthest lower entry first
        if ((m0 >> (ND - 1 - j)) & 1u) a0 = a0 + lo0[j] * xl0[j];
        if ((m1 >> (ND - 1 - j)) & 1u) a1 = a1 + lo1[j] * xl1[j];
    }
    if ((m0 >> (ND - 1)) & 1u) a0 = a0 + up[0].x * xd0;
    if ((m1 >> (ND - 1)) & 1u) a1 = a1 + up[0].y * xd1;
#pragma unroll
    for (int j = 1; j < ND; ++j) {
        if ((m0 >> (ND - 1 + j)) & 1u) a0 = a0 + up[j].x * xu0[j];
        if ((m1 >> (ND - 1 + j)) & 1u) a1 = a1 + up[j].y * xu1[j];
    }
    if (nv == 2)
        *reinterpret_cast<double2 *>(y + row) = make_double2(a0, a1);
    else if (nv == 1)
        y[row] = a0;
    double d = 0.0;
    if (nv > 0) d += xd0 * a0;
    if (nv > 1) d += xd1 * a1;
    const double sm = block_sum(d, slot);
    if (t == 0) part[chunk] = sm;
}

struct Csr {
    int n = 0, nnz = 0;
    std::vector<int> rp, cols;
    std::vector<double> vals;
};

static double face_value(long a, long b) { return -1.0 - 1e-3 * (double)((a + 3 * b) % 13); }  // a < b

static Csr poisson(int n)
{
    Csr A;
    const long N = (long)n * n * n;
    A.n = (int)N;
    A.rp.resize(N + 1);
    long e = 0;
    for (long c = 0; c < N; ++c) {
        const int i = c % n, j = (c / n) % n, k = c / ((long)n * n);
        A.rp[c] = (int)e;
        double dsum = 0;
        auto add = [&](long col) {
            const double v = face_value(std::min(c, col), std::max(c, col));
            A.cols.push_back((int)col);
            A.vals.push_back(v);
            dsum -= v;
            ++e;
        };
        if (k > 0) add(c - (long)n * n);
        if (j > 0) add(c - n);
        if (i > 0) add(c - 1);
        const long dpos = e;
        A.cols.push_back((int)c);
        A.vals.push_back(0.0);
        ++e;
        if (i < n - 1) add(c + 1);
        if (j < n - 1) add(c + n);
        if (k < n - 1) add(c + (long)n * n);
        A.vals[dpos] = dsum + 1e-3 * (1.0 + (c % 7) / 7.0);
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
    printf("poisson %d^3 (symmetric, varying coefficients): rows %d nnz %d\n", n, A.n, A.nnz);
    std::vector<double> x0(A.n), x1(A.n), yref(A.n);
    for (int i = 0; i < A.n; ++i) {
        x0[i] = sin(0.001 * i) + 0.5;
        x1[i] = cos(0.002 * i) - 0.25;
    }
    for (int r = 0; r < A.n; ++r) {
        double s = 0;
        for (int k = A.rp[r]; k < A.rp[r + 1]; ++k) s = s + A.vals[k] * x0[A.cols[k]];
        yref[r] = s;
    }
    const int nc = (A.n + CHUNK - 1) / CHUNK;
    // ---- full storage, pattern ids (the product's coding of this matrix) ----
    int wmax = 7;
    std::vector<ChunkHdr> phdr(nc);
    std::vector<int> ptab;
    std::vector<uint8_t> pcodes((size_t)nc * 2 * BLOCK + 16, 0);
    std::vector<double> svals((size_t)nc * CHUNK * wmax + 2, 0.0);
    for (int c = 0; c < nc; ++c) {
        ChunkHdr &h = phdr[c];
        h.val_off = (long)c * CHUNK * wmax;
        h.width = wmax;
        h.code_off = (long)c * 2 * BLOCK;
        h.code_stride = 2;
        h.dict_off = (int)ptab.size();
        std::vector<std::vector<int>> pats;
        for (int lr = 0; lr < CHUNK; ++lr) {
            const int r = c * CHUNK + lr;
            std::vector<int> pat(wmax, INT_MIN);
            if (r < A.n)
                for (int k = A.rp[r], s = 0; k < A.rp[r + 1]; ++k, ++s) {
                    pat[s] = A.cols[k] - r;
                    svals[h.val_off + (long)s * CHUNK + lr] = A.vals[k];
                }
            size_t id = 0;
            while (id < pats.size() && pats[id] != pat) ++id;
            if (id == pats.size()) pats.push_back(pat);
            pcodes[h.code_off + lr] = (uint8_t)id;
        }
        for (auto &p : pats) ptab.insert(ptab.end(), p.begin(), p.end());
        h.dict_len = (int)(pats.size() * wmax);
    }
    // ---- half storage ----
    SymArgs S;
    S.d[0] = 0; S.d[1] = 1; S.d[2] = n; S.d[3] = n * n;
    std::vector<double> planes((size_t)nc * ND * CHUNK + 2, 0.0);
    std::vector<uint8_t> mask((size_t)nc * CHUNK + 16, 0);
    for (int r = 0; r < A.n; ++r) {
        const int c = r / CHUNK, lr = r % CHUNK;
        for (int k = A.rp[r]; k < A.rp[r + 1]; ++k) {
            const int d = A.cols[k] - r;
            int j = 0;
            while (j < ND && S.d[j] != abs(d)) ++j;
            if (j == ND) { printf("unexpected offset %d\n", d); return 1; }
            if (d >= 0) {
                planes[(size_t)c * ND * CHUNK + (size_t)j * CHUNK + lr] = A.vals[k];
                mask[r] |= (uint8_t)(1u << (ND - 1 + j));
            } else {
                mask[r] |= (uint8_t)(1u << (ND - 1 - j));
            }
        }
    }
    ChunkHdr *d_phdr;
    int *d_ptab;
    uint8_t *d_pcodes, *d_mask;
    double *d_svals, *d_planes, *d_x0, *d_x1, *d_y, *d_part;
    CK(hipMalloc(&d_phdr, sizeof(ChunkHdr) * nc));
    CK(hipMalloc(&d_ptab, sizeof(int) * (ptab.size() + 1)));
    CK(hipMalloc(&d_pcodes, pcodes.size()));
    CK(hipMalloc(&d_mask, mask.size()));
    CK(hipMalloc(&d_svals, sizeof(double) * svals.size()));
    CK(hipMalloc(&d_planes, sizeof(double) * planes.size()));
    CK(hipMalloc(&d_x0, sizeof(double) * (A.n + 2)));
    CK(hipMalloc(&d_x1, sizeof(double) * (A.n + 2)));
    CK(hipMalloc(&d_y, sizeof(double) * (A.n + 2)));
    CK(hipMalloc(&d_part, sizeof(double) * (nc + 16)));
    CK(hipMemcpy(d_phdr, phdr.data(), sizeof(ChunkHdr) * nc, hipMemcpyHostToDevice));
    CK(hipMemcpy(d_ptab, ptab.data(), sizeof(int) * ptab.size(), hipMemcpyHostToDevice));
    CK(hipMemcpy(d_pcodes, pcodes.data(), pcodes.size(), hipMemcpyHostToDevice));
    CK(hipMemcpy(d_mask, mask.data(), mask.size(), hipMemcpyHostToDevice));
    CK(hipMemcpy(d_svals, svals.data(), sizeof(double) * svals.size(), hipMemcpyHostToDevice));
    CK(hipMemcpy(d_planes, planes.data(), sizeof(double) * planes.size(), hipMemcpyHostToDevice));
    CK(hipMemcpy(d_x0, x0.data(), sizeof(double) * A.n, hipMemcpyHostToDevice));
    CK(hipMemcpy(d_x1, x1.data(), sizeof(double) * A.n, hipMemcpyHostToDevice));
    const double moved_full = 8.0 * nc * CHUNK * wmax + 2.0 * BLOCK * nc + 16.0 * A.n + 32.0 * nc;
    const double moved_half = 8.0 * nc * CHUNK * ND + 1.0 * nc * CHUNK + 16.0 * A.n;
    printf("bytes per launch: full storage %.0f MB, half storage %.0f MB (CSR figure %.0f MB)\n", moved_full / 1e6,
           moved_half / 1e6, (12.0 * A.nnz + 20.0 * A.n) / 1e6);
    for (int rep = 0; rep < 3; ++rep) {
        time_it("full storage, ROW pattern ids (product)", A,
                [&](const double *x) {
                    hipLaunchKernelGGL((k_sell_pid<1>), dim3(xcd_grid(nc)), dim3(BLOCK), 0, 0, A.n, nc, d_phdr,
                                       d_ptab, d_pcodes, d_svals, x, d_y, d_part, (const int *)nullptr);
                },
                d_x0, d_x1, d_y, yref, reps, moved_full);
#define RUNSYM(XCD, VAR)                                                                                          \
        time_it("half storage, xcd" #XCD " var" #VAR, A,                                                          \
                [&](const double *x) {                                                                            \
                    hipLaunchKernelGGL((k_sym<XCD, VAR>), dim3(XCD ? xcd_grid(nc) : nc), dim3(BLOCK), 0, 0, A.n, nc, S, \
                                       d_mask, d_planes, x, d_y, d_part);                                         \
                },                                                                                                \
                d_x0, d_x1, d_y, yref, reps, moved_half)
        RUNSYM(1, 0);
        if (n & 1) continue;  // (the variants below assume even distances)
        RUNSYM(1, 1);
        RUNSYM(1, 2);
        RUNSYM(1, 3);
        RUNSYM(1, 4);
        RUNSYM(1, 5);
    }
    return 0;
}
