// tail_probe.hip -- development probe (not product): how much of a bandwidth-bound kernel's time is the drain of its
// last workgroups, as a function of the rows per workgroup?  The Krylov kernels give every workgroup one chunk of 512
// rows; a launch of W workgroups on S resident slots loses about S / (2 W) of its time to the tail (1 M rows: 1,954
// workgroups on ~1,800 slots).  Here: the byte pattern of k_cg_step1x (48 B per row: read p, x, r, d; write p, x) and
// of a heavier kernel (33 N matrix planes + 16 N vectors, the half-storage SpMV's bytes, no gathers) with 128 / 256 / 512 /
// 1024 rows per workgroup of 256 threads, at several system sizes.
//   hipcc --offload-arch=gfx950 -O3 tools/tail_probe.hip -o tools/bin/tail_probe && tools/bin/tail_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s line %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)
typedef double d2 __attribute__((ext_vector_type(2)));

// RPT rows per thread (1, 2, 4, 8 with 256 threads = 256 ... 2048 rows per workgroup); NP extra "matrix" planes
template <int RPT, int NP>
__global__ __launch_bounds__(256) void k_probe(long n, double *__restrict__ p, double *__restrict__ x,
                                               const double *__restrict__ r, const double *__restrict__ d,
                                               const double *__restrict__ planes, double a)
{
    const long base = (long)blockIdx.x * 256 * RPT + threadIdx.x;   // (coalesced: lane = row within a 256-row slice)
#pragma unroll
    for (int i = 0; i < RPT; ++i) {
        const long row = base + (long)i * 256;
        if (row >= n) return;
        double acc = r[row] * d[row];
#pragma unroll
        for (int k = 0; k < NP; ++k) acc += __builtin_nontemporal_load(planes + (long)k * n + row);
        const double pv = p[row];
        x[row] += a * pv;
        p[row] = acc + a * pv;
    }
}

template <int RPT, int NP>
static float run(long n, double *p, double *x, double *r, double *d, double *planes, int reps)
{
    const long rows_per_wg = 256L * RPT;
    const unsigned grid = (unsigned)((n + rows_per_wg - 1) / rows_per_wg);
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    for (int i = 0; i < 3; ++i) hipLaunchKernelGGL((k_probe<RPT, NP>), dim3(grid), dim3(256), 0, 0, n, p, x, r, d, planes, 1e-9);
    CK(hipEventRecord(e0));
    for (int i = 0; i < reps; ++i) hipLaunchKernelGGL((k_probe<RPT, NP>), dim3(grid), dim3(256), 0, 0, n, p, x, r, d, planes, 1e-9);
    CK(hipEventRecord(e1));
    CK(hipEventSynchronize(e1));
    float ms;
    CK(hipEventElapsedTime(&ms, e0, e1));
    return 1e3f * ms / reps;
}

int main()
{
    const long sizes[] = {1000000, 2097152, 2515456, 4096000, 10077696};
    const long nmax = 10077696;
    double *p, *x, *r, *d, *planes;
    CK(hipMalloc(&p, 8 * nmax)); CK(hipMalloc(&x, 8 * nmax)); CK(hipMalloc(&r, 8 * nmax)); CK(hipMalloc(&d, 8 * nmax));
    CK(hipMalloc(&planes, 8 * nmax * 4));
    CK(hipMemset(p, 0, 8 * nmax)); CK(hipMemset(x, 0, 8 * nmax)); CK(hipMemset(r, 0, 8 * nmax)); CK(hipMemset(d, 0, 8 * nmax));
    CK(hipMemset(planes, 0, 8 * nmax * 4));
    printf("rows per workgroup of 256 threads (256-row slices per thread):            256      512     1024     2048\n");
    for (long n : sizes) {
        printf("vector kernel (48 B/row), %8ld rows: us per launch            %8.1f %8.1f %8.1f %8.1f\n", n,
               run<1, 0>(n, p, x, r, d, planes, 200), run<2, 0>(n, p, x, r, d, planes, 200),
               run<4, 0>(n, p, x, r, d, planes, 200), run<8, 0>(n, p, x, r, d, planes, 200));
        printf("heavy kernel (80 B/row, 4 planes), %8ld rows: us per launch   %8.1f %8.1f %8.1f %8.1f\n", n,
               run<1, 4>(n, p, x, r, d, planes, 200), run<2, 4>(n, p, x, r, d, planes, 200),
               run<4, 4>(n, p, x, r, d, planes, 200), run<8, 4>(n, p, x, r, d, planes, 200));
    }
    return 0;
}
