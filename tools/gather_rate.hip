// scratch: how many scattered 8-byte gathers per second does an MI355X serve?  (the x gather of an SpMV on an irregular mesh)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <random>
constexpr int BLOCK = 256, PER = 16;
// every lane: PER indices (coalesced int loads, streamed), PER gathers of doubles, sum
template <bool NT>
__global__ __launch_bounds__(BLOCK) void k_gather(const int *__restrict__ idx, const double *__restrict__ x, double *__restrict__ out, long n_lanes)
{
    const long lane = (long)blockIdx.x * BLOCK + threadIdx.x;
    if (lane >= n_lanes) return;
    int c[PER];
#pragma unroll
    for (int k = 0; k < PER; ++k) c[k] = __builtin_nontemporal_load(idx + (long)k * n_lanes + lane);
    double v[PER];
#pragma unroll
    for (int k = 0; k < PER; ++k) v[k] = x[c[k]];
    double s = 0.0;
#pragma unroll
    for (int k = 0; k < PER; ++k) s += v[k];
    out[lane] = s;
}
int main()
{
    const long n_lanes = 4l << 20;  // 4 M lanes x 16 = 67 M gathers per launch
    const long n_x = 16l << 20;     // 128 MB of doubles
    int *d_idx; double *d_x, *d_out;
    (void)hipMalloc(&d_idx, n_lanes * PER * 4); (void)hipMalloc(&d_x, n_x * 8); (void)hipMalloc(&d_out, n_lanes * 8);
    (void)hipMemset(d_x, 0, n_x * 8);
    std::vector<int> h(n_lanes * PER);
    std::mt19937_64 rng(1);
    struct Case { const char *name; int mode; long window; } cases[] = {
        {"coalesced (lane l reads x[base + l])", 0, 0},
        {"random within 2 K doubles around the lane's position (16 KB: L1-resident)", 1, 2048},
        {"random within 32 K doubles (256 KB: one L2's share)", 1, 32768},
        {"random within 512 K doubles (4 MB)", 1, 524288},
        {"random within 2 M doubles (16 MB: a 2 M-row vector)", 1, 2097152},
        {"random within 16 M doubles (128 MB)", 1, n_x},
        {"7-point stencil of a shuffled 128^3 box, windows of 65536 (neighbour cells at random places of the window)", 2, 65536},
    };
    for (auto &cs : cases) {
        for (long lane = 0; lane < n_lanes; ++lane)
            for (int k = 0; k < PER; ++k) {
                long c;
                const long pos = lane * 4 % (n_x - 1);  // the lane's "row"
                if (cs.mode == 0) c = (lane + (long)k * 977) % n_x;
                else if (cs.mode == 1) { const long lo = cs.window >= n_x ? 0 : std::max(0l, std::min(n_x - cs.window, pos - cs.window / 2)); c = lo + (long)(rng() % (unsigned long)cs.window); }
                else { const long w0 = pos / cs.window * cs.window; c = std::min(n_x - 1, w0 + (long)(rng() % (unsigned long)cs.window)); }
                h[(size_t)k * n_lanes + lane] = (int)c;
            }
        (void)hipMemcpy(d_idx, h.data(), h.size() * 4, hipMemcpyHostToDevice);
        hipEvent_t a, b; (void)hipEventCreate(&a); (void)hipEventCreate(&b);
        const int grid = (int)((n_lanes + BLOCK - 1) / BLOCK);
        for (int i = 0; i < 2; ++i) hipLaunchKernelGGL(k_gather<false>, dim3(grid), dim3(BLOCK), 0, 0, d_idx, d_x, d_out, n_lanes);
        (void)hipEventRecord(a, 0);
        const int reps = 10;
        for (int i = 0; i < reps; ++i) hipLaunchKernelGGL(k_gather<false>, dim3(grid), dim3(BLOCK), 0, 0, d_idx, d_x, d_out, n_lanes);
        (void)hipEventRecord(b, 0); (void)hipEventSynchronize(b);
        float ms = 0; (void)hipEventElapsedTime(&ms, a, b);
        const double us = ms * 1e3 / reps, g = (double)n_lanes * PER / us / 1e3;
        printf("%-110s %8.1f us  %7.1f G gathers/s  (index + result streams alone: %.0f MB = %.1f us at 6.3 TB/s)\n", cs.name, us, g, n_lanes * (PER * 4 + 8) / 1e6, n_lanes * (PER * 4.0 + 8) / 6.3e6);
    }
    return 0;
}
