// coop_tune.hip -- what does a grid-wide barrier inside ONE cooperative persistent kernel cost on
// MI355X, against the dependent kernel launches it would replace?  ("one
// cooperative persistent kernel per batch of turns, grid barrier between the 5 phases".)
//
// A CG turn has 5 dependent phases; between two of them every workgroup needs data other workgroups
// (on other XCDs, behind other L2s) have just written: p before the SpMV gather, the per-chunk
// partials before the scalar logic, the scalars before the next vector update.  So the barrier has
// to carry a device-scope release (L2 write-back) and acquire (L2 invalidate), exactly what a kernel
// boundary does.  Measured here, per phase boundary, with the data hand-over checked:
//   (a) `launches`: K dependent launches of a kernel that writes its slot and reads its neighbour's
//   (b) `flat`    : one cooperative kernel, K barriers on one atomic counter
//   (c) `tree`    : one cooperative kernel, K barriers with one counter per XCD + a root counter
// for grids of 256 / 512 / 1024 workgroups of 256 threads and, as a proxy of a 128^3 turn, with a
// streaming pass over `words` doubles per phase (so the fence has dirty lines to write back).
//
//   hipcc -O3 --offload-arch=gfx950 tools/coop_tune.hip -o tools/bin/coop_tune && tools/bin/coop_tune
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <vector>

#define CHECK(x)                                                                        \
    do {                                                                                \
        hipError_t e_ = (x);                                                            \
        if (e_ != hipSuccess) {                                                         \
            fprintf(stderr, "%s: %s (%s:%d)\n", #x, hipGetErrorString(e_), __FILE__, __LINE__); \
            exit(1);                                                                    \
        }                                                                               \
    } while (0)

constexpr int BLOCK = 256;
constexpr long long TIMEOUT_TICKS = 2LL * 100000000LL;  // 2 s of the 100 MHz wall clock: never hang the box

struct Bar {
    unsigned *root;     // arrivals of XCD leaders (tree) or of all workgroups (flat)
    unsigned *xcd;      // 8 counters, 64 bytes apart
    unsigned *release;  // epoch published by the last arriver (tree)
    int *failed;
};

__device__ bool spin_until(const unsigned *addr, unsigned target)
{
    const long long t0 = wall_clock64();
    while (__hip_atomic_load(addr, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < target) {
        if (wall_clock64() - t0 > TIMEOUT_TICKS) return false;
        __builtin_amdgcn_s_sleep(1);
    }
    return true;
}

// epoch = number of barriers passed so far (same on every workgroup)
__device__ void barrier_flat(const Bar &b, unsigned epoch)
{
    __syncthreads();
    if (threadIdx.x == 0) {
        __threadfence();  // release: this workgroup's writes leave its XCD's L2
        __hip_atomic_fetch_add(b.root, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (!spin_until(b.root, (epoch + 1) * gridDim.x)) *b.failed = 1;
        __threadfence();  // acquire
    }
    __syncthreads();
}

__device__ void barrier_tree(const Bar &b, unsigned epoch)
{
    __syncthreads();
    if (threadIdx.x == 0) {
        __threadfence();
        const unsigned x = blockIdx.x % 8, per = (gridDim.x - x + 7) / 8;  // workgroups on this XCD
        const unsigned prev = __hip_atomic_fetch_add(b.xcd + 16 * x, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (prev + 1 == (epoch + 1) * per) {  // last of its XCD
            const unsigned r = __hip_atomic_fetch_add(b.root, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (r + 1 == (epoch + 1) * min(8u, gridDim.x))
                __hip_atomic_store(b.release, epoch + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        if (!spin_until(b.release, epoch + 1)) *b.failed = 1;
        __threadfence();
    }
    __syncthreads();
}

// one phase of work: stream `per` doubles of this workgroup's slice (read-modify-write), publish a stamp
__device__ void phase_work(double *v, long per, unsigned *stamps, unsigned value)
{
    double *mine = v + (long)blockIdx.x * per;
    for (long i = threadIdx.x; i < per; i += BLOCK) mine[i] = mine[i] * 1.0000001 + 1.0;
    if (threadIdx.x == 0) stamps[blockIdx.x] = value;
}

__global__ __launch_bounds__(BLOCK) void k_phase(double *v, long per, unsigned *stamps, unsigned value, int *bad)
{
    // the neighbour's stamp of the previous phase must be visible (kernel boundary = release/acquire)
    if (threadIdx.x == 0 && value > 1 && stamps[(blockIdx.x + 1) % gridDim.x] != value - 1) *bad = 1;
    __syncthreads();
    phase_work(v, per, stamps, value);
}

template <int TREE>
__global__ __launch_bounds__(BLOCK) void k_persistent(double *v, long per, unsigned *stamps, int phases, Bar b, int *bad)
{
    for (int ph = 1; ph <= phases; ++ph) {
        phase_work(v, per, stamps, (unsigned)ph);
        if (TREE)
            barrier_tree(b, (unsigned)(ph - 1));
        else
            barrier_flat(b, (unsigned)(ph - 1));
        if (threadIdx.x == 0 &&
            __hip_atomic_load(stamps + (blockIdx.x + 1) % gridDim.x, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != (unsigned)ph)
            *bad = 1;
        if (*b.failed) return;
    }
}

int main(int argc, char **argv)
{
    const int phases = argc > 1 ? atoi(argv[1]) : 200;
    hipStream_t st;
    CHECK(hipStreamCreate(&st));
    unsigned *ctr, *stamps;
    int *flags;
    CHECK(hipMalloc(&ctr, 4096));
    CHECK(hipMalloc(&stamps, 4096 * sizeof(unsigned)));
    CHECK(hipMalloc(&flags, 2 * sizeof(int)));
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0));
    CHECK(hipEventCreate(&e1));
    int max_blocks_per_cu = 0;
    CHECK(hipOccupancyMaxActiveBlocksPerMultiprocessor(&max_blocks_per_cu, k_persistent<1>, BLOCK, 0));
    hipDeviceProp_t prop;
    CHECK(hipGetDeviceProperties(&prop, 0));
    printf("# %s: %d CUs, %d resident workgroups of %d threads per CU for the persistent kernel\n", prop.gcnArchName,
           prop.multiProcessorCount, max_blocks_per_cu, BLOCK);
    printf("# us per phase boundary, %d phases; 'words' = doubles streamed (read+write) per phase in all\n", phases);
    printf("%8s %10s %12s %12s %12s   %s\n", "grid", "words", "launches", "coop flat", "coop tree", "hand-over");
    for (long words : {0L, 2097152L}) {  // 0 = pure synchronisation; 2M doubles = one 128^3 vector
        double *v;
        CHECK(hipMalloc(&v, (words + 4096) * sizeof(double)));
        CHECK(hipMemset(v, 0, (words + 4096) * sizeof(double)));
        for (int grid : {256, 512, 1024}) {
            if (grid > prop.multiProcessorCount * max_blocks_per_cu) continue;
            const long per = words / grid;
            float ms[3] = {0, 0, 0};
            int bad_any = 0;
            for (int variant = 0; variant < 3; ++variant) {
                for (int rep = 0; rep < 2; ++rep) {  // first repetition warms up
                    CHECK(hipMemsetAsync(ctr, 0, 4096, st));
                    CHECK(hipMemsetAsync(stamps, 0, 4096 * sizeof(unsigned), st));
                    CHECK(hipMemsetAsync(flags, 0, 2 * sizeof(int), st));
                    Bar b{ctr, ctr + 64, ctr + 512, flags + 1};
                    int *bad = flags;
                    CHECK(hipEventRecord(e0, st));
                    if (variant == 0) {
                        for (int ph = 1; ph <= phases; ++ph)
                            hipLaunchKernelGGL(k_phase, dim3(grid), dim3(BLOCK), 0, st, v, per, stamps, (unsigned)ph, bad);
                    } else {
                        int ph = phases;
                        void *args[] = {&v, (void *)&per, &stamps, &ph, &b, &bad};
                        CHECK(hipLaunchCooperativeKernel(variant == 1 ? (void *)k_persistent<0> : (void *)k_persistent<1>,
                                                         dim3(grid), dim3(BLOCK), args, 0, st));
                    }
                    CHECK(hipEventRecord(e1, st));
                    CHECK(hipEventSynchronize(e1));
                    CHECK(hipEventElapsedTime(&ms[variant], e0, e1));
                    int h[2];
                    CHECK(hipMemcpy(h, flags, sizeof(h), hipMemcpyDeviceToHost));
                    if (h[0]) bad_any |= 1;
                    if (h[1]) bad_any |= 2;
                }
            }
            printf("%8d %10ld %12.2f %12.2f %12.2f   %s\n", grid, words, 1e3 * ms[0] / phases, 1e3 * ms[1] / phases,
                   1e3 * ms[2] / phases, bad_any == 0 ? "ok" : (bad_any & 2 ? "TIMEOUT" : "STALE DATA SEEN"));
        }
        CHECK(hipFree(v));
    }
    return 0;
}
