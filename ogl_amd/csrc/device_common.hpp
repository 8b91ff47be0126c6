// device_common.hpp -- what the gfx950 (CDNA4) kernels of the Krylov inner loop share (kernels_*.hip).
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
#pragma once
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

#ifndef LEAD_SCOPE
#define LEAD_SCOPE __HIP_MEMORY_SCOPE_SYSTEM  // (agent scope on coarse-grained memory measured no faster)
#endif
// ------------------------------------------------------------------------------------------
// Leader finalisation (large single-rank systems; LeadBox, kernels.hpp).  The kernel that CONSUMES a finaliser's scalars
// runs the finaliser itself: its first 16 workgroups -- dispatched first -- are the 16 wavefronts of k_finalize's tree
// (same order of additions, same bits) and publish their sums in an uncached (fine-grained) mailbox; every workgroup
// polls the mailbox, adds the 16 sums and runs the scalar logic on its own copy of the scalars; workgroup 0 stores the
// new scalars for later kernels.  One launch and one dispatch gap fewer per finaliser.  Visibility: the mailbox words
// carry their own tag (the launch sequence number from the INPUT scalar slot, which no workgroup of this launch writes),
// 32 payload bits each, so no fence and no ordering between the words is needed; the partials and the input scalars
// were written by earlier kernels.  (A finaliser folded into the PRODUCER needs a release per workgroup: measured 4 x
// slower, HISTORY r5 item 14; ONE leader workgroup walking all partials in four batches behind the chip's row loads:
// 13 us against the 10 of the launch it replaces, round 6.)
// ------------------------------------------------------------------------------------------
// The finaliser's 16 wavefronts become the first 16 workgroups of the launch ("leaders"; 32 for two arrays: workgroup b is
// wavefront b % 16 of array b / 16): a leader stages the partials of virtual threads 64 w .. 64 w + 63 (partials v,
// v + 1024, ... each) through LDS with all of its 256 threads -- up to 8 loads per thread in ONE memory round trip for
// systems of up to 16.7 M rows, a tile more beyond -- adds them in the finaliser's order (one wavefront), runs the xor
// tree and publishes the sum.  EVERY workgroup then fetches the 16 (x K) sums and adds them left to right itself, as
// the folded kernels of small systems do with their own copies.
constexpr int LEAD_TILE = 32;                  // partials per virtual thread staged at once
constexpr int LEAD_STAGE = LEAD_TILE * WAVE;   // doubles of LDS (16 KB: eight workgroups per CU keep their place)
__device__ __forceinline__ void lead_wave_sums(const LeadBox &L, uint32_t tag, const double *__restrict__ part, int m, int w,
                                               int array, double *stage /* LEAD_STAGE */)
{
    constexpr int LOADS = LEAD_TILE * WAVE / BLOCK;  // per thread and tile
    static_assert(LOADS * BLOCK == LEAD_TILE * WAVE, "a tile is whole loads of the workgroup");
    const int t = threadIdx.x, lane = t & (WAVE - 1), wave = t / WAVE;
    const int per_thread = (m + FIN_BLOCK - 1) / FIN_BLOCK;  // partials of a virtual thread (the last ones may be missing)
    double s = 0.0;
    for (int k0 = 0; k0 < per_thread; k0 += LEAD_TILE) {
        double v[LOADS];
#pragma unroll
        for (int e = 0; e < LOADS; ++e) {
            const int idx = e * BLOCK + t, k = idx / WAVE, ln = idx % WAVE;
            const long i = (long)(w * WAVE + ln) + (long)FIN_BLOCK * (k0 + k);
            v[e] = i < m ? part[i] : 0.0;
        }
#pragma unroll
        for (int e = 0; e < LOADS; ++e) stage[e * BLOCK + t] = v[e];
        __syncthreads();
        if (wave == 0) {
            const int k_end = min(LEAD_TILE, per_thread - k0);
            for (int k = 0; k < k_end; ++k)
                if ((long)(w * WAVE + lane) + (long)FIN_BLOCK * (k0 + k) < m) s += stage[k * WAVE + lane];
        }
        __syncthreads();
    }
    if (wave == 0) {
        s = wave_sum(s);
        // every replica of the mailbox gets the two half-words (lane r -> replica r): the pollers spread over the
        // replicas, so that 19,683 workgroups do not fetch the same cache lines from one memory channel
        if (lane < LEAD_REPLICAS) {
            const unsigned long long bits = (unsigned long long)__double_as_longlong(s);
            unsigned long long *box = L.box + (size_t)lane * LEAD_REPLICA_STRIDE + array * 2 * FIN_WAVES + 2 * w;
            __hip_atomic_store(box, ((unsigned long long)tag << 32) | (uint32_t)bits, __ATOMIC_RELAXED, LEAD_SCOPE);
            __hip_atomic_store(box + 1, ((unsigned long long)tag << 32) | (uint32_t)(bits >> 32), __ATOMIC_RELAXED, LEAD_SCOPE);
        }
    }
}

// word i of the mailbox: [tag : 32 | payload : 32], one atomic 8-byte store / load each (payload: half of a double)
// Threads 0 .. n_words - 1 of a polling workgroup fetch one word each: half-word i of vals[] (the doubles the leaders
// published, in mailbox order); returns (to every thread) false when a leader never published (time-out: a defect, not
// a state of the solve -- the caller raises comm_error and leaves).
__device__ __forceinline__ bool lead_wait(const LeadBox &L, int n_words, uint32_t tag, double *vals, int *timed_out)
{
    if (threadIdx.x == 0) *timed_out = 0;
    __syncthreads();
    if ((int)threadIdx.x < n_words) {
        const unsigned long long *src = L.box + (size_t)(blockIdx.x % LEAD_REPLICAS) * LEAD_REPLICA_STRIDE + threadIdx.x;
        const long long t0 = wall_clock64();
        for (;;) {
            const unsigned long long w = __hip_atomic_load(src, __ATOMIC_RELAXED, LEAD_SCOPE);
            if ((uint32_t)(w >> 32) == tag) {
                reinterpret_cast<uint32_t *>(vals)[threadIdx.x] = (uint32_t)w;
                break;
            }
            if (wall_clock64() - t0 > L.timeout_ticks) {
                *timed_out = 1;
                break;
            }
            __builtin_amdgcn_s_sleep(1);
        }
    }
    __syncthreads();
    return *timed_out == 0;
}
// The leaders of a launch that finalises K partial arrays: workgroup b < 16 K is wavefront b % 16 of array b / 16.
template <int K>
__device__ __forceinline__ void lead_leaders(const LeadBox &L, uint32_t tag, const double *__restrict__ p0,
                                             const double *__restrict__ p1, const double *__restrict__ p2, int m,
                                             double *stage)
{
    if ((int)blockIdx.x < K * FIN_WAVES) {
        const int a = blockIdx.x / FIN_WAVES;
        lead_wave_sums(L, tag, a == 0 ? p0 : (a == 1 ? p1 : p2), m, blockIdx.x % FIN_WAVES, a, stage);
    }
}

// the 16 wavefront sums of array a, left to right (fin_block_sum); a rolled loop: one thread per workgroup runs it, and
// its registers would be every thread's (the step kernels live on 8 wavefronts per SIMD)
__device__ __forceinline__ double lead_total(const double *vals, int a)
{
    double sum = vals[a * FIN_WAVES];
#pragma unroll 1
    for (int w = 1; w < FIN_WAVES; ++w) sum += vals[a * FIN_WAVES + w];
    return sum;
}

// StoppingCriterion.C:71-151 on the device.  `norm` is sum|r| over all ranks.
__device__ inline void criterion_check(DevScalars *s, const DevCriterion &c, double norm, double *history)
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
__device__ inline bool peer_allreduce2(const PeerArgs &pa, double &v0, double &v1, unsigned *waited_out = nullptr)
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

inline int blocks_for(int64_t n) { return (int)((n + BLOCK - 1) / BLOCK); }

}  // namespace

}  // namespace ogl
