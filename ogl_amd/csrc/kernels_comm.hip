// kernels_comm.hip -- peer-mesh all-reduce and halo exchange kernels (mailboxes over hipIpc)
// (geometry, reduction tree and the -ffp-contract=off rule: device_common.hpp)
#include "device_common.hpp"

namespace ogl {

namespace {

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

}  // namespace

// ------------------------------------------------------------------------------------------
// launchers
// ------------------------------------------------------------------------------------------
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

}  // namespace ogl
