// setup_kernels.hip -- see setup_kernels.hpp.  Integer work, HBM- and atomics-bound; nothing here is on the
// per-turn path.  Determinism: atomics only decide where an entry waits inside its row before the row is
// sorted by a key that is unique per entry, so every array comes out the same on every run.
#include "setup_kernels.hpp"

#include <hipcub/hipcub.hpp>

#include <algorithm>

namespace ogl {

namespace {

constexpr int SCAN_BLOCK = 256;
constexpr int SCAN_ITEMS = 8;
constexpr int SCAN_TILE = SCAN_BLOCK * SCAN_ITEMS;

inline int blocks_for(int64_t n, int per_block = BLOCK) { return (int)((n + per_block - 1) / per_block); }

// exclusive scan of one value per thread over the workgroup; returns the exclusive prefix, *total = block sum
__device__ __forceinline__ int block_exclusive_scan(int v, int *total)
{
    __shared__ int wave_sums[SCAN_BLOCK / WAVE];
    const int lane = threadIdx.x & (WAVE - 1), wave = threadIdx.x / WAVE;
    int inc = v;
#pragma unroll
    for (int off = 1; off < WAVE; off <<= 1) {
        const int t = __shfl_up(inc, off, WAVE);
        if (lane >= off) inc += t;
    }
    if (lane == WAVE - 1) wave_sums[wave] = inc;
    __syncthreads();
    int base = 0, all = 0;
#pragma unroll
    for (int w = 0; w < SCAN_BLOCK / WAVE; ++w) {
        if (w < wave) base += wave_sums[w];
        all += wave_sums[w];
    }
    __syncthreads();
    *total = all;
    return base + inc - v;
}

__global__ __launch_bounds__(SCAN_BLOCK) void k_scan_tile_sums(const int32_t *__restrict__ in, int64_t n,
                                                               int32_t *__restrict__ sums)
{
    const int64_t base = (int64_t)blockIdx.x * SCAN_TILE + (int64_t)threadIdx.x * SCAN_ITEMS;
    int s = 0;
#pragma unroll
    for (int k = 0; k < SCAN_ITEMS; ++k)
        if (base + k < n) s += in[base + k];
    int total;
    (void)block_exclusive_scan(s, &total);
    if (threadIdx.x == 0) sums[blockIdx.x] = total;
}

// one workgroup: sums[0..nb) -> exclusive prefixes in place, sums[nb] = total
__global__ __launch_bounds__(SCAN_BLOCK) void k_scan_sums(int32_t *sums, int nb)
{
    int carry = 0;
    for (int i0 = 0; i0 < nb; i0 += SCAN_BLOCK) {
        const int i = i0 + threadIdx.x;
        const int v = i < nb ? sums[i] : 0;
        int total;
        const int ex = block_exclusive_scan(v, &total);
        if (i < nb) sums[i] = carry + ex;
        carry += total;
    }
    if (threadIdx.x == 0) sums[nb] = carry;
}

__global__ __launch_bounds__(SCAN_BLOCK) void k_scan_apply(const int32_t *in, int32_t *out, int64_t n,
                                                           const int32_t *__restrict__ sums, int nb)
{
    const int64_t base = (int64_t)blockIdx.x * SCAN_TILE + (int64_t)threadIdx.x * SCAN_ITEMS;
    int v[SCAN_ITEMS], s = 0;
#pragma unroll
    for (int k = 0; k < SCAN_ITEMS; ++k) {
        v[k] = base + k < n ? in[base + k] : 0;
        s += v[k];
    }
    int total;
    int run = block_exclusive_scan(s, &total) + sums[blockIdx.x];
#pragma unroll
    for (int k = 0; k < SCAN_ITEMS; ++k) {
        if (base + k < n) out[base + k] = run;
        run += v[k];
    }
    if (blockIdx.x == 0 && threadIdx.x == 0) out[n] = sums[nb];
}

// ---- pattern build ----
__global__ __launch_bounds__(BLOCK) void k_pat_count(PatternBuild b)
{
    const int i = blockIdx.x * BLOCK + threadIdx.x;
    if (i < b.n_rows) atomicAdd(&b.counts[i], 1);  // the diagonal
    if (i < b.n_faces) {
        const int lo = b.lower_addr[i], up = b.upper_addr[i];
        if (lo < 0 || lo >= b.n_rows || up < 0 || up >= b.n_rows) {
            b.flags[PAT_FLAG_OUT_OF_RANGE] = 1;
        } else {
            if (lo >= up) b.flags[PAT_FLAG_NONCONFORMING] = 1;
            atomicAdd(&b.counts[lo], 1);
            atomicAdd(&b.counts[up], 1);
        }
    }
    if (i < b.n_iface) atomicAdd(&b.counts[b.if_rows[i]], 1);
}

__global__ __launch_bounds__(BLOCK) void k_pat_fill(PatternBuild b)
{
    const int i = blockIdx.x * BLOCK + threadIdx.x;
    int *fill = b.diag_pos;  // zeroed scratch until k_pat_sort writes the diagonal positions
    const int after_neighbours = b.symmetric ? b.n_faces : 2 * b.n_faces;  // HostMatrixFreeFunctions.C:116
    if (i < b.n_rows) {  // diagonal -> source slot after_neighbours + row (:179-182)
        const int e = b.row_ptrs[i] + atomicAdd(&fill[i], 1);
        b.cols[e] = i;
        b.ldu_mapping[e] = after_neighbours + i;
    }
    if (i < b.n_faces) {
        const int lo = b.lower_addr[i], up = b.upper_addr[i];
        if (lo >= 0 && lo < b.n_rows && up >= 0 && up < b.n_rows) {
            int e = b.row_ptrs[lo] + atomicAdd(&fill[lo], 1);  // upper-triangle entry lives in row lower[f] (:123-124,188)
            b.cols[e] = up;
            b.ldu_mapping[e] = i;
            e = b.row_ptrs[up] + atomicAdd(&fill[up], 1);      // its transpose in row upper[f] (:139-140,164-165)
            b.cols[e] = lo;
            b.ldu_mapping[e] = b.symmetric ? i : b.n_faces + i;
        }
    }
    if (i < b.n_iface) {  // same-rank interface entry -> source slot after_neighbours + nrows + idx (HostMatrix.C:574)
        const int r = b.if_rows[i];
        const int e = b.row_ptrs[r] + atomicAdd(&fill[r], 1);
        b.cols[e] = b.if_cols[i];
        b.ldu_mapping[e] = after_neighbours + b.n_rows + i;
    }
}

// one thread per row: order the row by (column, source slot); rows are short (a cell's faces + 1)
__global__ __launch_bounds__(BLOCK) void k_pat_sort(int32_t n_rows, const int32_t *__restrict__ row_ptrs,
                                                    int32_t *cols, int32_t *perm, int32_t *diag_pos)
{
    const int r = blockIdx.x * BLOCK + threadIdx.x;
    if (r >= n_rows) return;
    const int k0 = row_ptrs[r], k1 = row_ptrs[r + 1];
    for (int i = k0 + 1; i < k1; ++i) {
        const int c = cols[i], p = perm[i];
        int j = i;
        while (j > k0 && (cols[j - 1] > c || (cols[j - 1] == c && perm[j - 1] > p))) {
            cols[j] = cols[j - 1];
            perm[j] = perm[j - 1];
            --j;
        }
        cols[j] = c;
        perm[j] = p;
    }
    int dp = -1;
    for (int k = k0; k < k1; ++k)
        if (cols[k] == r) {  // Csr::extract_diagonal takes the first (i, i) entry of a row
            dp = k;
            break;
        }
    diag_pos[r] = dp;
}

// ---- half storage ----
__global__ __launch_bounds__(BLOCK) void k_sym_distances(int32_t n_rows, const int32_t *__restrict__ row_ptrs,
                                                         const int32_t *__restrict__ cols, int32_t *table,
                                                         int32_t *flags)
{
    const int r = blockIdx.x * BLOCK + threadIdx.x;
    if (r >= n_rows) return;
    int last = -1;  // the previous distance of this row that was looked up (rows repeat few distances)
    int prev_col = -1;
    for (int k = row_ptrs[r]; k < row_ptrs[r + 1]; ++k) {
        const int c = cols[k];
        if (c <= prev_col) flags[SYM_FLAG_UNSORTED_ROW] = 1;  // the kernel sums in ascending column order: one entry per column
        prev_col = c;
        const int d = c - r;
        if (d < 0 || d == last) continue;
        last = d;
        bool found = false;
        for (int j = 0; j < SYM_TABLE && !found; ++j) {
            int cur = __hip_atomic_load(&table[j], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (cur == SYM_EMPTY) {  // claim the slot; a lost race leaves somebody else's distance in it
                const int old = atomicCAS(&table[j], SYM_EMPTY, d);
                cur = old == SYM_EMPTY ? d : old;
            }
            found = cur == d;
        }
        if (!found) flags[SYM_FLAG_TOO_MANY] = 1;
    }
}

__global__ __launch_bounds__(BLOCK) void k_sym_fill(int32_t n_rows, const int32_t *__restrict__ row_ptrs,
                                                    const int32_t *__restrict__ cols, SymDistances dist,
                                                    uint8_t *__restrict__ mask, int32_t *__restrict__ map,
                                                    int32_t *flags)
{
    const int r = blockIdx.x * BLOCK + threadIdx.x;
    if (r >= n_rows) return;
    const int nd = dist.nd;
    unsigned m = 0;
    for (int k = row_ptrs[r]; k < row_ptrs[r + 1]; ++k) {
        const int d = cols[k] - r;
        const int a = d < 0 ? -d : d;
        int j = 0;
        while (j < nd && dist.d[j] != a) ++j;
        if (j == nd) {  // a lower entry whose distance no upper entry has: no twin to read it from
            flags[SYM_FLAG_TOO_MANY] = 1;
            continue;
        }
        if (d >= 0) {  // plane j of this row holds the entry: slot -> position in the CSR values
            map[((size_t)(r / CHUNK_ROWS) * nd + j) * CHUNK_ROWS + (size_t)(r % CHUNK_ROWS)] = k;
            m |= 1u << (nd - 1 + j);
        } else {       // read where its twin (r - a, r) lives: plane j of row r - a -- which must hold it
            const int c = cols[k];
            bool twin = false;
            for (int kk = row_ptrs[c]; kk < row_ptrs[c + 1] && !twin; ++kk) twin = cols[kk] == r;
            if (!twin) flags[SYM_FLAG_TOO_MANY] = 1;
            m |= 1u << (nd - 1 - j);
        }
    }
    mask[r] = (uint8_t)m;
}

// ---- reverse Cuthill-McKee, level-synchronous ----
__global__ __launch_bounds__(BLOCK) void k_rcm_init(RcmWork w)
{
    const int i = blockIdx.x * BLOCK + threadIdx.x;
    if (i >= w.n_rows) return;
    w.lvl[i] = -1;
    w.key[i] = RCM_NO_KEY;
}

__global__ __launch_bounds__(BLOCK) void k_rcm_find_seed(RcmWork w)
{
    const int i = blockIdx.x * BLOCK + threadIdx.x;
    // (one candidate per wavefront reaches the cell: the lowest untouched index of the wavefront)
    unsigned long long mine = (i < w.n_rows && w.lvl[i] == -1) ? (unsigned long long)i : ~0ull;
#pragma unroll
    for (int off = WAVE / 2; off >= 1; off >>= 1) {
        const unsigned long long o = __shfl_xor(mine, off, WAVE);
        mine = o < mine ? o : mine;
    }
    if ((threadIdx.x & (WAVE - 1)) == 0 && mine != ~0ull) atomicMin(&w.cell[0], mine);
}

__global__ void k_rcm_start(RcmWork w, int32_t *list, int32_t pos, int32_t node, int32_t level)
{
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    list[pos] = node;
    w.lvl[node] = level;
}

// every untouched neighbour of a frontier node learns the earliest frontier position that reaches it
__global__ __launch_bounds__(BLOCK) void k_rcm_mark(RcmWork w, const int32_t *__restrict__ list, int32_t begin,
                                                    int32_t end)
{
    const int i = blockIdx.x * BLOCK + threadIdx.x;
    if (begin + i >= end) return;
    const int p = list[begin + i];
    for (int k = w.row_ptrs[p]; k < w.row_ptrs[p + 1]; ++k) {
        const int c = w.cols[k];
        if (w.lvl[c] == -1) atomicMin(&w.key[c], i);
    }
}

// children of frontier position i = untouched neighbours whose key is i (a column twice in a row counts once:
// rows are ordered by column, so repeats are adjacent)
__global__ __launch_bounds__(BLOCK) void k_rcm_count(RcmWork w, const int32_t *__restrict__ list, int32_t begin,
                                                     int32_t end)
{
    const int i = blockIdx.x * BLOCK + threadIdx.x;
    if (begin + i >= end) return;
    const int p = list[begin + i];
    int n = 0, prev = -1;
    for (int k = w.row_ptrs[p]; k < w.row_ptrs[p + 1]; ++k) {
        const int c = w.cols[k];
        if (c != prev && w.lvl[c] == -1 && w.key[c] == i) ++n;
        prev = c;
    }
    w.cnt[i] = n;
}

__global__ __launch_bounds__(BLOCK) void k_rcm_emit(RcmWork w, int32_t *list, int32_t begin, int32_t end,
                                                    int32_t level, int by_degree)
{
    const int i = blockIdx.x * BLOCK + threadIdx.x;
    if (begin + i >= end) return;
    const int p = list[begin + i];
    int32_t *out = list + end + w.cnt[i];  // (cnt holds the exclusive offsets by now)
    int n = 0, prev = -1;
    for (int k = w.row_ptrs[p]; k < w.row_ptrs[p + 1]; ++k) {
        const int c = w.cols[k];
        if (c != prev && w.lvl[c] == -1 && w.key[c] == i) {
            // insertion by (degree, index) resp. by index: the row arrives in ascending index order
            int j = n;
            if (by_degree) {
                const int d = w.row_ptrs[c + 1] - w.row_ptrs[c];
                while (j > 0) {
                    const int o = out[j - 1];
                    const int od = w.row_ptrs[o + 1] - w.row_ptrs[o];
                    if (od > d || (od == d && o > c)) {
                        out[j] = o;
                        --j;
                    } else {
                        break;
                    }
                }
            }
            out[j] = c;
            ++n;
        }
        prev = c;
    }
    for (int j = 0; j < n; ++j) w.lvl[out[j]] = level + 1;  // (only now: the tests above must see them untouched)
}

__global__ __launch_bounds__(BLOCK) void k_rcm_far_node(RcmWork w, const int32_t *__restrict__ list, int32_t begin,
                                                        int32_t end)
{
    const int i = blockIdx.x * BLOCK + threadIdx.x;
    unsigned long long mine = ~0ull;
    if (begin + i < end) {
        const int v = list[begin + i];
        mine = ((unsigned long long)(unsigned)(w.row_ptrs[v + 1] - w.row_ptrs[v]) << 32) | (unsigned)i;
    }
#pragma unroll
    for (int off = WAVE / 2; off >= 1; off >>= 1) {
        const unsigned long long o = __shfl_xor(mine, off, WAVE);
        mine = o < mine ? o : mine;
    }
    if ((threadIdx.x & (WAVE - 1)) == 0 && mine != ~0ull) atomicMin(&w.cell[1], mine);
}

__global__ __launch_bounds__(BLOCK) void k_rcm_reset(RcmWork w, const int32_t *__restrict__ list, int32_t count)
{
    const int i = blockIdx.x * BLOCK + threadIdx.x;
    if (i >= count) return;
    const int v = list[i];
    w.lvl[v] = -1;
    w.key[v] = RCM_NO_KEY;
}

__global__ __launch_bounds__(BLOCK) void k_rcm_finish(RcmWork w, int32_t *new_id)
{
    const int k = blockIdx.x * BLOCK + threadIdx.x;
    if (k < w.n_rows) new_id[w.order[w.n_rows - 1 - k]] = k;
}

// ---- pattern in a new numbering ----
__global__ __launch_bounds__(BLOCK) void k_renumber_lens(RenumberWork w)
{
    const int r = blockIdx.x * BLOCK + threadIdx.x;
    if (r >= w.n_rows) return;
    const int k = w.new_id[r];
    w.old_of[k] = r;
    w.row_ptrs_out[k] = w.row_ptrs[r + 1] - w.row_ptrs[r];  // (lengths; scanned in place afterwards)
}

__global__ __launch_bounds__(BLOCK) void k_renumber_fill(RenumberWork w)
{
    const int k = blockIdx.x * BLOCK + threadIdx.x;  // new row
    if (k >= w.n_rows) return;
    const int r = w.old_of[k];
    const int src = w.row_ptrs[r], len = w.row_ptrs[r + 1] - src, dst = w.row_ptrs_out[k];
    int dp = -1;
    for (int i = 0; i < len; ++i) {  // stable insertion by new column: equal columns keep the reference's order
        const int c = w.new_id[w.cols[src + i]], m = w.map[src + i];
        int j = i;
        while (j > 0 && w.cols_out[dst + j - 1] > c) {
            w.cols_out[dst + j] = w.cols_out[dst + j - 1];
            w.map_out[dst + j] = w.map_out[dst + j - 1];
            --j;
        }
        w.cols_out[dst + j] = c;
        w.map_out[dst + j] = m;
    }
    for (int i = 0; i < len; ++i)
        if (w.cols_out[dst + i] == k) {
            dp = dst + i;
            break;
        }
    w.diag_pos_out[k] = dp;
}

// ---- packed columns for the CSR-stream kernel ----
__global__ __launch_bounds__(BLOCK) void k_s21_plan(Stream21Build b, int n_chunks_)
{
    __shared__ int smin[BLOCK / WAVE], smax[BLOCK / WAVE], sfar[BLOCK / WAVE];
    __shared__ int s_base;
    const int c = blockIdx.x;
    if (c >= n_chunks_) return;
    const int r0 = c * CHUNK_ROWS, r1 = min(r0 + CHUNK_ROWS, b.n_rows);
    const int nz0 = b.row_ptrs[r0], nz1 = b.row_ptrs[r1];
    int lo = INT32_MAX, hi = INT32_MIN;
    for (int e = nz0 + threadIdx.x; e < nz1; e += BLOCK) {
        const int col = b.cols[e];
        lo = min(lo, col);
        hi = max(hi, col);
    }
#pragma unroll
    for (int off = WAVE / 2; off >= 1; off >>= 1) {
        lo = min(lo, __shfl_xor(lo, off, WAVE));
        hi = max(hi, __shfl_xor(hi, off, WAVE));
    }
    if ((threadIdx.x & (WAVE - 1)) == 0) {
        smin[threadIdx.x / WAVE] = lo;
        smax[threadIdx.x / WAVE] = hi;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        for (int w = 1; w < BLOCK / WAVE; ++w) {
            lo = min(lo, smin[w]);
            hi = max(hi, smax[w]);
        }
        if (nz1 == nz0) lo = hi = 0;
        // all columns within 2^21 of the smallest: that one is the base; else a window around the chunk's own rows
        s_base = ((long)hi - (long)lo < (1L << STREAM21_BITS)) ? lo : stream21_window_base(r0, b.n_rows);
    }
    __syncthreads();
    const int base = s_base;
    int far = 0;
    for (int e = nz0 + threadIdx.x; e < nz1; e += BLOCK) {
        const long d = (long)b.cols[e] - base;
        far += (d < 0 || d >= (1L << STREAM21_BITS)) ? 1 : 0;
    }
#pragma unroll
    for (int off = WAVE / 2; off >= 1; off >>= 1) far += __shfl_xor(far, off, WAVE);
    if ((threadIdx.x & (WAVE - 1)) == 0) sfar[threadIdx.x / WAVE] = far;
    __syncthreads();
    if (threadIdx.x == 0) {
        for (int w = 1; w < BLOCK / WAVE; ++w) far += sfar[w];
        b.chunks[c].base = base;
        b.chunks[c].far_n = 0;  // (counted up again by the fill)
        b.far[c] = far;
        const int t0 = nz0 & ~3;
        const int tiles = nz1 > nz0 ? (nz1 - t0 + STREAM21_TILE - 1) / STREAM21_TILE : 0;
        b.words[c] = tiles * STREAM21_GROUPS * BLOCK;
    }
}

__global__ __launch_bounds__(BLOCK) void k_s21_fill(Stream21Build b, int n_chunks_, uint4 *__restrict__ codes,
                                                     int *__restrict__ far_idx, int *__restrict__ far_col)
{
    const int c = blockIdx.x;
    if (c >= n_chunks_) return;
    const int r0 = c * CHUNK_ROWS, r1 = min(r0 + CHUNK_ROWS, b.n_rows);
    const int nz0 = b.row_ptrs[r0], nz1 = b.row_ptrs[r1];
    const int base = b.chunks[c].base, word_off = b.words[c], far_off = b.far[c];
    if (threadIdx.x == 0) {
        b.chunks[c].word_off = word_off;
        b.chunks[c].far_off = far_off;
    }
    const int tid = threadIdx.x;
    int tile = 0;
    for (int t0 = nz0 & ~3; t0 < nz1; t0 += STREAM21_TILE, ++tile)
        for (int g = 0; g < STREAM21_GROUPS; ++g) {
            unsigned long long code[6];
            for (int k = 0; k < 3; ++k)
                for (int j = 0; j < 2; ++j) {
                    const int e = t0 + ((g * 3 + k) * BLOCK + tid) * 2 + j;
                    unsigned long long cd = 0ull;
                    if (e >= nz0 && e < nz1) {
                        const int col = b.cols[e];
                        const long d = (long)col - base;
                        if (d < 0 || d >= (1L << STREAM21_BITS)) {
                            // far: coded as offset 0 (a valid column), listed with its entry and column -- the order
                            // within the chunk's list does not matter, every entry is put right on its own
                            const int at = far_off + atomicAdd(&b.chunks[c].far_n, 1);
                            far_idx[at] = e;
                            far_col[at] = col;
                        } else {
                            cd = (unsigned long long)d;
                        }
                    }
                    code[2 * k + j] = cd;
                }
            const unsigned long long lo = code[0] | (code[1] << 21) | (code[2] << 42) | (code[3] << 63);
            const unsigned long long hi = (code[3] >> 1) | (code[4] << 20) | (code[5] << 41);
            uint4 w;
            w.x = (unsigned)lo;
            w.y = (unsigned)(lo >> 32);
            w.z = (unsigned)hi;
            w.w = (unsigned)(hi >> 32);
            codes[(size_t)word_off + (size_t)(tile * STREAM21_GROUPS + g) * BLOCK + tid] = w;
        }
}

// ---- Hilbert keys (hilbert_key, host_matrix.cpp: J. Skilling, "Programming the Hilbert curve", AIP Conf. Proc. 707 (2004))
__device__ inline unsigned long long hilbert_key_dev(uint32_t x, uint32_t y, uint32_t z)
{
    constexpr int BITS = 16;
    uint32_t X[3] = {x, y, z};
    for (uint32_t Q = 1u << (BITS - 1); Q > 1; Q >>= 1) {
        const uint32_t P = Q - 1;
#pragma unroll
        for (int i = 0; i < 3; ++i) {
            if (X[i] & Q) {
                X[0] ^= P;
            } else {
                const uint32_t t = (X[0] ^ X[i]) & P;
                X[0] ^= t;
                X[i] ^= t;
            }
        }
    }
#pragma unroll
    for (int i = 1; i < 3; ++i) X[i] ^= X[i - 1];
    uint32_t t = 0;
    for (uint32_t Q = 1u << (BITS - 1); Q > 1; Q >>= 1)
        if (X[2] & Q) t ^= Q - 1;
#pragma unroll
    for (int i = 0; i < 3; ++i) X[i] ^= t;
    unsigned long long key = 0;
    for (int b = BITS - 1; b >= 0; --b)
#pragma unroll
        for (int i = 0; i < 3; ++i) key = (key << 1) | ((X[i] >> b) & 1u);
    return key;
}

__global__ __launch_bounds__(BLOCK) void k_hilbert_keys(int n, const double *__restrict__ centres, double lo0, double lo1,
                                                        double lo2, double scale, unsigned long long *__restrict__ keys,
                                                        int *__restrict__ cells)
{
    const int c = blockIdx.x * BLOCK + threadIdx.x;
    if (c >= n) return;
    const double lo[3] = {lo0, lo1, lo2};
    uint32_t q[3];
#pragma unroll
    for (int d = 0; d < 3; ++d) {
        const double v = (centres[3 * (size_t)c + d] - lo[d]) * scale;  // (-ffp-contract=off: the host's two roundings)
        q[d] = (uint32_t)fmin(65535.0, fmax(0.0, v));
    }
    keys[c] = hilbert_key_dev(q[0], q[1], q[2]);
    cells[c] = c;
}

__global__ __launch_bounds__(BLOCK) void k_invert_order(int n, const int *__restrict__ sorted_cells, int *__restrict__ new_id)
{
    const int k = blockIdx.x * BLOCK + threadIdx.x;
    if (k < n) new_id[sorted_cells[k]] = k;
}

// one workgroup per chunk of CHUNK_ROWS rows of the NEW numbering: smallest / largest new column of the chunk's entries;
// when they span 2^21 or more, the entries outside the chunk's window (stream21_window_base) are counted
__global__ __launch_bounds__(BLOCK) void k_curve_far_count(int n_rows, const int *__restrict__ row_ptrs,
                                                           const int *__restrict__ cols, const int *__restrict__ new_id,
                                                           const int *__restrict__ old_of, unsigned long long *far_out)
{
    __shared__ int s_lo[BLOCK], s_hi[BLOCK];
    __shared__ unsigned long long s_far[BLOCK];
    const int k0 = blockIdx.x * CHUNK_ROWS, k1 = min(n_rows, k0 + CHUNK_ROWS);
    int lo = n_rows, hi = 0;
    for (int k = k0 + (int)threadIdx.x; k < k1; k += BLOCK) {
        const int row = old_of[k];
        for (int e = row_ptrs[row]; e < row_ptrs[row + 1]; ++e) {
            const int c = new_id[cols[e]];
            lo = min(lo, c);
            hi = max(hi, c);
        }
    }
    s_lo[threadIdx.x] = lo;
    s_hi[threadIdx.x] = hi;
    __syncthreads();
    for (int w = BLOCK / 2; w > 0; w >>= 1) {
        if ((int)threadIdx.x < w) {
            s_lo[threadIdx.x] = min(s_lo[threadIdx.x], s_lo[threadIdx.x + w]);
            s_hi[threadIdx.x] = max(s_hi[threadIdx.x], s_hi[threadIdx.x + w]);
        }
        __syncthreads();
    }
    lo = s_lo[0];
    hi = s_hi[0];
    if (hi < lo || (long long)hi - lo < (1ll << STREAM21_BITS)) return;  // (workgroup-uniform)
    const long long base = stream21_window_base(k0, n_rows);
    unsigned long long far = 0;
    for (int k = k0 + (int)threadIdx.x; k < k1; k += BLOCK) {
        const int row = old_of[k];
        for (int e = row_ptrs[row]; e < row_ptrs[row + 1]; ++e) {
            const long long d = (long long)new_id[cols[e]] - base;
            far += (d < 0 || d >= (1ll << STREAM21_BITS)) ? 1 : 0;
        }
    }
    s_far[threadIdx.x] = far;
    __syncthreads();
    for (int w = BLOCK / 2; w > 0; w >>= 1) {
        if ((int)threadIdx.x < w) s_far[threadIdx.x] += s_far[threadIdx.x + w];
        __syncthreads();
    }
    if (threadIdx.x == 0 && s_far[0]) atomicAdd(far_out, s_far[0]);
}

}  // namespace

size_t scan_tmp_len(int64_t n) { return (size_t)((n + SCAN_TILE - 1) / SCAN_TILE) + 2; }

void launch_exclusive_scan(hipStream_t st, const int32_t *in, int32_t *out, int64_t n, int32_t *tmp)
{
    const int nb = (int)((n + SCAN_TILE - 1) / SCAN_TILE);
    if (nb == 0) {
        (void)hipMemsetAsync(out, 0, sizeof(int32_t), st);
        return;
    }
    hipLaunchKernelGGL(k_scan_tile_sums, dim3(nb), dim3(SCAN_BLOCK), 0, st, in, n, tmp);
    hipLaunchKernelGGL(k_scan_sums, dim3(1), dim3(SCAN_BLOCK), 0, st, tmp, nb);
    hipLaunchKernelGGL(k_scan_apply, dim3(nb), dim3(SCAN_BLOCK), 0, st, in, out, n, tmp, nb);
}

void launch_build_pattern(hipStream_t st, const PatternBuild &b)
{
    if (b.n_rows == 0) {
        (void)hipMemsetAsync(b.row_ptrs, 0, sizeof(int32_t), st);
        return;
    }
    const int64_t items = std::max<int64_t>(std::max(b.n_rows, b.n_faces), b.n_iface);
    (void)hipMemsetAsync(b.counts, 0, (size_t)b.n_rows * sizeof(int32_t), st);
    (void)hipMemsetAsync(b.diag_pos, 0, (size_t)b.n_rows * sizeof(int32_t), st);
    hipLaunchKernelGGL(k_pat_count, dim3(blocks_for(items)), dim3(BLOCK), 0, st, b);
    launch_exclusive_scan(st, b.counts, b.row_ptrs, b.n_rows, b.scan_tmp);
    hipLaunchKernelGGL(k_pat_fill, dim3(blocks_for(items)), dim3(BLOCK), 0, st, b);
    hipLaunchKernelGGL(k_pat_sort, dim3(blocks_for(b.n_rows)), dim3(BLOCK), 0, st, b.n_rows, b.row_ptrs, b.cols,
                       b.ldu_mapping, b.diag_pos);
}

void launch_sym_distances(hipStream_t st, int32_t n_rows, const int32_t *row_ptrs, const int32_t *cols,
                          int32_t *table, int32_t *flags)
{
    if (n_rows == 0) return;
    hipLaunchKernelGGL(k_sym_distances, dim3(blocks_for(n_rows)), dim3(BLOCK), 0, st, n_rows, row_ptrs, cols, table,
                       flags);
}

void launch_sym_fill(hipStream_t st, int32_t n_rows, const int32_t *row_ptrs, const int32_t *cols,
                     SymDistances dist, uint8_t *mask, int32_t *map, int32_t *flags)
{
    if (n_rows == 0) return;
    hipLaunchKernelGGL(k_sym_fill, dim3(blocks_for(n_rows)), dim3(BLOCK), 0, st, n_rows, row_ptrs, cols, dist, mask,
                       map, flags);
}

void launch_rcm_init(hipStream_t st, const RcmWork &w)
{
    if (w.n_rows) hipLaunchKernelGGL(k_rcm_init, dim3(blocks_for(w.n_rows)), dim3(BLOCK), 0, st, w);
}

void launch_rcm_find_seed(hipStream_t st, const RcmWork &w)
{
    (void)hipMemsetAsync(w.cell, 0xFF, sizeof(unsigned long long), st);
    if (w.n_rows) hipLaunchKernelGGL(k_rcm_find_seed, dim3(blocks_for(w.n_rows)), dim3(BLOCK), 0, st, w);
}

void launch_rcm_start(hipStream_t st, const RcmWork &w, int32_t *list, int32_t pos, int32_t node, int32_t level)
{
    hipLaunchKernelGGL(k_rcm_start, dim3(1), dim3(64), 0, st, w, list, pos, node, level);
}

void launch_rcm_level(hipStream_t st, const RcmWork &w, int32_t *list, int32_t begin, int32_t end, int32_t level,
                      bool by_degree)
{
    const int m = end - begin;
    if (m <= 0) return;
    const dim3 grid(blocks_for(m)), block(BLOCK);
    hipLaunchKernelGGL(k_rcm_mark, grid, block, 0, st, w, list, begin, end);
    hipLaunchKernelGGL(k_rcm_count, grid, block, 0, st, w, list, begin, end);
    launch_exclusive_scan(st, w.cnt, w.cnt, m, w.scan_tmp);
    hipLaunchKernelGGL(k_rcm_emit, grid, block, 0, st, w, list, begin, end, level, by_degree ? 1 : 0);
}

void launch_rcm_far_node(hipStream_t st, const RcmWork &w, const int32_t *list, int32_t begin, int32_t end)
{
    (void)hipMemsetAsync(w.cell + 1, 0xFF, sizeof(unsigned long long), st);
    if (end > begin)
        hipLaunchKernelGGL(k_rcm_far_node, dim3(blocks_for(end - begin)), dim3(BLOCK), 0, st, w, list, begin, end);
}

void launch_rcm_reset(hipStream_t st, const RcmWork &w, const int32_t *list, int32_t count)
{
    if (count > 0) hipLaunchKernelGGL(k_rcm_reset, dim3(blocks_for(count)), dim3(BLOCK), 0, st, w, list, count);
}

void launch_rcm_finish(hipStream_t st, const RcmWork &w, int32_t *new_id)
{
    if (w.n_rows) hipLaunchKernelGGL(k_rcm_finish, dim3(blocks_for(w.n_rows)), dim3(BLOCK), 0, st, w, new_id);
}

void launch_renumber_pattern(hipStream_t st, const RenumberWork &w)
{
    if (w.n_rows == 0) {
        (void)hipMemsetAsync(w.row_ptrs_out, 0, sizeof(int32_t), st);
        return;
    }
    hipLaunchKernelGGL(k_renumber_lens, dim3(blocks_for(w.n_rows)), dim3(BLOCK), 0, st, w);
    launch_exclusive_scan(st, w.row_ptrs_out, w.row_ptrs_out, w.n_rows, w.scan_tmp);
    hipLaunchKernelGGL(k_renumber_fill, dim3(blocks_for(w.n_rows)), dim3(BLOCK), 0, st, w);
}

void launch_stream21_plan(hipStream_t st, const Stream21Build &b)
{
    const int nc = (int)n_chunks(b.n_rows);
    if (nc == 0) return;
    hipLaunchKernelGGL(k_s21_plan, dim3(nc), dim3(BLOCK), 0, st, b, nc);
    launch_exclusive_scan(st, b.words, b.words, nc, b.scan_tmp);
    launch_exclusive_scan(st, b.far, b.far, nc, b.scan_tmp);
}

void launch_stream21_fill(hipStream_t st, const Stream21Build &b, uint4 *codes, int32_t *far_idx, int32_t *far_col)
{
    const int nc = (int)n_chunks(b.n_rows);
    if (nc == 0) return;
    hipLaunchKernelGGL(k_s21_fill, dim3(nc), dim3(BLOCK), 0, st, b, nc, codes, far_idx, far_col);
}

size_t hilbert_sort_temp_bytes(int32_t n)
{
    size_t bytes = 0;
    (void)hipcub::DeviceRadixSort::SortPairs(nullptr, bytes, (const unsigned long long *)nullptr, (unsigned long long *)nullptr,
                                             (const int *)nullptr, (int *)nullptr, n, 0, 48, (hipStream_t) nullptr);
    return bytes;
}

int hilbert_order_device(hipStream_t st, int32_t n, const double *centres, const double lo[3], double scale,
                         unsigned long long *keys, unsigned long long *keys_out, int32_t *cells, int32_t *cells_out,
                         void *temp, size_t temp_bytes, int32_t *new_id)
{
    if (n <= 0) return OGL_OK;
    hipLaunchKernelGGL(k_hilbert_keys, dim3(blocks_for(n)), dim3(BLOCK), 0, st, n, centres, lo[0], lo[1], lo[2], scale, keys,
                       cells);
    // (a radix sort is stable: cells with equal keys keep the caller's order, as std::sort of (key, cell) pairs gives)
    const hipError_t e = hipcub::DeviceRadixSort::SortPairs(temp, temp_bytes, keys, keys_out, cells, cells_out, n, 0, 48, st);
    if (e != hipSuccess) return fail(OGL_ERR_HIP, "radix sort of the Hilbert keys failed: %s", hipGetErrorString(e));
    hipLaunchKernelGGL(k_invert_order, dim3(blocks_for(n)), dim3(BLOCK), 0, st, n, cells_out, new_id);
    return OGL_OK;
}

void launch_curve_far_count(hipStream_t st, int32_t n_rows, const int32_t *row_ptrs, const int32_t *cols,
                            const int32_t *new_id, const int32_t *old_of, unsigned long long *far_out)
{
    if (n_rows <= 0) return;
    hipLaunchKernelGGL(k_curve_far_count, dim3((unsigned)n_chunks(n_rows)), dim3(BLOCK), 0, st, n_rows, row_ptrs, cols, new_id,
                       old_of, far_out);
}

}  // namespace ogl
