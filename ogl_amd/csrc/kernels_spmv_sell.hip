// kernels_spmv_sell.hip -- general half storage (symx), index-compressed chunked ELL (sell) and their value refresh
// (geometry, reduction tree and the -ffp-contract=off rule: device_common.hpp)
#include "device_common.hpp"

namespace ogl {

namespace {

// ------------------------------------------------------------------------------------------
// Half storage with per-chunk distances and explicit exceptions (SymxChunk, common.hpp; build_symx_layout).
// As k_spmv_sym, but the distances are the chunk's own (run-time values from its header, which also says where
// the twins of its lower entries live: in the planes of the chunk of row r - d or of the next one), and rows
// flagged in their mask byte carry explicit entries -- column + value lists -- that are merged into the row sum by
// column, so that every row is still summed in ascending column order and y keeps the bits of the other kernels.
// ------------------------------------------------------------------------------------------
// The explicit entries of a chunk (few: the couplings across block faces) are staged by its workgroup -- thread i
// takes entry i: column and value * x[column] into LDS -- while the plane loads are in flight; a row then walks its
// own entries in LDS.  Per row that costs two registers (next entry, end) where prefetching the first entries into
// registers cost 40 for the kernel, i.e. three of eight wavefronts per SIMD.  A chunk with more than
// SYMX_LDS_ENTRIES of them reads them from memory as it goes.
template <int MODE, bool IN_LDS>
__device__ __forceinline__ void symx_explicit(double &acc, int &k, int end, int limit, const int *lds_cols,
                                              const double *lds_prod, const int *__restrict__ ex_cols,
                                              const double *__restrict__ ex_vals, const double *__restrict__ x)
{
    while (k < end) {
        const int c = IN_LDS ? lds_cols[k] : ex_cols[k];
        if (c >= limit) break;
        const double p = IN_LDS ? lds_prod[k] : ex_vals[k] * x[c];
        acc = (MODE == SPMV_RESIDUAL) ? acc - p : acc + p;
        ++k;
    }
}

// FAST (as in k_spmv_sym; known per layout: every chunk's first distance is 1 and its further ones are even --
// blocks with even line lengths): the two rows of a lane are an aligned pair in every strip, so x and the lower
// values of the even distances come as one 16-byte load per pair and the d = 1 neighbours from the lane's own
// diagonal pair and plane-1 value.
template <int MODE, int NDOT, bool STREAM, bool FAST, bool GENERAL>
__global__ __launch_bounds__(BLOCK) void k_spmv_symx(int n_rows, int n_chunks, const SymxChunk *__restrict__ hdr,
                                                     const uint8_t *__restrict__ mask,
                                                     const double *__restrict__ planes,
                                                     const int *__restrict__ ex_rowptr,
                                                     const int *__restrict__ ex_cols,
                                                     const double *__restrict__ ex_vals,
                                                     const int *__restrict__ ex_lrow,
                                                     const double *__restrict__ x, const double *__restrict__ b,
                                                     double *__restrict__ y, const double *__restrict__ w,
                                                     double *__restrict__ dot_partials,
                                                     double *__restrict__ dot2_partials, const DevScalars *gate,
                                                     HaloFused hf)
{
    __shared__ double slot[N_WAVES];
    __shared__ double ys[CHUNK_ROWS];
    // GENERAL: the chunk's explicit entries as a list (column, product), walked by their rows; lean kernel: at most
    // one ahead of and one behind the planar entries of a row -- a slot per row for each, and a flag that says it is taken
    __shared__ int ex_c[GENERAL ? SYMX_LDS_ENTRIES : 1];
    __shared__ double ex_p[GENERAL ? SYMX_LDS_ENTRIES : 1];
    __shared__ double ex_a[GENERAL ? 1 : CHUNK_ROWS], ex_b[GENERAL ? 1 : CHUNK_ROWS];
    __shared__ unsigned char ex_fa[GENERAL ? 2 : CHUNK_ROWS], ex_fb[GENERAL ? 2 : CHUNK_ROWS];
    if (gate && gate->stop) return;
    // the headers are stored in dispatch order and name their chunk: one round trip (header -> data) instead of
    // two (order -> header -> data) in front of the loads.  The header is only ever indexed with compile-time
    // constants (every loop below is fully unrolled): it stays in scalar registers; a run-time index would push all
    // 96 bytes into scratch memory
    const SymxChunk h = hdr[blockIdx.x];
    const int chunk = h.chunk;
    if (chunk < 0 || chunk >= n_chunks) return;
    const int t = threadIdx.x;
    const RowPair rp = my_rows(chunk, n_rows);
    const int row = rp.row, r0 = chunk * CHUNK_ROWS;
    const unsigned mm = *reinterpret_cast<const unsigned short *>(mask + row);
    const unsigned m0 = mm & 0xffu, m1 = mm >> 8;
    const bool chunk_explicit = h.ex_rp_off >= 0;                                   // (workgroup-uniform)
    const bool has_explicit = GENERAL && chunk_explicit && ((m0 | m1) & SYMX_EXTRAS_BIT);   // this lane's rows
    const bool in_lds = h.ex_count <= SYMX_LDS_ENTRIES;
    // GENERAL: next explicit entry / end of the two rows, as indices into the staged entries (or into the arrays)
    int k0 = 0, e0 = 0, k1 = 0, e1 = 0;
    if (has_explicit) {
        const int *rpx = ex_rowptr + h.ex_rp_off + t * ROWS_PER_THREAD;
        const int shift = in_lds ? h.ex_begin : 0;
        k0 = rpx[0] - shift;
        e0 = k1 = rpx[1] - shift;
        e1 = rpx[2] - shift;
    }
    // the entry this thread stages (entry t of the chunk; the few chunks with more than 256 loop)
    int my_col = 0, my_lrow = 0;
    double my_val = 0.0;
    const bool stage = chunk_explicit && (in_lds || !GENERAL) && t < h.ex_count;
    if (stage) {
        my_col = ex_cols[h.ex_begin + t];
        my_val = ex_vals[h.ex_begin + t];
        if (!GENERAL) my_lrow = ex_lrow[h.ex_begin + t];
    }
    if (!GENERAL && chunk_explicit) {  // (own rows' flags down; the barrier below orders this before the staging)
        *reinterpret_cast<unsigned short *>(ex_fa + ROWS_PER_THREAD * t) = 0;
        *reinterpret_cast<unsigned short *>(ex_fb + ROWS_PER_THREAD * t) = 0;
    }
    double2 acc;
    acc.x = acc.y = 0.0;
    if (MODE == SPMV_RESIDUAL) acc = ld2(b, rp);
    // Distances, plane positions and twin bases padded to three with dummies (distance 0, plane 0) that no mask bit
    // refers to, and every load issued whatever the mask says, at an index clamped into its array: straight-line
    // code -- a run-time "does this plane exist" in front of each group of loads splits them into basic blocks that
    // wait for each other (tools/sym_tune.hip var1/var2)
    const int last = n_rows - 1, last_pair = last & ~1;  // (a pair load at the last even row: the vectors are allocated two past n_rows)
    int dj[4] = {0, 0, 0, 0};
    long pj[4] = {0, 0, 0, 0}, b0[4], b1[4];
#pragma unroll
    for (int j = 1; j < 4; ++j) {
        const bool have = j < h.nd;
        dj[j] = have ? h.d[j - 1] : 0;
        pj[j] = have ? (long)j * CHUNK_ROWS : 0;
        b0[j] = (have && h.lo_base[j - 1][0] >= 0) ? h.lo_base[j - 1][0] : h.val_off;
        b1[j] = (have && h.lo_base[j - 1][1] >= 0) ? h.lo_base[j - 1][1] : h.val_off;
    }
    // own planes: diagonal and upper entries of the two rows
    double2 up[4];
    const double *own = planes + h.val_off + t * ROWS_PER_THREAD;
#pragma unroll
    for (int j = 0; j < 4; ++j)
        up[j] = (STREAM && (j == 0 || (FAST && j == 1))) ? ld_pair_stream(own + pj[j])
                                                         : *reinterpret_cast<const double2 *>(own + pj[j]);
    const double2 xd = ld2(x, rp);
    double2 lo[4], xl[4], xu[4];
#pragma unroll
    for (int j = 1; j < 4; ++j) {
        const int d = dj[j];
        // first row of the chunk minus d: its chunk (floor division) is lo_base[.][0], the next one [.][1]
        const int cs0 = (r0 - d) >> 9;
        if (FAST && j >= 2) {  // even distance: rows (row - d, row + 1 - d) are an aligned pair of one chunk's plane
            const int rs = row - d;
            lo[j] = *reinterpret_cast<const double2 *>(planes + ((rs >> 9) == cs0 ? b0[j] : b1[j]) +
                                                       (rs & (CHUNK_ROWS - 1)));
            xl[j] = *reinterpret_cast<const double2 *>(x + min(max(rs, 0), last_pair));
            xu[j] = *reinterpret_cast<const double2 *>(x + min(row + d, last_pair));
        } else if (FAST) {     // d = 1: A(row + 1, row) is this lane's own plane-1 value of row
            const int rs = row - 1;
            const long at = ((rs >> 9) == cs0 ? b0[j] : b1[j]) + (rs & (CHUNK_ROWS - 1));
            if (STREAM) {      // A(row, row - 1) is the previous lane's second plane-1 value (lane 0: from memory)
                lo[j].x = __shfl_up(up[1].y, 1, WAVE);
                if ((t & (WAVE - 1)) == 0) lo[j].x = planes[at];
            } else {
                lo[j].x = planes[at];
            }
            xl[j].x = x[min(max(rs, 0), last)];
            lo[j].y = up[1].x;
            xl[j].y = xd.x;
            xu[j].x = xd.y;
            xu[j].y = x[min(row + 2, last)];
        } else {
            const int ra = row - d, rb = row + 1 - d;
            lo[j].x = planes[((ra >> 9) == cs0 ? b0[j] : b1[j]) + (ra & (CHUNK_ROWS - 1))];
            lo[j].y = planes[((rb >> 9) == cs0 ? b0[j] : b1[j]) + (rb & (CHUNK_ROWS - 1))];
            xl[j].x = x[min(max(ra, 0), last)];
            xl[j].y = x[min(max(rb, 0), last)];
            xu[j].x = x[min(row + d, last)];
            xu[j].y = x[min(row + 1 + d, last)];
        }
    }
    static_assert(CHUNK_ROWS == 512, "row >> 9 above");
    // staging of the chunk's explicit entries (their x gather is the last link of the header -> entry -> x chain; the
    // plane and x loads above are in flight meanwhile)
    if (GENERAL && chunk_explicit && in_lds) {
        for (int i = t; i < h.ex_count; i += BLOCK) {
            const int c = i == t ? my_col : ex_cols[h.ex_begin + i];
            const double v = i == t ? my_val : ex_vals[h.ex_begin + i];
            ex_c[i] = c;
            ex_p[i] = v * x[c];
        }
        __syncthreads();
    }
    if (!GENERAL && chunk_explicit) {
        __syncthreads();
        for (int i = t; i < h.ex_count; i += BLOCK) {
            const int c = i == t ? my_col : ex_cols[h.ex_begin + i];
            const double v = i == t ? my_val : ex_vals[h.ex_begin + i];
            const int lr = i == t ? my_lrow : ex_lrow[h.ex_begin + i];
            const double p = v * x[c];
            if (lr & SYMX_BEHIND_BIT) {
                ex_b[lr & (CHUNK_ROWS - 1)] = p;
                ex_fb[lr & (CHUNK_ROWS - 1)] = 1;
            } else {
                ex_a[lr] = p;
                ex_fa[lr] = 1;
            }
        }
        __syncthreads();
        const int l0 = ROWS_PER_THREAD * t;
        if (ex_fa[l0]) acc.x = (MODE == SPMV_RESIDUAL) ? acc.x - ex_a[l0] : acc.x + ex_a[l0];
        if (ex_fa[l0 + 1]) acc.y = (MODE == SPMV_RESIDUAL) ? acc.y - ex_a[l0 + 1] : acc.y + ex_a[l0 + 1];
    }
    // the row walk in ascending column order: the furthest lower entry first.  Explicit entries: the lean kernel has
    // added the one ahead of the planar entries above and adds the one behind them below; the general kernel merges
    // a row's list by column (an entry that repeats a column a plane holds comes after the plane's entry: `<` in
    // symx_explicit)
    auto ex_row0 = [&](int limit) {
        if (in_lds)
            symx_explicit<MODE, true>(acc.x, k0, e0, limit, ex_c, ex_p, ex_cols, ex_vals, x);
        else
            symx_explicit<MODE, false>(acc.x, k0, e0, limit, ex_c, ex_p, ex_cols, ex_vals, x);
    };
    auto ex_row1 = [&](int limit) {
        if (in_lds)
            symx_explicit<MODE, true>(acc.y, k1, e1, limit, ex_c, ex_p, ex_cols, ex_vals, x);
        else
            symx_explicit<MODE, false>(acc.y, k1, e1, limit, ex_c, ex_p, ex_cols, ex_vals, x);
    };
    // (the general kernel only sees chunks with SymxChunk::merge set: its rows' explicit entries are merged by column)
    const bool merge = has_explicit;
#pragma unroll
    for (int j = 3; j >= 1; --j) {
        if ((m0 >> (3 - j)) & 1u) {
            if (merge) ex_row0(row - dj[j]);
            const double p = lo[j].x * xl[j].x;
            acc.x = (MODE == SPMV_RESIDUAL) ? acc.x - p : acc.x + p;
        }
        if ((m1 >> (3 - j)) & 1u) {
            if (merge) ex_row1(row + 1 - dj[j]);
            const double p = lo[j].y * xl[j].y;
            acc.y = (MODE == SPMV_RESIDUAL) ? acc.y - p : acc.y + p;
        }
    }
    if ((m0 >> 3) & 1u) {
        if (merge) ex_row0(row);
        const double p = up[0].x * xd.x;
        acc.x = (MODE == SPMV_RESIDUAL) ? acc.x - p : acc.x + p;
    }
    if ((m1 >> 3) & 1u) {
        if (merge) ex_row1(row + 1);
        const double p = up[0].y * xd.y;
        acc.y = (MODE == SPMV_RESIDUAL) ? acc.y - p : acc.y + p;
    }
#pragma unroll
    for (int j = 1; j < 4; ++j) {
        if ((m0 >> (3 + j)) & 1u) {
            if (merge) ex_row0(row + dj[j]);
            const double p = up[j].x * xu[j].x;
            acc.x = (MODE == SPMV_RESIDUAL) ? acc.x - p : acc.x + p;
        }
        if ((m1 >> (3 + j)) & 1u) {
            if (merge) ex_row1(row + 1 + dj[j]);
            const double p = up[j].y * xu[j].y;
            acc.y = (MODE == SPMV_RESIDUAL) ? acc.y - p : acc.y + p;
        }
    }
    if (has_explicit) {
        ex_row0(INT32_MAX);
        ex_row1(INT32_MAX);
    }
    if (!GENERAL && chunk_explicit) {
        const int l0 = ROWS_PER_THREAD * t;
        if (ex_fb[l0]) acc.x = (MODE == SPMV_RESIDUAL) ? acc.x - ex_b[l0] : acc.x + ex_b[l0];
        if (ex_fb[l0 + 1]) acc.y = (MODE == SPMV_RESIDUAL) ? acc.y - ex_b[l0 + 1] : acc.y + ex_b[l0 + 1];
    }
    if (hf.chunk_bptr) halo_fused_add<MODE>(hf, chunk, acc.x, acc.y, ys);
    st2(y, rp, acc);
    if (NDOT >= 1) {
        const double2 vw = ld2(w, rp);
        double d = 0.0, d2 = 0.0;
        if (rp.n > 0) {
            d += vw.x * acc.x;
            d2 += acc.x * acc.x;
        }
        if (rp.n > 1) {
            d += vw.y * acc.y;
            d2 += acc.y * acc.y;
        }
        const double s = block_sum(d, slot);
        if (threadIdx.x == 0) dot_partials[chunk] = s;
        if (NDOT >= 2) {
            const double s2 = block_sum(d2, slot);
            if (threadIdx.x == 0) dot2_partials[chunk] = s2;
        }
    }
}

// ------------------------------------------------------------------------------------------
// Index-compressed chunked ELL SpMV (SellChunk, common.hpp).  Values: eight 16-byte loads per lane
// and group of 8 slots.  Columns, per chunk: pattern mode -- one 2-byte load brings the pattern ids
// of the lane's two rows, the pattern table in LDS gives the offsets; offset mode -- one 16-byte
// load brings 16 column codes, the offset dictionary in LDS turns a code into column = row +
// offset.  8.1 / 9 bytes per stored entry instead of 12 -- measured 137 / 146 us against 187 us
// for the CSR-stream kernel on the 216^3 matrix (profiles/spmv_tune_r01.txt).  Same per-row order
// as k_spmv_stream.
// ------------------------------------------------------------------------------------------
// STREAM: as in k_spmv_stream -- value planes and 16 / 32-bit code words of a matrix larger than the Infinity
// Cache are streamed past the caches.
template <int MODE, int NDOT, bool STREAM>
__global__ __launch_bounds__(BLOCK) void k_spmv_sell(int n_rows, int n_chunks,
                                                     const SellChunk *__restrict__ chunks,
                                                     const int *__restrict__ dict,
                                                     const uint8_t *__restrict__ codes,
                                                     const double *__restrict__ vals,
                                                     const int *__restrict__ spill_chunk_ptr,
                                                     const int *__restrict__ spill_rows,
                                                     const int *__restrict__ spill_ptrs,
                                                     const int *__restrict__ spill_cols,
                                                     const double *__restrict__ spill_vals,
                                                     const double *__restrict__ x,
                                                     const double *__restrict__ b,
                                                     double *__restrict__ y,
                                                     const double *__restrict__ w,
                                                     double *__restrict__ dot_partials,
                                                     double *__restrict__ dot2_partials,
                                                     const DevScalars *gate, int xgroup, HaloFused hf,
                                                     const uint16_t *__restrict__ rmap,
                                                     const int *__restrict__ block_order)
{
    __shared__ double slot[N_WAVES];
    __shared__ int stab[SELL_TABLE_INTS];
    static_assert(SELL_TABLE_INTS * sizeof(int) >= CHUNK_ROWS * sizeof(double), "the table doubles as the row-sum exchange");
    if (gate && gate->stop) return;
    const int chunk = block_order ? block_order[blockIdx.x] : xcd_chunk(blockIdx.x, xgroup);
    if (chunk < 0 || chunk >= n_chunks) return;
    const SellChunk h = chunks[chunk];
    const int t = threadIdx.x;
    // (DevSell::rmap: whose rows this thread's two slot rows hold -- asked for now, needed at the very end)
    unsigned own = 0;
    if (rmap) own = *reinterpret_cast<const unsigned *>(rmap + (long)chunk * CHUNK_ROWS + t * ROWS_PER_THREAD);
    // spill (the tails of this chunk's long rows): which row this thread will finish and where its
    // tail sits -- asked for now, so that the answers arrive while the planes are being worked on
    int sp0 = 0, sp1 = 0, s_row = 0, s_k0 = 0, s_k1 = 0;
    if (spill_chunk_ptr) {
        sp0 = spill_chunk_ptr[chunk];
        sp1 = spill_chunk_ptr[chunk + 1];
        if (sp0 + t < sp1) {
            s_row = spill_rows[sp0 + t];
            s_k0 = spill_ptrs[sp0 + t];
            s_k1 = spill_ptrs[sp0 + t + 1];
        }
    }
    if (h.mode() <= SELL_MODE_OFFSET8)  // (no table in delta / column mode)
        for (int i = t; i < h.dict_len(); i += BLOCK) stab[i] = dict[h.dict_off + i];
    __syncthreads();
    const RowPair rp = my_rows(chunk, n_rows);
    const int row = chunk * CHUNK_ROWS + t * ROWS_PER_THREAD;
    double2 acc;
    acc.x = acc.y = 0.0;
    if (MODE == SPMV_RESIDUAL) {
        if (rmap) {  // (b of the rows the slot rows hold)
            const int ra = chunk * CHUNK_ROWS + (int)(own & 0xffffu), rb = chunk * CHUNK_ROWS + (int)(own >> 16);
            acc.x = ra < n_rows ? b[ra] : 0.0;
            acc.y = rb < n_rows ? b[rb] : 0.0;
        } else {
            acc = ld2(b, rp);
        }
    }
    const double *v = vals + h.val_off + t * ROWS_PER_THREAD;
    // this wavefront runs to the longest of ITS rows: the planes beyond (padding up to the chunk's
    // longest row) are never touched.  Wave-uniform, so the loops below do not diverge.
    const int ww = h.wave_width(__builtin_amdgcn_readfirstlane(t / WAVE)), width = h.width();
    constexpr int BATCH = 8;
    // ... and every lane loads up to the longer of ITS two rows only (lengths: one byte per row in front
    // of the codes; not in pattern mode).  Padding that shares no 128-byte line with a slot in use costs
    // no memory traffic -- with the rows of a wavefront sorted by length that is nearly all of it.
    int ml = ww;
    if (h.mode() != SELL_MODE_PATTERN) {
        const unsigned ll =
            *reinterpret_cast<const unsigned short *>(codes + h.code_off - SELL_LEN_BYTES + t * ROWS_PER_THREAD);
        ml = (int)max(ll & 0xffu, ll >> 8);
    }
    if (h.mode() == SELL_MODE_DELTA16) {
        // delta mode: 16 bits per (row, slot), group-major 16-byte words of 4 slots x 2 rows; the
        // column of a slot is the running sum of the row's codes (first code relative to
        // row + dict_off).  The eight columns of a batch are formed first, then the eight gathers.
        static_assert(SELL_D16_GROUP * 2 == BATCH, "two code words per batch");
        const uint4 *cw = reinterpret_cast<const uint4 *>(codes + h.code_off) + t;
        int c0 = row + h.dict_off, c1 = row + 1 + h.dict_off;
        for (int s0 = 0; s0 < ww; s0 += BATCH) {
            const int g = s0 / SELL_D16_GROUP;
            uint4 wa, wb;
            wa.x = wa.y = wa.z = wa.w = 0xffffffffu;
            wb = wa;
            // (code words: read once per launch, whole lines per instruction -> streamed like the values)
            typedef unsigned u4v __attribute__((ext_vector_type(4)));
            if (s0 < ml) {
                const u4v tw = STREAM ? __builtin_nontemporal_load(reinterpret_cast<const u4v *>(cw + (long)g * BLOCK))
                                      : *reinterpret_cast<const u4v *>(cw + (long)g * BLOCK);
                wa.x = tw.x;
                wa.y = tw.y;
                wa.z = tw.z;
                wa.w = tw.w;
            }
            if (s0 + SELL_D16_GROUP < ml) {
                const u4v tw = STREAM ? __builtin_nontemporal_load(reinterpret_cast<const u4v *>(cw + (long)(g + 1) * BLOCK))
                                      : *reinterpret_cast<const u4v *>(cw + (long)(g + 1) * BLOCK);
                wb.x = tw.x;
                wb.y = tw.y;
                wb.z = tw.z;
                wb.w = tw.w;
            }
            const unsigned w8[BATCH] = {wa.x, wa.y, wa.z, wa.w, wb.x, wb.y, wb.z, wb.w};
            double2 vv[BATCH];
#pragma unroll
            for (int k = 0; k < BATCH; ++k) {
                vv[k].x = vv[k].y = 0.0;
                if (s0 + k < ml) vv[k] = STREAM ? ld_pair_stream(v + (long)(s0 + k) * CHUNK_ROWS) : *reinterpret_cast<const double2 *>(v + (long)(s0 + k) * CHUNK_ROWS);
            }
            int a0[BATCH], a1[BATCH];
            bool ok0[BATCH], ok1[BATCH];
#pragma unroll
            for (int k = 0; k < BATCH; ++k) {
                const unsigned d0 = w8[k] & 0xffffu, d1 = w8[k] >> 16;
                ok0[k] = (s0 + k < ml) && d0 != 0xffffu;
                ok1[k] = (s0 + k < ml) && d1 != 0xffffu;
                if (ok0[k]) c0 += (int)d0;
                if (ok1[k]) c1 += (int)d1;
                a0[k] = c0;
                a1[k] = c1;
            }
            double x0[BATCH], x1[BATCH];
#pragma unroll
            for (int k = 0; k < BATCH; ++k) {
                x0[k] = ok0[k] ? x[a0[k]] : 0.0;
                x1[k] = ok1[k] ? x[a1[k]] : 0.0;
            }
#pragma unroll
            for (int k = 0; k < BATCH; ++k) {
                if (ok0[k]) {
                    const double p = vv[k].x * x0[k];
                    acc.x = (MODE == SPMV_RESIDUAL) ? acc.x - p : acc.x + p;
                }
                if (ok1[k]) {
                    const double p = vv[k].y * x1[k];
                    acc.y = (MODE == SPMV_RESIDUAL) ? acc.y - p : acc.y + p;
                }
            }
        }
    } else if (h.mode() == SELL_MODE_COL32) {
        // column mode: plain 32-bit columns (-1 = padding), 16-byte words of 2 slots x 2 rows
        static_assert(SELL_C32_GROUP * 4 == BATCH, "four code words per batch");
        const int4 *cw = reinterpret_cast<const int4 *>(codes + h.code_off) + t;
        for (int s0 = 0; s0 < ww; s0 += BATCH) {
            const int g = s0 / SELL_C32_GROUP;
            int a0[BATCH], a1[BATCH];
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                int4 w;
                w.x = w.y = w.z = w.w = -1;
                if (s0 + SELL_C32_GROUP * q < ml) {
                    typedef int i4v __attribute__((ext_vector_type(4)));
                    const i4v tw = STREAM ? __builtin_nontemporal_load(reinterpret_cast<const i4v *>(cw + (long)(g + q) * BLOCK))
                                          : *reinterpret_cast<const i4v *>(cw + (long)(g + q) * BLOCK);
                    w.x = tw.x;
                    w.y = tw.y;
                    w.z = tw.z;
                    w.w = tw.w;
                }
                a0[2 * q] = w.x;
                a1[2 * q] = w.y;
                a0[2 * q + 1] = (s0 + 2 * q + 1 < ml) ? w.z : -1;
                a1[2 * q + 1] = (s0 + 2 * q + 1 < ml) ? w.w : -1;
            }
            double2 vv[BATCH];
#pragma unroll
            for (int k = 0; k < BATCH; ++k) {
                vv[k].x = vv[k].y = 0.0;
                if (s0 + k < ml) vv[k] = STREAM ? ld_pair_stream(v + (long)(s0 + k) * CHUNK_ROWS) : *reinterpret_cast<const double2 *>(v + (long)(s0 + k) * CHUNK_ROWS);
            }
            double x0[BATCH], x1[BATCH];
#pragma unroll
            for (int k = 0; k < BATCH; ++k) {
                x0[k] = a0[k] >= 0 ? x[a0[k]] : 0.0;
                x1[k] = a1[k] >= 0 ? x[a1[k]] : 0.0;
            }
#pragma unroll
            for (int k = 0; k < BATCH; ++k) {
                if (a0[k] >= 0) {
                    const double p = vv[k].x * x0[k];
                    acc.x = (MODE == SPMV_RESIDUAL) ? acc.x - p : acc.x + p;
                }
                if (a1[k] >= 0) {
                    const double p = vv[k].y * x1[k];
                    acc.y = (MODE == SPMV_RESIDUAL) ? acc.y - p : acc.y + p;
                }
            }
        }
    } else if (h.mode() == SELL_MODE_PATTERN) {
        // pattern mode: one byte per row -> `width` offsets of the row in the LDS table
        const unsigned short pp =
            *reinterpret_cast<const unsigned short *>(codes + h.code_off + t * ROWS_PER_THREAD);
        const int p0 = (int)(pp & 0xffu) * width, p1 = (int)(pp >> 8) * width;
        for (int s0 = 0; s0 < ww; s0 += BATCH) {
            double2 vv[BATCH];
            int d0[BATCH], d1[BATCH];
#pragma unroll
            for (int k = 0; k < BATCH; ++k) {
                const int s = min(s0 + k, ww - 1);  // clamp: always a valid plane
                vv[k] = STREAM ? ld_pair_stream(v + (long)s * CHUNK_ROWS) : *reinterpret_cast<const double2 *>(v + (long)s * CHUNK_ROWS);
                d0[k] = (s0 + k < ww) ? stab[p0 + s] : SELL_PAD_OFFSET;
                d1[k] = (s0 + k < ww) ? stab[p1 + s] : SELL_PAD_OFFSET;
            }
            double x0[BATCH], x1[BATCH];
#pragma unroll
            for (int k = 0; k < BATCH; ++k) {
                x0[k] = d0[k] != SELL_PAD_OFFSET ? x[row + d0[k]] : 0.0;
                x1[k] = d1[k] != SELL_PAD_OFFSET ? x[row + 1 + d1[k]] : 0.0;
            }
#pragma unroll
            for (int k = 0; k < BATCH; ++k) {
                if (d0[k] != SELL_PAD_OFFSET) {
                    const double p = vv[k].x * x0[k];
                    acc.x = (MODE == SPMV_RESIDUAL) ? acc.x - p : acc.x + p;
                }
                if (d1[k] != SELL_PAD_OFFSET) {
                    const double p = vv[k].y * x1[k];
                    acc.y = (MODE == SPMV_RESIDUAL) ? acc.y - p : acc.y + p;
                }
            }
        }
    } else {
        // offset mode: one byte per (row, slot) -> entry of the chunk's offset dictionary
        const uint8_t *c = codes + h.code_off + (long)t * h.code_stride();
        for (int s0 = 0; s0 < ww; s0 += BATCH) {
            uint4 cw;
            cw.x = cw.y = cw.z = cw.w = 0xffffffffu;
            if (s0 < ml) cw = *reinterpret_cast<const uint4 *>(c + 2 * s0);
            const unsigned w4[4] = {cw.x, cw.y, cw.z, cw.w};
            double2 vv[BATCH];
#pragma unroll
            for (int k = 0; k < BATCH; ++k) {
                vv[k].x = vv[k].y = 0.0;
                if (s0 + k < ml) vv[k] = STREAM ? ld_pair_stream(v + (long)(s0 + k) * CHUNK_ROWS) : *reinterpret_cast<const double2 *>(v + (long)(s0 + k) * CHUNK_ROWS);
            }
            double x0[BATCH], x1[BATCH];
            bool ok0[BATCH], ok1[BATCH];
#pragma unroll
            for (int k = 0; k < BATCH; ++k) {
                const unsigned pair = (w4[k / 2] >> (16 * (k & 1))) & 0xffffu;
                const unsigned c0 = pair & 0xffu, c1 = pair >> 8;
                // padding slots carry code 255; rows past n_rows only have padding slots
                ok0[k] = (s0 + k < ml) && c0 != 255u;
                ok1[k] = (s0 + k < ml) && c1 != 255u;
                x0[k] = ok0[k] ? x[row + stab[c0]] : 0.0;
                x1[k] = ok1[k] ? x[row + 1 + stab[c1]] : 0.0;
            }
#pragma unroll
            for (int k = 0; k < BATCH; ++k) {
                if (ok0[k]) {
                    const double p = vv[k].x * x0[k];
                    acc.x = (MODE == SPMV_RESIDUAL) ? acc.x - p : acc.x + p;
                }
                if (ok1[k]) {
                    const double p = vv[k].y * x1[k];
                    acc.y = (MODE == SPMV_RESIDUAL) ? acc.y - p : acc.y + p;
                }
            }
        }
    }
    // spill: the tails of this chunk's long rows (beyond the chunk's cap).  The row sums go through LDS
    // to the threads that walk the tails -- one row each, entries in stored order, so every row is still
    // summed left to right -- and back.  Workgroup-uniform branch; chunks without long rows skip it.
    if (sp1 > sp0) {
        double *ys = reinterpret_cast<double *>(stab);  // the table is not needed any more
        __syncthreads();
        ys[ROWS_PER_THREAD * t] = acc.x;
        ys[ROWS_PER_THREAD * t + 1] = acc.y;
        __syncthreads();
        for (int j = sp0 + t; j < sp1; j += BLOCK) {
            if (j != sp0 + t) {  // (more than BLOCK long rows in one chunk: the later ones were not prefetched)
                s_row = spill_rows[j];
                s_k0 = spill_ptrs[j];
                s_k1 = spill_ptrs[j + 1];
            }
            const int li = s_row - chunk * CHUNK_ROWS;
            double a = ys[li];
            constexpr int SB = 4;  // values, columns and x of SB entries in flight; the adds stay in order
            for (int k0 = s_k0; k0 < s_k1; k0 += SB) {
                double sv[SB], sx[SB];
                int sc[SB];
#pragma unroll
                for (int i = 0; i < SB; ++i) {
                    const int k = min(k0 + i, s_k1 - 1);
                    sv[i] = spill_vals[k];
                    sc[i] = spill_cols[k];
                }
#pragma unroll
                for (int i = 0; i < SB; ++i) sx[i] = x[sc[i]];
#pragma unroll
                for (int i = 0; i < SB; ++i)
                    if (k0 + i < s_k1) {
                        const double p = sv[i] * sx[i];
                        a = (MODE == SPMV_RESIDUAL) ? a - p : a + p;
                    }
            }
            ys[li] = a;
        }
        __syncthreads();
        acc.x = ys[ROWS_PER_THREAD * t];
        acc.y = ys[ROWS_PER_THREAD * t + 1];
    }
    if (rmap) {  // the sums go to the rows they belong to (workgroup-uniform branch)
        double *ys = reinterpret_cast<double *>(stab);  // the table is not needed any more
        __syncthreads();
        ys[own & 0xffffu] = acc.x;
        ys[own >> 16] = acc.y;
        __syncthreads();
        acc.x = ys[ROWS_PER_THREAD * t];
        acc.y = ys[ROWS_PER_THREAD * t + 1];
    }
    if (hf.chunk_bptr) halo_fused_add<MODE>(hf, chunk, acc.x, acc.y, reinterpret_cast<double *>(stab));
    st2(y, rp, acc);
    if (NDOT >= 1) {
        const double2 vw = ld2(w, rp);
        double d = 0.0, d2 = 0.0;
        if (rp.n > 0) {
            d += vw.x * acc.x;
            d2 += acc.x * acc.x;
        }
        if (rp.n > 1) {
            d += vw.y * acc.y;
            d2 += acc.y * acc.y;
        }
        const double s = block_sum(d, slot);
        if (threadIdx.x == 0) dot_partials[chunk] = s;
        if (NDOT >= 2) {
            const double s2 = block_sum(d2, slot);
            if (threadIdx.x == 0) dot2_partials[chunk] = s2;
        }
    }
}

__global__ __launch_bounds__(BLOCK) void k_gather_coeffs_masked(long n, const int *__restrict__ map,
                                                                const double *__restrict__ src,
                                                                double *__restrict__ out)
{
    const long i = ((long)blockIdx.x * BLOCK + threadIdx.x) * 2;
    if (i + 1 < n) {
        const int2 m = *reinterpret_cast<const int2 *>(map + i);
        double2 v;
        v.x = m.x >= 0 ? src[m.x] : 0.0;
        v.y = m.y >= 0 ? src[m.y] : 0.0;
        *reinterpret_cast<double2 *>(out + i) = v;
    } else if (i < n) {
        out[i] = map[i] >= 0 ? src[map[i]] : 0.0;
    }
}

// The same for the chunked layout (SellChunk): one workgroup fills all planes of its chunk, so the
// chunk's CSR value range (a few tens of KB) is fetched once into one XCD's L2 instead of once per
// plane (the flat kernel above amplified the reads 7x on the 7-point matrix).
__global__ __launch_bounds__(BLOCK) void k_gather_sell(int n_chunks, const SellChunk *__restrict__ chunks,
                                                       const int *__restrict__ map,
                                                       const double *__restrict__ src,
                                                       double *__restrict__ out)
{
    const int chunk = blockIdx.x;
    if (chunk >= n_chunks) return;
    const SellChunk h = chunks[chunk];
    const int width = h.width();
    for (int s = 0; s < width; ++s) {
        const long i = h.val_off + (long)s * CHUNK_ROWS + threadIdx.x * ROWS_PER_THREAD;
        const int2 m = *reinterpret_cast<const int2 *>(map + i);
        double2 v;
        v.x = m.x >= 0 ? src[m.x] : 0.0;
        v.y = m.y >= 0 ? src[m.y] : 0.0;
        *reinterpret_cast<double2 *>(out + i) = v;
    }
}

}  // namespace

// ------------------------------------------------------------------------------------------
// launchers
// ------------------------------------------------------------------------------------------
void launch_spmv_sell(hipStream_t st, const DevSell &A, int mode, const double *x, const double *b,
                      double *y, const SpmvDots &dots, const DevScalars *gate, const HaloFused &hf)
{
    if (A.n_rows == 0) return;
    const int nc = (int)n_chunks(A.n_rows);
    const int xg = A.xcd_group > 0 ? A.xcd_group : XCD_GROUP;
    const dim3 grid(A.block_order ? A.n_blocks : xcd_grid(nc, xg)), block(BLOCK);
#define OGL_SELL_K(MODE, NDOT, STREAM)                                                                   \
    hipLaunchKernelGGL((k_spmv_sell<MODE, NDOT, STREAM>), grid, block, 0, st, A.n_rows, nc, A.chunks,    \
                       A.dict, A.codes, A.vals, A.spill_chunk_ptr, A.spill_rows, A.spill_ptrs,          \
                       A.spill_cols, A.spill_vals, x, b, y, dots.with, dots.part, dots.part_yy, gate, xg, hf, \
                       A.rmap, A.block_order)
#define OGL_SELL(MODE, NDOT)               \
    do {                                   \
        if (A.stream)                      \
            OGL_SELL_K(MODE, NDOT, true);  \
        else                               \
            OGL_SELL_K(MODE, NDOT, false); \
    } while (0)
    if (mode == SPMV_RESIDUAL) {
        OGL_SELL(SPMV_RESIDUAL, 0);
    } else if (dots.part && dots.part_yy) {
        OGL_SELL(SPMV_PLAIN, 2);
    } else if (dots.part) {
        OGL_SELL(SPMV_PLAIN, 1);
    } else {
        OGL_SELL(SPMV_PLAIN, 0);
    }
#undef OGL_SELL
#undef OGL_SELL_K
}

void launch_spmv_symx(hipStream_t st, const DevSymx &A, int mode, const double *x, const double *b, double *y,
                      const SpmvDots &dots, const DevScalars *gate, const HaloFused &hf)
{
    if (A.n_rows == 0) return;
    const int nc = (int)n_chunks(A.n_rows);
    const dim3 block(BLOCK);
    // two launches at most: the chunks without explicit entries or with simple ones (lean kernel), then the others
#define OGL_SYMX_K(MODE, NDOT, STREAM, FAST, GENERAL, HDR, NB)                                                       \
    hipLaunchKernelGGL((k_spmv_symx<MODE, NDOT, STREAM, FAST, GENERAL>), dim3(NB), block, 0, st, A.n_rows, nc, HDR,    \
                       A.mask, A.planes, A.ex_rowptr, A.ex_cols, A.ex_vals, A.ex_lrow, x, b, y, dots.with, dots.part, \
                       dots.part_yy, gate, hf)
#define OGL_SYMX_G(MODE, NDOT, GENERAL, HDR, NB)               \
    do {                                                       \
        if (A.stream && A.fast)                                \
            OGL_SYMX_K(MODE, NDOT, true, true, GENERAL, HDR, NB);   \
        else if (A.stream)                                     \
            OGL_SYMX_K(MODE, NDOT, true, false, GENERAL, HDR, NB);  \
        else if (A.fast)                                       \
            OGL_SYMX_K(MODE, NDOT, false, true, GENERAL, HDR, NB);  \
        else                                                   \
            OGL_SYMX_K(MODE, NDOT, false, false, GENERAL, HDR, NB); \
    } while (0)
#define OGL_SYMX(MODE, NDOT)                                                             \
    do {                                                                                 \
        if (A.n_blocks > 0) OGL_SYMX_G(MODE, NDOT, false, A.chunks, A.n_blocks);         \
        if (A.n_blocks_general > 0) OGL_SYMX_G(MODE, NDOT, true, A.chunks_general, A.n_blocks_general); \
    } while (0)
    if (mode == SPMV_RESIDUAL) {
        OGL_SYMX(SPMV_RESIDUAL, 0);
    } else if (dots.part && dots.part_yy) {
        OGL_SYMX(SPMV_PLAIN, 2);
    } else if (dots.part) {
        OGL_SYMX(SPMV_PLAIN, 1);
    } else {
        OGL_SYMX(SPMV_PLAIN, 0);
    }
#undef OGL_SYMX
#undef OGL_SYMX_G
#undef OGL_SYMX_K
}

void launch_gather_coeffs_masked(hipStream_t st, int64_t n, const int32_t *map, const double *source,
                                 double *out)
{
    if (n == 0) return;
    hipLaunchKernelGGL(k_gather_coeffs_masked, dim3(blocks_for((n + 1) / 2)), dim3(BLOCK), 0, st,
                       (long)n, map, source, out);
}

void launch_gather_sell(hipStream_t st, int32_t n_chunks_, const SellChunk *chunks, const int32_t *map,
                        const double *source, double *out)
{
    if (n_chunks_ == 0) return;
    hipLaunchKernelGGL(k_gather_sell, dim3(n_chunks_), dim3(BLOCK), 0, st, n_chunks_, chunks, map,
                       source, out);
}

}  // namespace ogl
