// solver_precond.cpp -- Preconditioner::init_preconditioner with its caching rules, generation and application of
// block Jacobi and ISAI / GISAI (Preconditioner.H:47-105,225-258,353-431).  See solver.hpp, solver_internal.hpp.
#include "solver_internal.hpp"

#include <algorithm>
#include <cmath>
#include <cstring>
#include <vector>

using namespace ogl;

// ------------------------------------------------------------------------------------------
// Preconditioner::init_preconditioner (Preconditioner.H:353-431) with its caching rules:
//   * nothing stored yet      -> generate, store, counter := caching
//   * stored and counter > 0  -> counter -= 1, use the STORED one
//   * stored and counter == 0 -> counter := caching, generate a fresh one for this solve only;
//                                the stored object is not replaced (:411-413)
// The store is registry-wide (one key for all fields, :357), as in the reference.
// ------------------------------------------------------------------------------------------
int ogl_solver::generate_preconditioner(PrecondData &P)
{
    hipStream_t st = reg->stream;
    const size_t n = (size_t)pat.n_rows;
    // structures of a renumbered device copy (block-Jacobi blocks, ISAI(spd)'s triangle) in the CALLER's numbering:
    // the reference's operator (property precondCallerNumbering 0 = the backend's numbering, for A/B)
    const bool caller_numbering = prop("precondCallerNumbering", 1.0) != 0.0;
    const bool through_perm = pat.renumbered() && caller_numbering;
    if (P.struct_caller_numbering != caller_numbering) P.struct_pat_id = 0;  // (the switch was flipped: rebuild)
    P.struct_caller_numbering = caller_numbering;
    if (cfg.preconditioner == OGL_PRECOND_ISAI || cfg.preconditioner == OGL_PRECOND_GISAI) {
        // Isai<spd|general> with sparsity_power 1 and skip_sorting (Preconditioner.H:225-258).
        // Pattern of W on the host (tril(A) for spd, A for general), its transpose + map for spd,
        // values on the device (one dense solve per row).
        const bool spd = cfg.preconditioner == OGL_PRECOND_ISAI;
        const int32_t N = pat.n_rows;
        const int kind = spd ? 3 : 4;
        if (!P.has_structure(pat_id, kind, cfg.sparsity_power)) {
            P.struct_pat_id = 0;
            std::vector<int32_t> wrp, wc;
            ogl_label wide = -1;
            OGL_TRY(download_local_pattern(pat));
            if (!isai_pattern(pat, spd, cfg.sparsity_power, MAX_ISAI_HUGE_ROW, wrp, wc, wide, caller_numbering))
                return fail(OGL_ERR_UNSUPPORTED,
                            "preconditioner %s, sparsityPower %d: row %d of the approximate inverse has more than %d "
                            "pattern entries (lower sparsityPower)", spd ? "ISAI" : "GISAI", cfg.sparsity_power, wide,
                            MAX_ISAI_HUGE_ROW);
            int32_t max_row = 0;
            // rows solved by one wavefront each (33 .. 64 entries) / by one workgroup each in global scratch
            // (65 .. 2048 = MAX_ISAI_HUGE_ROW); the others: one thread
            std::vector<int32_t> wide_rows, huge_rows;
            std::vector<int64_t> huge_off;
            P.huge_batches.assign(1, 0);
            // scratch for the dense systems of the huge rows: batches of rows within a budget (property, bytes)
            const int64_t budget = (int64_t)(prop("isaiScratchBytes", 2147483648.0) / sizeof(double));
            int64_t used = 0, most = 0;
            for (int32_t r = 0; r < N; ++r) {
                const int32_t len = wrp[(size_t)r + 1] - wrp[(size_t)r];
                max_row = std::max(max_row, len);
                if (len > MAX_ISAI_ROW) {
                    const int64_t need = (int64_t)len * len;
                    if (used + need > budget && used > 0) {
                        P.huge_batches.push_back((int32_t)huge_rows.size());
                        used = 0;
                    }
                    huge_rows.push_back(r);
                    huge_off.push_back(used);
                    used += need;
                    most = std::max(most, used);
                } else if (len > ISAI_THREAD_ROW) {
                    wide_rows.push_back(r);
                }
            }
            P.huge_batches.push_back((int32_t)huge_rows.size());
            P.n_wide_rows = (int32_t)wide_rows.size();
            P.n_huge_rows = (int32_t)huge_rows.size();
            OGL_TRY(P.wide_rows.alloc(std::max<size_t>(1, wide_rows.size()), st));
            if (!wide_rows.empty())
                OGL_TRY(reg->stager.h2d(P.wide_rows.p, wide_rows.data(), wide_rows.size() * sizeof(int32_t), st));
            OGL_TRY(P.huge_rows.alloc(std::max<size_t>(1, huge_rows.size()), st));
            OGL_TRY(P.huge_off.alloc(std::max<size_t>(1, huge_off.size()), st));
            if (!huge_rows.empty()) {
                OGL_TRY(reg->stager.h2d(P.huge_rows.p, huge_rows.data(), huge_rows.size() * sizeof(int32_t), st));
                OGL_TRY(reg->stager.h2d(P.huge_off.p, huge_off.data(), huge_off.size() * sizeof(int64_t), st));
            }
            P.huge_scratch_len = most;
            P.huge_scratch.release();
            props["isaiWideRows"] = (double)wide_rows.size();
            props["isaiHugeRows"] = (double)huge_rows.size();
            const size_t wn = wc.size();
            // (property isaiSortRows 0: W / W^T on the compressed layout only where their own row order qualifies, A/B)
            const bool sort_w = prop("isaiSortRows", 1.0) != 0.0;
            OGL_TRY(P.w_row_ptrs.alloc((size_t)N + 1, st));
            OGL_TRY(P.w_cols.alloc(wn + NNZ_PAD, st));
            OGL_TRY(P.w_vals.alloc(wn + NNZ_PAD, st));
            OGL_TRY(reg->stager.h2d(P.w_row_ptrs.p, wrp.data(), wrp.size() * sizeof(int32_t), st));
            OGL_TRY(reg->stager.h2d(P.w_cols.p, wc.data(), wn * sizeof(int32_t), st));
            if (spd) {  // W^T: counting transpose keeps every row sorted by column
                std::vector<int32_t> trp((size_t)N + 1, 0), tc(wn), tmap(wn);
                for (size_t k = 0; k < wn; ++k) ++trp[wc[k] + 1];
                for (int32_t r = 0; r < N; ++r) trp[r + 1] += trp[r];
                std::vector<int32_t> fill(trp.begin(), trp.end() - 1);
                for (int32_t r = 0; r < N; ++r)
                    for (int32_t k = wrp[r]; k < wrp[r + 1]; ++k) {
                        const int32_t e = fill[wc[k]]++;
                        tc[e] = r;
                        tmap[e] = k;
                    }
                OGL_TRY(P.wt_row_ptrs.alloc((size_t)N + 1, st));
                OGL_TRY(P.wt_cols.alloc(wn + NNZ_PAD, st));
                OGL_TRY(P.wt_map.alloc(wn + NNZ_PAD, st));
                OGL_TRY(P.wt_vals.alloc(wn + NNZ_PAD, st));
                OGL_TRY(reg->stager.h2d(P.wt_row_ptrs.p, trp.data(), trp.size() * sizeof(int32_t), st));
                OGL_TRY(reg->stager.h2d(P.wt_cols.p, tc.data(), wn * sizeof(int32_t), st));
                OGL_TRY(reg->stager.h2d(P.wt_map.p, tmap.data(), wn * sizeof(int32_t), st));
                if (cfg.compress_indices) {
                    OGL_TRY(P.wt_sell.build(N, trp.data(), tc.data(), reg->stager, st));
                    if (!P.wt_sell.ready && sort_w) OGL_TRY(P.wt_sell.build(N, trp.data(), tc.data(), reg->stager, st, true));
                }
            }
            if (!spd || !cfg.compress_indices) P.wt_sell.ready = false;
            P.w_sell.ready = false;
            if (cfg.compress_indices) {
                OGL_TRY(P.w_sell.build(N, wrp.data(), wc.data(), reg->stager, st));
                if (!P.w_sell.ready && sort_w) OGL_TRY(P.w_sell.build(N, wrp.data(), wc.data(), reg->stager, st, true));
            }
            props["isaiWSorted"] = P.w_sell.sorted ? 1.0 : 0.0;
            props["isaiWtSorted"] = P.wt_sell.sorted ? 1.0 : 0.0;
            P.w_nnz = (int32_t)wn;
            P.w_max_row = max_row;
            props["isaiWCompressed"] = P.w_sell.ready ? 1.0 : 0.0;
            props["isaiWtCompressed"] = P.wt_sell.ready ? 1.0 : 0.0;
            P.struct_pat_id = pat_id;
            P.struct_kind = kind;
            P.struct_stride = cfg.sparsity_power;
        }
        launch_isai_generate(st, csr(), spd ? 1 : 0, P.w_row_ptrs.p, P.w_cols.p, P.w_vals.p,
                             P.w_max_row, P.wide_rows.p, P.n_wide_rows, prop("isaiGroupLanes", 1.0) != 0.0);
        // the dense systems of the huge rows live in a scratch of up to isaiScratchBytes that only this generation
        // needs: allocated here, released below (a field's own preconditioner plus the registry-wide cached one would
        // otherwise sit on 2 GiB each for the whole run)
        if (P.n_huge_rows > 0) OGL_TRY(P.huge_scratch.alloc((size_t)P.huge_scratch_len, st));
        for (size_t bt = 0; bt + 1 < P.huge_batches.size(); ++bt)  // (stream order: a batch reuses the scratch)
            launch_isai_generate_huge(st, csr(), spd ? 1 : 0, P.w_row_ptrs.p, P.w_cols.p, P.w_vals.p, P.huge_rows.p,
                                      P.huge_off.p, P.huge_batches[bt], P.huge_batches[bt + 1] - P.huge_batches[bt],
                                      P.huge_scratch.p);
        if (P.n_huge_rows > 0 && prop("isaiKeepScratch", 0.0) == 0.0) {
            OGL_HIP_CHECK(hipStreamSynchronize(st));
            P.huge_scratch.release();
        }
        if (spd) launch_gather_coeffs(st, P.w_nnz, P.wt_map.p, P.w_vals.p, P.wt_vals.p);
        P.w_sell.refresh(P.w_vals.p, st);
        if (spd) P.wt_sell.refresh(P.wt_vals.p, st);
        P.kind = spd ? 3 : 4;
        P.stride = cfg.sparsity_power;
    } else if (cfg.max_block_size == 1) {  // scalar Jacobi: 1 / diag
        OGL_TRY(P.values.alloc(n + 2, st));
        // the diagonal sits contiguous in the staged source of the last coefficient update when the device rows are the
        // caller's cells and no same-rank interface adds entries: 1/d from there (216^3: 124 -> ~25 us per generation,
        // 562 MB of CSR values not touched); otherwise from the CSR values through the diagonal positions -- same bits
        const bool from_source = source_diag_valid && d_new_id.n == 0 && pat.local_iface_nnz == 0 &&
                                 d_source.n >= (size_t)pat.diag_start() + (size_t)n && prop("jacobiFromSource", 1.0) != 0.0;
        props["jacobiFromSourceInUse"] = from_source ? 1.0 : 0.0;
        if (from_source)
            launch_jacobi_generate_diag(st, (int32_t)n, d_source.p + pat.diag_start(), P.values.p);
        else
            launch_jacobi_generate_pos(st, csr(), d_diag_pos.p, P.values.p);
        P.kind = 1;
        P.stride = 0;
    } else {
        // Jacobi factory with max_block_size = maxBlockSize, skip_sorting (Preconditioner.H:100-104)
        const size_t k = (size_t)cfg.max_block_size;
        if (!P.has_structure(pat_id, 2, cfg.max_block_size)) {
            P.struct_pat_id = 0;
            std::vector<int32_t> ptrs, row_block;
            OGL_TRY(download_local_pattern(pat));
            find_jacobi_blocks(pat, cfg.max_block_size, ptrs, row_block, caller_numbering);
            P.n_blocks = (int32_t)ptrs.size() - 1;
            P.uniform_blocks = true;
            for (int32_t b = 0; b + 1 < P.n_blocks; ++b)
                P.uniform_blocks = P.uniform_blocks && ptrs[(size_t)b + 1] - ptrs[(size_t)b] == cfg.max_block_size;
            if (P.n_blocks > 0)
                P.uniform_blocks = P.uniform_blocks && ptrs[(size_t)P.n_blocks - 1] == (P.n_blocks - 1) * cfg.max_block_size;
            OGL_TRY(P.block_ptrs.alloc(ptrs.size(), st));
            OGL_TRY(P.row_block.alloc(std::max<size_t>(1, row_block.size()), st));
            OGL_TRY(P.values.alloc(std::max<size_t>(1, (size_t)P.n_blocks * k * k), st));
            OGL_TRY(reg->stager.h2d(P.block_ptrs.p, ptrs.data(), ptrs.size() * sizeof(int32_t), st));
            OGL_TRY(reg->stager.h2d(P.row_block.p, row_block.data(),
                                    row_block.size() * sizeof(int32_t), st));
            P.struct_pat_id = pat_id;
            P.struct_kind = 2;
            P.struct_stride = cfg.max_block_size;
        }
        DevBlockJacobi J;
        J.n_rows = pat.n_rows;
        J.n_blocks = P.n_blocks;
        J.stride = cfg.max_block_size;
        J.block_ptrs = P.block_ptrs.p;
        J.row_block = P.row_block.p;
        J.blocks = P.values.p;
        if (through_perm) {  // blocks of the caller's numbering, reached through the permutation
            J.rows = d_new_id.p;
            J.pos = d_old_of.p;
            // staged apply (default): blocks stay block-major in the caller's order, the vectors are carried there and
            // back by two gather kernels -- one scattered access per row instead of one per block member (128^3
            // shuffled, BJ(4): 316 us per turn with the direct apply on block rows stored by device row)
            J.by_device_row = prop("bjStagedApply", 1.0) != 0.0 ? 0 : 1;
        }
        P.by_device_row = J.by_device_row != 0;
        P.through_perm = through_perm;
        P.perm_pat_id = through_perm ? pat_id : 0;

        launch_bj_generate(st, csr(), J, prop("bjGroupLanes", 1.0) != 0.0);
        P.kind = 2;
        P.stride = cfg.max_block_size;
    }
    P.n_rows = n;
    P.gen_pat_id = pat_id;
    P.gen_device_numbering =
        pat.renumbered() && !(P.kind == 2 && P.through_perm && !P.by_device_row);  // (see PrecondData::foreign_to)
    return OGL_OK;
}

void ogl_solver::apply_preconditioner(const double *in, double *out, const DevScalars *gate,
                                      double *dot_part)
{
    hipStream_t st = reg->stream;
    // the last kernel of the apply also leaves the partials of in . out
    SpmvDots last{};
    if (dot_part) {
        last.with = in;
        last.part = dot_part;
    }
    if (precond_data->kind == 3 || precond_data->kind == 4) {  // ISAI: one or two SpMVs
        DevCsr W;
        W.n_rows = pat.n_rows;
        W.nnz = precond_data->w_nnz;
        W.row_ptrs = precond_data->w_row_ptrs.p;
        W.cols = precond_data->w_cols.p;
        W.vals = precond_data->w_vals.p;
        const bool w_sell = cfg.compress_indices && precond_data->w_sell.ready;
        const bool general = precond_data->kind == 4;
        double *w_out = general ? out : d_isai_tmp.p;
        // (W / W^T are streamed past the caches exactly when the system matrix is: one working set, one policy)
        const bool stream_w = props.count("spmvStream") && props.at("spmvStream") == 1.0;
        W.stream = stream_w;
        if (w_sell)
            launch_spmv_sell(st, precond_data->w_sell.view(pat.n_rows, stream_w), SPMV_PLAIN, in, nullptr, w_out,
                             general ? last : SpmvDots{}, gate);
        else
            launch_spmv(st, W, SPMV_PLAIN, in, nullptr, w_out, general ? last : SpmvDots{}, gate);
        if (general) return;
        DevCsr WT = W;
        WT.row_ptrs = precond_data->wt_row_ptrs.p;
        WT.cols = precond_data->wt_cols.p;
        WT.vals = precond_data->wt_vals.p;
        if (cfg.compress_indices && precond_data->wt_sell.ready)
            launch_spmv_sell(st, precond_data->wt_sell.view(pat.n_rows, stream_w), SPMV_PLAIN, d_isai_tmp.p,
                             nullptr, out, last, gate);
        else
            launch_spmv(st, WT, SPMV_PLAIN, d_isai_tmp.p, nullptr, out, last, gate);
        return;
    }
    DevBlockJacobi J;
    J.n_rows = pat.n_rows;
    J.n_blocks = precond_data->n_blocks;
    J.stride = precond_data->stride;
    J.block_ptrs = precond_data->block_ptrs.p;
    J.row_block = precond_data->row_block.p;
    J.blocks = precond_data->values.p;
    J.uniform = precond_data->uniform_blocks ? 1 : 0;
    // (blocks kept in the caller's order -- also a stored object that a field WITHOUT a numbering of its own generated --
    //  are reached through this solver's permutation)
    if (pat.renumbered() && (precond_data->through_perm || precond_data->caller_order_blocks())) {
        J.rows = d_new_id.p;
        J.pos = d_old_of.p;
        J.by_device_row = precond_data->by_device_row ? 1 : 0;
        J.perm_xcd_group = (int32_t)prop("bjPermXcdGroup", 0.0);
        if (!precond_data->by_device_row) {
            // (property bjFusedPerm 0: the three-launch form with two staging vectors, for A/B)
            const bool fused_perm = prop("bjFusedPerm", 1.0) != 0.0 && in != out;
            launch_bj_apply_staged(st, J, in, out, dot_part, gate, fused_perm ? nullptr : d_bj_tmp0.p,
                                   fused_perm ? nullptr : d_bj_tmp1.p);
            return;
        }
    }
    launch_bj_apply(st, J, in, out, dot_part, gate);
}

int ogl_solver::init_preconditioner()
{
    precond = nullptr;
    precond_data = nullptr;
    if (cfg.preconditioner == OGL_PRECOND_NONE) return OGL_OK;  // :342
    const bool isai = cfg.preconditioner == OGL_PRECOND_ISAI || cfg.preconditioner == OGL_PRECOND_GISAI;
    if (cfg.preconditioner != OGL_PRECOND_BJ && !isai)
        return fail(OGL_ERR_UNSUPPORTED, "preconditioner kind %d is not built", cfg.preconditioner);
    if (isai && (cfg.sparsity_power < 1 || cfg.sparsity_power > 8))
        return fail(OGL_ERR_INVALID, "ISAI sparsityPower %d outside [1, 8]", cfg.sparsity_power);
    if (!isai && (cfg.max_block_size < 1 || cfg.max_block_size > MAX_JACOBI_BLOCK))
        return fail(OGL_ERR_INVALID, "BJ maxBlockSize %d outside [1, %d]", cfg.max_block_size,
                    MAX_JACOBI_BLOCK);
    if (cfg.preconditioner == OGL_PRECOND_ISAI)
        OGL_TRY(d_isai_tmp.alloc((size_t)pat.n_rows + 2, reg->stream));
    const int kind = isai ? (cfg.preconditioner == OGL_PRECOND_ISAI ? 3 : 4)
                          : (cfg.max_block_size == 1 ? 1 : 2);
    const int stride = kind == 2 ? cfg.max_block_size : (isai ? cfg.sparsity_power : 0);
    const int cache = (int)prop("preconditionerCaching", 0);
    const bool stored =
        reg->has_cached_precond && reg->cached_precond.matches(kind, (size_t)pat.n_rows, stride);
    // (the store is shared by all fields, Preconditioner.H:357: a stored object whose values live in ANOTHER pattern's
    //  device numbering -- inverse diagonal, W / W^T, block rows stored by device row, the backend's own blocks -- would be
    //  a silently permuted operator here: generate for this solve instead.  Blocks kept block-major in the caller's
    //  order are applied through this solver's permutation.)
    const bool foreign = stored && reg->cached_precond.foreign_to(pat_id, pat.renumbered());
    if (stored && cache > 0 && !foreign) {
        props["preconditionerCaching"] = cache - 1;
        precond_data = &reg->cached_precond;
    } else {
        props["preconditionerCaching"] = cfg.caching;
        PrecondData &P = stored ? own_precond : reg->cached_precond;
        OGL_TRY(generate_preconditioner(P));
        if (!stored) reg->has_cached_precond = true;
        precond_data = &P;
    }
    if (precond_data->kind == 1) precond = precond_data->values.p;
    if (pat.renumbered() && precond_data->kind == 2 && !precond_data->by_device_row &&
        (precond_data->through_perm || precond_data->caller_order_blocks())) {
        // (the staged apply's two vectors in the caller's order; also for a stored object another field generated)
        OGL_TRY(d_bj_tmp0.alloc((size_t)pat.n_rows + 2, reg->stream));
        OGL_TRY(d_bj_tmp1.alloc((size_t)pat.n_rows + 2, reg->stream));
    }
    return OGL_OK;
}
