"""One rank of a multi-process run of the sharded path (launched by test_distributed.py through
torch.distributed.run, gloo rendezvous on 127.0.0.1).

modes
  oracle    CPU only: the oracle's distributed CG (halo exchange + all-reduce through gloo)
            against the single-rank oracle on the assembled global system, plus the product's
            host-side pattern (ogl_host_pattern) against the oracle's per rank.
  gpu-host  every rank drives libogl_amd on the (shared) GPU with the host-buffer transport
            (forceHostBuffer: callbacks -> gloo) and must reproduce the distributed oracle run in
            the device's reduction order bit for bit.
  gpu-rccl  same through RCCL (needs one GPU per rank).
"""
import argparse
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

from ogl_amd import synthetic  # noqa: E402
from oracle import oracle as orc  # noqa: E402
from helpers import blocked, orc_ifaces, oracle_csr  # noqa: E402


def make_exchange(neigh_of_patch):
    """Blocked neighbour exchange over gloo: (neighbours, counts, send) -> recv."""
    def exchange(neighbours, counts, send):
        recv = np.zeros_like(send)
        offs = np.concatenate([[0], np.cumsum(counts)]).astype(int)
        reqs, bufs = [], []
        for i, nb in enumerate(neighbours):
            s = torch.from_numpy(np.ascontiguousarray(send[offs[i]:offs[i + 1]]))
            r = torch.zeros(int(counts[i]), dtype=torch.float64)
            reqs.append(dist.isend(s, int(nb)))
            reqs.append(dist.irecv(r, int(nb)))
            bufs.append((i, r, s))
        for q in reqs:
            q.wait()
        for i, r, _ in bufs:
            recv[offs[i]:offs[i + 1]] = r.numpy()
        return recv
    return exchange


def allreduce(a):
    t = torch.from_numpy(np.ascontiguousarray(a, dtype=np.float64).copy())
    dist.all_reduce(t, op=dist.ReduceOp.SUM)
    return t.numpy()


def allreduce_rank_order(a):
    """((0 + v_0) + v_1) + ... : the order of the peer-write all-reduce (device_common.hpp peer_allreduce2)."""
    mine = torch.from_numpy(np.ascontiguousarray(a, dtype=np.float64).copy())
    parts = [torch.zeros_like(mine) for _ in range(dist.get_world_size())]
    dist.all_gather(parts, mine)
    out = np.zeros_like(mine.numpy())
    for p in parts:
        out = out + p.numpy()
    return out


def oracle_dist_matrix(case, allreduce=allreduce):
    ifs = orc_ifaces(orc, case)
    rp, cols, vals = oracle_csr(orc, case)
    nl_rows, nl_cols, nl_perm = orc.init_non_local_sparsity(ifs)
    nl_vals = orc.update_non_local_matrix_data(ifs, nl_perm)
    ids, sizes, send_idxs = orc.create_communication_pattern(ifs)
    nl_rp = orc.rowptr_from_rows(case.n_cells, nl_rows)
    ex = make_exchange(None)
    A = orc.DistMatrix(rp, cols, vals, nl_rp, nl_cols, nl_vals, send_idxs, n_halo=nl_rows.size,
                       exchange=lambda s: ex(ids, sizes, s), allreduce=allreduce,
                       global_n=case.global_n)
    return A, (rp, cols, vals), (nl_rows, nl_cols, nl_perm, nl_vals), (ids, sizes, send_idxs)


def gather_global(case, x):
    """Assemble the global vector on every rank (test plumbing)."""
    out = np.zeros(case.global_n)
    out[case.global_index] = x
    return allreduce(out)


def run_case(args, rank, world, case, glob, xs_g, b_g, skw, state, round_no):
    # (the stand-in for librccl adds the ranks' contributions in rank order, like the peer mailboxes: tests/cpp/rccl_standin.cpp)
    standin = args.mode == "gpu-rccl" and bool(os.environ.get("OGL_RCCL_LIBRARY"))
    A, (rp, cols, vals), nl, comm = oracle_dist_matrix(
        case, allreduce_rank_order if (args.mode == "gpu-peer" or standin) else allreduce)
    if glob is None:
        # --no-global (full-size decomposed configs): nothing of the assembled system is built -- b = A x* through the
        # oracle's own distributed product (halo exchange over gloo), every check is rank-local or collective
        b = A.apply(synthetic.x_star(case.global_index, case.global_n))
    else:
        b = b_g[case.global_index]
    inv = orc.jacobi_generate_scalar(rp, cols, vals) if args.precond else None

    # ---- product host logic per rank vs the oracle (pure host, no GPU) ----
    from ogl_amd import capi
    d, loc, pnl, pcomm = capi.host_pattern(case)
    assert d.n_neighbours == comm[0].size and d.n_send == comm[2].size
    for a_, b_ in zip(pnl, nl[:3]):
        np.testing.assert_array_equal(a_, b_)
    for a_, b_ in zip(pcomm, comm):
        np.testing.assert_array_equal(a_, b_)

    # ---- distributed SpMV vs the global operator ----
    rng = np.random.default_rng(20241016)
    if glob is None:
        x_loc = np.random.default_rng(20241016 + rank).uniform(-1, 1, case.n_cells)
        y = A.apply(x_loc)
    else:
        xg = rng.uniform(-1, 1, glob.n_cells)
        x_loc = xg[case.global_index]
        y = A.apply(x_loc)
        np.testing.assert_allclose(gather_global(case, y), synthetic.apply_case(glob, xg), rtol=1e-13,
                                   atol=1e-13)

    solve = orc.bicgstab if args.asym else orc.cg
    if args.gmres:
        P = orc.Precond(rp, cols, vals, 1) if args.precond else None
        inv = P
        solve = lambda A_, b_, x_, P_, **kw: orc.gmres(A_, b_, x_, P_, krylov_dim=args.gmres, **kw)
    if args.mode == "oracle":
        res = solve(A, b, np.zeros_like(b), inv, **skw)
        g_rp, g_cols, g_vals = oracle_csr(orc, glob)
        G = orc.DistMatrix(g_rp, g_cols, g_vals)
        g_inv = orc.jacobi_generate_scalar(g_rp, g_cols, g_vals) if args.precond else None
        if args.gmres and args.precond:
            g_inv = orc.Precond(g_rp, g_cols, g_vals, 1)
        ref = solve(G, b_g, np.zeros_like(b_g), g_inv, **skw)
        xg_sol = gather_global(case, res.x)
        np.testing.assert_allclose(xg_sol, ref.x, atol=1e-9, rtol=0)
        np.testing.assert_allclose(xg_sol, xs_g, atol=1e-8, rtol=0)
        assert abs(res.n_iterations - ref.n_iterations) <= 2, (res.n_iterations, ref.n_iterations)
        m = min(20, res.history.size, ref.history.size)
        np.testing.assert_allclose(res.history[:m], ref.history[:m], rtol=1e-10)
        assert res.norm_factor == ref.norm_factor or abs(res.norm_factor / ref.norm_factor - 1) < 1e-13
        # every rank saw the same (all-reduced) history
        h = allreduce(res.history / world)
        np.testing.assert_allclose(h, res.history, rtol=1e-15)
    else:
        n_dev = torch.cuda.device_count()
        dev = rank % max(1, n_dev)
        reg = state.get("reg") or capi.Registry(device_id=dev)
        if "reg" in state:
            pass                                   # second round: same registry, same transport, same field
        elif args.mode in ("gpu-host", "gpu-peer"):
            ex = make_exchange(None)
            reg.set_host_comm(rank, world, allreduce, lambda nb, ct, s: ex(nb, ct, s))
            if args.mode == "gpu-peer":
                # scalar all-reduces through the peer mailboxes (hipIpc), inside the finaliser
                # kernels; the halo exchange stays on the host-buffer transport
                handles = [None] * world
                dist.all_gather_object(handles, reg.peer_handle())
                reg.peer_connect(rank, world, handles)
        else:
            uid = [capi.rccl_unique_id() if rank == 0 else None]
            dist.broadcast_object_list(uid, src=0)
            reg.init_rccl(rank, world, uid[0])
            info = reg.comm_info()
            assert info.transport == 2 and info.ranks_seen == world and info.n_ranks == world, (
                info.transport, info.ranks_seen, info.n_ranks)
        cfg = capi.default_config(
            solver=capi.SOLVER_GMRES if args.gmres else
            (capi.SOLVER_BICGSTAB if args.asym else capi.SOLVER_CG), krylov_dim=args.gmres,
            preconditioner=capi.PRECOND_BJ if args.precond else capi.PRECOND_NONE,
            tolerance=1e-11, rel_tol=0.0, max_iter=args.max_iter, export_res=1, adapt_min_iter=0,
            matrix_format=capi.FORMAT_CSR, force_host_buffer=int(args.mode != "gpu-rccl"),
            renumber=capi.RENUMBER_ON if args.renumber else capi.RENUMBER_OFF)
        state["reg"] = reg
        s = reg.solver("p", cfg)
        s.set_property("haloFused", float(args.halo_fused))
        s.set_property("fusedTurnMulti", float(args.fused_turn_multi))
        if args.peer_safe_wait >= 0:        # (-1: the library decides -- it finds ranks that share a device by itself)
            s.set_property("peerSafeWait", float(args.peer_safe_wait))
        s.set_matrix(case)
        new_id = s.renumbering()
        assert (new_id is not None) == bool(args.renumber and case.n_cells >= 2)
        if new_id is None:
            new_id = np.arange(case.n_cells)
        else:
            # the oracle gets this rank's system in the numbering the library chose (explicit input):
            # local block P A P^T, halo rows renamed, the send list in the same order with new names
            p_rp, p_cols, p_vals, _ = orc.permute_csr(rp, cols, vals, new_id)
            n_rows_, n_cols_, n_vals_, n_ord = orc.permute_non_local(nl[0], nl[1], nl[3], new_id)
            ex2 = make_exchange(None)
            ar = allreduce_rank_order if (args.mode == "gpu-peer" or standin) else allreduce
            A = orc.DistMatrix(p_rp, p_cols, p_vals, orc.rowptr_from_rows(case.n_cells, n_rows_), n_cols_,
                               n_vals_, new_id[comm[2]].astype(np.int32), n_halo=n_rows_.size,
                               exchange=lambda s_: ex2(comm[0], comm[1], s_), allreduce=ar,
                               global_n=case.global_n)
            nl = (n_rows_, n_cols_, nl[2][n_ord], n_vals_)
            if args.precond:
                inv = orc.jacobi_generate_scalar(p_rp, p_cols, p_vals)
                if args.gmres:
                    inv = orc.Precond(p_rp, p_cols, p_vals, 1)
        # device halo matrix == oracle's
        r_, c_, m_, v_ = s.non_local_matrix()
        np.testing.assert_array_equal(r_, nl[0])
        np.testing.assert_array_equal(c_, nl[1])
        np.testing.assert_array_equal(v_, nl[3])
        if os.environ.get("OGL_DIST_DEBUG"):
            import time
            t0 = time.time()
            yd = s.spmv(x_loc)
            print(f"rank {rank}: first spmv took {time.time() - t0:.2f} s", flush=True)
            bad = np.flatnonzero(yd != y)
            y_local = orc.spmv(rp, cols, vals, x_loc)          # without the non-local part
            nlr = np.asarray(nl[0])
            print(f"rank {rank}: {bad.size} rows differ; of them {int(np.isin(bad, nlr).sum())} are boundary rows "
                  f"(boundary rows: {np.unique(nlr).size}); equal to the LOCAL product on {int((yd[bad] == y_local[bad]).sum())}; "
                  f"first bad rows {bad[:6].tolist()} last {bad[-3:].tolist()}", flush=True)
            # which neighbour's block do the bad rows read?
            ids, sizes = np.asarray(comm[0]), np.asarray(comm[1])
            offs = np.concatenate([[0], np.cumsum(sizes)])
            nlc = np.asarray(nl[1])
            for i, nb in enumerate(ids):
                rows_i = nlr[(nlc >= offs[i]) & (nlc < offs[i + 1])]
                print(f"rank {rank}: neighbour {nb} block [{offs[i]}, {offs[i+1]}): {int(np.isin(rows_i, bad).sum())} of "
                      f"{rows_i.size} rows bad", flush=True)
            t0 = time.time()
            yd2 = s.spmv(x_loc)
            print(f"rank {rank}: second spmv: {int((yd2 != y).sum())} rows differ, took {time.time() - t0:.2f} s", flush=True)
        if args.renumber:
            np.testing.assert_allclose(s.spmv(x_loc), y, rtol=1e-13, atol=1e-13)
        else:
            np.testing.assert_array_equal(s.spmv(x_loc), y)
        x, perf = s.solve(b, np.zeros_like(b))
        hist = s.history()
        if args.expect_merged >= 0:
            # the multi-rank merged turn (step_1x inside the SpMV kernel, z put by step_2r: 4 launches) ran / did not
            assert s.get_property("fusedTurnInUse") == float(args.expect_merged), s.get_property("fusedTurnInUse")
        b_o = np.empty_like(b)
        b_o[new_id] = b
        with blocked(orc, capi.lib().ogl_reduction_chunk_rows()):
            ref = solve(A, b_o, np.zeros_like(b), inv, **skw)
        ref.x = ref.x[new_id]                      # back to the caller's order
        if args.mode in ("gpu-host", "gpu-peer") or standin:
            # same local trees, same order of the sum over ranks: bit-identical
            assert perf.n_iterations == (ref.n_iterations // 2 if (args.asym and not args.gmres)
                                         else ref.n_iterations)
            np.testing.assert_array_equal(hist, ref.history)
            np.testing.assert_array_equal(x, ref.x)
        else:
            m = min(hist.size, ref.history.size, 30)
            np.testing.assert_allclose(hist[:m], ref.history[:m], rtol=1e-10)
            np.testing.assert_allclose(x, ref.x, atol=1e-9, rtol=0)
        if args.max_iter >= 300 and glob is not None:
            np.testing.assert_allclose(gather_global(case, x), xs_g, atol=1e-8, rtol=0)


def main():
    if os.environ.get("OGL_DIST_FAULT_S"):      # a rank that hangs says where (python stack of every thread) before it is killed
        import faulthandler
        import signal
        import threading
        faulthandler.dump_traceback_later(float(os.environ["OGL_DIST_FAULT_S"]), exit=False)
        # ... and, with OGL_SEGV_TRACE=1, the native stack of the main thread (libogl_amd's SIGABRT handler)
        threading.Timer(float(os.environ["OGL_DIST_FAULT_S"]) + 1.0, lambda: signal.pthread_kill(threading.main_thread().ident, signal.SIGABRT)).start()
    ap = argparse.ArgumentParser()
    ap.add_argument("--mode", required=True)
    ap.add_argument("--shape", default="8,8,8")
    ap.add_argument("--procs", default="1,1,2")
    ap.add_argument("--precond", type=int, default=1)
    ap.add_argument("--asym", type=int, default=0)
    ap.add_argument("--gmres", type=int, default=0)
    ap.add_argument("--renumber", type=int, default=0,
                    help="1: the library renumbers every rank's device copy (config renumber = on); the "
                         "oracle then solves each rank's system permuted by the numbering the library reports")
    ap.add_argument("--halo-fused", type=int, default=1,
                    help="0: property haloFused off -- the peer-put transport's separate pack and finish kernels "
                         "instead of the puts folded into step_1x and the non-local part into the local SpMV kernel")
    ap.add_argument("--fused-turn-multi", type=int, default=1,
                    help="0: property fusedTurnMulti off -- the 5-launch multi-rank GKOCG turn (p put by step_1x) "
                         "instead of the merged 4-launch one (z put by step_2r)")
    ap.add_argument("--peer-safe-wait", type=int, default=0,
                    help="0 (default here: the fused waits are what these tests are for, and slab cuts do not starve): forced "
                         "off; -1: left to the library (peer_connect switches it on when two ranks report one PCI bus "
                         "id); 1: property peerSafeWait -- one workgroup waits for the neighbours' puts instead of every boundary "
                         "workgroup of the SpMV (ranks sharing a device with a cut that puts boundary rows into every chunk)")
    ap.add_argument("--expect-merged", type=int, default=-1,
                    help="0 / 1: assert that the merged turn did not run / ran (fusedTurnInUse)")
    ap.add_argument("--max-iter", type=int, default=300,
                    help="below 300: a fixed number of turns (large systems), no convergence check")
    ap.add_argument("--no-global", type=int, default=0,
                    help="1: do not assemble the global system on every rank (full-size decomposed configs: 20 M / 50 M "
                         "cells); b = A x* through the oracle's distributed product, rank-local and collective checks only")
    ap.add_argument("--shuffle", type=int, default=0,
                    help="W > 0: every rank's cells renamed at random inside windows of W cells (an unstructured numbering "
                         "per rank; interfaces keep their face order)")
    ap.add_argument("--relabel", type=int, default=0,
                    help="1 (gpu modes): after the first solve the LAST rank alone renames its cells (same "
                         "counts, new addressing) and every rank calls set_matrix + solve again: the rebuild "
                         "one rank asks for is collective, the others must re-upload their coefficients too")
    ap.add_argument("--random", type=int, default=-1,
                    help="seed: random irregular global system cut into contiguous row blocks of random "
                         "sizes (instead of the structured box of --shape/--procs)")
    args = ap.parse_args()
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    dist.init_process_group("gloo", rank=rank, world_size=world)
    if args.random >= 0:
        rng = np.random.default_rng(args.random)
        n_glob = int(rng.integers(600, 2500))
        glob = synthetic.random_global_case(n_glob, int(rng.integers(1, 5)), int(rng.choice([4, 60, 900])),
                                            symmetric=not args.asym, seed=args.random)
        cuts = np.sort(rng.choice(np.arange(1, n_glob), world - 1, replace=False)) if world > 1 else []
        bounds = [0, *[int(c) for c in cuts], n_glob]
        case = synthetic.partition_rows(glob, bounds, rank)
    else:
        gx, gy, gz = map(int, args.shape.split(","))
        px, py, pz = map(int, args.procs.split(","))
        assert px * py * pz == world
        kw = dict(symmetric=not args.asym)
        if args.asym:
            kw.update(off_upper=-0.9, off_lower=-1.1)
        case = synthetic.poisson_block(gx, gy, gz, px, py, pz, rank, **kw)
        glob = None if args.no_global else synthetic.poisson_block(gx, gy, gz, **kw)
        if args.shuffle:
            case = synthetic.renumber_case(case, args.shuffle, seed=20241016 + rank)
    xs_g = None if glob is None else synthetic.x_star(glob.global_index, glob.global_n)
    b_g = None if glob is None else synthetic.apply_case(glob, xs_g)
    skw = dict(tolerance=1e-11, rel_tol=0.0, max_iter=args.max_iter)
    cases = [case]
    if args.relabel:
        assert args.mode != "oracle"
        cases.append(synthetic.renumber_case(case, case.n_cells, seed=7) if rank == world - 1 else case)
    state = {}
    for round_no, case in enumerate(cases):
        run_case(args, rank, world, case, glob, xs_g, b_g, skw, state, round_no)
    if "reg" in state:
        state["reg"].close()
    dist.barrier()
    dist.destroy_process_group()
    print(f"rank {rank}: {args.mode} ok")


if __name__ == "__main__":
    main()
