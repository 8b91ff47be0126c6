"""Soak test of the peer mesh (launched by tools/gpu_stress.sh / test_distributed.py): R ranks on the
peer mesh solve the same distributed system over and over with changing stop positions, solvers and
right-hand sides; every solve must be bit-identical to the distributed oracle (rank-ordered sums).
Looks for rare ordering bugs in the mailbox / flag protocols."""
import argparse
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

from ogl_amd import capi, synthetic  # noqa: E402
from oracle import oracle as orc  # noqa: E402
from helpers import blocked  # noqa: E402
from dist_worker import allreduce, allreduce_rank_order, make_exchange, oracle_dist_matrix  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--solves", type=int, default=50)
    ap.add_argument("--seed", type=int, default=1)
    args = ap.parse_args()
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    dist.init_process_group("gloo", rank=rank, world_size=world)
    rng = np.random.default_rng(args.seed)                   # same stream on every rank

    def build(seed):
        """a new global system and partition: every field's pattern changes on every rank"""
        n_glob = int(rng.integers(1500, 4000))
        glob = synthetic.random_global_case(n_glob, 3, 80, symmetric=True, seed=seed)
        cuts = np.sort(rng.choice(np.arange(1, n_glob), world - 1, replace=False))
        bounds = [0, *[int(c) for c in cuts], n_glob]
        case = synthetic.partition_rows(glob, bounds, rank)
        A, (rp, cols, vals), nl, comm = oracle_dist_matrix(case, allreduce_rank_order)
        return n_glob, case, A, orc.jacobi_generate_scalar(rp, cols, vals), orc.Precond(rp, cols, vals, 1)

    n_glob, case, A, inv, P = build(args.seed)

    reg = capi.Registry(device_id=rank % max(1, torch.cuda.device_count()))
    ex = make_exchange(None)
    reg.set_host_comm(rank, world, allreduce, lambda nb, ct, s: ex(nb, ct, s))
    handles = [None] * world
    dist.all_gather_object(handles, reg.peer_handle())
    reg.peer_connect(rank, world, handles)
    solvers = {}
    for it in range(args.solves):
        if it and it % 12 == 0:                              # topology change: collective pattern rebuild
            n_glob, case, A, inv, P = build(args.seed + it)
            for sv in solvers.values():
                sv.set_matrix(case)
                assert sv.get_property("peerHalo") == 1.0
        kind = ["cg", "bicgstab", "gmres"][int(rng.integers(0, 3))]
        max_iter = int(rng.integers(1, 45))
        tol = float(10.0 ** rng.uniform(-12, -2))
        b = rng.uniform(-1, 1, n_glob)[case.global_index]
        kw = dict(tolerance=tol, rel_tol=0.0, max_iter=max_iter)
        cfg = capi.default_config(
            solver={"cg": capi.SOLVER_CG, "bicgstab": capi.SOLVER_BICGSTAB, "gmres": capi.SOLVER_GMRES}[kind],
            preconditioner=capi.PRECOND_BJ, krylov_dim=7, export_res=1, adapt_min_iter=0,
            update_init_guess=1, force_host_buffer=1, **kw)
        if kind not in solvers:
            solvers[kind] = reg.solver(kind, cfg).set_matrix(case)
            assert solvers[kind].get_property("peerHalo") == 1.0
        s = reg.solver(kind, cfg)                            # same field: persistent state, new controls
        x, perf = s.solve(b, np.zeros_like(b))
        with blocked(orc, capi.lib().ogl_reduction_chunk_rows()):
            if kind == "cg":
                ref = orc.cg(A, b, np.zeros_like(b), inv, **kw)
            elif kind == "bicgstab":
                ref = orc.bicgstab(A, b, np.zeros_like(b), inv, **kw)
            else:
                ref = orc.gmres(A, b, np.zeros_like(b), P, krylov_dim=7, **kw)
        np.testing.assert_array_equal(s.history(), ref.history, err_msg=f"solve {it} {kind} {max_iter}")
        np.testing.assert_array_equal(x, ref.x, err_msg=f"solve {it} {kind} {max_iter}")
    reg.close()
    dist.barrier()
    print(f"rank {rank}: {args.solves} solves ok")


if __name__ == "__main__":
    main()
