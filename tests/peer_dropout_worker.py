"""Two ranks on the peer mesh; rank 1 skips the solve.  Rank 0 must come back with OGL_ERR_COMM after
the (shortened) timeout instead of hanging (launched by test_distributed.py)."""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

from ogl_amd import capi, synthetic  # noqa: E402


def main():
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    dist.init_process_group("gloo", rank=rank, world_size=world)
    case = synthetic.poisson_block(8, 8, 8, 1, 1, world, rank)
    b = np.ones(case.n_cells)
    reg = capi.Registry(device_id=rank % max(1, torch.cuda.device_count()))

    def allreduce(a):
        t = torch.from_numpy(np.ascontiguousarray(a, dtype=np.float64).copy())
        dist.all_reduce(t)
        return t.numpy()

    reg.set_host_comm(rank, world, allreduce, lambda nb, ct, s: s)   # halo never goes this way here
    handles = [None] * world
    dist.all_gather_object(handles, reg.peer_handle())
    reg.peer_connect(rank, world, handles)
    cfg = capi.default_config(solver=capi.SOLVER_CG, preconditioner=capi.PRECOND_BJ, tolerance=1e-10,
                              rel_tol=0.0, max_iter=50, force_host_buffer=1)
    s = reg.solver("p", cfg).set_matrix(case)          # collective handshake: both ranks
    assert s.get_property("peerHalo") == 1.0
    if rank == 0:
        t0 = time.time()
        try:
            s.solve(b, np.zeros_like(b))
            raise SystemExit("rank 0: the solve returned although rank 1 never took part")
        except capi.OglError as e:
            assert e.status == capi.ERR_COMM, e
            assert time.time() - t0 < 30, "timeout not honoured"
            print(f"rank 0: failed loudly after {time.time() - t0:.1f} s: {e}")
    dist.barrier()
    print(f"rank {rank}: dropout ok")


if __name__ == "__main__":
    main()
