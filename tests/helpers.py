"""Shared helpers for the parity tests: build the oracle's view of a synthetic case."""
import numpy as np


def orc_ifaces(oracle, case):
    return [oracle.Iface(f.kind, f.face_cells, f.bou_coeffs, f.neighb_proc, f.neighb_patch)
            for f in case.interfaces]


def oracle_csr(oracle, case, host_path=False, scaling=1.0):
    """(rowptr, cols, vals) of the local matrix as the reference assembles it."""
    ifs = orc_ifaces(oracle, case)
    rows, cols, perm = oracle.init_local_sparsity_pattern(case.n_cells, case.upper_addr,
                                                          case.lower_addr, case.symmetric, ifs)
    vals = oracle.update_local_matrix_data(case.diag, case.upper, case.lower, ifs, perm,
                                           host_path=host_path, scaling=scaling)
    return oracle.rowptr_from_rows(case.n_cells, rows), cols, vals


def oracle_matrix(oracle, case, **kw):
    rowptr, cols, vals = oracle_csr(oracle, case, **kw)
    return oracle.DistMatrix(rowptr, cols, vals), (rowptr, cols, vals)


class blocked:
    """Run the oracle's reductions in the device's fixed tree (bit-exact comparisons)."""

    def __init__(self, oracle, chunk_rows):
        self.oracle, self.chunk_rows = oracle, chunk_rows

    def __enter__(self):
        self.oracle.set_reduction(self.oracle.REDUCE_BLOCKED, self.chunk_rows)

    def __exit__(self, *a):
        self.oracle.set_reduction(self.oracle.REDUCE_SEQUENTIAL)


def rel_dev(a, b):
    a, b = np.asarray(a, float), np.asarray(b, float)
    return np.abs(a - b) / np.maximum(np.abs(b), 1e-300)


def to_new(v, new_id):
    """A caller-order vector in the renumbered order: out[new_id[c]] = v[c]."""
    out = np.empty_like(np.asarray(v))
    out[new_id] = v
    return out


def oracle_matrix_renumbered(oracle, case, new_id, **kw):
    """The oracle's matrix of `case` permuted by the numbering the product reports
    (Solver.renumbering()): DistMatrix + (rowptr, cols, vals) of P A P^T."""
    rowptr, cols, vals = oracle_csr(oracle, case, **kw)
    rp, cc, vv, _ = oracle.permute_csr(rowptr, cols, vals, new_id)
    return oracle.DistMatrix(rp, cc, vv), (rp, cc, vv)


def oracle_precond_renumbered(oracle, case, rp, cols, vals, new_id, max_block_size=1, isai=None, sparsity_power=1, **kw):
    """The reference's preconditioner -- generated on the matrix OpenFOAM hands over, i.e. in the CALLER's numbering
    (Preconditioner.H:91-105, :225-241) -- expressed on the system (rp, cols, vals) the library renumbered by new_id:
    block-Jacobi blocks are runs of consecutive CALLER rows, ISAI(spd) takes tril(A) by the caller's index."""
    rp0, c0, v0 = oracle_csr(oracle, case, **kw)
    return oracle.Precond(rp, cols, vals, max_block_size, isai=isai, sparsity_power=sparsity_power,
                          caller=(rp0, c0, v0, new_id))
