"""The EXACT reduction mode of the oracle (ORC_REDUCE_EXACT: TwoSum / TwoProduct accumulation of every dot, norm1, sum
and SpMV row sum, rounded once) -- the arbiter between two summation orders (SURVEY.md section 7
"hard parts": reduction order).

north_star asks for the reference executor's iterate history "to 1e-12 rel".  The reference executor sums left to
right; the HIP kernels sum in a fixed tree.  Neither is exact.  This file pins, on the CPU, (a) that the EXACT mode is
exact (against rational arithmetic), (b) which of the two orders is closer to the exact history of a GKOCG solve and by
how much -- the statement BASELINE.md makes; tests/test_gpu_exact_arbiter.py repeats (b) with the device's history in
place of the oracle's BLOCKED one (the two are bit-equal, tests/test_gpu_fullsize_oracle.py).
Reference: StoppingCriterion/StoppingCriterion.C:92-113 (the norm the history holds), [UPSTREAM] gko::solver::Cg.
"""
from fractions import Fraction

import numpy as np
import pytest

from ogl_amd import synthetic
from helpers import oracle_matrix


class exact:
    def __init__(self, oracle):
        self.o = oracle

    def __enter__(self):
        self.o.set_reduction(self.o.REDUCE_EXACT)

    def __exit__(self, *a):
        self.o.set_reduction(self.o.REDUCE_SEQUENTIAL)


def test_exact_mode_is_exact_against_rational_arithmetic(oracle):
    rng = np.random.default_rng(20241016)
    n = 30000
    # (cancelling terms of very different magnitude: the plain sums lose digits, the exact one must not)
    a = rng.uniform(-1, 1, n) * 10.0 ** rng.integers(-6, 7, n)
    b = rng.uniform(-1, 1, n)
    fa, fb = [Fraction(v) for v in a.tolist()], [Fraction(v) for v in b.tolist()]
    with exact(oracle):
        assert oracle.dot(a, b) == float(sum(x * y for x, y in zip(fa, fb)))
        assert oracle.vsum(a) == float(sum(fa))
        assert oracle.norm1(a) == float(sum(abs(x) for x in fa))
    assert oracle.dot(a, b) != float(sum(x * y for x, y in zip(fa, fb)))      # ... which the left-to-right sum misses
    # SpMV rows
    case = synthetic.poisson_case(6, symmetric=False)
    _, (rp, cols, vals) = oracle_matrix(oracle, case)
    x = rng.uniform(-1, 1, case.n_cells) * 10.0 ** rng.integers(-8, 9, case.n_cells)
    with exact(oracle):
        y = oracle.spmv(rp, cols, vals, x)
    for r in range(case.n_cells):
        s = sum(Fraction(float(vals[k])) * Fraction(float(x[cols[k]])) for k in range(rp[r], rp[r + 1]))
        assert y[r] == float(s)


def histories(oracle, edge, chunk_rows=512, **kw):
    case = synthetic.poisson_case(edge)
    A, (rp, cols, vals) = oracle_matrix(oracle, case)
    b, _ = synthetic.rhs_for_x_star(case)
    inv = oracle.jacobi_generate_scalar(rp, cols, vals)
    out = {}
    for name, mode in (("seq", oracle.REDUCE_SEQUENTIAL), ("tree", oracle.REDUCE_BLOCKED), ("exact", oracle.REDUCE_EXACT)):
        oracle.set_reduction(mode, chunk_rows)
        try:
            out[name] = oracle.cg(A, b, np.zeros_like(b), inv, **kw).history
        finally:
            oracle.set_reduction(oracle.REDUCE_SEQUENTIAL)
    return out


def envelopes(h):
    m = min(v.size for v in h.values())
    ex = h["exact"][:m]
    d_seq = np.maximum.accumulate(np.abs(h["seq"][:m] - ex) / ex)
    d_tree = np.maximum.accumulate(np.abs(h["tree"][:m] - ex) / ex)
    return ex / ex[0], d_seq, d_tree


# |tree - exact| <= C_ARBITER * max(|seq - exact|, 1e-13), on the running maxima over the checks so far (a single check
# of a CG history that has amplified rounding noise is a coin toss between any two orders: measured worst single-check
# ratio 50, worst running-maximum ratio 3.1 at 32^3 / 1.1 at 64^3)
C_ARBITER = 8.0


@pytest.mark.parametrize("edge", [24, 32])
def test_device_tree_is_no_further_from_exact_than_the_reference_order(oracle, edge):
    h = histories(oracle, edge, tolerance=1e-9, rel_tol=0.0, max_iter=1000)
    rel, d_seq, d_tree = envelopes(h)
    assert h["seq"].size == h["tree"].size == h["exact"].size                # same iteration count in all three orders
    assert (d_tree <= C_ARBITER * np.maximum(d_seq, 1e-13)).all()
    # while the residual is above 1e-3 of its start the tree order is the CLOSER one, by more than an order of
    # magnitude (pairwise sums: error ~ log2(n) eps; left to right: ~ sqrt(n) eps) ...
    early = rel > 1e-3
    assert early.sum() >= 20
    assert d_tree[early].max() <= 0.1 * d_seq[early].max() and d_tree[early].max() <= 1e-14
    # ... and north_star's 1e-12 holds for BOTH orders against the exact history there
    assert d_seq[early].max() <= 1e-12
    print(f"{edge}^3: above 1e-3 of the start seq {d_seq[early].max():.2e} tree {d_tree[early].max():.2e}; "
          f"whole solve seq {d_seq[-1]:.2e} tree {d_tree[-1]:.2e}")
