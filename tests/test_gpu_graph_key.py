"""A captured batch of GKOCG turns (hipGraph) replays kernel launches with the arguments they had at capture time.  Its key
is the hash of every view struct the launchers read (csrc/launch_key.hpp): when a field's layout changes under a live
graph -- half storage <-> compressed full storage <-> plain CSR <-> Ell, the library's own numbering switched on, a
pattern rebuilt with the same sizes -- the graph must be captured again, and the solve must carry the bits of the same
solve without a graph.  One solver object per field name lives through all of it, as in a run whose dictionary or mesh
changes between time steps (lookup-or-create by name: DevicePersistent/Base/Base.H:75-115, HostMatrix.C:79-87)."""
import numpy as np
import pytest

from ogl_amd import capi, synthetic

pytestmark = pytest.mark.gpu


def box(gx, gy, gz, symmetric=True):
    return synthetic.poisson_block(gx, gy, gz, symmetric=symmetric)


STEPS = [
    # (label, case factory, config overrides)
    ("half storage", lambda: box(32, 64, 32), {}),
    ("same sizes, other box", lambda: box(64, 32, 32), {}),                      # pattern rebuilt, buffers of equal size
    ("compressed full storage", lambda: box(64, 32, 32), dict(symmetric_half=0)),
    ("plain CSR", lambda: box(64, 32, 32), dict(compress_indices=0)),
    ("Ell", lambda: box(64, 32, 32), dict(matrix_format=capi.FORMAT_ELL)),
    ("shuffled cells, library renumbers", lambda: synthetic.renumber_case(box(40, 40, 40), 4096), dict(renumber=capi.RENUMBER_ON)),
    ("shuffled cells, caller's numbering", lambda: synthetic.renumber_case(box(40, 40, 40), 4096), dict(renumber=capi.RENUMBER_OFF)),
    ("half storage again", lambda: box(32, 64, 32), {}),
    ("no preconditioner", lambda: box(32, 64, 32), dict(preconditioner=capi.PRECOND_NONE)),
]


def test_every_layout_change_under_a_live_graph_is_seen():
    reg = capi.Registry()
    base = dict(solver=capi.SOLVER_CG, preconditioner=capi.PRECOND_BJ, tolerance=0.0, rel_tol=0.0, max_iter=70, export_res=1,
                adapt_min_iter=0, update_init_guess=1, matrix_format=capi.FORMAT_CSR)
    captures = 0.0
    for label, make, over in STEPS:
        case = make()
        b = synthetic.rhs_for_x_star(case)[0]
        got = {}
        for graph in (1.0, 0.0):
            # ONE field name per mode: the graph-on solver keeps its captured graph from step to step
            s = reg.solver(f"p_graph_{int(graph)}", capi.default_config(**{**base, **over}))
            s.set_property("hipGraph", graph)
            s.set_matrix(case)
            x, perf = s.solve(b, np.zeros_like(b))
            got[graph] = (x, perf.n_iterations, s.history().copy())
            if graph:
                assert s.get_property("hipGraphCaptures") >= captures + 1.0, (label, captures)   # the change was seen
                # (the second solve of a field takes its own preconditioner object instead of the registry's stored one,
                #  Preconditioner.H:411-413: one more capture; from then on nothing changes any more)
                x2, _ = s.solve(b, np.zeros_like(b))
                captures = s.get_property("hipGraphCaptures")
                x3, _ = s.solve(b, np.zeros_like(b))
                assert s.get_property("hipGraphCaptures") == captures, label
                np.testing.assert_array_equal(x2, x)
                np.testing.assert_array_equal(x3, x)
        assert got[1.0][1] == got[0.0][1] == 71, label
        np.testing.assert_array_equal(got[1.0][2], got[0.0][2], err_msg=label)
        np.testing.assert_array_equal(got[1.0][0], got[0.0][0], err_msg=label)
    reg.close()
