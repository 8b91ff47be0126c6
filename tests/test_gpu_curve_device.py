"""The Hilbert-curve candidate of the numbering policy (`renumber auto | on` with `ogl_ldu_view::cell_centres`) is formed on the
device since round 6 -- keys by Skilling's transpose, a stable radix sort, the inverse permutation, and the count of entries
that would leave their chunk's window of packed columns (setup_kernels.hip) -- and must be the host's (host_matrix.cpp
hilbert_order, choose_numbering) entry for entry: the same numbering, the same counts, ties in the caller's order.  The
reference reads no geometry (this is the build's own addition); what is pinned here is that the
two implementations of it cannot drift apart."""
import numpy as np
import pytest

from ogl_amd import capi, synthetic

pytestmark = pytest.mark.gpu


def numbering(reg, name, case, on_device):
    s = reg.solver(name, capi.default_config(solver=capi.SOLVER_CG, preconditioner=capi.PRECOND_BJ, renumber=capi.RENUMBER_ON))
    s.set_property("curveOnDevice", 1.0 if on_device else 0.0)
    s.set_matrix(case)
    props = {k: s.get_property(k) for k in ("renumberedAlongCurve", "gatherSectorRatioCurve", "gatherSectorRatioRcm",
                                            "curveFarEntries", "renumbered")}
    return s.renumbering(), props


def cases():
    vor = synthetic.voronoi_case(60000, with_centres=True)
    box = synthetic.renumber_case(synthetic.poisson_block(40, 36, 30, with_centres=True), 4096)
    # many cells per lattice point of the curve: centres snapped to a coarse grid -> equal keys, the caller's order decides
    ties = synthetic.renumber_case(synthetic.poisson_block(32, 32, 32, with_centres=True), 2048)
    ties.centres = np.floor(ties.centres / 4.0) * 4.0
    flat = synthetic.renumber_case(synthetic.poisson_block(64, 64, 1, with_centres=True), 512)   # zero extent along z
    return {"voronoi": vor, "shuffled box": box, "ties": ties, "flat": flat}


@pytest.mark.parametrize("which", ["voronoi", "shuffled box", "ties", "flat"])
def test_device_curve_is_the_hosts(which):
    case = cases()[which]
    reg = capi.Registry()
    dev, pd = numbering(reg, "dev", case, True)
    host, ph = numbering(reg, "host", case, False)
    assert pd == ph, (pd, ph)
    if dev is None or host is None:
        assert dev is None and host is None
    else:
        np.testing.assert_array_equal(dev, host)
    if pd["renumberedAlongCurve"] == 1.0:      # the numbering in use IS the curve's: compare with the pure host function too
        np.testing.assert_array_equal(dev, capi.host_hilbert_order(case.centres))
    reg.close()


def test_far_entry_count_above_two_million_cells():
    """Above 2^21 cells the policy counts the entries that would fall outside their chunk's 2^21-column window along the
    curve (it gives the curve up beyond 2 %): the device's count is the host's."""
    case = synthetic.renumber_case(synthetic.poisson_block(132, 128, 128, with_centres=True), 65536)
    assert case.n_cells > (1 << 21)
    reg = capi.Registry()
    dev, pd = numbering(reg, "dev", case, True)
    host, ph = numbering(reg, "host", case, False)
    assert pd == ph, (pd, ph)
    np.testing.assert_array_equal(dev, host)
    reg.close()
