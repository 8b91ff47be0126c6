"""A time-step loop as OpenFOAM runs it -- per step and field: constructor (lookup-or-create by name), new coefficients,
solve (lduLduBase.H:189-308, HostMatrix.C:15-96) -- leaves nothing behind.  The reference's device objects are created once
per field and live in the objectRegistry (DevicePersistent/Base/Base.H:53-137, HostMatrix.C:79-95); here every allocation
of the library is booked in a ledger (csrc/ledger.hpp, ogl_memory_ledger_read) and the tests assert on it EXACTLY: zero
drift over the steps, zero left after the last registry closes.  What the driver reports in use (hipMemGetInfo) also
moves with the runtime's own pools in 2 MiB granules, so it is judged by its slope over many marks, in a fresh process
(tests/soak_worker.py), not by the difference of two readings."""
import gc
import json
import os
import subprocess
import sys

import numpy as np
import pytest

from ogl_amd import capi, synthetic

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import soak_worker  # noqa: E402


def test_time_steps_leave_no_memory_behind():
    """In this process (whatever else the suite left on the device): the ledger does not move over 140 time steps of four
    solver / preconditioner pairs with the preconditioner regenerated every step (caching 0)."""
    gc.collect()  # (registries other tests dropped without closing go now, not between the two marks)
    reg = capi.Registry()
    fields = soak_worker.fields_of(24)
    marks = {}
    for step in range(161):
        soak_worker.one_step(reg, fields, step)
        if step in (20, 160):
            marks[step] = capi.memory_ledger().as_dict()
    reg.close()
    for k in soak_worker.LEDGER_EXACT:
        assert marks[160][k] == marks[20][k], (k, marks)
    assert marks[160]["unknown_frees"] == 0


def test_soak_in_a_fresh_process():
    """300 steps x 4 fields in a child process without torch: ledger drift exactly 0, everything returned at close, and
    the driver's in-use figure without a trend (a 15 kB/step leak -- 2 MiB over 140 steps, what round 5's red run would
    have been had it been a leak -- fails this; one 2 MiB granule of the runtime's pools does not)."""
    out = subprocess.run([sys.executable, os.path.join(HERE, "soak_worker.py"), "300", "25"], capture_output=True, text=True,
                         timeout=900)
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-2000:]
    s = json.loads(out.stdout.strip().splitlines()[-1])
    assert s["summary"] and s["solves"] == 301 * 4
    assert all(v == 0 for v in s["ledger_drift"].values()), s
    assert all(v == 0 for v in s["ledger_after_close"].values()), s
    assert abs(s["driver_slope_bytes_per_step"]) < 6144, s
    assert s["rss_growth_kb"] < 16384, s


def test_registries_come_and_go_without_residue():
    """A registry per run (objectRegistry analogue, DevicePersistent/Base/Base.H:53-137): creating one, solving on it and
    closing it -- stream, events, pinned staging buffers, every field's device arrays -- gives everything back."""
    case = synthetic.poisson_case(20)
    b = synthetic.rhs_for_x_star(case)[0]
    gc.collect()
    base = capi.memory_ledger().as_dict()
    for i in range(41):
        reg = capi.Registry()
        for name, pc in (("p", capi.PRECOND_BJ), ("q", capi.PRECOND_ISAI)):
            s = reg.solver(name, capi.default_config(preconditioner=pc, tolerance=1e-8, rel_tol=0.0, max_iter=200)).set_matrix(case)
            x, perf = s.solve(b, np.zeros_like(b))
            assert perf.final_residual < 1e-8
        inside = capi.memory_ledger().as_dict()
        assert inside["device_bytes"] > base["device_bytes"] and inside["pinned_bytes"] > base["pinned_bytes"]
        reg.close()
        now = capi.memory_ledger().as_dict()
        for k in soak_worker.LEDGER_EXACT:
            assert now[k] == base[k], (i, k, now, base)
    assert capi.memory_ledger().unknown_frees == 0
