"""Frozen self-generated fixtures (tools/gen_golden.py).

PROVENANCE: both files hold the ORACLE's outputs, not the reference's (the reference holds no
Krylov numbers and its free functions do not build here).  They guard against an accidental change
of the oracle AND the kernels at once; the host-matrix cases are cross-checked by the product's
independently written algorithm (ogl_amd/csrc/host_matrix.cpp).
"""
import json
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))

import gen_golden  # noqa: E402
from ogl_amd import capi  # noqa: E402


def load(name):
    with open(os.path.join(ROOT, "tests", "golden", name)) as f:
        return json.load(f)


def unhex(a):
    return np.array([float.fromhex(v) for v in a], dtype=np.float64)


KRYLOV = load("krylov_histories.json")["cases"]
HOST = load("host_matrix_generated.json")["cases"]


@pytest.mark.parametrize("name", [c[0] for c in gen_golden.KRYLOV_CASES])
def test_oracle_reproduces_the_frozen_history(oracle, name):
    g = KRYLOV[name]
    spec = next(c for c in gen_golden.KRYLOV_CASES if c[0] == name)
    oracle.set_reduction(oracle.REDUCE_SEQUENTIAL)
    case, xs, b, res = gen_golden.krylov_case(*spec[1:])
    assert res.n_iterations == g["n_iterations"]
    np.testing.assert_array_equal(res.history, unhex(g["history"]))
    assert res.norm_factor == float.fromhex(g["norm_factor"])
    assert res.initial_residual == float.fromhex(g["initial_residual"])
    assert res.final_residual == float.fromhex(g["final_residual"])
    np.testing.assert_array_equal(res.x[:: max(1, case.n_cells // 16)][:16], unhex(g["x_probe"]))
    assert float(np.sum(res.x)) == float.fromhex(g["x_checksum"])
    assert np.abs(res.x - xs).max() < 1e-6


@pytest.mark.parametrize("name", [c[0] for c in gen_golden.HOST_CASES])
def test_host_matrix_fixture_oracle_and_product(oracle, name):
    g = HOST[name]
    spec = next(c for c in gen_golden.HOST_CASES if c[0] == name)
    case = gen_golden.host_case(spec[1], spec[2])
    assert case.lower_addr.tolist() == g["lower_addr"] and case.upper_addr.tolist() == g["upper_addr"]
    # --- the oracle still gives what was frozen
    from helpers import orc_ifaces
    ifs = orc_ifaces(oracle, case)
    rows, cols, perm = oracle.init_local_sparsity_pattern(case.n_cells, case.upper_addr, case.lower_addr,
                                                          case.symmetric, ifs)
    assert rows.tolist() == g["rows"] and cols.tolist() == g["cols"] and perm.tolist() == g["ldu_mapping"]
    vals = oracle.update_local_matrix_data(case.diag, case.upper, case.lower, ifs, perm)
    np.testing.assert_array_equal(vals, unhex(g["coeffs"]))
    # --- the product's independent algorithm gives the same
    d, loc, nl, comm = capi.host_pattern(case)
    assert loc[0].tolist() == g["rows"] and loc[1].tolist() == g["cols"]
    assert loc[2].tolist() == g["ldu_mapping"]
    assert nl[0].tolist() == g["non_local"]["rows"] and nl[1].tolist() == g["non_local"]["cols"]
    assert nl[2].tolist() == g["non_local"]["ldu_mapping"]
    assert comm[0].tolist() == g["comm"]["target_ids"]
    assert comm[1].tolist() == g["comm"]["target_sizes"]
    assert comm[2].tolist() == g["comm"]["send_idxs"]
    # coefficients through the product's host update functions (reorderOnHost path, scale 1)
    iface = np.concatenate([-1.0 * f.bou_coeffs for f in case.interfaces if f.kind == 1] or [np.zeros(0)])
    if iface.size:
        if case.symmetric:
            pv = capi.host_symmetric_update_w_interface(loc[2], 1.0, case.diag, case.upper, iface)
        else:
            pv = capi.host_non_symmetric_update_w_interface(loc[2], 1.0, case.diag, case.upper,
                                                            case.lower, iface)
    elif case.symmetric:
        pv = capi.host_symmetric_update(loc[2], 1.0, case.diag, case.upper)
    else:
        pv = capi.host_non_symmetric_update(loc[2], 1.0, case.diag, case.upper, case.lower)
    np.testing.assert_array_equal(pv, unhex(g["coeffs"]))
    # structural properties the reference's data_validation.py checks on its .mtx dumps
    # (test/data_validation.py:55-111): sorted row-major, positive diagonal, negative off-diagonal
    r, c, v = np.array(g["rows"]), np.array(g["cols"]), unhex(g["coeffs"])
    assert np.all(np.diff(r) >= 0)
    assert np.all(v[r == c] > 0) and np.all(v[r != c] < 0)
