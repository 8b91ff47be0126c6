"""Pin the oracle on the reference's own known-answer vectors.

Vectors: tests/golden/host_matrix_kat.json (from unitTests/test_HostMatrix.C:8-107 and
SURVEY.md §10).  These tests read like the reference's gtest cases on purpose.
"""
import numpy as np
import pytest


def test_symmetric_update(oracle, golden):
    g = golden["symmetric_update"]
    res = oracle.symmetric_update(g["permute"], g["scale"], g["diag"], g["upper"])
    assert res.tolist() == [float(v) for v in g["expected"]]


def test_symmetric_update_ignores_scale(oracle, golden):
    # SURVEY.md §10.4 / HostMatrixFreeFunctions.C:27-28: scale only acts as a truth value
    g = golden["symmetric_update"]
    res = oracle.symmetric_update(g["permute"], -1.0, g["diag"], g["upper"])
    assert res.tolist() == [float(v) for v in g["expected"]]


def test_non_symmetric_update(oracle, golden):
    g = golden["non_symmetric_update"]
    res = oracle.non_symmetric_update(g["permute"], g["scale"], g["diag"], g["upper"], g["lower"])
    assert res.tolist() == [float(v) for v in g["expected"]]


def test_non_symmetric_update_scales(oracle, golden):
    g = golden["non_symmetric_update"]
    res = oracle.non_symmetric_update(g["permute"], -2.0, g["diag"], g["upper"], g["lower"])
    assert res.tolist() == [-2.0 * v for v in g["expected"]]


@pytest.mark.parametrize("case", ["init_local_sparsity", "init_local_sparsity_asym",
                                  "init_local_sparsity_box2"])
def test_init_local_sparsity(oracle, golden, case):
    g = golden[case]
    rows, cols, perm = oracle.init_local_sparsity(g["nrows"], g["upper"], g["lower"],
                                                  g["is_symmetric"])
    assert rows.tolist() == g["rows"]
    assert cols.tolist() == g["cols"]
    assert perm.tolist() == g["permute"]


def test_update_w_interface_variants(oracle, golden):
    # no golden vector exists for the *_w_interface variants; check them against the plain
    # ones (same decode, HostMatrixFreeFunctions.C:32-82) plus an interface tail
    g = golden["non_symmetric_update"]
    perm = g["permute"] + [17, 18]
    iface = [7.0, 9.0]
    res = oracle.non_symmetric_update_w_interface(perm, 2.0, g["diag"], g["upper"], g["lower"],
                                                  iface)
    assert res.tolist() == [2.0 * v for v in g["expected"]] + [14.0, 18.0]
    g = golden["symmetric_update"]
    perm = g["permute"] + [11, 12]
    res = oracle.symmetric_update_w_interface(perm, -1.0, g["diag"], g["upper"], iface)
    assert res.tolist() == [-1.0 * v for v in g["expected"]] + [-7.0, -9.0]


def test_box_generator_matches_golden_addressing(golden):
    from ogl_amd import synthetic
    lo, up = synthetic.box_faces(2, 2, 2)
    g = golden["init_local_sparsity_box2"]
    assert lo.tolist() == g["lower"] and up.tolist() == g["upper"]
