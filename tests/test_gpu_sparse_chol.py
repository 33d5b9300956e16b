"""BA_SOLVER_SPARSE_SCHUR (nested dissection + multifrontal Cholesky of the reduced system) against the dense Cholesky
and the oracle.

The reference solves with Ceres' SPARSE_NORMAL_CHOLESKY (src/main.cpp:4706): a direct method, so every solver here that
is direct must give the same iterates to rounding.
"""
import numpy as np
import pytest

from bundleadjustment_amd import capi, host, solver
from util import assert_solve_parity

pytestmark = pytest.mark.gpu


def _opt(ls, iters=None):
    o = solver.default_options()
    o.linear_solver = ls
    if iters is not None:
        o.max_num_iterations = iters
    return o


CASES = {
    # 40 stations: a handful of leaves under one or two separators
    "small": dict(n_cam=1, n_sta=40, n_pts=600, rays_per_point=5, seed=2, ap_free_mask=host.AP_K1K2P1P2),
    # separators wider than one panel of 128, several levels
    "levels": dict(n_cam=1, n_sta=900, n_pts=9000, rays_per_point=5, seed=3, ap_free_mask=host.AP_K1K2P1P2),
    "camera_constant": dict(n_cam=1, n_sta=300, n_pts=3000, rays_per_point=4, seed=4, ap_free_mask=0, iop_free_mask=0),
    "four_cameras": dict(n_cam=4, n_sta=350, n_pts=5000, rays_per_point=6, seed=5, ap_free_mask=host.AP_K1K2P1P2),
    "ragged": dict(n_cam=2, n_sta=331, n_pts=3500, rays_per_point=5, seed=6, ap_free_mask=host.AP_K1K2P1P2, outlier_fraction=0.05),
}


@pytest.mark.parametrize("case", sorted(CASES))
def test_sparse_cholesky_equals_dense_cholesky(case):
    net = host.synth(**CASES[case]).net
    with solver.Problem(net) as P:
        d = P.solve(_opt(capi.SOLVER_DENSE_SCHUR))
    with solver.Problem(net) as P:
        s = P.solve(_opt(capi.SOLVER_SPARSE_SCHUR))
    assert s[4]["linear_solver_used"] == capi.SOLVER_SPARSE_SCHUR
    assert_solve_parity(s, d)


def test_sparse_cholesky_against_the_oracle():
    from oracle import py_oracle as po
    net = host.synth(n_cam=1, n_sta=130, n_pts=700, rays_per_point=6, seed=3, control_every=6).net
    opt = _opt(capi.SOLVER_SPARSE_SCHUR)
    with solver.Problem(net) as P:
        g = P.solve(opt)
    o = po.solve(net, _opt(capi.SOLVER_DENSE_SCHUR), po.SCHUR_DENSE)
    assert o[4]["termination"] == capi.CONVERGENCE
    assert_solve_parity(g, o)


def test_unused_stations_and_two_solves():
    # stations nobody observes stay out of the tree and keep their values; a second solve reuses the tree
    import dataclasses
    net = host.synth(n_cam=1, n_sta=260, n_pts=3000, rays_per_point=5, seed=8, ap_free_mask=host.AP_K1K2P1P2).net
    keep = ~np.isin(net.obs_sta, [5, 77, 201])
    kw = {k: getattr(net, k)[keep] for k in ("obs_pt", "obs_sta", "obs_x", "obs_y", "obs_sx", "obs_sy", "obs_cx", "obs_cy")}
    net = dataclasses.replace(net, **kw)
    with solver.Problem(net) as P:
        d = P.solve(_opt(capi.SOLVER_DENSE_SCHUR))
        s1 = P.solve(_opt(capi.SOLVER_SPARSE_SCHUR))
        s2 = P.solve(_opt(capi.SOLVER_SPARSE_SCHUR))
    assert_solve_parity(s1, d)
    assert np.array_equal(s1[0], s2[0]) and np.array_equal(s1[1], s2[1])  # same bits twice
    assert np.array_equal(s1[0][[5, 77, 201]], net.eop[[5, 77, 201]])


def test_a_network_without_block_form_is_refused():
    # every point seen from every station: the reduced system is dense, no block-sparse form is kept
    net = host.synth(n_cam=1, n_sta=30, n_pts=300, rays_per_point=30, seed=1).net
    with solver.Problem(net) as P:
        try:
            P.solve(_opt(capi.SOLVER_SPARSE_SCHUR, 2))
        except solver.BaError as e:
            assert e.code == capi.ERR_UNSUPPORTED


def test_a_factorisation_that_breaks_down_is_an_invalid_step():
    # All 16 additional parameters free with the reference's Gauss-Newton options: the radial monomials are numerically
    # collinear, a pivot of the camera block is not positive (tests/test_oracle_solver.py).  The sparse solver's report comes
    # back with the scalars of the step, not through a wait of its own: the step must still count as invalid, the radius
    # shrink, and the solve end as the dense solver's and the oracle's do.
    net = host.synth_config("cfg2", scale=0.025, ap_free_mask=host.AP_ALL16).net
    res = {}
    for ls in (capi.SOLVER_SPARSE_SCHUR, capi.SOLVER_DENSE_SCHUR, capi.SOLVER_PCG_SCHUR):
        opt = solver.default_options()
        opt.linear_solver = ls
        with solver.Problem(net) as P:
            res[ls] = P.solve(opt)
    s, d = res[capi.SOLVER_SPARSE_SCHUR][4], res[capi.SOLVER_DENSE_SCHUR][4]
    assert s["termination"] == capi.FAILURE and d["termination"] == capi.FAILURE
    assert s["num_unsuccessful_steps"] == d["num_unsuccessful_steps"] == solver.default_options().max_num_consecutive_invalid_steps - 1
    # (at which iteration a pivot first falls below zero is a matter of rounding: the factorisations eliminate in different orders)
    assert all(r["step_is_valid"] == 0 for r in res[capi.SOLVER_SPARSE_SCHUR][5][-s["num_unsuccessful_steps"]:])
    assert all(r["step_is_valid"] == 0 for r in res[capi.SOLVER_DENSE_SCHUR][5][-d["num_unsuccessful_steps"]:])
    # the PCG cannot start without the factor of the camera block: the same end, by the same rule
    g = res[capi.SOLVER_PCG_SCHUR][4]
    assert g["termination"] == capi.FAILURE and g["num_unsuccessful_steps"] == s["num_unsuccessful_steps"]
    # and the oracle (Ceres' loop restated) ends the same way
    from oracle import py_oracle as po
    o = po.solve(net, capi.default_options(capi.SOLVER_DENSE_SCHUR), po.SCHUR_DENSE)
    assert o[4]["termination"] == capi.FAILURE and o[4]["num_unsuccessful_steps"] == s["num_unsuccessful_steps"]
