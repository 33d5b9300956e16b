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
