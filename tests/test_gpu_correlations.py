"""The camera rows of the covariance matrix (ba_covariance_camera_rows: the blocks the reference's correlation analysis
takes from ceres::Covariance, src/main.cpp:5720-5778, 6154-6189, 6423-6436, 5592-5598) against the oracle's inverse of
the full normal matrix, the three reduced solvers against each other, and the correlation tables end to end.
Parity with Ceres itself is unpinned (the reference holds no covariance output)."""
import numpy as np
import pytest

from bundleadjustment_amd import capi, host, solver
from oracle import py_oracle as po

pytestmark = pytest.mark.gpu


def options(ls):
    opt = solver.default_options()
    opt.linear_solver = ls
    return opt


def as_correlations(rows, var):
    """Rows divided by the standard deviations of both sides: entries of one row span many orders of magnitude."""
    par, ce, cx, cc = rows
    ve, vx, vi, va = var
    sd_cam = np.array([np.sqrt(vi[q // 19, q % 19] if q % 19 < 3 else va[q // 19, q % 19 - 3]) for q in par])
    with np.errstate(divide="ignore", invalid="ignore"):
        e = ce / np.sqrt(ve)[None] / sd_cam[:, None, None]
        x = None if cx is None else cx / np.sqrt(vx)[None] / sd_cam[:, None, None]
    c = cc / np.outer(sd_cam, sd_cam)
    return np.nan_to_num(e), None if x is None else np.nan_to_num(x), c


@pytest.mark.parametrize("which", ["cfg1", "three_cameras"])
def test_camera_rows_against_the_oracle(which):
    if which == "cfg1":
        net = host.synth_config("cfg1").net
    else:
        net = host.synth(n_cam=3, n_sta=12, n_pts=150, rays_per_point=6, ap_free_mask=host.AP_K1K2P1P2, seed=5).net
    with solver.Problem(net) as P:
        P.set_parameters()
        P.solve_resident(options(capi.SOLVER_DENSE_SCHUR))
        eop, xyz, iop, ap = P.get_parameters()
        var = P.covariance()
        got = P.covariance_camera_rows()
        assert np.array_equal(np.stack(P.get_parameters()[2]), iop)  # the parameters are left alone
    want = po.covariance_rows(net, eop, xyz, iop, ap)
    wvar = po.covariance(net, eop, xyz, iop, ap)
    assert np.array_equal(got[0], want[0])
    assert len(got[0]) == (10 if which == "cfg1" else 21)
    for g, w, name in zip(as_correlations(got, wvar), as_correlations(want, wvar), ("eop", "xyz", "cam")):
        assert g.shape == w.shape, name
        assert np.max(np.abs(g - w)) <= 2e-8, (name, np.max(np.abs(g - w)))
    # the diagonal of the camera corner is the variance ba_covariance reports; the corner is symmetric
    for k, q in enumerate(got[0]):
        c, q = int(q) // 19, int(q) % 19
        v = var[2][c, q] if q < 3 else var[3][c, q - 3]
        assert abs(got[3][k, k] - v) <= 1e-9 * v
    assert np.allclose(got[3], got[3].T, rtol=1e-9, atol=0)
    # and the tables the reference prints, from GPU figures only, against the same from the oracle's
    for camera in range(net.n_cam):
        tg, tw = host.correlations(var, got, camera=camera), host.correlations(wvar, want, camera=camera)
        for k in tw:
            assert np.array_equal(np.isnan(tg[k]), np.isnan(tw[k])), k
            assert np.allclose(tg[k], tw[k], rtol=0, atol=1e-7, equal_nan=True), k


def test_the_three_reduced_solvers_give_the_same_rows():
    # 300 stations (order 1807): dense factor, sparse direct factor (one factorisation, a pair of tree solves per row)
    # and one PCG solve per row; 1e-6 for the PCG, which stops at a relative residual of 1e-12
    net = host.synth(n_cam=1, n_sta=300, n_pts=3000, rays_per_point=5, seed=11, ap_free_mask=host.AP_K1K2P1P2).net
    res = {}
    try:
        for how in (0, 1, 2):
            solver.tuning("rows_solver", how)
            with solver.Problem(net) as P:
                P.set_parameters()
                P.solve_resident(options(capi.SOLVER_PCG_SCHUR))
                if how == 0:
                    var = P.covariance()
                res[how] = P.covariance_camera_rows()
                if how == 1:
                    assert P.sparse_solver_stats()["fronts"] >= 3
    finally:
        solver.tuning("rows_solver", -1)
    base = as_correlations(res[0], var)
    for how, tol in ((1, 1e-8), (2, 1e-6)):
        assert np.array_equal(res[how][0], res[0][0])
        for g, w, name in zip(as_correlations(res[how], var), base, ("eop", "xyz", "cam")):
            assert np.max(np.abs(g - w)) <= tol, (how, name, np.max(np.abs(g - w)))
    used = np.bincount(net.obs_sta, minlength=net.n_sta) > 0
    assert np.all(res[0][1][:, ~used] == 0.0)
    # correlations are correlations
    for part in base:
        assert np.max(np.abs(part)) <= 1.0 + 1e-9


def test_rows_of_a_network_without_free_camera_parameters_and_sizing_call():
    net = host.synth(n_cam=1, n_sta=9, n_pts=80, rays_per_point=5, ap_free_mask=0, iop_free_mask=0, seed=7, control_every=8).net
    with solver.Problem(net) as P:
        P.set_parameters()
        par, ce, cx, cc = P.covariance_camera_rows()
    assert len(par) == 0 and ce.shape == (0, 9, 6) and cc.shape == (0, 0)
    net = host.synth_config("cfg1").net
    with solver.Problem(net) as P:
        with pytest.raises(solver.BaError):
            P.covariance_camera_rows()  # no parameters set
        P.set_parameters()
        par, ce, cx, cc = P.covariance_camera_rows(points=False)
        assert cx is None and len(par) == 10 and list(par) == [0, 1, 2] + [3 + k for k in range(7)]
