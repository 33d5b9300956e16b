"""Variances from the dissection of the station graph (ba_covariance, route 2) against the dense inverse (route 1) and the oracle.

The reference asks ceres::Covariance for the diagonal blocks only (src/main.cpp:5261-5279, 5443-5565); both routes
return exactly those, so they must agree to the rounding of two different factorisations of the same matrix.
"""
import numpy as np
import pytest

from bundleadjustment_amd import capi, host, solver

pytestmark = pytest.mark.gpu


def _options(ls):
    opt = solver.default_options()
    opt.linear_solver = ls
    return opt


def _both_routes(net, max_cluster, iters=3, ls=capi.SOLVER_PCG_SCHUR):
    with solver.Problem(net) as P:
        opt = _options(ls)
        opt.max_num_iterations = iters
        P.set_parameters()
        P.solve_resident(opt)
        P.covariance_route(1)
        dense = P.covariance()
        assert P.covariance_stats()["route"] == 1
        P.covariance_route(2, max_cluster)
        sel = P.covariance()
        st = P.covariance_stats()
        assert st["route"] == 2
    return dense, sel, st


def _assert_same(got, want, tol):
    for g, w, name in zip(got, want, ("eop", "xyz", "iop", "ap")):
        assert g.shape == w.shape
        assert np.array_equal(g == 0.0, w == 0.0), name
        nz = w != 0.0
        if nz.any():
            err = np.abs(g[nz] - w[nz]) / w[nz]
            assert np.all(err <= tol), (name, float(err.max()))


@pytest.mark.parametrize("case", ["one_camera", "camera_constant", "four_cameras", "ragged"])
def test_dissection_equals_dense_inverse(case):
    # a few hundred stations cut into clusters of at most 40 (so that several levels of cuts happen, clusters and the
    # boundary are smaller than the blocked factorisation's minimum order and get identity rows)
    if case == "one_camera":
        net = host.synth(n_cam=1, n_sta=400, n_pts=4000, rays_per_point=5, seed=2, ap_free_mask=host.AP_K1K2P1P2).net
    elif case == "camera_constant":
        net = host.synth(n_cam=1, n_sta=300, n_pts=3000, rays_per_point=4, seed=3, ap_free_mask=0, iop_free_mask=0).net
    elif case == "four_cameras":
        net = host.synth(n_cam=4, n_sta=350, n_pts=5000, rays_per_point=6, seed=4, ap_free_mask=host.AP_K1K2P1P2).net
    else:
        # 331 stations: the last row of the grid is incomplete and the clusters have different sizes
        net = host.synth(n_cam=2, n_sta=331, n_pts=3500, rays_per_point=5, seed=5, ap_free_mask=host.AP_K1K2P1P2, outlier_fraction=0.05).net
    dense, sel, st = _both_routes(net, 40)
    assert st["clusters"] >= 4 and st["boundary_order"] > 0 and st["largest_cluster_order"] >= 256
    _assert_same(sel, dense, 1e-8)


def test_dissection_against_the_oracle():
    from oracle import py_oracle as po
    net = host.synth(n_cam=1, n_sta=130, n_pts=700, rays_per_point=6, seed=3, control_every=6).net
    with solver.Problem(net) as P:
        g = P.solve(_options(capi.SOLVER_DENSE_SCHUR))
        P.set_parameters(*g[:4])
        P.covariance_route(2, 16)
        got = P.covariance()
        assert P.covariance_stats()["clusters"] >= 4
    want = po.covariance(net, *g[:4])
    _assert_same(got, want, 1e-8)


def test_one_cluster_and_unused_stations():
    # cluster size above the network size: one cluster, an empty boundary but for the camera columns; stations nobody
    # observes stay out and get zeros
    net = host.synth(n_cam=1, n_sta=120, n_pts=1500, rays_per_point=5, seed=6, ap_free_mask=host.AP_K1K2P1P2).net
    import dataclasses
    keep = (net.obs_sta != 17) & (net.obs_sta != 93)
    kw = {k: getattr(net, k)[keep] for k in ("obs_pt", "obs_sta", "obs_x", "obs_y", "obs_sx", "obs_sy", "obs_cx", "obs_cy")}
    net = dataclasses.replace(net, **kw)
    dense, sel, st = _both_routes(net, 4000)
    assert st["clusters"] == 1
    assert np.all(sel[0][[17, 93]] == 0.0)
    _assert_same(sel, dense, 1e-8)


def test_clusters_with_unused_stations():
    import dataclasses
    net = host.synth(n_cam=1, n_sta=260, n_pts=3000, rays_per_point=5, seed=8, ap_free_mask=host.AP_K1K2P1P2).net
    keep = ~np.isin(net.obs_sta, [5, 77, 201])
    kw = {k: getattr(net, k)[keep] for k in ("obs_pt", "obs_sta", "obs_x", "obs_y", "obs_sx", "obs_sy", "obs_cx", "obs_cy")}
    net = dataclasses.replace(net, **kw)
    dense, sel, st = _both_routes(net, 30)
    assert st["clusters"] >= 4
    assert np.all(sel[0][[5, 77, 201]] == 0.0)
    _assert_same(sel, dense, 1e-8)


def test_no_separators_is_refused_with_a_message():
    # every point is seen from every station: the station graph is complete
    net = host.synth(n_cam=1, n_sta=30, n_pts=300, rays_per_point=30, seed=1).net
    with solver.Problem(net) as P:
        P.set_parameters()
        P.covariance_route(2, 8)
        with pytest.raises(solver.BaError) as e:
            P.covariance()
        assert e.value.code == capi.ERR_UNSUPPORTED and "separator" in str(e.value)
        P.covariance_route(0)
        ve, vx, vi, va = P.covariance()  # the automatic choice takes the dense inverse here
        assert P.covariance_stats()["route"] == 1 and np.all(ve > 0.0)


def test_route_arguments_are_checked():
    net = host.synth_config("cfg1").net
    with solver.Problem(net) as P:
        for bad in ((3, 0), (-1, 0), (2, -5)):
            with pytest.raises(solver.BaError) as e:
                P.covariance_route(*bad)
            assert e.value.code == capi.ERR_INVALID
