"""Hand-over between adjustments of the outer loop (src/main.cpp:3040-3056, :7544-7613): new image corrections go
into the problem that is resident on the GPU instead of through temp.pho and a rebuilt ceres::Problem."""
import numpy as np
import pytest

from bundleadjustment_amd import capi, host, solver
from oracle import py_oracle as po

from util import assert_solve_parity, modified

pytestmark = pytest.mark.gpu


def test_set_corrections_equals_a_rebuilt_problem():
    net = host.synth(n_cam=2, n_sta=10, n_pts=120, rays_per_point=5, seed=4).net
    rng = np.random.default_rng(1)
    cx, cy = rng.normal(0, 0.4, net.n_obs), rng.normal(0, 0.4, net.n_obs)
    opt = solver.default_options()
    with solver.Problem(net) as P:          # built with zero corrections, then updated in place
        P.set_corrections(cx, cy)
        c1, r1 = P.evaluate()
        g1 = P.solve(opt)
    net2 = modified(net, obs_cx=cx, obs_cy=cy)
    with solver.Problem(net2) as Q:         # built with the corrections
        c2, r2 = Q.evaluate()
        g2 = Q.solve(opt)
    assert c1 == c2 and np.array_equal(r1, r2)
    for a, b in zip(g1[:4], g2[:4]):
        assert np.array_equal(a, b)
    assert g1[4]["final_cost"] == g2[4]["final_cost"] and g1[4]["num_iterations"] == g2[4]["num_iterations"]
    assert_solve_parity(g1, po.solve(net2, opt, po.SCHUR_DENSE))
    with solver.Problem(net) as P:
        with pytest.raises(solver.BaError, match="not finite"):
            P.set_corrections(np.full(net.n_obs, np.nan), cy)


def test_outer_loop_with_a_regressor_removes_an_unmodelled_image_error():
    # the image points carry a smooth systematic error that no parameter of the adjustment can take up: a k-nearest-
    # neighbour regression of the residuals over the image plane, fed back as corrections (the scheme of the
    # reference's outer loop, MLMODE 1), takes it out; the problem on the GPU is never rebuilt
    neighbors = pytest.importorskip("sklearn.neighbors")
    hn = host.synth(n_cam=1, n_sta=16, n_pts=600, rays_per_point=8, seed=12, ap_free_mask=host.AP_K1K2P1P2, outlier_fraction=0.0)
    x, y = hn.net.obs_x, hn.net.obs_y
    net = modified(hn.net, obs_x=x + 2.0 * np.sin(y / 200.0), obs_y=y + 2.0 * np.cos(x / 200.0))
    N = net.n_obs
    opt = solver.default_options()
    cx, cy = np.zeros(N), np.zeros(N)
    rmse = []
    uv = np.column_stack([net.obs_x, net.obs_y])
    with solver.Problem(net) as P:
        P.set_parameters()
        for it in range(5):
            P.set_corrections(cx, cy)
            sm, _ = P.solve_resident(opt)
            assert sm["termination"] == capi.CONVERGENCE
            eop, xyz, iop, ap = P.get_parameters()
            _, res = P.evaluate(eop, xyz, iop, ap, apply_loss=True)
            vx, vy = res[0:2 * N:2] * net.obs_sx, res[1:2 * N:2] * net.obs_sy
            rmse.append(float(np.sqrt(np.mean(vx ** 2 + vy ** 2) / 2)))
            pred = neighbors.KNeighborsRegressor(n_neighbors=30).fit(uv, np.column_stack([vx, vy])).predict(uv)
            cx, cy = cx + pred[:, 0], cy + pred[:, 1]      # v = computed - corr - observed: corrections accumulate
    # loss-weighted residuals, as the reference's image.jck holds them: 0.63 with the systematic error in, towards
    # the 0.35 that the noise of the points (sigma 0.5) leaves in this network once it is out
    assert rmse[0] > 0.55 and rmse[-1] < 0.45, rmse
    assert all(b < a for a, b in zip(rmse, rmse[1:])), rmse
