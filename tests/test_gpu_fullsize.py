"""Full-size runs (the 10M-observation network the benchmark is quoted on): the oracle is too
slow here, so the checks are size-independent properties of the path."""
import numpy as np
import pytest

from bundleadjustment_amd import capi, host, solver
from util import RTOL, rel

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def big():
    # noise-free observations: the truth is the exact minimiser, cost there = priors only
    hn = host.synth_config("cfg4", noise_sigma=0.5, outlier_fraction=0.0)
    return hn.net


def test_cost_at_truth_and_descent(big):
    net = big
    t = net.truth
    with solver.Problem(net) as P:
        c_start, _ = P.evaluate(residuals=False)
        c_truth, res = P.evaluate(t["eop"], t["xyz"], t["iop"], t["ap"], apply_loss=False)
        # at the truth the image residuals are the injected N(0,1) noise (in units of sigma)
        img = res[:2 * net.n_obs]
        assert abs(np.mean(img)) < 5e-3 and abs(np.std(img) - 1.0) < 5e-3
        # idempotence: evaluating twice gives the same bits
        c2, _ = P.evaluate(t["eop"], t["xyz"], t["iop"], t["ap"], residuals=False)
        assert c2 == c_truth
        opt = solver.default_options()
        opt.linear_solver = capi.SOLVER_PCG_SCHUR
        opt.max_num_iterations = 2
        opt.pcg_rtol = 1e-10
        P.set_parameters()
        sm, rows = P.solve_resident(opt)
        assert rows[0]["cost"] == pytest.approx(c_start, rel=1e-12)   # iteration 0 = plain evaluation
        assert all(r["step_is_successful"] for r in rows)
        assert rows[-1]["cost"] < 0.12 * rows[0]["cost"]
        # two Gauss-Newton steps from a 5 mm / 0.05 deg perturbation land close to the noise floor
        assert rows[-1]["cost"] < 1.2 * c_truth
        eop, xyz, iop, ap = P.get_parameters()
        assert np.sqrt(np.mean((xyz - t["xyz"]) ** 2)) < np.sqrt(np.mean((net.xyz - t["xyz"]) ** 2))


def test_sparsity_counts_full_size(big):
    net = big
    with solver.Problem(net) as P:
        nr, nc, rows, cols = P.sparsity()
    N, S, Pn = net.n_obs, net.n_sta, net.n_pts
    assert nr == 2 * N + 3 * Pn + 3 and nc == 6 * S + 3 * Pn + 3 + 4
    assert rows[-1] == 2 * N * (6 + 3 + 3 + 4) + 9 * Pn + 9
    # checksum of checksums: every image row references its own station and point columns
    first = cols[rows[:2 * N:2]]
    assert np.array_equal(first, 6 * net.obs_sta)


# ------------------------------------------------------------------------------------------------
# the other networks of the benchmark definition at their full size
# ------------------------------------------------------------------------------------------------
@pytest.mark.timeout(600)
def test_cfg2_full_size_solve_parity():
    # 4 cameras, 200 stations, 20 000 points, 1 000 000 observations, dense Cholesky of order 1248: every iteration's
    # cost and the final parameters against the oracle (about half a minute of one CPU core)
    from oracle import py_oracle as po
    from util import assert_solve_parity
    net = host.synth_config("cfg2").net
    assert net.n_obs == 1000000 and net.n_sta == 200 and net.n_cam == 4
    opt = solver.default_options()
    with solver.Problem(net) as P:
        g = P.solve(opt)
    assert g[4]["linear_solver_used"] == capi.SOLVER_DENSE_SCHUR
    o = po.solve(net, opt, po.SCHUR_DENSE)
    assert o[4]["termination"] == capi.CONVERGENCE
    assert_solve_parity(g, o)


@pytest.mark.timeout(900)
@pytest.mark.parametrize("name", ["cfg3", "cfg5"])
def test_full_size_properties(name):
    # cfg3: 1800 stations, 1M points, 5M observations, camera held constant.  cfg5: 50 000 stations, 10M points, 50M
    # observations (about 35 GB on the device).  The oracle is too slow here: size-independent properties instead.
    hn = host.synth_config(name, outlier_fraction=0.0)
    net, t = hn.net, hn.net.truth
    N, S, Pn = net.n_obs, net.n_sta, net.n_pts
    assert (N, S, Pn) == {"cfg3": (5000000, 1800, 1000000), "cfg5": (50000000, 50000, 10000000)}[name]
    with solver.Problem(net) as P:
        c_start, _ = P.evaluate(residuals=False)
        c_truth, res = P.evaluate(t["eop"], t["xyz"], t["iop"], t["ap"], apply_loss=False)
        img = res[:2 * N]
        assert abs(np.mean(img)) < 5e-3 and abs(np.std(img) - 1.0) < 5e-3   # the injected noise, in units of sigma
        assert res.size == 2 * N + 3 * Pn + 3
        del res, img
        c2, _ = P.evaluate(t["eop"], t["xyz"], t["iop"], t["ap"], apply_loss=False, residuals=False)
        assert c2 == c_truth                                                 # same bits twice
        opt = solver.default_options()
        opt.linear_solver = capi.SOLVER_PCG_SCHUR                           # (the default takes the sparse direct solver at cfg3's size)
        opt.max_num_iterations = 2
        opt.pcg_rtol = 1e-10
        P.set_parameters()
        sm, rows = P.solve_resident(opt)
        assert sm["linear_solver_used"] == capi.SOLVER_PCG_SCHUR
        assert rows[0]["cost"] == pytest.approx(c_start, rel=1e-12)
        assert all(r["step_is_successful"] for r in rows)
        assert rows[-1]["cost"] < 0.12 * rows[0]["cost"] and rows[-1]["cost"] < 1.2 * c_truth
        # the multilevel preconditioner keeps the iteration count flat in the network size
        assert all(0 < r["linear_iterations"] < 400 for r in rows[1:]), [r["linear_iterations"] for r in rows]
        free = 0 if name == "cfg3" else 7
        assert sm["num_effective_parameters_reduced"] == 6 * S + 3 * Pn + free
        eop, xyz, iop, ap = P.get_parameters()
        assert np.sqrt(np.mean((xyz - t["xyz"]) ** 2)) < np.sqrt(np.mean((net.xyz - t["xyz"]) ** 2))
        if name == "cfg5":
            # variances at this size: the dense inverse of the reduced system (order 300 007, two matrices of 720 GB) does
            # not fit, so ba_covariance cuts the station graph into clusters and inverts those and the boundary system;
            # the camera parameters' variances also come from one PCG solve per camera column, to compare
            import time
            t0 = time.time()
            ve, vx, di, da = P.covariance()
            st = P.covariance_stats()
            print(f"cfg5 variances by dissection: {time.time() - t0:.1f} s, {st}")
            assert st["route"] == 2 and st["clusters"] >= 32 and st["largest_cluster_order"] <= 6 * 1536
            vi, va = P.covariance_cameras()
            k = [2, 3, 5, 6]  # k1 k2 p1 p2
            assert np.all(np.isfinite(vi)) and np.all(vi > 0.0) and np.all(va[0, k] > 0.0) and np.all(np.delete(va[0], k) == 0.0)
            assert np.all(np.abs(di - vi) <= 1e-6 * vi) and np.all(np.abs(da[0, k] - va[0, k]) <= 1e-6 * va[0, k])
            assert np.all(np.delete(da[0], k) == 0.0)
            # five times the observations of the 10M network, same geometry: the camera is known better, not worse
            assert np.all(np.sqrt(vi) < 1e-1)
            assert ve.shape == (S, 6) and np.all(np.isfinite(ve)) and np.all(ve > 0.0)
            assert vx.shape == (Pn, 3) and np.all(np.isfinite(vx)) and np.all(vx > 0.0)
            assert np.all(vx <= net.pt_sigma ** 2 * (1.0 + 1e-9))       # an observation can only add information
            # stations in the middle of the block are held by neighbours on every side: better known than the median edge station
            gx = int(np.ceil(np.sqrt(S)))
            ix, iy = np.arange(S) % gx, np.arange(S) // gx
            edge = (ix < 2) | (iy < 2) | (ix >= gx - 2) | (iy >= (S - 1) // gx - 1)
            assert np.median(ve[~edge, 3:].sum(axis=1)) < np.median(ve[edge, 3:].sum(axis=1))
        if name == "cfg3":
            assert np.array_equal(iop, net.iop) and np.array_equal(ap, net.ap)  # constant blocks stay
            nr, nc, rws, cols = P.sparsity()
            assert nr == 2 * N + 3 * Pn + 3 and nc == 6 * S + 3 * Pn + 3 + 16
            assert rws[-1] == 2 * N * (6 + 3) + 9 * Pn


@pytest.mark.timeout(900)
@pytest.mark.parametrize("name", ["cfg3", "cfg4"])
def test_full_size_dissection_equals_dense_inverse(name):
    # 1800 stations (reduced order 10 800) and 10 000 stations (order 60 007): both fit the dense inverse, so the
    # dissection route can be held against it entry by entry: all station, camera and point variances to 1e-8
    import time
    hn = host.synth_config(name, outlier_fraction=0.0)
    net = hn.net
    with solver.Problem(net) as P:
        opt = solver.default_options()
        opt.max_num_iterations = 3
        P.set_parameters()
        P.solve_resident(opt)
        P.covariance_route(1)
        t0 = time.time()
        dense = P.covariance()
        t1 = time.time()
        P.covariance_route(2)
        sel = P.covariance()
        t2 = time.time()
        st = P.covariance_stats()
    print(f"{name}: dense inverse {t1 - t0:.1f} s, dissection {t2 - t1:.1f} s, {st}")
    assert st["route"] == 2 and st["clusters"] >= 2
    for g, w, what in zip(sel, dense, ("eop", "xyz", "iop", "ap")):
        assert np.array_equal(g == 0.0, w == 0.0), what
        nz = w != 0.0
        if nz.any():
            err = np.abs(g[nz] - w[nz]) / w[nz]
            assert np.all(err <= 1e-8), (what, float(err.max()))


@pytest.mark.timeout(600)
def test_cfg3_full_size_variances():
    # 1800 stations (reduced system of order 10 800, inverted densely on the matrix cores), one million points: the
    # oracle's full inverse is out of reach, so properties.  Every variance is positive; an observation can only add
    # information, so no point ends up less certain than its prior; the control points (the tight priors) end up better
    # known on average than the tie points; the constant camera has no variance.
    hn = host.synth_config("cfg3", outlier_fraction=0.0)
    net = hn.net
    with solver.Problem(net) as P:
        opt = solver.default_options()
        opt.max_num_iterations = 3
        P.set_parameters()
        P.solve_resident(opt)
        ve, vx, vi, va = P.covariance()
    assert ve.shape == (net.n_sta, 6) and vx.shape == (net.n_pts, 3)
    assert np.all(np.isfinite(ve)) and np.all(ve > 0.0)
    assert np.all(np.isfinite(vx)) and np.all(vx > 0.0)
    assert np.all(vx <= net.pt_sigma ** 2 * (1.0 + 1e-9))
    assert np.all(vi == 0.0) and np.all(va == 0.0)
    tight = net.pt_sigma[:, 0] < np.median(net.pt_sigma[:, 0])
    if tight.any() and (~tight).any():
        assert np.mean(vx[tight]) < np.mean(vx[~tight])
    # the same call again gives the same bits
    with solver.Problem(net) as P:
        P.set_parameters()
        P.solve_resident(opt)
        ve2, vx2, _, _ = P.covariance()
    assert np.array_equal(ve, ve2) and np.array_equal(vx, vx2)


@pytest.mark.timeout(600)
def test_cfg4_full_size_variances(big):
    # 10 000 stations: reduced system of order 60 007, two dense matrices of 28.8 GB each, factored and inverted on the
    # matrix cores (a few seconds).  Properties as for the 5M-observation network, plus one independent route: the camera
    # parameters' variances from one PCG solve per camera column (ba_covariance_cameras) against the dense inverse.
    net = big
    with solver.Problem(net) as P:
        opt = solver.default_options()
        opt.max_num_iterations = 3
        P.set_parameters()
        P.solve_resident(opt)
        ve, vx, vi, va = P.covariance()
        wi, wa = P.covariance_cameras()
        import time
        t0 = time.time()
        par, ce, cx, cc = P.covariance_camera_rows()
        print(f"camera rows of the 10M network: {time.time() - t0:.2f} s")
    assert ve.shape == (net.n_sta, 6) and vx.shape == (net.n_pts, 3)
    assert np.all(np.isfinite(ve)) and np.all(ve > 0.0)
    assert np.all(np.isfinite(vx)) and np.all(vx > 0.0)
    assert np.all(vx <= net.pt_sigma ** 2 * (1.0 + 1e-9))
    free = [2, 3, 5, 6]  # k1 k2 p1 p2
    assert np.all(vi > 0.0) and np.all(va[0, free] > 0.0) and np.all(np.delete(va[0], free) == 0.0)
    np.testing.assert_allclose(wi, vi, rtol=1e-6)
    np.testing.assert_allclose(wa, va, rtol=1e-6)
    # The camera rows of Cx (one sparse direct factorisation, a pair of tree solves and one back substitution per row):
    # their corner carries the same variances, and every entry is a covariance -- |correlation| <= 1 against the
    # variances of the other route, for all 60 000 station parameters and 6 million point coordinates.
    assert list(par) == [0, 1, 2] + [3 + k for k in free]
    v_cam = np.concatenate([vi[0], va[0, free]])
    np.testing.assert_allclose(np.diag(cc), v_cam, rtol=1e-8)
    for k in range(len(par)):
        assert np.max(np.abs(ce[k]) / np.sqrt(ve * v_cam[k])) <= 1.0 + 1e-9
        assert np.max(np.abs(cx[k]) / np.sqrt(vx * v_cam[k])) <= 1.0 + 1e-9
    tab = host.correlations((ve, vx, vi, va), (par, ce, cx, cc))
    assert np.all(np.isfinite(tab["eop_iop"])) and np.all(tab["eop_iop"][0] >= tab["eop_iop"][1]) and np.all(tab["xyz_iop"][0] <= 1.0)


@pytest.mark.timeout(900)
@pytest.mark.parametrize("name", ["cfg3", "cfg4"])
def test_full_size_pcg_step_equals_dense_cholesky_step(name, big):
    # One LM step of the whole network (cfg3: reduced system of order 10 800; cfg4: order 60 007, the benchmark's) taken
    # twice on the GPU: with the PCG on the block-sparse system -- three-level hierarchy, structures built on the device,
    # batches of iterations replayed as graphs -- and with the dense Cholesky factorisation on the matrix cores, which
    # shares none of that.  Same candidate cost and same parameters after the step to 1e-9.
    net = big if name == "cfg4" else host.synth_config("cfg3", outlier_fraction=0.0).net
    out = {}
    for ls in (capi.SOLVER_PCG_SCHUR, capi.SOLVER_DENSE_SCHUR, capi.SOLVER_SPARSE_SCHUR):
        with solver.Problem(net) as P:
            opt = solver.default_options()
            opt.linear_solver = ls
            opt.max_num_iterations = 1
            opt.pcg_rtol = 1e-12
            opt.pcg_max_iters = 20000
            opt.function_tolerance = opt.parameter_tolerance = opt.gradient_tolerance = 0.0
            P.set_parameters()
            sm, rows = P.solve_resident(opt)
            assert sm["linear_solver_used"] == ls and len(rows) == 2 and rows[1]["step_is_successful"]
            out[ls] = (rows, P.get_parameters())
    (rs, ps), (rd, pd) = out[capi.SOLVER_SPARSE_SCHUR], out[capi.SOLVER_DENSE_SCHUR]
    assert rs[0]["cost"] == rd[0]["cost"]
    assert abs(rs[1]["cost"] - rd[1]["cost"]) <= RTOL * rd[1]["cost"], (rs[1]["cost"], rd[1]["cost"])
    for a, b, what in zip(ps, pd, ("eop", "xyz", "iop", "ap")):
        assert rel(a, b) < RTOL, ("sparse", what, rel(a, b))
    (rp, pp), (rd, pd) = out[capi.SOLVER_PCG_SCHUR], out[capi.SOLVER_DENSE_SCHUR]
    assert rp[0]["cost"] == rd[0]["cost"]
    assert abs(rp[1]["cost"] - rd[1]["cost"]) <= RTOL * rd[1]["cost"], (rp[1]["cost"], rd[1]["cost"])
    assert abs(rp[1]["step_norm"] - rd[1]["step_norm"]) <= 1e-9 * rd[1]["step_norm"]
    for nm, a, b in zip(("eop", "xyz", "iop", "ap"), pp, pd):
        assert rel(a, b) < RTOL, (nm, rel(a, b))
    # and the step is a real one: the cost falls by an order of magnitude
    assert rd[1]["cost"] < 0.12 * rd[0]["cost"]
