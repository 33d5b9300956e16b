"""Parity of the HIP path (through the C ABI) with the CPU oracle on the same inputs.
Tolerances: index/sparsity data bit-exact; floating point 1e-9 relative (stated in util.RTOL)
on per-iteration cost, final parameters and sigma0; the per-observation kernel is held to
1e-11 (it has no accumulated sums)."""
import numpy as np
import pytest

from bundleadjustment_amd import capi, host, solver
from oracle import py_oracle as po

from util import RTOL, assert_solve_parity, modified, rel

pytestmark = pytest.mark.gpu


def gpu_solve(net, opt):
    with solver.Problem(net) as P:
        return P.solve(opt)


def options(linear_solver, **kw):
    o = solver.default_options()
    o.linear_solver = linear_solver
    for k, v in kw.items():
        setattr(o, k, v)
    return o


# ------------------------------------------------------------------------------------------------
# per-observation kernel
# ------------------------------------------------------------------------------------------------
def golden_network(golden):
    """Every golden case as its own station, point and camera (all parameters differ per observation)."""
    cs = golden["collinearity"]
    n = len(cs)
    obs = np.array([c["obs"] for c in cs])
    return capi.Network(
        obs_pt=np.arange(n), obs_sta=np.arange(n), sta_cam=np.arange(n),
        obs_x=obs[:, 0], obs_y=obs[:, 1], obs_sx=obs[:, 2], obs_sy=obs[:, 3], obs_cx=obs[:, 4], obs_cy=obs[:, 5],
        pt_prior=np.array([c["xyz"] for c in cs]), pt_sigma=np.full((n, 3), 10.0),
        iop_prior=np.array([c["iop"] for c in cs]), iop_sigma=np.tile([10.0, 10.0, 1000.0], (n, 1)),
        ap_free_mask=np.full(n, 0xFFFF), iop_free_mask=np.full(n, 7),
        eop=np.array([c["eop"] for c in cs]), xyz=np.array([c["xyz"] for c in cs]),
        iop=np.array([c["iop"] for c in cs]), ap=np.array([c["ap"] for c in cs]))


def test_residual_jacobian_kernel_vs_golden(golden):
    # golden = the reference's functors (src/main.cpp:2033-2120) and their complex-step derivative
    net = golden_network(golden)
    with solver.Problem(net) as P:
        P.set_parameters()
        r, J = P.residual_jacobian()
    rg = np.array([c["r_mls"] for c in golden["collinearity"]]).reshape(-1)
    Jg = np.array([c["J"] for c in golden["collinearity"]]).reshape(-1, 2, 28)
    # residuals are differences of numbers ~1e3: absolute 1e-11 in units of sigma
    assert np.max(np.abs(r - rg)) < 1e-10
    scale = np.maximum(np.abs(Jg), 1e-9 * np.max(np.abs(Jg), axis=(1, 2), keepdims=True))
    assert np.max(np.abs(J - Jg) / scale) < 1e-9
    assert np.array_equal(J[:, 1, 12:14], np.zeros((len(r) // 2, 2)))  # a1, a2 do not enter y (:2090)


def test_residual_jacobian_kernel_vs_oracle():
    net = host.synth_config("cfg2", scale=0.02).net
    with solver.Problem(net) as P:
        P.set_parameters()
        r, J = P.residual_jacobian()
    ro, Jo = po.residual_jacobian_all(net)
    assert np.max(np.abs(r - ro)) < 1e-10
    scale = np.maximum(np.abs(Jo), 1e-9 * np.max(np.abs(Jo), axis=(1, 2), keepdims=True))
    assert np.max(np.abs(J - Jo) / scale) < 1e-10


@pytest.mark.parametrize("apply_loss", [True, False])
def test_evaluate_cost_and_residual_order(apply_loss):
    # problem.Evaluate, src/main.cpp:4853: rows = image (file order) | XYZ priors | IOP priors
    net = host.synth(n_cam=2, n_sta=9, n_pts=50, rays_per_point=5, seed=9, outlier_fraction=0.3).net
    with solver.Problem(net) as P:
        c, r = P.evaluate(apply_loss=apply_loss)
    co, ro = po.evaluate(net, apply_loss=apply_loss)
    assert abs(c - co) <= 1e-13 * co
    assert np.max(np.abs(r - ro)) < 1e-10


# ------------------------------------------------------------------------------------------------
# sparsity pattern / index maps: bit-exact
# ------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("masks", [(host.AP_REFERENCE_DEFAULT, 7), (host.AP_ALL16, 7), (0, 0), (host.AP_BROWN7, 5), (0, 7)])
def test_sparsity_bit_exact(masks):
    net = host.synth(n_cam=3, n_sta=7, n_pts=30, rays_per_point=4, seed=2, ap_free_mask=masks[0], iop_free_mask=masks[1]).net
    with solver.Problem(net) as P:
        nr, nc, rows, cols = P.sparsity()
    onr, onc, orows, ocols = po.sparsity(net)
    assert (nr, nc) == (onr, onc)
    assert np.array_equal(rows, orows) and np.array_equal(cols, ocols)


# ------------------------------------------------------------------------------------------------
# full solves
# ------------------------------------------------------------------------------------------------
ALL_SOLVERS = [capi.SOLVER_DENSE_SCHUR, capi.SOLVER_PCG_SCHUR, capi.SOLVER_PCG_IMPLICIT]


@pytest.mark.parametrize("ls", ALL_SOLVERS)
def test_cfg1_solve_parity_and_sigma0(ls):
    # configs[0]: the reference's CPU-runnable case, full size; oracle in its most literal mode
    hn = host.synth_config("cfg1")
    net = hn.net
    opt = options(ls)
    g = gpu_solve(net, opt)
    o = po.solve(net, opt, po.FULL_DENSE)
    assert o[4]["termination"] == capi.CONVERGENCE
    assert_solve_parity(g, o)
    sg, so = capi.BaSummary(), capi.BaSummary()
    for s, d in ((sg, g[4]), (so, o[4])):
        for k, v in d.items():
            if hasattr(s, k):
                setattr(s, k, v)
    s0g, redg = host.sigma0(sg, net.n_pts, hn.num_ap_correction())
    s0o, redo = host.sigma0(so, net.n_pts, hn.num_ap_correction())
    assert redg == redo and abs(s0g - s0o) <= RTOL * s0o
    # the adjustment recovers the truth: better than the start values
    assert np.sqrt(np.mean((g[1] - net.truth["xyz"]) ** 2)) < 0.6 * np.sqrt(np.mean((net.xyz - net.truth["xyz"]) ** 2))


@pytest.mark.parametrize("name,scale", [("cfg2", 0.05), ("cfg3", 0.01), ("cfg4", 0.005)])
@pytest.mark.parametrize("ls", ALL_SOLVERS)
def test_scaled_configs_solve_parity(name, scale, ls):
    net = host.synth_config(name, scale=scale).net
    opt = options(ls)
    g = gpu_solve(net, opt)
    o = po.solve(net, opt, po.SCHUR_DENSE)
    assert o[4]["termination"] == capi.CONVERGENCE
    assert_solve_parity(g, o)


def test_default_options_pick_the_solver_by_size():
    # the reference fixes its linear solver (src/main.cpp:4706), so its callers never choose one: the default
    # options must work at any size.  Dense up to capi.AUTO_DENSE_ORDER reduced unknowns, PCG above.
    small = host.synth_config("cfg2", scale=0.05).net
    assert 6 * small.n_sta + 10 <= capi.AUTO_DENSE_ORDER
    opt = solver.default_options()
    assert opt.linear_solver == capi.SOLVER_AUTO
    g = gpu_solve(small, opt)
    assert g[4]["linear_solver_used"] == capi.SOLVER_DENSE_SCHUR and g[4]["total_linear_iterations"] == 0
    assert_solve_parity(g, po.solve(small, opt, po.SCHUR_DENSE))
    large = host.synth(n_cam=1, n_sta=700, n_pts=20000, rays_per_point=6, seed=3).net
    assert 6 * large.n_sta > capi.AUTO_DENSE_ORDER
    g = gpu_solve(large, opt)
    assert g[4]["linear_solver_used"] == capi.SOLVER_SPARSE_SCHUR and g[4]["total_linear_iterations"] == 0
    assert g[4]["termination"] == capi.CONVERGENCE
    explicit = gpu_solve(large, options(capi.SOLVER_SPARSE_SCHUR))
    assert g[4]["final_cost"] == explicit[4]["final_cost"]
    pcg = gpu_solve(large, options(capi.SOLVER_PCG_SCHUR))
    assert pcg[4]["total_linear_iterations"] > 0 and abs(pcg[4]["final_cost"] - g[4]["final_cost"]) <= 1e-9 * g[4]["final_cost"]
    huge = host.synth(n_cam=1, n_sta=42000, n_pts=200000, rays_per_point=4, seed=3).net
    assert 6 * huge.n_sta > capi.AUTO_SPARSE_ORDER
    opt2 = solver.default_options()
    opt2.max_num_iterations = 2
    g = gpu_solve(huge, opt2)
    assert g[4]["linear_solver_used"] == capi.SOLVER_PCG_SCHUR and g[4]["total_linear_iterations"] > 0


def test_multilevel_preconditioner_same_answer_fewer_iterations():
    # 300 stations: three levels of the aggregation hierarchy; the preconditioner must not change what
    # the PCG converges to, only how fast
    net = host.synth_config("cfg4", scale=0.03).net
    opt_ml = options(capi.SOLVER_PCG_SCHUR, preconditioner=0)
    opt_bj = options(capi.SOLVER_PCG_SCHUR, preconditioner=1)
    g_ml = gpu_solve(net, opt_ml)
    g_bj = gpu_solve(net, opt_bj)
    o = po.solve(net, opt_ml, po.SCHUR_DENSE)
    assert o[4]["termination"] == capi.CONVERGENCE
    assert_solve_parity(g_ml, o)
    assert_solve_parity(g_bj, o)
    assert g_ml[4]["total_linear_iterations"] < 0.6 * g_bj[4]["total_linear_iterations"]


def test_multilevel_preconditioner_several_cameras():
    # more than one camera: the camera border comes from the general (probing) path
    net = host.synth(n_cam=3, n_sta=90, n_pts=6000, rays_per_point=6, ap_free_mask=host.AP_K1K2P1P2, seed=5).net
    opt = options(capi.SOLVER_PCG_SCHUR)
    g = gpu_solve(net, opt)
    o = po.solve(net, opt, po.SCHUR_DENSE)
    assert o[4]["termination"] == capi.CONVERGENCE
    assert_solve_parity(g, o)


def test_long_rows_of_the_reduced_matrix():
    # points seen from 40 stations each: station rows of the block-sparse reduced matrix with more than 64 blocks -- the
    # product kernel of the PCG keeps a row's first 64 column indices in registers and 21 blocks in flight per pass
    net = host.synth(n_cam=1, n_sta=100, n_pts=600, rays_per_point=40, ap_free_mask=host.AP_K1K2P1P2, seed=11).net
    pt, st = np.asarray(net.obs_pt), np.asarray(net.obs_sta)
    seen = np.zeros((net.n_sta, net.n_sta), dtype=bool)
    for p_ in range(net.n_pts):
        ss = st[pt == p_]
        seen[np.ix_(ss, ss)] = True
    assert seen.sum(axis=1).max() > 64
    o = po.solve(net, options(capi.SOLVER_PCG_SCHUR), po.SCHUR_DENSE)
    assert o[4]["termination"] == capi.CONVERGENCE
    for precond in (0, 1):
        assert_solve_parity(gpu_solve(net, options(capi.SOLVER_PCG_SCHUR, preconditioner=precond)), o)


@pytest.mark.parametrize("which", ["cfg1", "three_cameras", "fixed_camera"])
def test_covariance_after_the_solve(which):
    # a-posteriori variances (src/main.cpp:5261-5279, 5443-5565) at the adjusted parameters: the GPU's
    # Schur-complement route against the oracle's inverse of the full normal matrix.  1e-8: the inverse
    # amplifies rounding by the condition number of the network -- the oracle and an independent numpy inverse of
    # the same matrix differ by up to 1.5e-10 on these networks (tests/test_oracle_covariance.py), the GPU and the
    # oracle by up to 4e-10 (scripts/cov_gap.py).
    if which == "cfg1":
        net = host.synth_config("cfg1").net
    elif which == "three_cameras":
        net = host.synth(n_cam=3, n_sta=12, n_pts=150, rays_per_point=6, ap_free_mask=host.AP_K1K2P1P2, seed=5).net
    else:
        # every 8th point is a control point: with two of 80 the datum is so weak (station sigmas of kilometres)
        # that the comparison measures the conditioning of the network, not the code
        net = host.synth(n_cam=1, n_sta=9, n_pts=80, rays_per_point=5, ap_free_mask=0, iop_free_mask=0, seed=7,
                         outlier_fraction=0.1, control_every=8).net
    with solver.Problem(net) as P:
        P.set_parameters()
        P.solve_resident(options(capi.SOLVER_DENSE_SCHUR))
        eop, xyz, iop, ap = P.get_parameters()
        got = P.covariance()
    want = po.covariance(net, eop, xyz, iop, ap)
    for g, w, name in zip(got, want, ("eop", "xyz", "iop", "ap")):
        assert g.shape == w.shape
        assert np.array_equal(g == 0.0, w == 0.0), name
        nz = w != 0.0
        assert np.all(np.abs(g[nz] - w[nz]) <= 1e-8 * w[nz]), (name, np.max(np.abs(g[nz] - w[nz]) / w[nz]))


@pytest.mark.parametrize("which", ["cfg1", "three_cameras"])
def test_camera_variances_by_iterative_solves_equal_the_dense_inverse(which):
    # ba_covariance_cameras (one PCG solve per camera column) against ba_covariance (dense inverse); 1e-6: the PCG
    # stops at a relative residual of 1e-12 and the condition number of these networks takes the rest
    if which == "cfg1":
        net = host.synth_config("cfg1").net
    else:
        net = host.synth(n_cam=3, n_sta=12, n_pts=150, rays_per_point=6, ap_free_mask=host.AP_K1K2P1P2, seed=5).net
    with solver.Problem(net) as P:
        P.set_parameters()
        P.solve_resident(options(capi.SOLVER_DENSE_SCHUR))
        _, _, vi, va = P.covariance()
        wi, wa = P.covariance_cameras()
        eop, xyz, iop, ap = P.get_parameters()
        assert np.array_equal(np.stack(P.get_parameters()[2]), iop)  # the parameters are left alone
    for g, w in ((wi, vi), (wa, va)):
        assert np.array_equal(g == 0.0, w == 0.0)
        nz = w != 0.0
        assert np.all(np.abs(g[nz] - w[nz]) <= 1e-6 * w[nz]), np.max(np.abs(g[nz] - w[nz]) / w[nz])


def test_variances_of_a_network_beyond_the_old_dense_limit():
    # 8407 reduced unknowns: blocked Cholesky and inverse on the matrix cores.  The dense inverse and the iterative
    # route (one PCG solve per camera column) must agree on the camera parameters.
    net = host.synth(n_cam=1, n_sta=1400, n_pts=6000, rays_per_point=4, seed=1, ap_free_mask=host.AP_K1K2P1P2).net
    with solver.Problem(net) as P:
        P.set_parameters()
        P.solve_resident(options(capi.SOLVER_PCG_SCHUR))
        ve, vx, di, da = P.covariance()
        vi, va = P.covariance_cameras()
    free = [2, 3, 5, 6]  # k1 k2 p1 p2
    assert np.all(vi > 0.0) and np.all(va[0, free] > 0.0)
    assert np.all(np.delete(va[0], free) == 0.0)
    # the posterior is tighter than the prior (sigma 10, 10, 1000 on xp, yp, c: src/main.cpp:4612-4623)
    assert np.all(np.sqrt(vi[0]) < np.array([10.0, 10.0, 1000.0]))
    assert np.all(np.abs(di - vi) <= 1e-6 * vi) and np.all(np.abs(da[0, free] - va[0, free]) <= 1e-6 * va[0, free])
    used = np.bincount(net.obs_sta, minlength=net.n_sta) > 0
    assert np.all(ve[used] > 0.0) and np.all(ve[~used] == 0.0) and np.all(vx > 0.0)
    # a point is known no worse than its prior says
    assert np.all(vx <= net.pt_sigma ** 2 * (1 + 1e-9))


@pytest.mark.parametrize("n_sta", [60, 130])
def test_blocked_cholesky_solve_and_covariance_against_the_oracle(n_sta):
    # reduced systems of order 367 and 787: the blocked factorisation (panels of 128, a ragged last panel), its
    # triangular solves and its inverse, against the oracle's solve and its inverse of the full normal matrix
    net = host.synth(n_cam=1, n_sta=n_sta, n_pts=400 if n_sta == 60 else 700, rays_per_point=6, seed=3, control_every=6).net
    assert 6 * net.n_sta + 7 >= 256
    opt = options(capi.SOLVER_DENSE_SCHUR)
    with solver.Problem(net) as P:
        g = P.solve(opt)
        P.set_parameters(*g[:4])
        got = P.covariance()
    o = po.solve(net, opt, po.SCHUR_DENSE)
    assert o[4]["termination"] == capi.CONVERGENCE
    assert_solve_parity(g, o)
    want = po.covariance(net, *g[:4])
    for a, w, name in zip(got, want, ("eop", "xyz", "iop", "ap")):
        assert np.array_equal(a == 0.0, w == 0.0), name
        nz = w != 0.0
        assert np.all(np.abs(a[nz] - w[nz]) <= 1e-8 * w[nz]), (name, np.max(np.abs(a[nz] - w[nz]) / w[nz]))


def test_covariance_too_large_is_refused():
    # 50 000 stations: the dense inverse would need 2 x 720 GB; asked for that route only, the call says so
    net = host.synth(n_cam=1, n_sta=50000, n_pts=60000, rays_per_point=3, seed=1).net
    with solver.Problem(net) as P:
        P.set_parameters()
        P.covariance_route(1)
        with pytest.raises(solver.BaError) as e:
            P.covariance()
        assert e.value.code == capi.ERR_UNSUPPORTED


@pytest.mark.parametrize("ls", ALL_SOLVERS)
@pytest.mark.parametrize("clamp", [True, False])
def test_rejected_steps(ls, clamp):
    toy = host.synth(n_cam=1, n_sta=3, n_pts=8, rays_per_point=3, ap_free_mask=host.ap_mask("k1"), control_every=1,
                     control_sigma=10.0, outlier_fraction=0.2, seed=11).net
    bad = modified(toy, eop=toy.eop + np.array([0.8, -0.8, 1.5, 2000.0, -2000.0, 3000.0]))
    kw = dict(max_num_iterations=12)
    if not clamp:
        kw.update(min_lm_diagonal=1e-6, max_lm_diagonal=1e32)  # Ceres' own defaults: real LM damping
    opt = options(ls, **kw)
    g = gpu_solve(bad, opt)
    o = po.solve(bad, opt, po.FULL_DENSE)
    assert any(r["step_is_successful"] == 0 for r in o[5])
    # far from the solution the problem is badly conditioned: 1e-7 on the costs
    assert_solve_parity(g, o, cost_rtol=1e-7, par_rtol=1e-6)


def test_edge_cases_unused_blocks():
    # a station nobody observes from, a point nobody observes, a camera without stations' observations
    base = host.synth(n_cam=1, n_sta=6, n_pts=40, rays_per_point=4, seed=4, control_every=4).net
    S, P = base.n_sta, base.n_pts
    net = modified(
        base,
        sta_cam=np.concatenate([base.sta_cam, [1]]),           # extra station using camera 1
        eop=np.vstack([base.eop, base.eop[:1] + 1.0]),
        xyz=np.vstack([base.xyz, [[1.0, 2.0, 3.0]]]),          # extra point, prior only
        pt_prior=np.vstack([base.pt_prior, [[1.5, 2.5, 2.0]]]),
        pt_sigma=np.vstack([base.pt_sigma, [[10.0, 20.0, 30.0]]]),
        iop=np.vstack([base.iop, [[1.0, -1.0, 3100.0]]]),      # camera 1: never observed through
        iop_prior=np.vstack([base.iop_prior, [[0.0, 0.0, 3000.0]]]),
        iop_sigma=np.vstack([base.iop_sigma, [[10.0, 10.0, 1000.0]]]),
        ap=np.vstack([base.ap, np.zeros((1, 16))]),
        ap_free_mask=np.array([host.AP_BROWN7, host.AP_BROWN7]), iop_free_mask=np.array([7, 7]))
    assert net.n_sta == S + 1 and net.n_pts == P + 1 and net.n_cam == 2
    for ls in ALL_SOLVERS:
        opt = options(ls)
        g = gpu_solve(net, opt)
        o = po.solve(net, opt, po.FULL_DENSE)
        assert_solve_parity(g, o)
        assert np.array_equal(g[0][S], net.eop[S])                    # unobserved station untouched
        assert np.allclose(g[1][P], net.pt_prior[P], atol=1e-6)        # prior-only point goes to its prior
        assert np.allclose(g[2][1], net.iop_prior[1], atol=1e-6)       # prior-only camera goes to its prior
        assert np.array_equal(g[3][1], net.ap[1])                      # its AP block is not in the program
        assert g[4]["num_parameters_reduced"] == 6 * S + 3 * (P + 1) + 19 + 3


def test_duplicate_observations_of_a_point_from_one_station():
    # the same target measured twice in one image: both rows couple into the station's own diagonal block
    base = host.synth(n_cam=1, n_sta=6, n_pts=40, rays_per_point=4, seed=14, control_every=4,
                      ap_free_mask=host.AP_K1K2P1P2).net
    dup = np.arange(0, base.n_obs, 7)
    cat = lambda a, b: np.concatenate([a, b])
    net = modified(base, obs_pt=cat(base.obs_pt, base.obs_pt[dup]), obs_sta=cat(base.obs_sta, base.obs_sta[dup]),
                   obs_x=cat(base.obs_x, base.obs_x[dup] + 0.3), obs_y=cat(base.obs_y, base.obs_y[dup] - 0.2),
                   obs_sx=cat(base.obs_sx, base.obs_sx[dup]), obs_sy=cat(base.obs_sy, base.obs_sy[dup]),
                   obs_cx=cat(base.obs_cx, base.obs_cx[dup]), obs_cy=cat(base.obs_cy, base.obs_cy[dup]))
    for ls in ALL_SOLVERS:
        opt = options(ls)
        g = gpu_solve(net, opt)
        o = po.solve(net, opt, po.FULL_DENSE)
        assert_solve_parity(g, o)


def test_constant_blocks_and_partial_masks():
    net = host.synth(n_cam=2, n_sta=8, n_pts=60, rays_per_point=6, seed=6, control_every=4,
                     ap_free_mask=host.ap_mask("k1", "p1"), iop_free_mask=4).net  # only c free
    net = modified(net, ap_free_mask=np.array([host.ap_mask("k1", "p1"), 0]), iop_free_mask=np.array([4, 0]))
    for ls in ALL_SOLVERS:
        opt = options(ls)
        g = gpu_solve(net, opt)
        o = po.solve(net, opt, po.FULL_DENSE)
        assert_solve_parity(g, o)
        assert np.array_equal(g[2][1], net.iop[1]) and np.array_equal(g[3][1], net.ap[1])
        assert np.array_equal(g[2][0][:2], net.iop[0][:2])


def test_empirical_terms_held_constant_at_non_zero_values():
    # no empirical term is estimated, but AP[7] and AP[10] are not zero: the kernels must not take the variant that
    # leaves those terms out (it is chosen from the values, not from the masks alone)
    net = host.synth(n_cam=1, n_sta=10, n_pts=120, rays_per_point=5, seed=6, ap_free_mask=host.AP_K1K2P1P2).net
    ap = net.ap.copy()
    ap[0, 7], ap[0, 10] = 2e-3, -3e-5
    net2 = modified(net, ap=ap)
    with solver.Problem(net2) as P:
        c, r = P.evaluate()
        g = P.solve(options(capi.SOLVER_DENSE_SCHUR))
        # ba_evaluate with other values than the resident ones: zero empirical terms here
        c0, _ = P.evaluate(ap=net.ap)
    co, ro = po.evaluate(net2)
    assert abs(c - co) <= 1e-13 * co and np.max(np.abs(r - ro)) < 1e-10
    assert abs(c0 - po.evaluate(net)[0]) <= 1e-13 * co and abs(c0 - c) > 1e-6 * c
    o = po.solve(net2, options(capi.SOLVER_DENSE_SCHUR), po.SCHUR_DENSE)
    assert_solve_parity(g, o)
    assert g[3][0, 7] == 2e-3 and g[3][0, 10] == -3e-5  # held


def test_no_observations_and_empty_problem():
    net = host.synth(n_cam=1, n_sta=3, n_pts=5, rays_per_point=2, seed=8).net
    empty = modified(net, obs_pt=np.zeros(0, np.int32), obs_sta=np.zeros(0, np.int32), obs_x=np.zeros(0),
                     obs_y=np.zeros(0), obs_sx=np.zeros(0), obs_sy=np.zeros(0), obs_cx=np.zeros(0), obs_cy=np.zeros(0))
    with solver.Problem(empty) as P:
        c, r = P.evaluate()
        co, ro = po.evaluate(empty)
        assert c == pytest.approx(co, abs=1e-15) and np.allclose(r, ro)
        out = P.solve(options(capi.SOLVER_PCG_SCHUR))
    o = po.solve(empty, options(capi.SOLVER_PCG_SCHUR), po.SCHUR_DENSE)
    assert out[4]["stop_reason"] == o[4]["stop_reason"]
    assert np.array_equal(out[0], empty.eop)


def test_error_behaviour():
    net = host.synth(n_cam=1, n_sta=3, n_pts=5, rays_per_point=2, seed=8).net
    bad = modified(net, obs_pt=np.where(np.arange(net.n_obs) == 1, 99, net.obs_pt))
    with pytest.raises(solver.BaError, match="point index 99"):
        solver.Problem(bad)
    with pytest.raises(solver.BaError, match="standard deviation"):
        solver.Problem(modified(net, obs_sx=np.zeros(net.n_obs)))
    with pytest.raises(solver.BaError, match="model"):
        solver.Problem(capi.Network(**{k: getattr(net, k) for k in __import__("util").FIELDS}, model=8))
    with solver.Problem(net) as P:
        with pytest.raises(solver.BaError, match="ba_set_parameters"):
            P.solve_resident(options(capi.SOLVER_DENSE_SCHUR))
        o = options(capi.SOLVER_DENSE_SCHUR)
        o.linear_solver = 7
        P.set_parameters()
        with pytest.raises(solver.BaError, match="linear_solver"):
            P.solve_resident(o)


def test_resident_solve_equals_in_place_solve():
    net = host.synth_config("cfg1", scale=0.5).net
    opt = options(capi.SOLVER_PCG_SCHUR)
    with solver.Problem(net) as P:
        a = P.solve(opt)
        P.set_parameters()
        sm, rows = P.solve_resident(opt)
        b = P.get_parameters()
        for x, y in zip(a[:4], b):
            assert np.array_equal(x, y)  # same kernels, same order: bitwise
        assert sm["final_cost"] == a[4]["final_cost"]
        # solving again from the solution stops at once and changes nothing
        sm2, rows2 = P.solve_resident(opt)
        assert sm2["num_iterations"] <= 2
        assert rel(P.get_parameters()[1], b[1]) < 1e-12


def test_many_camera_columns_through_the_pcg_solvers():
    # 9 cameras x (xp yp c + a1 a2 k1 k2 k3 p1 p2) = 90 free camera columns: more than the block-sparse PCG kernels
    # hold in shared memory, so PCG_SCHUR must take the matrix-free operator (and give the same answer)
    net = host.synth(n_cam=9, n_sta=45, n_pts=1500, rays_per_point=8, ap_free_mask=host.AP_BROWN7, seed=21).net
    assert sum(bin(int(m)).count("1") for m in net.ap_free_mask) + 3 * net.n_cam == 90
    with solver.Problem(net) as P:
        g = P.solve(options(capi.SOLVER_PCG_SCHUR, pcg_max_iters=6000))
        with pytest.raises(solver.BaError) as e:
            P.covariance_cameras()
        assert e.value.code == capi.ERR_UNSUPPORTED
    o = po.solve(net, options(capi.SOLVER_PCG_SCHUR, pcg_max_iters=6000), po.SCHUR_DENSE)
    assert o[4]["termination"] == capi.CONVERGENCE
    assert_solve_parity(g, o, par_rtol=1e-8)
    d = gpu_solve(net, options(capi.SOLVER_DENSE_SCHUR))
    assert_solve_parity(d, o, par_rtol=1e-8)


def test_index_structures_built_on_the_device_match_the_host_construction():
    # The same network set up twice: by the host loops (the rule for small networks) and by the sorts and scans on the
    # device (the rule from 65 536 observations on).  Both must give the same arrays, so everything downstream --
    # evaluation, sparsity pattern, every iteration of the solve -- is the same bits.  Three cameras and few rays per
    # point make station pairs of several sizes; points without observations and a station that two observations of one
    # point share exercise the corners.
    hn = host.synth(n_cam=3, n_sta=90, n_pts=3000, rays_per_point=5, seed=11, ap_free_mask=host.AP_K1K2P1P2, control_every=9)
    net = hn.net
    keep = np.ones(net.n_obs, dtype=bool)
    keep[net.obs_pt == 17] = False                      # a point without observations
    keep[(net.obs_pt % 50 == 3) & (np.arange(net.n_obs) % 2 == 0)] = False   # points with fewer rays than the others
    net = capi.Network(**{**{f: getattr(net, f) for f in ("sta_cam", "pt_prior", "pt_sigma", "iop_prior", "iop_sigma", "ap_free_mask",
                                                             "iop_free_mask", "eop", "xyz", "iop", "ap", "huber_a", "model")},
                          **{f: getattr(net, f)[keep] for f in ("obs_pt", "obs_sta", "obs_x", "obs_y", "obs_sx", "obs_sy", "obs_cx", "obs_cy")}})
    dup = np.nonzero(net.obs_pt == 40)[0]
    net.obs_sta[dup[1]] = net.obs_sta[dup[0]]           # two observations of one point from the same station
    out = {}
    for mode in ("host", "gpu"):
        solver.tuning("setup", {"host": 0, "gpu": 1}[mode])
        opt = solver.default_options()
        opt.linear_solver = capi.SOLVER_PCG_SCHUR
        opt.max_num_iterations = 6
        with solver.Problem(net) as P:
            c, res = P.evaluate()
            sp = P.sparsity()
            sol = P.solve(opt)
        out[mode] = (c, res, sp, sol)
    solver.tuning("setup", -1)
    a, b = out["host"], out["gpu"]
    assert a[0] == b[0] and np.array_equal(a[1], b[1])
    assert a[2][0] == b[2][0] and a[2][1] == b[2][1] and np.array_equal(a[2][2], b[2][2]) and np.array_equal(a[2][3], b[2][3])
    for x, y in zip(a[3][:4], b[3][:4]):
        assert np.array_equal(x, y)
    assert [r["cost"] for r in a[3][5]] == [r["cost"] for r in b[3][5]]
    assert [r["linear_iterations"] for r in a[3][5]] == [r["linear_iterations"] for r in b[3][5]]
