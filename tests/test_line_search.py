"""The bounds-constrained route of the trust-region loop on the CPU: the step-size contraction of the Armijo search
(oracle and product, each against numpy) and the oracle's bounded solve.  The route restates Ceres 1.13
(trust_region_minimizer.cc DoLineSearch, line_search.cc, polynomial.cc, parameter_block.h Plus); the reference holds no
recorded output of a bounded solve: parity unpinned against Ceres itself, like the rest of the loop."""
import numpy as np
import pytest

from bundleadjustment_amd import capi, host, solver
from oracle import py_oracle as po

DBL_MAX = np.finfo(np.float64).max


def numpy_contract(samples, lo, hi):
    """Minimiser over [lo, hi] of the polynomial through values and slopes, with numpy.linalg and numpy.roots."""
    deg = 2 * len(samples) - 1
    A, rhs = [], []
    for x, f, g in samples:
        A.append([x ** (deg - j) for j in range(deg + 1)])
        rhs.append(f)
        A.append([(deg - j) * x ** (deg - j - 1) if j < deg else 0.0 for j in range(deg + 1)])
        rhs.append(g)
    c = np.linalg.solve(np.array(A), np.array(rhs))
    # as Ceres' MinimizePolynomial: the real parts of ALL roots of the derivative, complex pairs included.  The real
    # part of a complex pair is no stationary point, so it can never beat the interval's ends and the real stationary
    # points inside, which the oracle and the product offer: same minimum, asserted below for both.
    cands = [(lo + hi) / 2, lo, hi] + [z.real for z in np.roots(np.polyder(c)) if lo <= z.real <= hi]
    vals = [np.polyval(c, x) for x in cands]
    return c, min(vals)


def cases(n, seed):
    rng = np.random.default_rng(seed)
    for t in range(n):
        f0, g0, a = rng.uniform(1, 10), -rng.uniform(0.1, 5), rng.uniform(0.2, 1.0)
        cur = [a, f0 + rng.uniform(0, 5), rng.uniform(-3, 8), 1, 1]
        prev = [a / rng.uniform(0.61, 0.99), f0 + rng.uniform(0, 9), rng.uniform(-3, 12), 1, 1] if t % 2 else None
        yield [0.0, f0, g0, 1, 1], prev, cur, 1e-3 * a, 0.6 * a


@pytest.mark.parametrize("which", ["oracle", "product"])
def test_contraction_minimises_the_interpolating_polynomial(which):
    f = po.interpolating_step if which == "oracle" else solver.line_search_contract
    for start, prev, cur, lo, hi in cases(600, 3):
        got = f(start, prev, cur, lo, hi)
        assert lo <= got <= hi
        samples = [tuple(s[:3]) for s in (start, cur, prev) if s is not None]
        c, best = numpy_contract(samples, lo, hi)
        # cubic (two samples) or quintic (three): the value reached is the smallest on the interval
        assert abs(np.polyval(c, got) - best) <= 1e-11 * max(1.0, abs(best))


def test_contraction_oracle_and_product_agree():
    for start, prev, cur, lo, hi in cases(600, 4):
        a, b = po.interpolating_step(start, prev, cur, lo, hi), solver.line_search_contract(start, prev, cur, lo, hi)
        assert abs(a - b) <= 1e-12 * a


@pytest.mark.parametrize("which", ["oracle", "product"])
def test_contraction_halves_the_step_when_the_trial_is_not_finite(which):
    f = po.interpolating_step if which == "oracle" else solver.line_search_contract
    assert f([0, 5.0, -1.0, 1, 1], None, [0.8, 0.0, 0.0, 0, 0], 8e-4, 0.48) == 0.4
    assert f([0, 5.0, -1.0, 1, 1], None, [0.8, 0.0, 0.0, 0, 0], 8e-4, 0.3) == 0.3  # clamped to the interval


def small_net():
    return host.synth(n_cam=2, n_sta=12, n_pts=150, rays_per_point=6, ap_free_mask=host.AP_K1K2P1P2, seed=5).net


def lm_options(max_iter=100):
    """Ceres' own LM diagonal limits: the reference clamps the diagonal to 1e-150 (src/main.cpp:4712-4713), which leaves
    a plain Gauss-Newton step whose projection need not descend once a bound is active."""
    o = capi.default_options()
    o.min_lm_diagonal, o.max_lm_diagonal, o.max_num_iterations = 1e-6, 1e32, max_iter
    return o


def test_oracle_inactive_bounds_change_nothing_but_the_route():
    net = small_net()
    opt = capi.default_options()
    free = po.solve(net, opt, po.SCHUR_DENSE)
    lo = np.tile([-100.0, -100.0, 0.0], (net.n_cam, 1))
    hi = np.tile([100.0, 100.0, DBL_MAX], (net.n_cam, 1))
    boxed = po.solve(net, opt, po.SCHUR_DENSE, iop_bounds=(lo, hi))
    # every Gauss-Newton step of this network passes the sufficient-decrease test at full length
    assert boxed[4]["num_iterations"] == free[4]["num_iterations"]
    for a, b in zip(free[:4], boxed[:4]):
        assert np.array_equal(a, b)
    assert [r["cost"] for r in free[5]] == [r["cost"] for r in boxed[5]]


def test_oracle_active_bounds_are_kept_and_reached():
    net = small_net()
    free = po.solve(net, lm_options(), po.SCHUR_DENSE)
    lo = np.tile([-100.0, -100.0, 0.0], (net.n_cam, 1))
    hi = np.tile([100.0, 100.0, DBL_MAX], (net.n_cam, 1))
    hi[0, 0] = free[2].reshape(-1, 3)[0, 0] - 1.0  # xp of camera 0 may not reach its free optimum
    lo[1, 2] = free[2].reshape(-1, 3)[1, 2] + 5.0  # nor c of camera 1
    boxed = po.solve(net, lm_options(), po.SCHUR_DENSE, iop_bounds=(lo, hi))
    iop = boxed[2].reshape(-1, 3)
    assert np.all(iop >= lo) and np.all(iop <= hi)
    assert iop[0, 0] == hi[0, 0] and iop[1, 2] == lo[1, 2]
    assert boxed[4]["termination"] == capi.CONVERGENCE
    assert free[4]["final_cost"] < boxed[4]["final_cost"] < 1.01 * free[4]["final_cost"]
    costs = [r["cost"] for r in boxed[5] if r["step_is_successful"]]
    assert all(b <= a for a, b in zip(costs, costs[1:]))
    # some accepted steps were shortened by the search: their decrease is far below the model's
    assert sum(1 for r in boxed[5][1:] if r["step_is_successful"] and r["relative_decrease"] < 0.5) > 5


def test_oracle_start_point_is_projected():
    net = small_net()
    lo = np.tile([-DBL_MAX, -DBL_MAX, 0.0], (net.n_cam, 1))
    hi = np.tile([DBL_MAX, DBL_MAX, DBL_MAX], (net.n_cam, 1))
    lo[0, 0] = net.iop[0, 0] + 2.0  # the start value of xp lies below its lower bound
    opt = lm_options(0)
    out = po.solve(net, opt, po.SCHUR_DENSE, iop_bounds=(lo, hi))
    assert out[2].reshape(-1, 3)[0, 0] == lo[0, 0]
    moved = net.iop.copy()
    moved[0, 0] = lo[0, 0]
    cost, _ = po.evaluate(net, iop=moved)
    assert abs(out[4]["initial_cost"] - cost) <= 1e-12 * cost


def test_oracle_infeasible_boxes_are_errors():
    net = small_net()
    lo = np.tile([-100.0, -100.0, 0.0], (net.n_cam, 1))
    hi = np.tile([100.0, 100.0, DBL_MAX], (net.n_cam, 1))
    hi[1, 1] = lo[1, 1]
    with pytest.raises(RuntimeError, match="empty interval"):
        po.solve(net, capi.default_options(), po.SCHUR_DENSE, iop_bounds=(lo, hi))
    from util import modified
    fixed = modified(net, iop_free_mask=np.array([0, 7], dtype=np.uint8))
    hi[1, 1] = 100.0
    hi[0, 2] = net.iop[0, 2] - 1.0  # a constant block outside its box
    with pytest.raises(RuntimeError, match="outside"):
        po.solve(fixed, capi.default_options(), po.SCHUR_DENSE, iop_bounds=(lo, hi))
