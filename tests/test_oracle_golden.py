"""The CPU oracle against the golden vectors made from the reference's own functors
(tests/golden/make_golden.py), and against the functors themselves where they are built."""
import numpy as np
import pytest

from oracle import py_oracle as po


def test_residuals_bit_exact(golden):
    # src/main.cpp:2033-2120 (with correction) and :1315-1401 (without)
    for c in golden["collinearity"]:
        r, _ = po.collinearity(c["eop"], c["xyz"], c["iop"], c["ap"], c["obs"], jac=False)
        assert r.tolist() == c["r_mls"]
        obs0 = list(c["obs"][:4]) + [0.0, 0.0]
        r0, _ = po.collinearity(c["eop"], c["xyz"], c["iop"], c["ap"], obs0, jac=False)
        assert r0.tolist() == c["r_plain"]


def test_jacobian_matches_complex_step(golden):
    worst = 0.0
    for c in golden["collinearity"]:
        _, J = po.collinearity(c["eop"], c["xyz"], c["iop"], c["ap"], c["obs"])
        Jg = np.array(c["J"]).reshape(2, 28)
        assert np.array_equal(Jg == 0.0, J == 0.0)  # same structural zeros (a1,a2 rows of y)
        nz = Jg != 0.0
        worst = max(worst, np.max(np.abs(J[nz] - Jg[nz]) / np.abs(Jg[nz])))
    assert worst < 1e-11, worst  # tolerance: analytic derivative vs complex step, fp64


def test_golden_covers_both_huber_branches_and_duplicated_term(golden):
    norms = [np.hypot(*c["r_mls"]) for c in golden["collinearity"]]
    assert min(norms) < 1.0 < max(norms)
    assert any(c["ap"][10] != 0.0 for c in golden["collinearity"])
    assert any(c["obs"][4] != 0.0 for c in golden["collinearity"])
    # AP[10] multiplies rr^7 + rr^8 (src/main.cpp:1375): its Jacobian column must be x_bar*(rr^7+rr^8)/sigma
    c = next(c for c in golden["collinearity"] if c["ap"][10] != 0.0)
    xb = (c["obs"][0] - c["iop"][0]) / 1000.0
    yb = (c["obs"][1] - c["iop"][1]) / 1000.0
    rr = xb * xb + yb * yb
    J = np.array(c["J"]).reshape(2, 28)
    assert J[0, 12 + 10] == pytest.approx(xb * (rr ** 7 + rr ** 8) / c["obs"][2], rel=1e-12)


def test_constrain_point_bit_exact(golden):
    # src/main.cpp:1057-1085
    lib = po.lib()
    for c in golden["constrain_point"]:
        p, p0, sg = (np.array(c[k]) for k in ("p", "p0", "sigma"))
        r = np.zeros(3)
        lib.bao_constrain_point(po._dp(p), po._dp(p0), po._dp(sg), po._dp(r))
        assert r.tolist() == c["r"]


@pytest.mark.skipif(po.ref_lib() is None, reason="reference functors not built (no /root/reference here)")
def test_oracle_equals_reference_functor_on_fresh_inputs():
    rng = np.random.default_rng(7)
    for _ in range(200):
        eop = np.concatenate([rng.uniform(-0.5, 0.5, 2), rng.uniform(-3, 3, 1), rng.normal(0, 3e3, 2), [rng.uniform(8e3, 1.2e4)]])
        xyz = rng.normal(0, 3e3, 3)
        xyz[2] = rng.uniform(-1e3, 1e3)
        iop = np.array([rng.uniform(-5, 5), rng.uniform(-5, 5), rng.uniform(2e3, 4e3)])
        ap = rng.normal(0, 1, 16) * 10.0 ** -np.concatenate([[2, 2, 0.5, 2, 3, 1.5, 1.5], np.arange(5, 14)])
        obs = np.array([rng.uniform(-2e3, 2e3), rng.uniform(-2e3, 2e3), 0.5, 0.7, rng.normal(), rng.normal()])
        r, J = po.collinearity(eop, xyz, iop, ap, obs)
        assert r.tolist() == po.ref_collinearity(eop, xyz, iop, ap, obs, mls=True).tolist()
        Jr = po.ref_collinearity_jacobian(eop, xyz, iop, ap, obs)
        nz = Jr != 0
        assert np.max(np.abs(J[nz] - Jr[nz]) / np.abs(Jr[nz])) < 1e-10


# ---- the other projection functors (src/main.cpp:2133-2562, :1412-1490, :2700-2790), models 1..7 of ba_inputs.model

def test_projection_models_residuals_bit_exact(projection_golden):
    seen = set()
    for c in projection_golden["cases"]:
        r, _ = po.projection(c["model"], c["eop"], c["xyz"], c["iop"], c["ap"], c["obs"], jac=False)
        assert r.tolist() == c["r"], c["model"]
        seen.add(c["model"])
    assert seen == {1, 2, 3, 4, 5, 6, 7}


def test_projection_models_jacobian_matches_complex_step(projection_golden):
    worst = {}
    for c in projection_golden["cases"]:
        _, J = po.projection(c["model"], c["eop"], c["xyz"], c["iop"], c["ap"], c["obs"])
        Jg = np.array(c["J"]).reshape(2, 28)
        assert np.array_equal(Jg == 0.0, J == 0.0)
        nz = Jg != 0.0
        # relative to the largest entry of the same parameter block: the kappa column of a point near the axis
        # is a difference of two products and carries their rounding
        scale = np.empty_like(Jg)
        for lo, hi in ((0, 3), (3, 6), (6, 9), (9, 12), (12, 28)):
            scale[:, lo:hi] = np.max(np.abs(Jg[:, lo:hi]), axis=1, keepdims=True)
        worst[c["model"]] = max(worst.get(c["model"], 0.0), np.max(np.abs(J[nz] - Jg[nz]) / scale[nz]))
    assert max(worst.values()) < 1e-11, worst  # tolerance: analytic derivative vs complex step, fp64


def test_model_zero_of_the_general_function_is_collinearity(golden):
    for c in golden["collinearity"][:64]:
        r, J = po.projection(0, c["eop"], c["xyz"], c["iop"], c["ap"], c["obs"])
        r0, J0 = po.collinearity(c["eop"], c["xyz"], c["iop"], c["ap"], c["obs"])
        assert r.tolist() == r0.tolist() and np.array_equal(J, J0)


@pytest.mark.skipif(po.ref_lib() is None, reason="reference functors not built (no /root/reference here)")
@pytest.mark.parametrize("model", [1, 2, 3, 4, 5, 6, 7])
def test_projection_oracle_equals_reference_functor_on_fresh_inputs(model):
    rng = np.random.default_rng(70 + model)
    for _ in range(100):
        eop = np.concatenate([rng.uniform(-0.5, 0.5, 2), rng.uniform(-3, 3, 1), rng.normal(0, 3e3, 2), [rng.uniform(8e3, 1.2e4)]])
        xyz = rng.normal(0, 6e3, 3)
        xyz[2] = rng.uniform(-1e3, 1e3)
        iop = np.array([rng.uniform(-5, 5), rng.uniform(-5, 5), rng.uniform(0.05, 0.1) if model == 6 else rng.uniform(800, 1500)])
        ap = rng.normal(0, 1, 16) * 10.0 ** -np.concatenate([[2, 2, 0.5, 2, 3, 1.5, 1.5], np.arange(5, 14)])
        obs = np.array([rng.uniform(-1e3, 1e3), rng.uniform(-1e3, 1e3), 0.5, 0.7, rng.normal(), rng.normal()])
        if model == 6:
            obs[4:] = 0.0  # collinearityOrthographic takes no corrections (src/main.cpp:1414)
        r, J = po.projection(model, eop, xyz, iop, ap, obs)
        assert r.tolist() == po.ref_projection(model, eop, xyz, iop, ap, obs).tolist()
        Jr = po.ref_projection_jacobian(model, eop, xyz, iop, ap, obs)
        nz = Jr != 0
        assert np.max(np.abs(J[nz] - Jr[nz]) / np.abs(Jr[nz])) < 1e-9
