"""Full-size runs (the 10M-observation network the benchmark is quoted on): the oracle is too
slow here, so the checks are size-independent properties of the path."""
import numpy as np
import pytest

from bundleadjustment_amd import capi, host, solver

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
