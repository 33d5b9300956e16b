"""An independent numpy restatement of the trust-region loop, used only to cross-check the C
oracle on tiny problems (dense Jacobian, numpy linear algebra, Python control flow).
It shares with the oracle only the per-observation cost function (bao_collinearity), which is
pinned separately by the golden vectors.  [Ceres 1.13: TrustRegionMinimizer, LevenbergMarquardtStrategy,
HuberLoss, Corrector]"""
import numpy as np

from oracle import py_oracle as po


def _layout(net):
    S, P, C = net.n_sta, net.n_pts, net.n_cam
    used_sta = np.zeros(S, bool)
    used_sta[net.obs_sta] = True
    used_cam = np.zeros(C, bool)
    used_cam[net.sta_cam[net.obs_sta]] = True
    cols = {}
    n = 0
    for i in range(S):
        for k in range(6):
            cols[("eop", i, k)] = n
            n += 1
    for j in range(P):
        for k in range(3):
            cols[("xyz", j, k)] = n
            n += 1
    for c in range(C):
        for k in range(3):
            if net.iop_free_mask[c] & (1 << k):
                cols[("iop", c, k)] = n
                n += 1
        if used_cam[c]:
            for k in range(16):
                if net.ap_free_mask[c] & (1 << k):
                    cols[("ap", c, k)] = n
                    n += 1
    return cols, n, used_sta, used_cam


def _evaluate(net, x, cols, n, jac=True):
    eop, xyz, iop, ap = x
    N, P, C = net.n_obs, net.n_pts, net.n_cam
    r_all, J_all = po.residual_jacobian_all(net, eop, xyz, iop, ap)
    a = net.huber_a
    rows = []
    res = []
    cost = 0.0
    fixed = 0.0
    for i_obs in range(N):
        r = r_all[2 * i_obs:2 * i_obs + 2].copy()
        s = float(r @ r)
        if a > 0 and s > a * a:
            rho0, rho1 = 2 * a * np.sqrt(s) - a * a, a / np.sqrt(s)
        else:
            rho0, rho1 = s, 1.0
        cost += 0.5 * rho0
        sc = np.sqrt(rho1)
        res.extend(sc * r)
        if jac:
            i, j = net.obs_sta[i_obs], net.obs_pt[i_obs]
            c = net.sta_cam[i]
            for row in range(2):
                Jr = np.zeros(n)
                Jd = sc * J_all[i_obs, row]
                for k in range(6):
                    Jr[cols[("eop", i, k)]] = Jd[k]
                for k in range(3):
                    Jr[cols[("xyz", j, k)]] = Jd[6 + k]
                for k in range(3):
                    if ("iop", c, k) in cols:
                        Jr[cols[("iop", c, k)]] = Jd[9 + k]
                for k in range(16):
                    if ("ap", c, k) in cols:
                        Jr[cols[("ap", c, k)]] = Jd[12 + k]
                rows.append(Jr)
    pr = (xyz - net.pt_prior) / net.pt_sigma
    cost += 0.5 * float(np.sum(pr ** 2))
    for j in range(P):
        for k in range(3):
            res.append(pr[j, k])
            if jac:
                Jr = np.zeros(n)
                Jr[cols[("xyz", j, k)]] = 1.0 / net.pt_sigma[j, k]
                rows.append(Jr)
    ir = (iop - net.iop_prior) / net.iop_sigma
    for c in range(C):
        v = 0.5 * float(np.sum(ir[c] ** 2))
        if net.iop_free_mask[c] & 7:
            cost += v
            for k in range(3):
                res.append(ir[c, k])
                if jac:
                    Jr = np.zeros(n)
                    if ("iop", c, k) in cols:
                        Jr[cols[("iop", c, k)]] = 1.0 / net.iop_sigma[c, k]
                    rows.append(Jr)
        else:
            fixed += v
    return cost, fixed, np.array(res), (np.array(rows) if jac else None)


def _plus(net, x, delta, cols):
    eop, xyz, iop, ap = (a.copy() for a in x)
    for (kind, b, k), col in cols.items():
        {"eop": eop, "xyz": xyz, "iop": iop, "ap": ap}[kind][b, k] += delta[col]
    return eop, xyz, iop, ap


def _ambient(net, x, used_sta, used_cam):
    eop, xyz, iop, ap = x
    parts = [eop[used_sta].ravel(), xyz.ravel()]
    for c in range(net.n_cam):
        if net.iop_free_mask[c] & 7:
            parts.append(iop[c])
        if net.ap_free_mask[c] and used_cam[c]:
            parts.append(ap[c])
    return np.concatenate(parts)


def solve(net, opt):
    cols, n, used_sta, used_cam = _layout(net)
    dead = [cols[("eop", i, k)] for i in range(net.n_sta) if not used_sta[i] for k in range(6)]
    x = net.params_copy()
    cost, fixed, res, J = _evaluate(net, x, cols, n)
    scale = 1.0 / (1.0 + np.sqrt(np.sum(J * J, axis=0))) if opt.jacobi_scaling else np.ones(n)
    J = J * scale
    radius, dec = opt.initial_trust_region_radius, 2.0
    reuse = False
    x_norm = np.linalg.norm(_ambient(net, x, used_sta, used_cam))
    log = [dict(iteration=0, cost=cost + fixed, ok=1)]
    it, invalid = 0, 0
    diag = None
    while True:
        if it >= opt.max_num_iterations or radius <= opt.min_trust_region_radius:
            break
        it += 1
        if not reuse:
            diag = np.clip(np.sum(J * J, axis=0), opt.min_lm_diagonal, opt.max_lm_diagonal)
        D2 = np.sqrt(diag / radius) ** 2
        H = J.T @ J + np.diag(D2)
        for c in dead:
            H[c, c] = 1.0
        g = J.T @ res
        reuse = True
        try:
            L = np.linalg.cholesky(H)
            y = np.linalg.solve(L.T, np.linalg.solve(L, g))
            step = -y
            m = J @ step
            mcc = -float(m @ (res + 0.5 * m))
        except np.linalg.LinAlgError:
            mcc = -1.0
        if not mcc > 0:
            invalid += 1
            if invalid >= opt.max_num_consecutive_invalid_steps:
                break
            radius /= dec
            dec *= 2
            log.append(dict(iteration=it, cost=cost + fixed, ok=0))
            continue
        invalid = 0
        cand = _plus(net, x, step * scale, cols)
        ccost, _, _, _ = _evaluate(net, cand, cols, n, jac=False)
        step_norm = np.linalg.norm(_ambient(net, x, used_sta, used_cam) - _ambient(net, cand, used_sta, used_cam))
        if step_norm <= opt.parameter_tolerance * (x_norm + opt.parameter_tolerance):
            break
        if abs(cost - ccost) <= opt.function_tolerance * cost:
            break
        rho = (cost - ccost) / mcc
        if rho > opt.min_relative_decrease:
            x = cand
            x_norm = np.linalg.norm(_ambient(net, x, used_sta, used_cam))
            cost, fixed, res, J = _evaluate(net, x, cols, n)
            J = J * scale
            radius = min(opt.max_trust_region_radius, radius / max(1.0 / 3.0, 1.0 - (2 * rho - 1) ** 3))
            dec, reuse = 2.0, False
            log.append(dict(iteration=it, cost=cost + fixed, ok=1))
        else:
            radius /= dec
            dec *= 2
            log.append(dict(iteration=it, cost=ccost + fixed, ok=0))
    return x, log
