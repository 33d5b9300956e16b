#!/usr/bin/env python3
"""Generates tests/golden/collinearity_golden.json and tests/golden/projection_golden.json from the
reference's own cost functors.

Run in the build container, where /root/reference is mounted and `make -C oracle` has built
oracle/_ref/libba_ref.so out of src/main.cpp:1315-1401, :2033-2120, :1057-1085 (see
oracle/ref/build_ref.sh).  The fixture holds numbers only: inputs, the residuals of both
functors evaluated in double, and the 2x28 Jacobian obtained by complex-step differentiation
of the reference functor (std::complex<double> instantiation, h = 1e-30).

projection_golden.json holds the same for the seven other projection functors (src/main.cpp:2133-2562:
collinearity stereographic and the four fisheye models; :1412-1490 collinearity orthographic; :2700-2790
omnidirectional), at off-axis angles up to 69 degrees.

reporting_golden.json holds inputs and outputs of the reference's statistics and angle helpers
(src/main.cpp:791-823, 884-1019).

    python tests/golden/make_golden.py
"""
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import py_oracle as po  # noqa: E402

SEED = 0xBA5EBA11
N_PER_CLASS = 48


def rot(om, ph, ka):
    so, co, sp, cp, sk, ck = np.sin(om), np.cos(om), np.sin(ph), np.cos(ph), np.sin(ka), np.cos(ka)
    return np.array([[cp * ck, co * sk + so * sp * ck, so * sk - co * sp * ck],
                     [-cp * sk, co * ck - so * sp * sk, so * ck + co * sp * sk],
                     [sp, -so * cp, co * cp]])


def make_case(rng, ap_kind, resid_kind, with_corr):
    # a station looking roughly down the -Z axis of its own frame at a point 8-12 m away
    eop = np.array([rng.uniform(-0.4, 0.4), rng.uniform(-0.4, 0.4), rng.uniform(-np.pi, np.pi),
                    rng.uniform(-5e3, 5e3), rng.uniform(-5e3, 5e3), rng.uniform(8e3, 12e3)])
    R = rot(*eop[:3])
    cam = np.array([rng.uniform(-5e3, 5e3), rng.uniform(-5e3, 5e3), -rng.uniform(8e3, 12e3)])
    xyz = eop[3:] + R.T @ cam
    iop = np.array([rng.uniform(-5, 5), rng.uniform(-5, 5), rng.uniform(2500, 3500)])
    ap = np.zeros(16)
    if ap_kind in ("brown", "all"):
        ap[:7] = [rng.normal(0, 1e-2), rng.normal(0, 1e-2), rng.normal(0, 0.5), rng.normal(0, 2e-2),
                  rng.normal(0, 5e-4), rng.normal(0, 5e-2), rng.normal(0, 5e-2)]
    if ap_kind == "all":
        ap[7:] = rng.normal(0, 1.0, 9) * 10.0 ** -np.arange(5, 14)
        ap[10] = rng.choice([-1, 1]) * rng.uniform(0.5, 2) * 1e-7  # pins the duplicated rr^7 + rr^8 term
    sx, sy = rng.uniform(0.2, 1.0, 2)
    corr = rng.normal(0, 0.7, 2) if with_corr else np.zeros(2)
    obs = np.array([0.0, 0.0, sx, sy, corr[0], corr[1]])
    # fixed point so that the residual is ~0 (the distortion depends on the observed coordinate)
    for _ in range(8):
        r = po.ref_collinearity(eop, xyz, iop, ap, obs, mls=True)
        obs[0] += r[0] * sx
        obs[1] += r[1] * sy
    if resid_kind == "small":   # |r| < 1  -> quadratic branch of Huber(1)
        t = rng.uniform(0, 2 * np.pi)
        m = rng.uniform(0.05, 0.7)
        obs[0] += m * np.cos(t) * sx
        obs[1] += m * np.sin(t) * sy
    else:                       # |r| > 1  -> linear branch
        obs[0] += rng.choice([-1, 1]) * rng.uniform(2, 20) * sx
        obs[1] += rng.choice([-1, 1]) * rng.uniform(2, 20) * sy
    return eop, xyz, iop, ap, obs


MODEL_NAMES = {1: "collinearityStereographicMachineLearnedSimple :2133-2214",
               2: "fisheyeEquidistantMachineLearnedSimple :2226-2301",
               3: "fisheyeEquisolidAngleMachineLearnedSimple :2312-2387",
               4: "fisheyeOrthographicMachineLearnedSimple :2398-2473",
               5: "fisheyeStereographicMachineLearnedSimple :2484-2562",
               6: "collinearityOrthographic :1412-1490 (no correction arguments)",
               7: "omniCollinearityMachineLearnedSimple :2700-2790"}


def make_projection_case(rng, model, ap_kind, resid_kind, with_corr):
    eop = np.array([rng.uniform(-0.4, 0.4), rng.uniform(-0.4, 0.4), rng.uniform(-np.pi, np.pi),
                    rng.uniform(-5e3, 5e3), rng.uniform(-5e3, 5e3), rng.uniform(8e3, 12e3)])
    R = rot(*eop[:3])
    depth, off_axis, azimuth = rng.uniform(8e3, 12e3), rng.uniform(0.01, 1.2), rng.uniform(0, 2 * np.pi)
    cam = np.array([depth * np.tan(off_axis) * np.cos(azimuth), depth * np.tan(off_axis) * np.sin(azimuth), -depth])
    xyz = eop[3:] + R.T @ cam
    # orthographic: image = c x object coordinates, so c is a scale of order 0.1
    iop = np.array([rng.uniform(-5, 5), rng.uniform(-5, 5), rng.uniform(0.05, 0.1) if model == 6 else rng.uniform(800, 1500)])
    ap = np.zeros(16)
    if ap_kind in ("brown", "all"):
        ap[:7] = [rng.normal(0, 1e-2), rng.normal(0, 1e-2), rng.normal(0, 0.5), rng.normal(0, 2e-2),
                  rng.normal(0, 5e-4), rng.normal(0, 5e-2), rng.normal(0, 5e-2)]
    if ap_kind == "all":
        ap[7:] = rng.normal(0, 1.0, 9) * 10.0 ** -np.arange(5, 14)
    sx, sy = rng.uniform(0.2, 1.0, 2)
    corr = rng.normal(0, 0.7, 2) if (with_corr and model != 6) else np.zeros(2)
    obs = np.array([0.0, 0.0, sx, sy, corr[0], corr[1]])
    for _ in range(12):
        r = po.ref_projection(model, eop, xyz, iop, ap, obs)
        obs[0] += r[0] * sx
        obs[1] += r[1] * sy
    if resid_kind == "small":
        t, m = rng.uniform(0, 2 * np.pi), rng.uniform(0.05, 0.7)
        obs[0] += m * np.cos(t) * sx
        obs[1] += m * np.sin(t) * sy
    else:
        obs[0] += rng.choice([-1, 1]) * rng.uniform(2, 20) * sx
        obs[1] += rng.choice([-1, 1]) * rng.uniform(2, 20) * sy
    return eop, xyz, iop, ap, obs


def main_projection():
    rng = np.random.default_rng(SEED + 1)
    cases = []
    for model in sorted(MODEL_NAMES):
        for ap_kind in ("zero", "brown", "all"):
            for resid_kind in ("small", "large"):
                for i in range(12):
                    eop, xyz, iop, ap, obs = make_projection_case(rng, model, ap_kind, resid_kind, bool(i % 2))
                    cases.append({
                        "model": model, "ap_kind": ap_kind, "resid_kind": resid_kind,
                        "eop": eop.tolist(), "xyz": xyz.tolist(), "iop": iop.tolist(), "ap": ap.tolist(),
                        "obs": obs.tolist(),
                        "r": po.ref_projection(model, eop, xyz, iop, ap, obs).tolist(),
                        "J": po.ref_projection_jacobian(model, eop, xyz, iop, ap, obs).reshape(-1).tolist(),
                    })
    out = {
        "generator": "tests/golden/make_golden.py",
        "source": "reference functors src/main.cpp:2133-2562, 1412-1490, 2700-2790 compiled by oracle/ref/build_ref.sh",
        "models": {str(k): v for k, v in MODEL_NAMES.items()},
        "jacobian_method": "complex step, std::complex<double>, h=1e-30 (atan2 continued to first order)",
        "seed": SEED + 1,
        "cases": cases,
    }
    path = os.path.join(ROOT, "tests", "golden", "projection_golden.json")
    with open(path, "w") as f:
        json.dump(out, f)
    print(f"wrote {path}: {len(cases)} cases")


def main_reporting():
    """Inputs and outputs of the reference's statistics and angle helpers (src/main.cpp:791-823, 884-1019)."""
    rng = np.random.default_rng(SEED + 2)
    stats = []
    for n in (1, 2, 3, 10, 11, 64, 257):
        for kind in ("normal", "ties"):
            v = rng.normal(1.5, 3.0, n)
            if kind == "ties":
                v = np.round(v)
            stats.append({"v": v.tolist(), "stats": po.ref_statistics(v).tolist(),  # median mean stdev min max rmse
                          "median": po.ref_scalar("ref_median", v, int(n)), "stddev": po.ref_scalar("ref_stddev", v, int(n)),
                          "sort_index": po.ref_sort_index(v).tolist()})
    angles = []
    for i in range(40):
        eop, xyz, iop, ap, obs = make_projection_case(rng, 2, ("zero", "brown", "all")[i % 3], "small", False)
        angles.append({"eop": eop.tolist(), "xyz": xyz.tolist(), "iop": iop.tolist(), "ap": ap.tolist(), "x": obs[0], "y": obs[1],
                       "incidence": po.ref_scalar("ref_incidence_angle", eop, xyz),
                       "refraction": po.ref_scalar("ref_refraction_angle", float(obs[0]), float(obs[1]), iop, ap),
                       "radial": po.ref_scalar("ref_radial_distance", float(obs[0]), float(obs[1]), iop, ap)})
    ic = []
    for _ in range(8):
        n, mse, k = float(rng.integers(100, 10 ** 7)), float(rng.uniform(0.1, 1e6)), float(rng.integers(7, 10 ** 5))
        ic.append({"n": n, "mse": mse, "k": k, "aic": po.ref_scalar("ref_aic", n, mse, k), "bic": po.ref_scalar("ref_bic", n, mse, k)})
    out = {"generator": "tests/golden/make_golden.py",
           "source": "reference helpers src/main.cpp:791-823, 884-1019 compiled by oracle/ref/build_ref.sh",
           "seed": SEED + 2, "statistics": stats, "angles": angles, "information_criteria": ic}
    path = os.path.join(ROOT, "tests", "golden", "reporting_golden.json")
    with open(path, "w") as f:
        json.dump(out, f)
    print(f"wrote {path}: {len(stats)} statistics, {len(angles)} angle and {len(ic)} AIC/BIC cases")


def main():
    if po.ref_lib() is None:
        sys.exit("oracle/_ref/libba_ref.so is missing: run `make -C oracle` where /root/reference exists")
    rng = np.random.default_rng(SEED)
    cases = []
    for ap_kind in ("zero", "brown", "all"):
        for resid_kind in ("small", "large"):
            for i in range(N_PER_CLASS):
                with_corr = bool(i % 2)
                eop, xyz, iop, ap, obs = make_case(rng, ap_kind, resid_kind, with_corr)
                r_mls = po.ref_collinearity(eop, xyz, iop, ap, obs, mls=True)
                obs0 = obs.copy()
                obs0[4:] = 0.0
                r_plain = po.ref_collinearity(eop, xyz, iop, ap, obs0, mls=False)
                J = po.ref_collinearity_jacobian(eop, xyz, iop, ap, obs)
                cases.append({
                    "ap_kind": ap_kind, "resid_kind": resid_kind,
                    "eop": eop.tolist(), "xyz": xyz.tolist(), "iop": iop.tolist(), "ap": ap.tolist(),
                    "obs": obs.tolist(),
                    "r_mls": r_mls.tolist(),      # collinearityMachineLearnedSimple, src/main.cpp:2033-2120
                    "r_plain": r_plain.tolist(),  # collinearity (corrections zero), src/main.cpp:1315-1401
                    "J": J.reshape(-1).tolist(),  # complex step of the former, row-major 2x28
                })
    priors = []
    for _ in range(16):
        p, p0 = rng.normal(0, 5e3, 3), rng.normal(0, 5e3, 3)
        sg = 10.0 ** rng.uniform(-1, 4, 3)
        priors.append({"p": p.tolist(), "p0": p0.tolist(), "sigma": sg.tolist(),
                       "r": po.ref_constrain_point(p, p0, sg).tolist()})  # src/main.cpp:1057-1085
    out = {
        "generator": "tests/golden/make_golden.py",
        "source": "reference functors src/main.cpp:1315-1401, 2033-2120, 1057-1085 compiled by oracle/ref/build_ref.sh",
        "jacobian_method": "complex step, std::complex<double>, h=1e-30",
        "seed": SEED,
        "collinearity": cases,
        "constrain_point": priors,
    }
    path = os.path.join(ROOT, "tests", "golden", "collinearity_golden.json")
    with open(path, "w") as f:
        json.dump(out, f)
    print(f"wrote {path}: {len(cases)} collinearity cases, {len(priors)} prior cases")


if __name__ == "__main__":
    main()
    main_projection()
    main_reporting()
