"""Largest relative gap between ba_covariance (GPU, Schur-complement route) and the oracle's inverse of the full normal
matrix, on the networks of tests/test_gpu_parity.py::test_covariance_after_the_solve and two larger ones."""
import os
import sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bundleadjustment_amd import capi, host, solver
from oracle import py_oracle as po

nets = {"three_cameras": host.synth(n_cam=3, n_sta=12, n_pts=150, rays_per_point=6, ap_free_mask=host.AP_K1K2P1P2, seed=5).net,
        "fixed_camera": host.synth(n_cam=1, n_sta=9, n_pts=80, rays_per_point=5, ap_free_mask=0, iop_free_mask=0, seed=7,
                                   outlier_fraction=0.1, control_every=8).net,
        "cfg1": host.synth_config("cfg1").net,
        "60_stations": host.synth(n_cam=1, n_sta=60, n_pts=600, rays_per_point=5, ap_free_mask=host.AP_K1K2P1P2, seed=13, control_every=6).net}
for name, net in nets.items():
    opt = solver.default_options()
    opt.linear_solver = capi.SOLVER_DENSE_SCHUR
    with solver.Problem(net) as P:
        P.set_parameters()
        P.solve_resident(opt)
        eop, xyz, iop, ap = P.get_parameters()
        got = P.covariance()
    want = po.covariance(net, eop, xyz, iop, ap)
    worst = {}
    for g, w, k in zip(got, want, ("eop", "xyz", "iop", "ap")):
        nz = w != 0.0
        if nz.any():
            worst[k] = float(np.max(np.abs(g[nz] - w[nz]) / w[nz]))
    print(name, {k: "%.1e" % v for k, v in worst.items()}, flush=True)
