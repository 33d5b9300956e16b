"""A-posteriori variances of the 10M-observation network at full size: three LM iterations, then ba_covariance (dense inverse of
the reduced system of order 60 007 on the matrix cores) and ba_covariance_cameras (one PCG solve per camera column); prints the
times, a few properties and the gap between the two routes for the camera parameters."""
import time, sys
import numpy as np
sys.path.insert(0, ".")
from bundleadjustment_amd import capi, host, solver
t0 = time.time()
hn = host.synth_config("cfg4", outlier_fraction=0.0)
net = hn.net
print("synth", time.time() - t0, flush=True)
with solver.Problem(net) as P:
    opt = solver.default_options()
    opt.max_num_iterations = 3
    P.set_parameters()
    t0 = time.time()
    sm, rows = P.solve_resident(opt)
    print("solve", time.time() - t0, [r["cost"] for r in rows], flush=True)
    t0 = time.time()
    ve, vx, vi, va = P.covariance()
    print("cov", time.time() - t0, flush=True)
print(ve.shape, vx.shape, vi, va[0, :8])
print("finite", np.all(np.isfinite(ve)), np.all(np.isfinite(vx)), "pos", np.all(ve > 0), np.all(vx > 0))
print("post<=prior", np.all(vx <= net.pt_sigma ** 2 * (1 + 1e-9)))
print("sd eop median", np.sqrt(np.median(ve, axis=0)), "sd xyz median", np.sqrt(np.median(vx, axis=0)))
with solver.Problem(net) as P:
    P.set_parameters(); P.solve_resident(opt)
    t0 = time.time(); wi, wa = P.covariance_cameras(); print("cov cameras", time.time() - t0)
print("rel gap iop", np.max(np.abs(wi / vi - 1)), "ap", np.max(np.abs(wa[0, [2, 3, 5, 6]] / va[0, [2, 3, 5, 6]] - 1)))
