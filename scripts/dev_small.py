"""Development: per-iteration times of the small and mid networks with the dense reduced solve."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bundleadjustment_amd import host, capi, solver
for name in ("cfg1", "cfg2"):
    net = host.synth_config(name, scale=1.0).net
    P = solver.Problem(net); P.set_parameters()
    opt = solver.default_options(); opt.linear_solver = capi.SOLVER_DENSE_SCHUR; opt.max_num_iterations = 8
    P.solve_resident(opt); P.set_parameters()
    t = time.perf_counter(); sm, rows = P.solve_resident(opt); w = time.perf_counter() - t
    print(name, net.n_obs, "iterations", sm["num_iterations"], "wall %.2f ms" % (w * 1e3), "per iteration %.3f ms" % (w * 1e3 / sm["num_iterations"]),
          ["%.2f %.2f" % (r['t_linearize_s'] * 1e3, r['t_solve_s'] * 1e3) for r in rows[1:4]], flush=True)
