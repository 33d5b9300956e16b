"""LM iterations on a benchmark network with the sparse direct solver and with the PCG: time per phase, costs."""
import os
import sys
import time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bundleadjustment_amd import capi, host, solver

name = sys.argv[1] if len(sys.argv) > 1 else "cfg4"
iters = int(sys.argv[2]) if len(sys.argv) > 2 else 6
only = sys.argv[3] if len(sys.argv) > 3 else ""
scale = float(sys.argv[4]) if len(sys.argv) > 4 else 1.0
net = host.synth_config(name, scale=scale).net
print(name, "scale", scale, "stations", net.n_sta, "observations", net.n_obs)
with solver.Problem(net) as P:
    for ls, label in ((capi.SOLVER_SPARSE_SCHUR, "sparse"), (capi.SOLVER_PCG_SCHUR, "pcg")):
        if only and label != only:
            continue
        for rep in range(2):
            P.set_parameters()
            opt = solver.default_options()
            opt.linear_solver = ls
            opt.pcg_rtol = 1e-12
            opt.max_num_iterations = iters
            opt.function_tolerance = opt.parameter_tolerance = opt.gradient_tolerance = 0.0
            t0 = time.perf_counter()
            sm, rows = P.solve_resident(opt)
            dt = time.perf_counter() - t0
        print(label, "%.2f ms/step" % (1e3 * dt / iters), "solve phase ms", ["%.2f" % (1e3 * r["t_solve_s"]) for r in rows[1:]],
              "linearise ms", ["%.2f" % (1e3 * r["t_linearize_s"]) for r in rows[1:]])
        if label == "sparse":
            print("   tree", P.sparse_solver_stats())
        print("   costs", ["%.9e" % r["cost"] for r in rows])
