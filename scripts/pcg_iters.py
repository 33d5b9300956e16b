"""PCG iterations per LM iteration on the benchmark network (pcg_rtol 1e-12), for preconditioner experiments."""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bundleadjustment_amd import capi, host, solver

name = sys.argv[1] if len(sys.argv) > 1 else "cfg4"
net = host.synth_config(name).net
with solver.Problem(net) as P:
    P.set_parameters()
    opt = solver.default_options()
    opt.linear_solver = capi.SOLVER_PCG_SCHUR
    opt.pcg_rtol = 1e-12
    opt.max_num_iterations = 4
    sm, rows = P.solve_resident(opt)
    print(name, "pcg iterations", [r["linear_iterations"] for r in rows], "cost", rows[-1]["cost"])
    print("iteration %.2f us" % (1e6 * P.time_kernel(5, 200)))
