"""Time stamps inside the kernels of one PCG iteration on the benchmark network (ba_time_kernel, which = 9)."""
import sys
import os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bundleadjustment_amd import capi, host, solver

scale = float(sys.argv[1]) if len(sys.argv) > 1 else 1.0
net = host.synth_config("cfg4", scale=scale).net
with solver.Problem(net) as P:
    P.set_parameters()
    opt = solver.default_options()
    opt.linear_solver = capi.SOLVER_PCG_SCHUR
    opt.max_num_iterations = 1
    sm, rows = P.solve_resident(opt)
    print("pcg iterations", sm["total_linear_iterations"])
    P.set_parameters()
    for which, name in ((6, "spmv"), (5, "iteration")):
        print(name, "%.2f us" % (1e6 * P.time_kernel(which, 200)))
    P.time_kernel(9, 4)
