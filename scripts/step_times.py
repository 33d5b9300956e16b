"""Wall-clock phases of the LM iterations on a benchmark network (columns of ba_iter_row) and kernel timings."""
import os
import sys
import time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bundleadjustment_amd import capi, host, solver

name = sys.argv[1] if len(sys.argv) > 1 else "cfg4"
scale = float(sys.argv[2]) if len(sys.argv) > 2 else 1.0
net = host.synth_config(name, scale=scale).net
t = time.time()
with solver.Problem(net) as P:
    print("ba_create %.3f s" % (time.time() - t))
    opt = solver.default_options()
    opt.max_num_iterations = 8
    for rep in range(2):
        P.set_parameters()
        t = time.time()
        sm, rows = P.solve_resident(opt)
        print("solve %.1f ms, %d iterations, %d pcg" % (1e3 * (time.time() - t), sm["num_iterations"], sm["total_linear_iterations"]))
    for r in rows:
        print("it %2d lin %6.2f solve %6.2f eval %5.2f ms  pcg %3d  cost %.6e" % (r["iteration"], 1e3 * r["t_linearize_s"], 1e3 * r["t_solve_s"],
                                                                                   1e3 * r["t_eval_s"], r["linear_iterations"], r["cost"]))
    P.set_parameters()
    for which, nm in ((7, "k_point_elim"), (8, "k_obs_rows"), (4, "k_schur_blocks"), (3, "k_cost_obs"), (6, "spmv"), (5, "pcg iteration")):
        try:
            print("%-16s %8.1f us" % (nm, 1e6 * P.time_kernel(which, 20)))
        except Exception as e:
            print(nm, e)
