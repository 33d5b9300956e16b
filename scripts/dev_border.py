import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bundleadjustment_amd import host, capi, solver
for kw in (dict(), dict(ap_free_mask=0, iop_free_mask=0)):
    net = host.synth_config("cfg4", scale=1.0, **kw).net
    P = solver.Problem(net); P.set_parameters()
    opt = solver.default_options(); opt.linear_solver = capi.SOLVER_PCG_SCHUR; opt.max_num_iterations = 4; opt.pcg_rtol = 1e-12; opt.pcg_max_iters = 5000
    sm, rows = P.solve_resident(opt)
    print(kw, [r['linear_iterations'] for r in rows], ["%.4f" % r['t_solve_s'] for r in rows], flush=True)
    P.close()
