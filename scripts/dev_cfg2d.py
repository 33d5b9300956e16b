import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bundleadjustment_amd import host, capi, solver
net = host.synth_config("cfg2", scale=1.0).net
P = solver.Problem(net); P.set_parameters()
opt = solver.default_options(); opt.linear_solver = capi.SOLVER_DENSE_SCHUR; opt.max_num_iterations = 6
sm, rows = P.solve_resident(opt)
print([ "%.4f %.4f" % (r['t_linearize_s'], r['t_solve_s']) for r in rows])
