import sys
sys.path.insert(0, "/root/repo")
from bundleadjustment_amd import host, solver, capi
hn = host.synth_config("cfg4")
P = solver.Problem(hn.net)
P.set_parameters()
o = solver.default_options(); o.linear_solver = capi.SOLVER_PCG_SCHUR; o.pcg_rtol = 1e-12; o.max_num_iterations = 2
o.function_tolerance = 0.0; o.parameter_tolerance = 0.0; o.gradient_tolerance = 0.0
P.solve_resident(o)
P.time_kernel(9, 1)
P.close()
