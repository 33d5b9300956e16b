import os, sys, time, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bundleadjustment_amd import host, capi, solver
name, scale, iters = sys.argv[1], float(sys.argv[2]), int(sys.argv[3])
rtol = float(sys.argv[4]) if len(sys.argv) > 4 else 1e-12
t=time.time(); hn = host.synth_config(name, scale=scale); net = hn.net
print(name, net.n_obs, net.n_sta, net.n_pts, net.n_cam, "gen %.1fs"%(time.time()-t), flush=True)
t=time.time(); P = solver.Problem(net); P.set_parameters(); print("create %.2fs"%(time.time()-t), flush=True)
for which,nm,bytes_per_obs in ((0,"linearize",0),(1,"point_apply",0),(2,"station_gather",0),(3,"cost",0),(4,"schur_blocks",0),(5,"cg2_iteration",0),(6,"spmv",0)):
    dt = P.time_kernel(which, 5)
    print("kernel %-15s %.3f ms  %.2f Gobs/s"%(nm, dt*1e3, net.n_obs/dt/1e9), flush=True)
opt = solver.default_options(); opt.linear_solver = capi.SOLVER_PCG_SCHUR; opt.max_num_iterations = iters; opt.pcg_rtol = rtol; opt.pcg_max_iters=5000
P.set_parameters()
t=time.time(); sm, rows = P.solve_resident(opt); w=time.time()-t
print("solve wall %.3f s"%w, sm['stop_reason_name'], sm['num_iterations'], "lin", sm['total_linear_iterations'])
for r in rows: print("  it %d ok=%d cost %.12e dc %.3e |g| %.3e step %.3e rho %.3f lin %d  t_lin %.4f t_solve %.4f t_eval %.4f"%(r['iteration'],r['step_is_successful'],r['cost'],r['cost_change'],r['gradient_max_norm'],r['step_norm'],r['relative_decrease'],r['linear_iterations'],r['t_linearize_s'],r['t_solve_s'],r['t_eval_s']))
