import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bundleadjustment_amd import host, capi, solver
name = sys.argv[1] if len(sys.argv) > 1 else "cfg2"
net = host.synth_config(name, scale=1.0).net
print(name, net.n_obs, net.n_sta, net.n_pts, net.n_cam, flush=True)
for ls, nm in ((capi.SOLVER_DENSE_SCHUR, "dense"), (capi.SOLVER_PCG_SCHUR, "pcg")):
    P = solver.Problem(net); P.set_parameters()
    opt = solver.default_options(); opt.linear_solver = ls; opt.max_num_iterations = 6
    t = time.time(); sm, rows = P.solve_resident(opt); w = time.time() - t
    print(nm, "wall %.3f" % w, sm['stop_reason_name'], "final", sm['final_cost'], flush=True)
    for r in rows: print("   it %d cost %.10e lin %d t_lin %.4f t_solve %.4f t_eval %.4f" % (r['iteration'], r['cost'], r['linear_iterations'], r['t_linearize_s'], r['t_solve_s'], r['t_eval_s']))
    P.close()
