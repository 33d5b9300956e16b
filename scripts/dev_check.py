import sys, time, numpy as np
import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bundleadjustment_amd import host, capi, solver
from oracle import py_oracle as po
def show(tag, sm, rows):
    print(tag, sm['stop_reason_name'], sm['num_iterations'], "%.10e"%sm['initial_cost'], "%.10e"%sm['final_cost'], "lin", sm['total_linear_iterations'], "t %.3f"%sm['t_total_s'])
    for r in rows: print("  it %d ok=%d cost %.12e dc %.3e |g| %.3e step %.3e rho %.3f rad %.2e lin %d"%(r['iteration'],r['step_is_successful'],r['cost'],r['cost_change'],r['gradient_max_norm'],r['step_norm'],r['relative_decrease'],r['trust_region_radius'],r['linear_iterations']))
name, scale = sys.argv[1], float(sys.argv[2])
hn = host.synth_config(name, scale=scale)
net = hn.net
print(name, net.n_obs, net.n_sta, net.n_pts, net.n_cam)
P = solver.Problem(net)
P.set_parameters()
r, J = P.residual_jacobian()
ro, Jo = po.residual_jacobian_all(net)
print("resid max abs diff", np.abs(r-ro).max(), "J max rel diff", (np.abs(J-Jo)/(np.abs(Jo)+1e-300*0+1e-12*np.abs(Jo).max(axis=(0,1),keepdims=False))).max())
c, res = P.evaluate()
co, reso = po.evaluate(net)
print("cost", c, co, abs(c-co)/co, "res diff", np.abs(res-reso).max())
for ls, mode in ((capi.SOLVER_PCG_SCHUR, po.SCHUR_DENSE), (capi.SOLVER_DENSE_SCHUR, po.SCHUR_DENSE)):
    opt = solver.default_options(); opt.linear_solver = ls
    t=time.time()
    eop,xyz,iop,ap,sm,rows = P.solve(opt)
    print("gpu wall", time.time()-t)
    show("GPU ls=%d"%ls, sm, rows)
    if ls == capi.SOLVER_PCG_SCHUR:
        eo,xo,io,ao,smo,rowso = po.solve(net, opt, mode)
        show("ORACLE", smo, rowso)
    for a,b,nm in ((eop,eo,'eop'),(xyz,xo,'xyz'),(iop,io,'iop'),(ap,ao,'ap')):
        print("  ", nm, "max rel", np.max(np.abs(a-b)/np.maximum(np.abs(b),1e-3)))
    print("  cost rel per iter", [abs(x['cost']-y['cost'])/y['cost'] for x,y in zip(rows,rowso)])
