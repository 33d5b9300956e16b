"""Cost of the first solve of a problem (the sparse solver's tree or the PCG hierarchy is built then) against later ones:
    python scripts/first_solve.py cfg4   ->  10M network: first one-iteration solve 0.030 s, later ones 0.016 s; 50M network (PCG): 0.205 / 0.069 s."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bundleadjustment_amd import capi, host, solver
net = host.synth_config(sys.argv[1] if len(sys.argv) > 1 else "cfg4").net
t0 = time.perf_counter()
P = solver.Problem(net)
t1 = time.perf_counter()
P.set_parameters()
opt = solver.default_options(); opt.max_num_iterations = 1
ts = []
for rep in range(3):
    P.set_parameters()
    a = time.perf_counter(); P.solve_resident(opt); ts.append(time.perf_counter() - a)
print("create %.3f s; one-iteration solves: first %.3f s, second %.3f s, third %.3f s" % (t1 - t0, ts[0], ts[1], ts[2]))
P.close()
