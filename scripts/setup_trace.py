import sys, time
sys.path.insert(0, "/root/repo")
from bundleadjustment_amd import host, solver
net = host.synth_config("cfg4").net
for k in range(2):
    t = time.time()
    P = solver.Problem(net)
    print("ba_create %.3f s" % (time.time() - t), flush=True)
    P.close()
