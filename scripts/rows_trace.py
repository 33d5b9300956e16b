"""Phase clocks of the observation kernel that also forms the station sums (ba_time_kernel, which = 12), and its time."""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bundleadjustment_amd import host, solver

name = sys.argv[1] if len(sys.argv) > 1 else "cfg4"
scale = float(sys.argv[2]) if len(sys.argv) > 2 else 1.0
net = host.synth_config(name, scale=scale).net
with solver.Problem(net) as P:
    P.set_parameters()
    for which, nm in ((7, "k_point_elim"), (8, "k_obs_rows"), (4, "k_schur_blocks")):
        print("%-16s %8.1f us" % (nm, 1e6 * P.time_kernel(which, 20)))
    P.time_kernel(12, 1)
