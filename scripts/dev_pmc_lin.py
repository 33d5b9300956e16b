"""Development: run the two linearisation kernels a few times on cfg4 (for rocprofv3 --pmc passes)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bundleadjustment_amd import host, solver
scale = float(sys.argv[1]) if len(sys.argv) > 1 else 1.0
net = host.synth_config("cfg4", scale=scale).net
P = solver.Problem(net); P.set_parameters()
for which in (7, 8):
    print(which, P.time_kernel(which, 3), flush=True)
