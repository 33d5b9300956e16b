"""Dense Cholesky of the reduced camera system on the matrix cores: time and fp64 rate (ba_time_kernel 10 / 11)."""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bundleadjustment_amd import capi, host, solver

PEAK = 78.6e12  # fp64 matrix peak of MI355X, flop/s
cases = [("cfg1", dict()), ("cfg2", dict()), ("4000", dict(n_cam=1, n_sta=666, n_pts=20000, rays_per_point=6, ap_free_mask=0, iop_free_mask=0)),
         ("8190", dict(n_cam=1, n_sta=1365, n_pts=40000, rays_per_point=6, ap_free_mask=0, iop_free_mask=0)),
         ("30000", dict(n_cam=1, n_sta=5000, n_pts=100000, rays_per_point=6, ap_free_mask=0, iop_free_mask=0))]
if len(sys.argv) > 1:
    cases = [c for c in cases if c[0] in sys.argv[1:]]
for name, kw in cases:
    net = (host.synth_config(name) if name.startswith("cfg") else host.synth(**kw)).net
    with solver.Problem(net) as P:
        P.set_parameters()
        free = sum(bin(int(a)).count("1") + bin(int(b) & 7).count("1") for a, b in zip(net.ap_free_mask, net.iop_free_mask))
        n = 6 * net.n_sta + free
        tf = P.time_kernel(10, 5)
        ts = P.time_kernel(11, 5)
        flops = n ** 3 / 3.0
        print("%-5s n = %5d  factorisation %8.3f ms  %6.2f TFLOP/s  (%.1f %% of %.1f)   with solves %8.3f ms" %
              (name, n, tf * 1e3, flops / tf * 1e-12, 100 * flops / tf / PEAK, PEAK * 1e-12, ts * 1e3), flush=True)
