import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
mode = sys.argv[1]
if mode != "none":
    import torch
    print("torch", torch.__version__, "avail", torch.cuda.is_available() if mode == "init" else "-")
    if mode == "init":
        torch.cuda.set_device(0); x = torch.zeros(4, device="cuda"); print(x.sum().item())
from bundleadjustment_amd import host, solver
net = host.synth(n_cam=1, n_sta=3, n_pts=8, rays_per_point=3).net
try:
    with solver.Problem(net) as P:
        print("ok", P.evaluate()[0])
except Exception as e:
    print("ERR", e)
os.system("grep -E 'amdhip|rccl|hsa-runtime' /proc/%d/maps | awk '{print $6}' | sort -u" % os.getpid())
