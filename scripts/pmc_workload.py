"""Short workload for the rocprofv3 counter passes: one network, a few LM iterations through the solver.

    rocprofv3 --pmc FETCH_SIZE -d gpurun_out/pmc_fetch --output-format csv -- python3 scripts/pmc_workload.py cfg4 1.0 2
    rocprofv3 --pmc WRITE_SIZE -d gpurun_out/pmc_write --output-format csv -- python3 scripts/pmc_workload.py cfg4 1.0 2

(each counter in a run of its own; scripts/summarise_profiles.py turns the two into profiles/pmc_traffic.json)."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bundleadjustment_amd import capi, host, solver  # noqa: E402

config = sys.argv[1] if len(sys.argv) > 1 else "cfg4"
scale = float(sys.argv[2]) if len(sys.argv) > 2 else 1.0
iters = int(sys.argv[3]) if len(sys.argv) > 3 else 2
net = host.synth_config(config, scale=scale).net
opt = solver.default_options()
opt.max_num_iterations = iters
opt.function_tolerance = opt.parameter_tolerance = opt.gradient_tolerance = 0.0
with solver.Problem(net) as P:
    P.set_parameters()
    sm, rows = P.solve_resident(opt)
    print(config, "solver", sm["linear_solver_used"], "iterations", sm["num_iterations"] - 1, "final cost", sm["final_cost"],
          "pcg iterations", sm["total_linear_iterations"])
