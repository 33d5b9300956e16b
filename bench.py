#!/usr/bin/env python3
"""Benchmark of the Levenberg-Marquardt inner loop on MI355X.

    python bench.py --gpus N --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N --steps K --warmup W

A step is one LM iteration (linearise + reduced solve + candidate evaluation) of the 10M-observation
synthetic network ("cfg4": 1 camera, 10k stations, 2M points, IOP + k1 k2 p1 p2 free) with the
parameters resident in HBM.  With N > 1 the same network is split by points over the ranks
(strong scaling) and the reduced camera system is summed with RCCL.  Rank 0 prints one JSON line.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec peak


def measured_traffic(kernel, config, scale, world):
    """HBM bytes per launch of `kernel` from the committed rocprofv3 --pmc summary (FETCH_SIZE / WRITE_SIZE
    passes of this same workload, corrected as the summary says); None when the run is not that workload."""
    path = os.path.join(ROOT, "profiles", "pmc_traffic.json")
    if world != 1 or scale != 1.0 or not os.path.exists(path):
        return None
    with open(path) as f:
        tab = json.load(f)
    if tab.get("config") != config:
        return None
    ent = tab.get("kernels", {}).get(kernel)
    return None if ent is None else ent["bytes_per_launch"]


def kernel_bytes(which, n_obs, n_pts, n_sta, n_blocks=0, n_pairs=0, ncam=0):
    """Algorithmic bytes one launch must move (per-unit figures x units).
    0 linearize      : per observation 56 (record) + 48 (station) + 24 (point) + 16 (residual) + 144 (coupling block W),
                       per point 72 (V, g), per station 216 (U, g).  It runs as two kernels (7 per point, 8 per
                       observation) that are also listed on their own; the record is read by both, which the
                       figure of the pair does not count twice
    1 point_apply    : per observation Js 96 + Jp 48 + station index 4 + p_i 48 + b 16, per point Vinv 48
    2 station_gather : per observation Js 96 + b 16 + index 4, per station q 48
    3 cost           : per observation record 56 + point 24
    4 schur_blocks   : every 144-B coupling block once (however many pairs it takes part in: re-reads are the kernel's
                       own doing), per 6x6 block 288 written twice (the block and its mirror image)
    5 pcg_iteration  : the block-sparse matrix once (288 B per block + 4-B column index), the border once,
                       and the six vectors of the iteration (x r z p q read, x r z p q written: 10 x 48 B per station);
                       the multilevel preconditioner's transfer operators are not counted (they are a design choice,
                       not something the iteration must move)"""
    if which == 0:
        return 288.0 * n_obs + 72.0 * n_pts + 216.0 * n_sta
    if which == 7:  # per-point kernel alone: record, station, point in; V and g out
        return 128.0 * n_obs + 72.0 * n_pts
    if which == 8:  # per-observation kernel alone: record, station, point in; residual and coupling block out
        return 288.0 * n_obs
    if which == 1:
        return 212.0 * n_obs + 48.0 * n_pts
    if which == 2:
        return 116.0 * n_obs + 48.0 * n_sta
    if which == 3:
        return 80.0 * n_obs
    if which == 4:
        return 144.0 * n_obs + 2 * 288.0 * n_blocks
    if which == 5:
        return 292.0 * n_blocks + 48.0 * ncam * n_sta + 480.0 * n_sta
    return 292.0 * n_blocks + 48.0 * ncam * n_sta + 96.0 * n_sta  # 6 spmv: matrix, border, p read, q written


JS_PAIR = "Jacobian + Schur assembly (k_point_elim + k_obs_rows + k_schur_blocks)"
KERNEL_NAMES = {0: "linearisation (k_point_elim + k_obs_rows)", 7: "k_point_elim", 8: "k_obs_rows", 1: "k_point_apply", 2: "k_station_pass<GATHER>", 3: "k_cost_obs",
                13: "sparse Cholesky factorisation (front assembly + k_front_diag + k_front_trsm + k_front_syrk per panel and level)",
                14: "sparse Cholesky factorisation + forward and backward solves on the tree",
                4: "k_schur_blocks", 5: "pcg iteration (k_bsr_spmv + k_cg2_update1 + k_amg_restrict x levels + k_amg_prolong + k_amg_update2)", 6: "k_bsr_spmv"}


def structure_counts(net):
    """Observation pairs and distinct station pairs (6x6 blocks incl. the diagonal ones) of a network."""
    order = np.argsort(net.obs_pt, kind="stable")
    deg = np.bincount(net.obs_pt, minlength=net.n_pts).astype(np.int64)
    n_pairs = int(np.sum(deg * (deg - 1) // 2))
    sta = net.obs_sta[order].astype(np.int64)
    start = np.concatenate([[0], np.cumsum(deg)])
    keys = []
    dmax = int(deg.max()) if len(deg) else 0
    if len(deg) and np.all(deg == dmax):  # uniform degree: vectorised
        m = sta.reshape(-1, dmax)
        for a in range(dmax):
            for b in range(a + 1, dmax):
                lo, hi = np.minimum(m[:, a], m[:, b]), np.maximum(m[:, a], m[:, b])
                keys.append(np.unique(lo * net.n_sta + hi))
        uniq = np.unique(np.concatenate(keys)) if keys else np.zeros(0, np.int64)
        off = int(np.sum(uniq // net.n_sta != uniq % net.n_sta))
    else:
        seen = set()
        for j in range(net.n_pts):
            ss = sta[start[j]:start[j + 1]]
            for a in range(len(ss)):
                for b in range(a + 1, len(ss)):
                    if ss[a] != ss[b]:
                        seen.add((min(ss[a], ss[b]), max(ss[a], ss[b])))
        off = len(seen)
    return n_pairs, 2 * off + net.n_sta


def cpu_baseline(config, cpu_scale, pcg_rtol, omp=False):
    """The CPU oracle (plain C; one thread, or its OpenMP build on all the cores this process may use) on a shrunk
    copy of the same network."""
    from bundleadjustment_amd import capi, host
    from oracle import py_oracle as po
    hn = host.synth_config(config, scale=cpu_scale)
    net = hn.net
    opt = capi.default_options(linear_solver=capi.SOLVER_PCG_SCHUR, pcg_rtol=pcg_rtol, pcg_max_iters=5000)
    opt.max_num_iterations = 2
    opt.function_tolerance = 0.0
    opt.parameter_tolerance = 0.0
    opt.gradient_tolerance = 0.0
    t0 = time.time()
    *_, sm, rows = po.solve(net, opt, po.SCHUR_PCG, omp=omp)
    t_solve = time.time() - t0
    iters = max(1, sm["num_iterations"] - 1)
    t0 = time.time()
    po.linearize_once(net, omp=omp)
    t_lin = time.time() - t0
    return net.n_obs, iters / t_solve, net.n_obs / t_lin, sm["total_linear_iterations"] / iters


def cpu_full_size(config, pcg_rtol, budget_s, predicted_s, gpu_costs=None):
    """One LM iteration of the all-cores oracle on the WHOLE network (not a sample), when the sample predicts that it
    fits the time budget.  Returns a dict for the JSON line.  gpu_costs: the costs of iterations 0 and 1 of the GPU's
    solve from the same start values; the oracle's are compared with them (relative 1e-9) and the result is recorded."""
    if predicted_s > budget_s:
        return {"value": None, "skipped": f"predicted {predicted_s:.0f} s for one iteration, budget {budget_s:.0f} s"}
    from bundleadjustment_amd import capi, host
    from oracle import py_oracle as po
    net = host.synth_config(config).net
    opt = capi.default_options(linear_solver=capi.SOLVER_PCG_SCHUR, pcg_rtol=pcg_rtol, pcg_max_iters=20000)
    opt.max_num_iterations = 1
    opt.function_tolerance = opt.parameter_tolerance = opt.gradient_tolerance = 0.0
    t0 = time.time()
    *_, sm, rows = po.solve(net, opt, po.SCHUR_PCG, omp=True)
    t_solve = time.time() - t0
    t0 = time.time()
    po.linearize_once(net, omp=True)
    t_lin = time.time() - t0
    # the solve holds the evaluation of the start point and the Jacobi-scaling pass besides the iteration itself
    # (three linearisations in all): they are taken off with the measured time of one
    t_iter = max(t_solve - 2.0 * t_lin, 1e-9)
    check = None
    if gpu_costs is not None and len(rows) >= 2:
        cpu_costs = [rows[0]["cost"], rows[1]["cost"]]
        rel = [abs(a - b) / abs(b) for a, b in zip(gpu_costs, cpu_costs)]
        check = {"iteration_costs_gpu": list(gpu_costs), "iteration_costs_cpu": cpu_costs, "relative_difference": rel,
                 "tolerance": 1e-9, "passed": bool(max(rel) <= 1e-9)}
    return {"value": 1.0 / t_iter, "unit": "iter/s", "cores": po.omp_threads(), "kind": "port", "full_size_parity": check,
            "sample": f"the OpenMP build of the oracle on the whole network ({net.n_obs} observations), one LM iteration measured: "
                      f"{t_iter:.1f} s with {sm['total_linear_iterations']} block-Jacobi PCG iterations (solve {t_solve:.1f} s minus two "
                      f"of the three linearisations at {t_lin:.2f} s each); Jacobian pass {net.n_obs / t_lin:.3e} obs/s",
            "jacobian_obs_per_s": net.n_obs / t_lin, "sample_observations": net.n_obs}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    # defaults: the eight iterations this network needs to converge from the generator's start values (the solve also
    # holds one evaluation of the start point and the Jacobi scaling pass, as every real solve does)
    ap.add_argument("--steps", type=int, default=8)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--config", default="cfg4")
    ap.add_argument("--scale", type=float, default=1.0, help="shrink the network (stations and points) for quick runs")
    ap.add_argument("--pcg-rtol", type=float, default=1e-12, help="PCG stopping tolerance; 1e-12 is what the parity tests use")
    ap.add_argument("--solver", choices=("auto", "pcg", "sparse"), default="auto",
                    help="reduced solve of the networks beyond the dense limit: what BA_SOLVER_AUTO picks (the sparse direct "
                         "Cholesky up to order 250 000, the PCG beyond), or one of the two by name")
    ap.add_argument("--cpu-scale", type=float, default=0.02, help="size of the CPU-baseline sample relative to the config")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-full-budget", type=float, default=90.0,
                    help="seconds the one full-size iteration of the all-cores CPU baseline may take (0: skip it)")
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            sys.exit(f"bench.py --gpus {args.gpus} must be launched with torch.distributed.run --nproc-per-node {args.gpus}")
        args.gpus = world

    import torch
    import torch.distributed as dist
    if not torch.cuda.is_available():
        sys.exit("bench.py needs an MI355X: the solver has no CPU path")
    # BA_BENCH_COMM=gloo: rehearsal of the N > 1 path on a box with fewer GPUs than ranks (RCCL wants a GPU per rank):
    # the ranks share the GPUs, torch.distributed runs over gloo and the solver's reductions are staged through the host.
    rehearsal = world > 1 and os.environ.get("BA_BENCH_COMM") == "gloo"
    if rehearsal:
        local_rank %= torch.cuda.device_count()
    torch.cuda.set_device(local_rank)
    from bundleadjustment_amd import capi, host, solver

    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if rehearsal:
            dist.init_process_group("gloo")
        else:
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
    red_dev = "cpu" if rehearsal else "cuda"  # where the few scalars that the ranks compare live

    hn = host.synth_config(args.config, scale=args.scale)
    full = hn.net
    # the rank's points and their observations, cut by the host library (the same partition as sharding.shard_network)
    net = hn.shard(rank, world)[0].net if world > 1 else full
    P = solver.Problem(net)
    if world > 1 and rehearsal:
        def reduce_fn(buf, op):
            dist.all_reduce(torch.from_numpy(buf), op=dist.ReduceOp.MAX if op == 1 else dist.ReduceOp.SUM)
        P.set_allreduce_host_staged(reduce_fn)
    elif world > 1:
        ids = [solver.rccl_unique_id() if rank == 0 else None]
        dist.broadcast_object_list(ids, src=0)
        P.use_rccl(world, rank, ids[0])
    if world > 1:
        # rows of the PCG's product split over the ranks when the library's model says it pays (it does not at 10 000
        # stations); BA_BENCH_PCG_ROWS=always|never overrides
        P.shard_reduced_solve(rank, world, {"always": 1, "never": -1}.get(os.environ.get("BA_BENCH_PCG_ROWS", ""), 0))
    P.set_parameters()

    free_cam = int(sum(bin(int(a)).count("1") + bin(int(b) & 7).count("1") for a, b in zip(full.ap_free_mask, full.iop_free_mask)))
    n_red = 6 * full.n_sta + free_cam
    dense = n_red <= capi.AUTO_DENSE_ORDER  # what BA_SOLVER_AUTO picks: the two small networks
    sparse = not dense and (args.solver == "sparse" or (args.solver == "auto" and n_red <= capi.AUTO_SPARSE_ORDER))
    ls_main = capi.SOLVER_DENSE_SCHUR if dense else capi.SOLVER_SPARSE_SCHUR if sparse else capi.SOLVER_PCG_SCHUR

    def options(iters, ls=None):
        o = solver.default_options()
        o.linear_solver = ls_main if ls is None else ls
        o.pcg_rtol = args.pcg_rtol
        o.pcg_max_iters = 20000
        o.max_num_iterations = iters
        # run exactly `iters` iterations: the stopping tests are off, the arithmetic is unchanged
        o.function_tolerance = 0.0
        o.parameter_tolerance = 0.0
        o.gradient_tolerance = 0.0
        return o

    def barrier():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    if args.warmup > 0:
        P.solve_resident(options(args.warmup))
        P.set_parameters()
    barrier()
    t0 = time.perf_counter()
    sm, rows = P.solve_resident(options(args.steps))
    barrier()
    elapsed = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device=red_dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    steps_done = sm["num_iterations"] - 1
    if steps_done != args.steps:
        sys.exit(f"solver ran {steps_done} iterations instead of {args.steps}: {sm}")

    per_rank = None
    if world > 1:
        # Where the time of a rank goes, from one more solve outside the timed region with events around every
        # reduction over the ranks: host-clock phases of the iteration rows, and the reductions inside them (a
        # collective also waits for the slowest rank, so load imbalance shows as reduction time).
        P.set_parameters()
        P.reduction_stats()
        P.reduction_timing(True)
        sm2, rows2 = P.solve_resident(options(args.steps))
        n_red, red_bytes, red_s = P.reduction_stats()
        P.reduction_timing(False)
        mine = {"rank": rank, "observations": int(net.n_obs), "points": int(net.n_pts),
                "linearise_ms_per_step": 1e3 * sum(r["t_linearize_s"] for r in rows2) / args.steps,
                "solve_ms_per_step": 1e3 * sum(r["t_solve_s"] for r in rows2) / args.steps,
                "evaluate_ms_per_step": 1e3 * sum(r["t_eval_s"] for r in rows2) / args.steps,
                "reductions_per_step": n_red / args.steps, "reduction_MB_per_step": red_bytes / args.steps / 1e6,
                "reduction_ms_per_step": 1e3 * red_s / args.steps,
                "pcg_product_rows": list(P.reduced_solve_rows())}
        per_rank = [None] * world
        dist.all_gather_object(per_rank, mine)

    if dense:
        # the reduced system is factored on the matrix cores: report that kernel chain against the fp64 MFMA peak
        n_red = 6 * full.n_sta + free_cam
        t_fac, t_lin = P.time_kernel(10, 5), P.time_kernel(0, 10)
        flops = n_red ** 3 / 3.0
        out = {"metric": "LM iter/s", "value": args.steps / elapsed, "unit": "iter/s", "n_gpus": world, "steps": args.steps,
               "warmup": args.warmup, "ms_per_step": 1e3 * elapsed / args.steps, "higher_is_better": True, "scaling": "strong",
               "vs_baseline": None, "dtype": "f64", "data": "synthetic",
               "config": {"workload": f"{args.config}: {full.n_cam} camera(s), {full.n_sta} stations, {full.n_pts} points, {full.n_obs} "
                                      f"observations, point elimination + dense Cholesky of order {n_red} on the matrix cores",
                          "scale": args.scale, "sharding": f"points/{world}"},
               "jacobian_obs_per_s": full.n_obs / t_lin, "final_cost": sm["final_cost"],
               "roofline": {"bound": "mfma", "kernel": "dense Cholesky (k_chol_diag + k_gemm_nt<TRSM> + k_gemm_nt<SYRK>)",
                            "achieved": flops / t_fac / 1e12, "peak": 78.6, "unit": "TFLOP/s", "frac": flops / t_fac / 78.6e12,
                            "traffic": None, "ms": 1e3 * t_fac, "flops": flops}}
        if per_rank is not None:
            out["per_rank"] = per_rank
        if rank == 0:
            print(json.dumps(out), flush=True)
        P.close()
        if world > 1:
            dist.destroy_process_group()
        return

    # the other sparse solver on the same network, outside the timed region: its time per step and its final cost
    other = None
    if world == 1:
        ls_other = capi.SOLVER_PCG_SCHUR if sparse else capi.SOLVER_SPARSE_SCHUR
        try:
            P.set_parameters()
            P.solve_resident(options(1, ls_other))  # its structures (hierarchy or tree) are built once, untimed
            P.set_parameters()
            barrier()
            t1 = time.perf_counter()
            sm_o, rows_o = P.solve_resident(options(args.steps, ls_other))
            barrier()
            other = {"solver": "pcg (multilevel preconditioner)" if sparse else "sparse Cholesky (nested dissection, multifrontal)",
                     "ms_per_step": 1e3 * (time.perf_counter() - t1) / args.steps, "final_cost": sm_o["final_cost"],
                     "linear_iterations_per_step": sm_o["total_linear_iterations"] / args.steps,
                     "final_cost_relative_difference": abs(sm_o["final_cost"] - sm["final_cost"]) / sm["final_cost"]}
        except solver.BaError as e:
            other = {"error": str(e)}
        # leave the problem as the timed solve left its structures: the kernels below are timed on the main solver's
        P.set_parameters()
        P.solve_resident(options(1))
        P.set_parameters()

    # per-kernel timing with HIP events on the solver's stream (live, same process)
    which_all = (0, 7, 8, 4, 14, 13, 3) if sparse else (0, 7, 8, 4, 6, 5, 3)
    ktime = {w: P.time_kernel(w, 10) for w in which_all}
    if world > 1:
        t = torch.tensor([ktime[w] for w in which_all], dtype=torch.float64, device=red_dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        ktime = {w: float(t[i].item()) for i, w in enumerate(which_all)}
    lin_iters = sm["total_linear_iterations"]
    n_lin = sum(1 for r in rows if r["step_is_successful"]) + 1
    calls = {0: n_lin, 7: n_lin, 8: n_lin, 4: n_lin, 5: lin_iters, 6: lin_iters, 3: args.steps, 13: args.steps, 14: args.steps}
    tree = P.sparse_solver_stats() if sparse else None
    n_pairs, n_blocks = structure_counts(net)
    ncam = int(sum(bin(int(a)).count("1") + bin(int(b) & 7).count("1") for a, b in zip(net.ap_free_mask, net.iop_free_mask)))
    full_pairs, full_blocks = (n_pairs, n_blocks) if world == 1 else structure_counts(full)

    def kb(w):
        # the PCG iteration runs on the whole reduced system on every rank; the other kernels on the shard
        if w in (5, 6):
            return kernel_bytes(w, full.n_obs, full.n_pts, full.n_sta, full_blocks, full_pairs, ncam)
        return kernel_bytes(w, net.n_obs, net.n_pts, net.n_sta, n_blocks, n_pairs, ncam)

    # The roofline entry is the pair the benchmark definition asks about: the Jacobian kernels and the Schur assembly
    # (residuals + analytic Jacobians + point elimination + the block-sparse reduced system), as one figure: their
    # algorithmic bytes over the sum of their launch times.  Every kernel is listed on its own below.
    step_s = elapsed / args.steps
    t_pair = ktime[7] + ktime[8] + ktime[4]
    b_pair = kb(0) + kb(4)
    achieved = b_pair / t_pair / 1e9
    traffic = [measured_traffic(KERNEL_NAMES[w], args.config, args.scale, world) for w in (7, 8, 4)]
    kernels = []
    for w in which_all:
        if w in (13, 14):
            # not a memory-bound kernel: the operations of the factorisation against the fp64 peak of the matrix cores
            kernels.append({"kernel": KERNEL_NAMES[w], "ms": ktime[w] * 1e3, "launches_in_timed_region": calls[w],
                            "flops": tree["factor_flops"], "TFLOPs": tree["factor_flops"] / ktime[w] / 1e12,
                            "frac_of_mfma_peak": tree["factor_flops"] / ktime[w] / 78.6e12, "share_of_step_time": ktime[w] * calls[w] / elapsed,
                            "note": "a chain of %d panel steps (diagonal block, rows below, trailing update) over %d levels of the "
                                    "elimination tree: bound by the length of the chain, not by the operations" % (tree["critical_panels"], tree["levels"])})
            continue
        b = kb(w)
        kernels.append({"kernel": KERNEL_NAMES[w], "ms": ktime[w] * 1e3, "launches_in_timed_region": calls[w],
                        "algorithmic_GB": b / 1e9, "GBps": b / ktime[w] / 1e9, "frac": b / ktime[w] / 1e9 / HBM_PEAK_GBS,
                        "traffic_GB": (lambda t: None if t is None else t / 1e9)(measured_traffic(KERNEL_NAMES[w], args.config, args.scale, world)),
                        # of the timed region (rows 0 and 5 are sums of other rows and are not added to anything)
                        "share_of_step_time": ktime[w] * calls[w] / elapsed})

    # What N GPUs can do with this network, from this run's own measurements (N = 1: measured; N > 1: the work that
    # divides is this rank's share times N).  Points and their observations are divided; the reduced solve runs on
    # every rank on the summed system; four sums over the ranks per LM iteration, of which one is large.
    lin_ms = 1e3 * sum(r["t_linearize_s"] for r in rows[1:]) / max(len(rows) - 1, 1)
    solve_ms = 1e3 * sum(r["t_solve_s"] + r["t_eval_s"] for r in rows[1:]) / max(len(rows) - 1, 1)
    pcg_ms = 1e3 * ktime[14] if sparse else 1e3 * ktime[5] * lin_iters / args.steps
    divides_ms = world * (1e3 * (ktime[7] + ktime[8] + ktime[4] + ktime[3]))  # Jacobians, Schur assembly, candidate cost
    red_bytes = [8.0 * (39 * full.n_sta + 3 * ncam + 1), 16.0, 8.0 * (36 * full_blocks + 6 * full.n_sta * ncam + max(64, ncam * ncam)), 64.0]

    def t_allreduce_ms(nbytes, n):  # ring over xGMI links of about 150 GB/s + 30 us per collective
        return 0.0 if n == 1 else 1e3 * (30e-6 + 2.0 * (n - 1) / n * nbytes / 150e9)
    # PCG, hierarchy, folds, back substitution ...: the rest of the measured step (less the model's own price of the sums
    # when this run had some)
    other_ms = max(1e3 * step_s - divides_ms / world - sum(t_allreduce_ms(b, world) for b in red_bytes), 0.0)
    scaling_model = {
        "assumptions": "points and observations divide by N (Jacobian kernels, Schur assembly, candidate cost); everything else "
                       "of a step -- the solve of the summed reduced system (sparse Cholesky or PCG), folds, back substitution -- is counted as "
                       "not dividing (the back substitution does divide: the prediction errs on the slow side); four all-reduces per "
                       "LM iteration priced as a ring over 150 GB/s links + 30 us each",
        "measured_on_n_gpus": world, "ms_per_step_measured": 1e3 * step_s,
        "divides_ms_at_1_gpu": divides_ms, "does_not_divide_ms": other_ms, "reduced_solve_ms_per_step": pcg_ms,
        "linearise_phase_ms": lin_ms, "solve_phase_ms": solve_ms,
        "reduction_bytes_per_lm_iteration": red_bytes,
        "predicted_ms_per_step": {str(n): divides_ms / n + other_ms + sum(t_allreduce_ms(b, n) for b in red_bytes) for n in (1, 2, 4, 8)},
    }
    scaling_model["predicted_speedup_at_8"] = scaling_model["predicted_ms_per_step"]["1"] / scaling_model["predicted_ms_per_step"]["8"]

    out = {
        "metric": "LM iter/s (10M-obs synthetic network)" if args.config == "cfg4" and args.scale == 1.0 else f"LM iter/s ({args.config}, scale {args.scale})",
        "value": args.steps / elapsed,
        "unit": "iter/s",
        "n_gpus": world,
        "steps": args.steps,
        "warmup": args.warmup,
        "ms_per_step": 1e3 * elapsed / args.steps,
        "higher_is_better": True,
        "scaling": "strong",
        "vs_baseline": None,
        "dtype": "f64",
        "data": "synthetic",
        "config": {"workload": f"{args.config}: {full.n_cam} camera(s), {full.n_sta} stations, {full.n_pts} points, "
                               f"{full.n_obs} observations, self-calibrating (IOP + k1 k2 p1 p2), point elimination + " +
                               ("sparse direct Cholesky of the block-sparse reduced system (nested dissection, multifrontal, dense fronts on the "
                                "matrix cores)" if sparse else
                                "PCG with a multilevel (smoothed aggregation) preconditioner on the block-sparse reduced system"),
                   "scale": args.scale, "pcg_rtol": args.pcg_rtol, "sharding": f"points/{world}",
                   "reduced_solver": "sparse_cholesky" if sparse else "pcg"},
        "jacobian_obs_per_s": full.n_obs / ktime[0],
        "pcg_iterations_per_step": lin_iters / args.steps,
        "final_cost": sm["final_cost"],
        "roofline": {"bound": "hbm", "kernel": JS_PAIR, "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                     "frac": achieved / HBM_PEAK_GBS, "traffic": None if any(t is None for t in traffic) else sum(traffic),
                     "ms": 1e3 * t_pair, "algorithmic_bytes": b_pair},
        "kernels": kernels,
        "scaling_model": scaling_model,
    }
    if tree is not None:
        out["elimination_tree"] = tree
    if other is not None:
        out["other_reduced_solver"] = other
    if per_rank is not None:
        out["per_rank"] = per_rank
    parity_failed = None
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        # the CPU legs take a minute or two and print nothing: a line on stderr every 30 s shows a watcher that the run is alive
        import threading
        cpu_done = threading.Event()

        def heartbeat():
            t_cpu = time.perf_counter()
            while not cpu_done.wait(30.0):
                print(f"bench.py: CPU baseline running ({time.perf_counter() - t_cpu:.0f} s)", file=sys.stderr, flush=True)
        threading.Thread(target=heartbeat, daemon=True).start()
        # Three legs.  The sample legs (a shrunk copy of the network, one thread and all cores) take seconds; their
        # extrapolation to the full network is linear in the observations and is known to flatter the CPU, because the
        # oracle's block-Jacobi PCG needs more iterations on the larger system.  The leg that counts is therefore the
        # third: one LM iteration of the all-cores build on the WHOLE network, measured.  It is also a check: the oracle's
        # costs of iterations 0 and 1 must equal the GPU's from the same start values to 1e-9.
        n_cpu, it_s, obs_s, cg = cpu_baseline(args.config, args.cpu_scale * args.scale, args.pcg_rtol)
        one_thread = {
            "value": it_s * n_cpu / full.n_obs, "unit": "iter/s", "cores": 1, "kind": "port",
            "sample": f"oracle (C, 1 thread), same generator at {n_cpu} observations: {it_s:.3f} LM iter/s measured with "
                      f"{cg:.0f} PCG iterations per step, scaled linearly in observations to {full.n_obs} (an extrapolation: "
                      f"optimistic for the CPU, see extrapolation_error); Jacobian pass {obs_s:.3e} obs/s",
            "jacobian_obs_per_s": obs_s, "measured_iter_per_s_on_sample": it_s, "sample_observations": n_cpu}
        out["cpu_baseline"] = one_thread
        out["extra"] = {}
        try:  # the other legs are reported only: their absence must not cost the benchmark line
            from oracle import py_oracle as po
            n_cpu, it_s, obs_s, cg = cpu_baseline(args.config, args.cpu_scale * args.scale, args.pcg_rtol, omp=True)
            all_cores = {
                "value": it_s * n_cpu / full.n_obs, "unit": "iter/s", "cores": po.omp_threads(), "kind": "port",
                "sample": f"the same oracle built with OpenMP (loops over observations and points; OMP_NUM_THREADS as set), "
                          f"same sample of {n_cpu} observations: {it_s:.3f} LM iter/s measured, scaled linearly to {full.n_obs}; "
                          f"Jacobian pass {obs_s:.3e} obs/s",
                "jacobian_obs_per_s": obs_s, "measured_iter_per_s_on_sample": it_s, "sample_observations": n_cpu}
            out["extra"]["cpu_all_cores_extrapolated_from_sample"] = all_cores
            if args.scale == 1.0 and args.cpu_full_budget > 0.0:
                gpu_costs = [rows[0]["cost"], rows[1]["cost"]] if len(rows) >= 2 else None
                full_leg = cpu_full_size(args.config, args.pcg_rtol, args.cpu_full_budget, full.n_obs / (it_s * n_cpu), gpu_costs)
                if full_leg.get("value"):
                    # the measured figure is the baseline; the extrapolated ones move to `extra` with their error beside them
                    err = all_cores["value"] / full_leg["value"]
                    all_cores["extrapolation_error"] = f"{err:.1f} x the measured full-size figure"
                    one_thread["extrapolation_error"] = f"the all-cores extrapolation from the same sample is {err:.1f} x its measured full-size figure; the same holds here"
                    out["extra"]["cpu_1_thread_extrapolated_from_sample"] = one_thread
                    out["cpu_baseline"] = full_leg
                else:
                    out["extra"]["cpu_all_cores_full_size"] = full_leg
                chk = full_leg.get("full_size_parity")
                if chk is not None and not chk["passed"]:
                    parity_failed = f"full-size parity check failed: GPU costs {chk['iteration_costs_gpu']} vs CPU {chk['iteration_costs_cpu']}"
        except (RuntimeError, OSError, MemoryError) as e:
            out["extra"]["cpu_all_cores_error"] = str(e)
        cpu_done.set()
    if rank == 0:
        print(json.dumps(out), flush=True)
    P.close()
    if world > 1:
        dist.destroy_process_group()
    if parity_failed:
        sys.exit("bench.py: " + parity_failed)


if __name__ == "__main__":
    main()
