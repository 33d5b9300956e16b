"""Two ranks sharing the one GPU of the test box: each owns a shard of the points and the reduced
camera system is summed over the ranks (gloo, staged through the host).  The result must match the
CPU oracle's solve of the whole network like a single-rank run does."""
import os
import socket

import numpy as np
import pytest
import torch.multiprocessing as mp

pytestmark = pytest.mark.gpu


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _make_net(n_cam=1):
    from bundleadjustment_amd import host
    # few points per station pair, so that the two shards see different sets of station pairs
    return host.synth(n_cam=n_cam, n_sta=120, n_pts=500, rays_per_point=4, seed=77, ap_free_mask=host.AP_K1K2P1P2,
                      control_every=8).net


def _box(net):
    """A box on xp yp c whose upper bound on camera 0's xp is active at the solution."""
    lo = np.tile([-100.0, -100.0, 0.0], (net.n_cam, 1))
    hi = np.tile([100.0, 100.0, np.finfo(np.float64).max], (net.n_cam, 1))
    hi[0, 0] = net.truth["iop"][0, 0] - 1.0
    return lo, hi


def _options(ls, bounded):
    from bundleadjustment_amd import capi
    opt = capi.default_options(linear_solver=ls)
    if bounded:  # LM damping, so that the projected steps keep descending (see test_gpu_bounds.py)
        opt.min_lm_diagonal, opt.max_lm_diagonal, opt.max_num_iterations = 1e-6, 1e32, 12
    return opt


def _worker(rank, world, port, ls, n_cam, out, bounded=False):
    import torch
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.cuda.set_device(0)
    from bundleadjustment_amd import sharding, solver
    net = _make_net(n_cam)
    shard, sl = sharding.shard_network(net, rank, world)

    def reduce_fn(buf, op):
        t = torch.from_numpy(buf)
        dist.all_reduce(t, op=dist.ReduceOp.MAX if op == 1 else dist.ReduceOp.SUM)

    with solver.Problem(shard) as P:
        P.set_allreduce_host_staged(reduce_fn)
        if bounded:
            P.set_iop_bounds(*_box(net))
        opt = _options(ls, bounded)
        eop, xyz, iop, ap, sm, rows = P.solve(opt)
    gathered = [None] * world
    dist.all_gather_object(gathered, (sl.start, sl.stop, xyz))
    if rank == 0:
        full_xyz = np.zeros_like(net.xyz)
        for a, b, x in gathered:
            full_xyz[a:b] = x
        out.put((eop, full_xyz, iop, ap, sm, rows))
    dist.destroy_process_group()


# one camera: the fused single-camera border and the camera coarse space; three cameras: the general border
@pytest.mark.parametrize("ls,n_cam,bounded", [(0, 1, False), (1, 1, False), (2, 1, False), (0, 3, False), (1, 3, False), (1, 1, True)])
def test_two_ranks_match_oracle(ls, n_cam, bounded):
    from oracle import py_oracle as po
    from bundleadjustment_amd import capi
    from util import assert_solve_parity
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, ls, n_cam, q, bounded)) for r in range(2)]
    for p in procs:
        p.start()
    res = q.get(timeout=240)
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    net = _make_net(n_cam)
    opt = _options(ls, bounded)
    o = po.solve(net, opt, po.SCHUR_DENSE, iop_bounds=_box(net) if bounded else None)
    # the summary sizes of a shard count its own points only: compare costs, steps and parameters
    g = list(res)
    for k in ("num_residuals_reduced", "num_parameters_reduced", "num_effective_parameters_reduced", "num_parameters",
              "num_residuals"):
        g[4][k] = o[4][k]
    assert_solve_parity(tuple(g), o)


def test_rccl_single_rank():
    """The RCCL callback itself (one rank: a communicator of size 1) leaves the solve unchanged."""
    from bundleadjustment_amd import solver
    net = _make_net()
    opt = solver.default_options()
    opt.linear_solver = 1
    with solver.Problem(net) as P:
        a = P.solve(opt)
    with solver.Problem(net) as P:
        P.use_rccl(1, 0, solver.rccl_unique_id())
        b = P.solve(opt)
    for x, y in zip(a[:4], b[:4]):
        assert np.array_equal(x, y)
    assert a[4]["final_cost"] == b[4]["final_cost"]
