"""The N > 1 path on CPU: world_size-2 gloo run of the host-side sharding logic.  Every rank takes
its shard of the points, the shards are checked to partition the network, and the sum over ranks
(gloo all_reduce) of the shard costs equals the cost of the whole network (the oracle is the
checker on both sides)."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from bundleadjustment_amd import host, sharding
from oracle import py_oracle as po


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, out):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    net = host.synth(n_cam=2, n_sta=12, n_pts=200, rays_per_point=5, seed=33).net
    shard, sl = sharding.shard_network(net, rank, world)
    # shard consistency
    assert shard.n_sta == net.n_sta and shard.n_cam == net.n_cam
    assert shard.n_pts == sl.stop - sl.start
    sel = (net.obs_pt >= sl.start) & (net.obs_pt < sl.stop)
    assert shard.n_obs == int(sel.sum())
    assert np.array_equal(shard.obs_x, net.obs_x[sel]) and np.array_equal(shard.obs_pt, net.obs_pt[sel] - sl.start)
    # cost: image rows and point priors are local, the IOP priors are counted once
    cost, res = po.evaluate(shard, apply_loss=True)
    iop_rows = res[2 * shard.n_obs + 3 * shard.n_pts:]
    local = cost - 0.5 * float(np.sum(iop_rows ** 2))
    t = torch.tensor([local, float(shard.n_obs), float(shard.n_pts)], dtype=torch.float64)
    dist.all_reduce(t)
    full_cost, full_res = po.evaluate(net, apply_loss=True)
    full_iop = 0.5 * float(np.sum(full_res[2 * net.n_obs + 3 * net.n_pts:] ** 2))
    ok = abs(t[0].item() + full_iop - full_cost) <= 1e-12 * full_cost
    ok = ok and int(t[1].item()) == net.n_obs and int(t[2].item()) == net.n_pts
    # every rank sees the same verdict
    v = torch.tensor([1.0 if ok else 0.0], dtype=torch.float64)
    dist.all_reduce(v, op=dist.ReduceOp.MIN)
    if rank == 0:
        out.put(bool(v.item() == 1.0))
    dist.destroy_process_group()


def test_two_rank_gloo_sharding():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(120)
        assert p.exitcode == 0
    assert q.get(timeout=5) is True


@pytest.mark.parametrize("world", [1, 2, 3, 8])
def test_point_ranges_balance_and_cover(world):
    net = host.synth(n_cam=1, n_sta=20, n_pts=501, rays_per_point=4, seed=2).net
    b = sharding.point_ranges(net.obs_pt, net.n_pts, world)
    assert b[0] == 0 and b[-1] == net.n_pts and np.all(np.diff(b) >= 0)
    counts = [int(((net.obs_pt >= b[r]) & (net.obs_pt < b[r + 1])).sum()) for r in range(world)]
    assert sum(counts) == net.n_obs
    assert max(counts) - min(counts) <= 2 * 4 + 2  # within a couple of points of each other
    total_pts = sum(sharding.shard_network(net, r, world)[0].n_pts for r in range(world))
    assert total_pts == net.n_pts


@pytest.mark.parametrize("world", [1, 2, 3, 8])
def test_host_library_shards_like_the_python_partition(world):
    """ba_net_point_ranges / ba_net_shard (the C++ host's own path) against sharding.shard_network."""
    hn = host.synth(n_cam=2, n_sta=20, n_pts=501, rays_per_point=4, seed=2)
    net = hn.net
    assert np.array_equal(hn.point_ranges(world), sharding.point_ranges(net.obs_pt, net.n_pts, world))
    for rank in range(world):
        sh, sl = hn.shard(rank, world)
        ref, rsl = sharding.shard_network(net, rank, world)
        assert (sl.start, sl.stop) == (rsl.start, rsl.stop)
        for f in ("obs_pt", "obs_sta", "sta_cam", "obs_x", "obs_y", "obs_sx", "obs_sy", "obs_cx", "obs_cy", "pt_prior", "pt_sigma",
                  "iop_prior", "iop_sigma", "ap_free_mask", "iop_free_mask", "eop", "xyz", "iop", "ap"):
            a, b = getattr(sh.net, f), getattr(ref, f)
            assert np.array_equal(np.asarray(a).reshape(-1), np.asarray(b).reshape(-1)), f
        assert sh.net.model == ref.model and sh.net.huber_a == ref.huber_a
        # the ids of the shard's records are those of the records it took
        sel = (net.obs_pt >= sl.start) & (net.obs_pt < sl.stop)
        assert np.array_equal(sh.ids("pho_target"), hn.ids("pho_target")[sel])
        assert np.array_equal(sh.ids("xyz_target"), hn.ids("xyz_target")[sl])
        assert np.array_equal(sh.net.truth["xyz"], net.truth["xyz"][sl])
        sh.close()


def test_shard_of_points_without_observations():
    """Unobserved points weigh one unit each, so ranges stay non-degenerate; bad arguments are refused."""
    hn = host.synth(n_cam=1, n_sta=6, n_pts=40, rays_per_point=3, seed=5)
    b = hn.point_ranges(4)
    assert b[0] == 0 and b[-1] == 40 and np.all(np.diff(b) > 0)
    with pytest.raises(RuntimeError):
        hn.shard(4, 4)
    with pytest.raises(RuntimeError):
        hn.point_ranges(0)
