"""Partition of a network over ranks for the multi-GPU solve.

Points are independent given the station and camera unknowns, so every rank gets a contiguous
range of points (balanced by observation count) together with ALL observations of those points;
stations and cameras are replicated.  The ranks then only have to sum the reduced camera system
(see ba_set_allreduce in include/ba_gpu.h)."""
import numpy as np

from .capi import Network


def point_ranges(obs_pt, n_pts, world):
    """[world+1] point boundaries such that every rank holds about the same number of observations."""
    deg = np.bincount(obs_pt, minlength=n_pts).astype(np.int64)
    # a point without observations still costs a little (its prior): weight 1 keeps ranges non-degenerate
    w = np.cumsum(deg + 1)
    total = int(w[-1]) if n_pts else 0
    bounds = [0]
    for r in range(1, world):
        bounds.append(int(np.searchsorted(w, total * r / world, side="left")))
    bounds.append(n_pts)
    return np.maximum.accumulate(np.asarray(bounds, dtype=np.int64))


def shard_network(net: Network, rank: int, world: int):
    """Returns (shard, point_slice): the sub-network of `rank` and the slice of the global point
    arrays it owns.  With world == 1 the network itself is returned."""
    if world == 1:
        return net, slice(0, net.n_pts)
    b = point_ranges(net.obs_pt, net.n_pts, world)
    lo, hi = int(b[rank]), int(b[rank + 1])
    sel = np.nonzero((net.obs_pt >= lo) & (net.obs_pt < hi))[0]
    # cameras that the whole network observes: a rank without rows of one of them still carries its columns
    in_use = np.zeros(net.n_cam, dtype=np.uint8)
    in_use[net.sta_cam[net.obs_sta]] = 1
    shard = Network(
        obs_pt=net.obs_pt[sel] - lo, obs_sta=net.obs_sta[sel], sta_cam=net.sta_cam,
        obs_x=net.obs_x[sel], obs_y=net.obs_y[sel], obs_sx=net.obs_sx[sel], obs_sy=net.obs_sy[sel],
        obs_cx=net.obs_cx[sel], obs_cy=net.obs_cy[sel],
        pt_prior=net.pt_prior.reshape(-1, 3)[lo:hi], pt_sigma=net.pt_sigma.reshape(-1, 3)[lo:hi],
        iop_prior=net.iop_prior, iop_sigma=net.iop_sigma, ap_free_mask=net.ap_free_mask,
        iop_free_mask=net.iop_free_mask, eop=net.eop, xyz=net.xyz.reshape(-1, 3)[lo:hi], iop=net.iop, ap=net.ap,
        huber_a=net.huber_a, model=net.model, cam_in_use=in_use)
    return shard, slice(lo, hi)
