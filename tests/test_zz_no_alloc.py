"""ba_evaluate_noalloc (the entry point that allocates nothing: its work memory is one fixed array in the library's
device image) against ba_evaluate, through the C ABI.  Two calls in a row on one GPU, the second on a smaller network,
so that what the first call left in the shared array would show."""
import ctypes as C

import numpy as np
import pytest

from bundleadjustment_amd import capi, host, solver

pytestmark = pytest.mark.gpu


def _device_inputs(net, torch):
    """BaInputs whose arrays live in device memory (torch only holds the memory) + the tensors that keep them alive."""
    dev = torch.device("cuda:0")
    keep = {}

    def put(name, a):
        keep[name] = torch.from_numpy(np.ascontiguousarray(a)).to(dev)
        return keep[name].data_ptr()

    s = capi.BaInputs()
    for k in ("obs_pt", "obs_sta", "sta_cam"):
        setattr(s, k, C.cast(put(k, getattr(net, k)), capi.c_int32_p))
    for k in ("obs_x", "obs_y", "obs_sx", "obs_sy", "obs_cx", "obs_cy", "pt_prior", "pt_sigma", "iop_prior", "iop_sigma"):
        setattr(s, k, C.cast(put(k, getattr(net, k)), capi.c_double_p))
    s.huber_a = float(net.huber_a)
    s.model = int(net.model)
    par = [put(k, getattr(net, k)) for k in ("eop", "xyz", "iop", "ap")]
    return s, par, keep


def _both(net, torch, apply_loss):
    with solver.Problem(net) as p:
        c_ref, r_ref = p.evaluate(apply_loss=apply_loss)
    inp, par, keep = _device_inputs(net, torch)
    n_res = 2 * net.n_obs + 3 * net.n_pts + 3 * net.n_cam
    cost = torch.full((2,), -1.0, dtype=torch.float64, device="cuda:0")
    res = torch.full((n_res,), 7.0, dtype=torch.float64, device="cuda:0")
    torch.cuda.synchronize()
    stream = torch.cuda.current_stream().cuda_stream
    solver.evaluate_noalloc(net.dims(), inp, *par, apply_loss, cost.data_ptr(), res.data_ptr(), stream)
    torch.cuda.synchronize()
    c = cost.cpu().numpy()
    assert c[1] == 0.0
    # same kernels' arithmetic per row: the rows are equal to the last bit or two, the cost to summation order
    np.testing.assert_allclose(res.cpu().numpy(), r_ref, rtol=1e-12, atol=1e-12)
    assert abs(c[0] - c_ref) <= 1e-12 * abs(c_ref)
    # cost-only call (no residual vector)
    cost.fill_(-1.0)
    solver.evaluate_noalloc(net.dims(), inp, *par, apply_loss, cost.data_ptr(), None, stream)
    torch.cuda.synchronize()
    assert abs(cost.cpu().numpy()[0] - c_ref) <= 1e-12 * abs(c_ref)
    del keep
    return c_ref


def test_noalloc_matches_ba_evaluate_twice_second_smaller():
    import torch
    big = host.synth_config("cfg2", scale=0.25).net            # 4 cameras, empirical terms, 250k observations
    small = host.synth_config("cfg1").net                      # 1 camera, 10k observations
    c1 = _both(big, torch, True)
    c2 = _both(small, torch, True)
    assert c1 > 0 and c2 > 0
    _both(small, torch, False)
    # a fisheye network through the same entry point
    fish = host.synth_config("cfg1", model=4).net
    _both(fish, torch, True)


def test_noalloc_limits_and_bad_rows():
    import torch
    net = host.synth_config("cfg1").net
    inp, par, keep = _device_inputs(net, torch)
    cost = torch.zeros(2, dtype=torch.float64, device="cuda:0")
    d = net.dims()
    d.n_sta = 65537
    with pytest.raises(solver.BaError) as e:
        solver.evaluate_noalloc(d, inp, *par, 1, cost.data_ptr(), None, None)
    assert e.value.code == capi.ERR_UNSUPPORTED and "65536" in str(e.value)
    d = net.dims()
    d.n_cam = 65
    with pytest.raises(solver.BaError) as e:
        solver.evaluate_noalloc(d, inp, *par, 1, cost.data_ptr(), None, None)
    assert e.value.code == capi.ERR_UNSUPPORTED
    # an observation that points outside the point array is skipped and counted, not dereferenced
    bad = capi.Network(**{k: getattr(net, k).copy() for k in
                          ("obs_pt", "obs_sta", "sta_cam", "obs_x", "obs_y", "obs_sx", "obs_sy", "obs_cx", "obs_cy", "pt_prior",
                           "pt_sigma", "iop_prior", "iop_sigma", "ap_free_mask", "iop_free_mask", "eop", "xyz", "iop", "ap")})
    bad.obs_pt[5] = net.n_pts + 3
    bad.obs_sta[9] = -1
    inp, par, keep = _device_inputs(bad, torch)
    solver.evaluate_noalloc(bad.dims(), inp, *par, 1, cost.data_ptr(), None, None)
    torch.cuda.synchronize()
    assert cost.cpu().numpy()[1] == 2.0
