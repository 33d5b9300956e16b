"""The C-ABI shared library loads and exports every symbol include/ba_gpu.h declares
(no compute calls: this runs without a GPU)."""
import ctypes
import os
import re

from bundleadjustment_amd import build, solver

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared(header):
    txt = open(os.path.join(ROOT, "include", header)).read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    return sorted(set(re.findall(r"\b(ba_[a-z0-9_]+)\s*\(", txt)) - {"ba_allreduce_fn"})


def test_gpu_library_exports_header_symbols():
    build.build_gpu()  # hipcc cross-compiles for gfx950 without a device
    lib = ctypes.CDLL(solver.lib_path())
    names = declared("ba_gpu.h")
    assert set(names) == set(solver.EXPORTS), (names, solver.EXPORTS)
    for n in names:
        assert hasattr(lib, n), n


def test_host_library_exports_header_symbols():
    lib = ctypes.CDLL(build.build_host())
    names = [n for n in declared("ba_host.h") if n not in declared("ba_gpu.h")]
    assert "ba_net_read" in names and "ba_sigma0" in names
    for n in names:
        assert hasattr(lib, n), n


def test_default_options_are_the_reference_values():
    o = solver.default_options()  # src/main.cpp:4705-4719 over Ceres 1.13 defaults
    assert (o.min_lm_diagonal, o.max_lm_diagonal, o.max_num_iterations) == (1e-150, 1e-150, 100)
    assert (o.initial_trust_region_radius, o.max_trust_region_radius, o.min_trust_region_radius) == (1e4, 1e16, 1e-32)
    assert (o.function_tolerance, o.gradient_tolerance, o.parameter_tolerance) == (1e-6, 1e-10, 1e-8)
    assert o.min_relative_decrease == 1e-3 and o.jacobi_scaling == 1 and o.max_num_consecutive_invalid_steps == 5


def test_create_without_device_fails_loudly():
    import pytest
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    from bundleadjustment_amd import host
    net = host.synth(n_cam=1, n_sta=2, n_pts=4, rays_per_point=2).net
    with pytest.raises(solver.BaError, match="no HIP device|no CPU path"):
        solver.Problem(net)
