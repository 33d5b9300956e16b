"""Python binding of the HIP solver library (include/ba_gpu.h, libba_gpu.so).

Every call goes through the C ABI; there is no fallback: if the shared library is missing or
no HIP device is visible the calls raise."""
import ctypes as C
import os

import numpy as np

from .capi import (BaDims, BaInputs, BaIterRow, BaLsSample, BaOptions, BaSummary, Network, c_double_p, c_int32_p,
                   rows_to_dicts, summary_to_dict)

_PKG = os.path.dirname(os.path.abspath(__file__))
_lib = None

ALLREDUCE_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, c_double_p, C.c_int64, C.c_int32, C.c_void_p)

KERNEL_LINEARIZE, KERNEL_POINT_APPLY, KERNEL_STATION_GATHER, KERNEL_COST = 0, 1, 2, 3

# every entry point include/ba_gpu.h declares
EXPORTS = ("ba_default_options", "ba_create", "ba_destroy", "ba_set_parameters", "ba_get_parameters", "ba_set_corrections", "ba_covariance_cameras",
           "ba_set_iop_bounds", "ba_line_search_contract",
           "ba_solve_resident", "ba_solve", "ba_evaluate", "ba_covariance", "ba_sparsity", "ba_residual_jacobian", "ba_time_kernel",
           "ba_set_allreduce", "ba_copy_to_host", "ba_copy_to_device", "ba_rccl_unique_id", "ba_use_rccl", "ba_evaluate_noalloc",
           "ba_device_count", "ba_set_device", "ba_reduction_timing", "ba_reduction_stats",
           "ba_shard_reduced_solve", "ba_reduced_solve_rows", "ba_set_covariance_route", "ba_covariance_stats", "ba_sparse_solver_stats", "ba_tuning", "ba_covariance_camera_rows",
           "ba_last_error")


def lib_path():
    return os.path.join(_PKG, "libba_gpu.so")


def lib():
    global _lib
    if _lib is None:
        path = lib_path()
        if not os.path.exists(path):
            raise RuntimeError(f"{path} is missing: run `python -m bundleadjustment_amd.build` (there is no CPU path)")
        # One HIP runtime per process: PyTorch bundles its own libamdhip64/librccl, and a process
        # that loads this library first and torch afterwards ends up with a device-less runtime.
        # Loading torch first makes both share torch's copy (same SONAMEs).
        try:
            import torch  # noqa: F401
        except ImportError:
            pass
        L = C.CDLL(path)
        L.ba_last_error.restype = C.c_char_p
        L.ba_create.argtypes = [C.POINTER(BaDims), C.POINTER(BaInputs), C.POINTER(C.c_void_p)]
        L.ba_destroy.argtypes = [C.c_void_p]
        L.ba_destroy.restype = None
        L.ba_default_options.argtypes = [C.POINTER(BaOptions)]
        L.ba_default_options.restype = None
        L.ba_set_parameters.argtypes = [C.c_void_p, c_double_p, c_double_p, c_double_p, c_double_p]
        L.ba_get_parameters.argtypes = [C.c_void_p, c_double_p, c_double_p, c_double_p, c_double_p]
        L.ba_set_corrections.argtypes = [C.c_void_p, c_double_p, c_double_p]
        L.ba_covariance_cameras.argtypes = [C.c_void_p, c_double_p, c_double_p]
        L.ba_covariance_camera_rows.argtypes = [C.c_void_p, c_int32_p, c_int32_p, c_double_p, c_double_p, c_double_p]
        L.ba_set_iop_bounds.argtypes = [C.c_void_p, c_double_p, c_double_p]
        L.ba_set_covariance_route.argtypes = [C.c_void_p, C.c_int32, C.c_int32]
        L.ba_covariance_stats.argtypes = [C.c_void_p] + [c_int32_p] * 4
        L.ba_tuning.argtypes = [C.c_char_p, C.c_double]
        L.ba_sparse_solver_stats.argtypes = [C.c_void_p] + [c_int32_p] * 4 + [c_double_p] * 2
        L.ba_line_search_contract.argtypes = [C.POINTER(BaLsSample)] * 3 + [C.c_double, C.c_double]
        L.ba_line_search_contract.restype = C.c_double
        L.ba_solve_resident.argtypes = [C.c_void_p, C.POINTER(BaOptions), C.POINTER(BaSummary), C.POINTER(BaIterRow), C.c_int32]
        L.ba_solve.argtypes = [C.c_void_p, C.POINTER(BaOptions), c_double_p, c_double_p, c_double_p, c_double_p,
                               C.POINTER(BaSummary), C.POINTER(BaIterRow), C.c_int32]
        L.ba_evaluate.argtypes = [C.c_void_p, c_double_p, c_double_p, c_double_p, c_double_p, C.c_int, c_double_p, c_double_p]
        L.ba_sparsity.argtypes = [C.c_void_p, c_int32_p, c_int32_p, C.POINTER(C.c_int64), c_int32_p, c_int32_p]
        L.ba_residual_jacobian.argtypes = [C.c_void_p, c_double_p, c_double_p]
        L.ba_time_kernel.argtypes = [C.c_void_p, C.c_int32, C.c_int32, c_double_p]
        L.ba_set_allreduce.argtypes = [C.c_void_p, ALLREDUCE_FN, C.c_void_p]
        L.ba_copy_to_host.argtypes = [C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p]
        L.ba_copy_to_device.argtypes = [C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p]
        L.ba_rccl_unique_id.argtypes = [C.c_void_p]
        L.ba_device_count.argtypes = [C.POINTER(C.c_int32)]
        L.ba_reduction_timing.argtypes = [C.c_void_p, C.c_int32]
        L.ba_reduction_stats.argtypes = [C.c_void_p, C.POINTER(C.c_int64), C.POINTER(C.c_double), C.POINTER(C.c_double)]
        L.ba_set_device.argtypes = [C.c_int32]
        L.ba_shard_reduced_solve.argtypes = [C.c_void_p, C.c_int32, C.c_int32, C.c_int32]
        L.ba_reduced_solve_rows.argtypes = [C.c_void_p, C.POINTER(C.c_int32), C.POINTER(C.c_int32)]
        L.ba_use_rccl.argtypes = [C.c_void_p, C.c_int32, C.c_int32, C.c_void_p]
        L.ba_evaluate_noalloc.argtypes = [C.POINTER(BaDims), C.POINTER(BaInputs)] + [C.c_void_p] * 4 + [C.c_int] + [C.c_void_p] * 3
        _lib = L
    return _lib


class BaError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__(f"ba_gpu error {code}: {msg}")
        self.code = code


def _check(rc):
    if rc != 0:
        raise BaError(rc, lib().ba_last_error().decode())


def _dp(a):
    return a.ctypes.data_as(c_double_p)


def default_options():
    o = BaOptions()
    lib().ba_default_options(C.byref(o))
    return o


class Problem:
    """One bundle-adjustment problem resident on the current HIP device (ba_create ... ba_destroy)."""

    def __init__(self, net: Network):
        self.net = net
        self._h = C.c_void_p()
        d, inp = net.dims(), net.inputs()
        _check(lib().ba_create(C.byref(d), C.byref(inp), C.byref(self._h)))
        self._cb = None

    def close(self):
        if self._h:
            lib().ba_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()

    def set_parameters(self, eop=None, xyz=None, iop=None, ap=None):
        n = self.net
        eop, xyz, iop, ap = (np.ascontiguousarray(a if a is not None else b, dtype=np.float64)
                             for a, b in ((eop, n.eop), (xyz, n.xyz), (iop, n.iop), (ap, n.ap)))
        _check(lib().ba_set_parameters(self._h, _dp(eop), _dp(xyz), _dp(iop), _dp(ap)))

    def set_corrections(self, obs_cx, obs_cy):
        """New constant image corrections (columns 7-8 of the .pho file) for the problem on the device."""
        cx, cy = (np.ascontiguousarray(a, dtype=np.float64) for a in (obs_cx, obs_cy))
        if cx.shape != (self.net.n_obs,) or cy.shape != (self.net.n_obs,):
            raise ValueError("one correction per observation")
        _check(lib().ba_set_corrections(self._h, _dp(cx), _dp(cy)))

    def set_iop_bounds(self, lower=None, upper=None):
        """Box on xp, yp, c per camera ([C,3] each, None = unbounded): the SetParameterLowerBound/UpperBound calls of
        src/main.cpp:4252-4256.  A finite bound on a variable block switches the solve to the bounds-constrained route."""
        lo, hi = (None if b is None else np.ascontiguousarray(b, dtype=np.float64).reshape(self.net.n_cam, 3) for b in (lower, upper))
        _check(lib().ba_set_iop_bounds(self._h, None if lo is None else _dp(lo), None if hi is None else _dp(hi)))

    def get_parameters(self):
        n = self.net
        eop, xyz, iop, ap = (np.zeros_like(a) for a in (n.eop, n.xyz, n.iop, n.ap))
        _check(lib().ba_get_parameters(self._h, _dp(eop), _dp(xyz), _dp(iop), _dp(ap)))
        return eop, xyz, iop, ap

    def solve_resident(self, options, rows_cap=256):
        sm = BaSummary()
        rows = (BaIterRow * rows_cap)()
        _check(lib().ba_solve_resident(self._h, C.byref(options), C.byref(sm), rows, rows_cap))
        self.last_summary = sm
        return summary_to_dict(sm), rows_to_dicts(rows, min(sm.num_iterations, rows_cap))

    def solve(self, options, rows_cap=256):
        """ceres::Solve semantics: starts from the network's parameters, returns the adjusted copies."""
        eop, xyz, iop, ap = self.net.params_copy()
        sm = BaSummary()
        rows = (BaIterRow * rows_cap)()
        _check(lib().ba_solve(self._h, C.byref(options), _dp(eop), _dp(xyz), _dp(iop), _dp(ap), C.byref(sm), rows, rows_cap))
        self.last_summary = sm
        return eop, xyz, iop, ap, summary_to_dict(sm), rows_to_dicts(rows, min(sm.num_iterations, rows_cap))

    def evaluate(self, eop=None, xyz=None, iop=None, ap=None, apply_loss=True, residuals=True):
        n = self.net
        eop, xyz, iop, ap = (np.ascontiguousarray(a if a is not None else b, dtype=np.float64)
                             for a, b in ((eop, n.eop), (xyz, n.xyz), (iop, n.iop), (ap, n.ap)))
        cost = C.c_double()
        res = np.zeros(2 * n.n_obs + 3 * n.n_pts + 3 * n.n_cam) if residuals else None
        _check(lib().ba_evaluate(self._h, _dp(eop), _dp(xyz), _dp(iop), _dp(ap), int(bool(apply_loss)), C.byref(cost),
                                 _dp(res) if residuals else None))
        return cost.value, res

    def covariance_cameras(self):
        """(var_iop [C,3], var_ap [C,16]) at the parameters the problem holds; works for networks of any size."""
        n = self.net
        vi, va = np.zeros((n.n_cam, 3)), np.zeros((n.n_cam, 16))
        _check(lib().ba_covariance_cameras(self._h, _dp(vi), _dp(va)))
        return vi, va

    def covariance_camera_rows(self, points=True):
        """Covariances of the free camera parameters with all parameters: (cam_param [n], cov_eop [n,S,6],
        cov_xyz [n,P,3] or None, cov_cam [n,n]); cam_param = 19 * camera + parameter (0..2 xp yp c, 3..18 AP)."""
        net = self.net
        n = C.c_int32()
        _check(lib().ba_covariance_camera_rows(self._h, C.byref(n), None, None, None, None))
        n = n.value
        par = np.zeros(max(n, 1), dtype=np.int32)
        ce, cc = np.zeros((n, net.n_sta, 6)), np.zeros((n, n))
        cx = np.zeros((n, net.n_pts, 3)) if points else None
        _check(lib().ba_covariance_camera_rows(self._h, None, par.ctypes.data_as(c_int32_p), _dp(ce), _dp(cx) if points else None, _dp(cc)))
        return par[:n], ce, cx, cc

    def covariance_route(self, route=0, max_cluster_stations=0):
        """0: dense inverse while it fits, dissection of the station graph beyond; 1: dense only; 2: dissection."""
        _check(lib().ba_set_covariance_route(self._h, int(route), int(max_cluster_stations)))

    def sparse_solver_stats(self):
        """Of the sparse direct solver's elimination tree: dict(fronts, levels, largest_front, critical_panels, front_bytes, factor_flops)."""
        v = [C.c_int32() for _ in range(4)]
        d = [C.c_double() for _ in range(2)]
        _check(lib().ba_sparse_solver_stats(self._h, *[C.byref(x) for x in v], *[C.byref(x) for x in d]))
        return dict(zip(("fronts", "levels", "largest_front", "critical_panels", "front_bytes", "factor_flops"),
                        [x.value for x in v] + [x.value for x in d]))

    def covariance_stats(self):
        """Of the last covariance(): dict(route, clusters, boundary_order, largest_cluster_order)."""
        v = [C.c_int32() for _ in range(4)]
        _check(lib().ba_covariance_stats(self._h, *[C.byref(x) for x in v]))
        return dict(zip(("route", "clusters", "boundary_order", "largest_cluster_order"), (x.value for x in v)))

    def covariance(self):
        """Variances of all parameters at the values the problem holds: (var_eop [S,6], var_xyz [P,3], var_iop [C,3], var_ap [C,16])."""
        n = self.net
        ve, vx = np.zeros((n.n_sta, 6)), np.zeros((n.n_pts, 3))
        vi, va = np.zeros((n.n_cam, 3)), np.zeros((n.n_cam, 16))
        _check(lib().ba_covariance(self._h, _dp(ve), _dp(vx), _dp(vi), _dp(va)))
        return ve, vx, vi, va

    def sparsity(self):
        nr, nc, nnz = C.c_int32(), C.c_int32(), C.c_int64()
        _check(lib().ba_sparsity(self._h, C.byref(nr), C.byref(nc), C.byref(nnz), None, None))
        rows = np.zeros(nr.value + 1, dtype=np.int32)
        cols = np.zeros(nnz.value, dtype=np.int32)
        _check(lib().ba_sparsity(self._h, C.byref(nr), C.byref(nc), C.byref(nnz), rows.ctypes.data_as(c_int32_p),
                                 cols.ctypes.data_as(c_int32_p)))
        return nr.value, nc.value, rows, cols

    def residual_jacobian(self, jacobian=True):
        n = self.net
        r = np.zeros(2 * n.n_obs)
        J = np.zeros((n.n_obs, 2, 28)) if jacobian else None
        _check(lib().ba_residual_jacobian(self._h, _dp(r), _dp(J) if jacobian else None))
        return r, J

    def time_kernel(self, which, repeats=10):
        t = C.c_double()
        _check(lib().ba_time_kernel(self._h, which, repeats, C.byref(t)))
        return t.value

    def set_allreduce(self, fn):
        """fn(device_ptr:int, count:int, op:int, stream:int) -> int, called on the solver's stream."""
        def tramp(user, buf, count, op, stream):
            try:
                return int(fn(C.cast(buf, C.c_void_p).value, int(count), int(op), stream or 0))
            except Exception:  # never let an exception cross the C boundary
                import traceback
                traceback.print_exc()
                return -1
        self._cb = ALLREDUCE_FN(tramp)
        _check(lib().ba_set_allreduce(self._h, self._cb, None))

    def set_allreduce_host_staged(self, reduce_fn):
        """reduce_fn(numpy array, op) -> reduces the array in place over all ranks (op 0 sum, 1 max).
        The device buffer is copied to the host, reduced there, and copied back: for tests and for
        back ends without device-side collectives."""
        def cb(ptr, count, op, stream):
            buf = np.empty(count, dtype=np.float64)
            _check(lib().ba_copy_to_host(buf.ctypes.data, ptr, count, stream))
            reduce_fn(buf, op)
            _check(lib().ba_copy_to_device(ptr, buf.ctypes.data, count, stream))
            return 0
        self.set_allreduce(cb)

    def reduction_timing(self, on=True):
        _check(lib().ba_reduction_timing(self._h, int(bool(on))))

    def reduction_stats(self):
        """(calls, bytes, seconds on the stream) of the reductions over the ranks since the last call; resets them."""
        n, b, t = C.c_int64(), C.c_double(), C.c_double()
        _check(lib().ba_reduction_stats(self._h, C.byref(n), C.byref(b), C.byref(t)))
        return n.value, b.value, t.value

    def shard_reduced_solve(self, rank, n_ranks, mode=0):
        """This problem is rank `rank` of `n_ranks`; the block-sparse PCG forms its product by rows never (-1), when the
        library's model says it pays (0) or always (1)."""
        _check(lib().ba_shard_reduced_solve(self._h, rank, n_ranks, mode))

    def reduced_solve_rows(self):
        """Station rows of the product that the last block-sparse PCG solve gave this rank; (0, 0): all of them on every rank."""
        a, b = C.c_int32(), C.c_int32()
        _check(lib().ba_reduced_solve_rows(self._h, C.byref(a), C.byref(b)))
        return a.value, b.value

    def use_rccl(self, n_ranks, rank, unique_id: bytes):
        buf = C.create_string_buffer(unique_id, 128)
        _check(lib().ba_use_rccl(self._h, n_ranks, rank, buf))


def evaluate_noalloc(dims, inputs, eop, xyz, iop, ap, apply_loss, cost, residuals=None, stream=None):
    """ba_evaluate_noalloc: every pointer (those inside `inputs` too) is a device address, given as an int; cost [2] and
    residuals are device outputs.  Queues kernels on `stream` (a hipStream_t as int, None = default) and returns."""
    _check(lib().ba_evaluate_noalloc(C.byref(dims), C.byref(inputs), eop, xyz, iop, ap, int(bool(apply_loss)), cost, residuals, stream))


def device_count() -> int:
    n = C.c_int32()
    _check(lib().ba_device_count(C.byref(n)))
    return n.value


def set_device(device: int):
    """The GPU of the calling thread: before the first Problem of a process that owns one GPU of several."""
    _check(lib().ba_set_device(int(device)))


def tuning(name, value):
    """Process-wide comparison switch (see ba_tuning in include/ba_gpu.h)."""
    _check(lib().ba_tuning(name.encode(), float(value)))


def rccl_unique_id() -> bytes:
    buf = C.create_string_buffer(128)
    _check(lib().ba_rccl_unique_id(buf))
    return buf.raw


def line_search_contract(start, previous, current, min_step, max_step):
    """Next step size of the Armijo search (samples: x, value, gradient, value valid, gradient valid; previous may be None)."""
    def sample(t):
        return None if t is None else BaLsSample(float(t[0]), float(t[1]), float(t[2]), int(t[3]), int(t[4]))
    s, p, c = sample(start), sample(previous), sample(current)
    return lib().ba_line_search_contract(C.byref(s), None if p is None else C.byref(p), C.byref(c), float(min_step), float(max_step))
