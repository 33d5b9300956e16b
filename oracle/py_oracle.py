"""ctypes access to the CPU oracle (oracle/libba_oracle.so) and, when built, to the reference's
own functors (oracle/_ref/libba_ref.so).

TEST INFRASTRUCTURE ONLY: imported by tests/, __graft_entry__.smoke() and the cpu_baseline leg
of bench.py.  Nothing under bundleadjustment_amd/ imports this module.
"""
import ctypes as C
import os

import numpy as np

from bundleadjustment_amd.capi import (BaDims, BaInputs, BaIterRow, BaOptions, BaSummary, c_double_p,
                                       c_int32_p, rows_to_dicts, summary_to_dict)

_HERE = os.path.dirname(os.path.abspath(__file__))
FULL_DENSE, SCHUR_DENSE, SCHUR_PCG = 0, 1, 2

_lib = None
_lib_omp = None
_ref = None


def _load(name):
    path = os.path.join(_HERE, name)
    if not os.path.exists(path):
        raise RuntimeError(f"{path} is missing: run `make -C oracle` (or __graft_entry__.build())")
    L = C.CDLL(path)
    L.bao_last_error.restype = C.c_char_p
    L.bao_linearize_once.restype = C.c_double
    return L


def lib():
    """The checker: one thread, every sum in the written order."""
    global _lib
    if _lib is None:
        _lib = _load("libba_oracle.so")
    return _lib


def lib_omp():
    """The same source built with OpenMP: only for timing the all-cores CPU baseline, never for checking."""
    global _lib_omp
    if _lib_omp is None:
        _lib_omp = _load("libba_oracle_omp.so")
    return _lib_omp


def omp_threads():
    return int(lib_omp().bao_num_threads())


def ref_lib():
    """The reference functors compiled from /root/reference (None when not built)."""
    global _ref
    if _ref is None:
        path = os.path.join(_HERE, "_ref", "libba_ref.so")
        if not os.path.exists(path):
            return None
        _ref = C.CDLL(path)
    return _ref


def _dp(a):
    return a.ctypes.data_as(c_double_p)


def _check(rc):
    if rc != 0:
        raise RuntimeError(f"oracle error {rc}: {lib().bao_last_error().decode()}")


def collinearity(eop, xyz, iop, ap, obs, jac=True):
    eop, xyz, iop, ap, obs = (np.ascontiguousarray(a, dtype=np.float64) for a in (eop, xyz, iop, ap, obs))
    r = np.zeros(2)
    J = np.zeros((2, 28)) if jac else None
    lib().bao_collinearity(_dp(eop), _dp(xyz), _dp(iop), _dp(ap), _dp(obs), _dp(r), _dp(J) if jac else None)
    return r, J


def projection(model, eop, xyz, iop, ap, obs, jac=True):
    """Residual and 2x28 Jacobian of one observation under projection model 0..5."""
    eop, xyz, iop, ap, obs = (np.ascontiguousarray(a, dtype=np.float64) for a in (eop, xyz, iop, ap, obs))
    r = np.zeros(2)
    J = np.zeros((2, 28)) if jac else None
    lib().bao_projection(int(model), _dp(eop), _dp(xyz), _dp(iop), _dp(ap), _dp(obs), _dp(r), _dp(J) if jac else None)
    return r, J


def index_first_match(ids, keys):
    ids = np.ascontiguousarray(ids, dtype=np.int32)
    keys = np.ascontiguousarray(keys, dtype=np.int32)
    out = np.zeros(len(keys), dtype=np.int32)
    lib().bao_index_first_match(ids.ctypes.data_as(c_int32_p), len(ids), keys.ctypes.data_as(c_int32_p),
                                len(keys), out.ctypes.data_as(c_int32_p))
    return out


def residual_jacobian_all(net, eop=None, xyz=None, iop=None, ap=None):
    eop = net.eop if eop is None else eop
    xyz = net.xyz if xyz is None else xyz
    iop = net.iop if iop is None else iop
    ap = net.ap if ap is None else ap
    d, inp = net.dims(), net.inputs()
    r = np.zeros(2 * net.n_obs)
    J = np.zeros((net.n_obs, 2, 28))
    _check(lib().bao_residual_jacobian_all(C.byref(d), C.byref(inp), _dp(eop), _dp(xyz), _dp(iop), _dp(ap),
                                           _dp(r), _dp(J)))
    return r, J


def evaluate(net, eop=None, xyz=None, iop=None, ap=None, apply_loss=True):
    eop = net.eop if eop is None else eop
    xyz = net.xyz if xyz is None else xyz
    iop = net.iop if iop is None else iop
    ap = net.ap if ap is None else ap
    d, inp = net.dims(), net.inputs()
    cost = C.c_double()
    res = np.zeros(2 * net.n_obs + 3 * net.n_pts + 3 * net.n_cam)
    _check(lib().bao_evaluate(C.byref(d), C.byref(inp), _dp(eop), _dp(xyz), _dp(iop), _dp(ap),
                              int(bool(apply_loss)), C.byref(cost), _dp(res)))
    return cost.value, res


def sparsity(net):
    d, inp = net.dims(), net.inputs()
    nr, nc, nnz = C.c_int32(), C.c_int32(), C.c_int64()
    _check(lib().bao_sparsity(C.byref(d), C.byref(inp), C.byref(nr), C.byref(nc), C.byref(nnz), None, None))
    rows = np.zeros(nr.value + 1, dtype=np.int32)
    cols = np.zeros(nnz.value, dtype=np.int32)
    _check(lib().bao_sparsity(C.byref(d), C.byref(inp), C.byref(nr), C.byref(nc), C.byref(nnz),
                              rows.ctypes.data_as(c_int32_p), cols.ctypes.data_as(c_int32_p)))
    return nr.value, nc.value, rows, cols


def solve(net, options, mode=SCHUR_DENSE, rows_cap=256, iop_bounds=None, omp=False):
    """Returns (eop, xyz, iop, ap, summary dict, list of iteration-row dicts).
    iop_bounds = (lower [C,3], upper [C,3]): the box of src/main.cpp:4252-4256."""
    d, inp = net.dims(), net.inputs()
    eop, xyz, iop, ap = net.params_copy()
    sm = BaSummary()
    rows = (BaIterRow * rows_cap)()
    lo = hi = None
    if iop_bounds is not None:
        lo, hi = (np.ascontiguousarray(b, dtype=np.float64).reshape(net.n_cam, 3) for b in iop_bounds)
    _check((lib_omp() if omp else lib()).bao_solve_bounded(C.byref(d), C.byref(inp), C.byref(options), int(mode),
                                   _dp(lo) if lo is not None else None, _dp(hi) if hi is not None else None,
                                   _dp(eop), _dp(xyz), _dp(iop), _dp(ap), C.byref(sm), rows, rows_cap))
    return eop, xyz, iop, ap, summary_to_dict(sm), rows_to_dicts(rows, min(sm.num_iterations, rows_cap))


def interpolating_step(lower, previous, current, x_min, x_max):
    """One backtracking move of the Armijo search; samples are (x, value, gradient, value_ok, gradient_ok)."""
    L = lib()
    L.bao_interpolating_step.restype = C.c_double
    arr = [None if s is None else np.ascontiguousarray(s, dtype=np.float64) for s in (lower, previous, current)]
    return L.bao_interpolating_step(*[None if a is None else _dp(a) for a in arr], C.c_double(x_min), C.c_double(x_max))


def covariance(net, eop=None, xyz=None, iop=None, ap=None):
    """Variances of all parameters at the given point: (var_eop [S,6], var_xyz [P,3], var_iop [C,3], var_ap [C,16])."""
    eop = net.eop if eop is None else eop
    xyz = net.xyz if xyz is None else xyz
    iop = net.iop if iop is None else iop
    ap = net.ap if ap is None else ap
    d, inp = net.dims(), net.inputs()
    ve, vx = np.zeros((net.n_sta, 6)), np.zeros((net.n_pts, 3))
    vi, va = np.zeros((net.n_cam, 3)), np.zeros((net.n_cam, 16))
    _check(lib().bao_covariance(C.byref(d), C.byref(inp), _dp(eop), _dp(xyz), _dp(iop), _dp(ap), _dp(ve), _dp(vx), _dp(vi), _dp(va)))
    return ve, vx, vi, va


def covariance_rows(net, eop=None, xyz=None, iop=None, ap=None):
    """Covariances of the free camera parameters with all parameters: (cam_param [n], cov_eop [n,S,6], cov_xyz [n,P,3],
    cov_cam [n,n]); cam_param = 19 * camera + parameter (0..2 xp yp c, 3..18 AP)."""
    eop = net.eop if eop is None else eop
    xyz = net.xyz if xyz is None else xyz
    iop = net.iop if iop is None else iop
    ap = net.ap if ap is None else ap
    d, inp = net.dims(), net.inputs()
    cap = 19 * net.n_cam
    par, n = np.zeros(cap, dtype=np.int32), C.c_int32(0)
    ce, cx, cc = np.zeros((cap, net.n_sta, 6)), np.zeros((cap, net.n_pts, 3)), np.zeros(cap * cap)
    _check(lib().bao_covariance_rows(C.byref(d), C.byref(inp), _dp(eop), _dp(xyz), _dp(iop), _dp(ap), C.byref(n),
                                     par.ctypes.data_as(C.POINTER(C.c_int32)), _dp(ce), _dp(cx), _dp(cc)))
    n = n.value
    return par[:n].copy(), ce[:n].copy(), cx[:n].copy(), cc[:n * n].reshape(n, n).copy()


def linearize_once(net, omp=False):
    d, inp = net.dims(), net.inputs()
    return (lib_omp() if omp else lib()).bao_linearize_once(C.byref(d), C.byref(inp), _dp(net.eop), _dp(net.xyz), _dp(net.iop), _dp(net.ap))


# ---- the reference functors themselves (present only where /root/reference was available at build time)

def ref_collinearity(eop, xyz, iop, ap, obs, mls=True):
    L = ref_lib()
    eop, xyz, iop, ap, obs = (np.ascontiguousarray(a, dtype=np.float64) for a in (eop, xyz, iop, ap, obs))
    r = np.zeros(2)
    fn = L.ref_collinearity_mls if mls else L.ref_collinearity
    fn(_dp(eop), _dp(xyz), _dp(iop), _dp(ap), _dp(obs), _dp(r))
    return r


def ref_collinearity_jacobian(eop, xyz, iop, ap, obs, h=1e-30):
    L = ref_lib()
    eop, xyz, iop, ap, obs = (np.ascontiguousarray(a, dtype=np.float64) for a in (eop, xyz, iop, ap, obs))
    J = np.zeros((2, 28))
    L.ref_collinearity_mls_jacobian(_dp(eop), _dp(xyz), _dp(iop), _dp(ap), _dp(obs), C.c_double(h), _dp(J))
    return J


def ref_projection(model, eop, xyz, iop, ap, obs):
    L = ref_lib()
    eop, xyz, iop, ap, obs = (np.ascontiguousarray(a, dtype=np.float64) for a in (eop, xyz, iop, ap, obs))
    r = np.zeros(2)
    if L.ref_projection(int(model), _dp(eop), _dp(xyz), _dp(iop), _dp(ap), _dp(obs), _dp(r)) != 0:
        raise ValueError(f"no reference functor for model {model}")
    return r


def ref_projection_jacobian(model, eop, xyz, iop, ap, obs, h=1e-30):
    L = ref_lib()
    eop, xyz, iop, ap, obs = (np.ascontiguousarray(a, dtype=np.float64) for a in (eop, xyz, iop, ap, obs))
    J = np.zeros((2, 28))
    if L.ref_projection_jacobian(int(model), _dp(eop), _dp(xyz), _dp(iop), _dp(ap), _dp(obs), C.c_double(h), _dp(J)) != 0:
        raise ValueError(f"no reference functor for model {model}")
    return J


def ref_statistics(v):
    """median, mean, stdev, min, max, RMSE of the reference's calcStatistics (src/main.cpp:884)."""
    L = ref_lib()
    v = np.ascontiguousarray(v, dtype=np.float64)
    out = np.zeros(6)
    L.ref_statistics(_dp(v), len(v), _dp(out))
    return out


def ref_sort_index(v):
    L = ref_lib()
    v = np.ascontiguousarray(v, dtype=np.float64)
    out = np.zeros(len(v), dtype=np.int32)
    L.ref_sort_index(_dp(v), len(v), out.ctypes.data_as(c_int32_p))
    return out


def ref_scalar(name, *args):
    """ref_median / ref_stddev (array, n) and ref_aic / ref_bic / ref_incidence_angle / ref_refraction_angle /
    ref_radial_distance with their C arguments; arrays as numpy float64."""
    L = ref_lib()
    fn = getattr(L, name)
    fn.restype = C.c_double
    conv = []
    for a in args:
        if isinstance(a, np.ndarray):
            conv.append(_dp(np.ascontiguousarray(a, dtype=np.float64)))
        elif isinstance(a, int):
            conv.append(C.c_int(a))
        else:
            conv.append(C.c_double(a))
    return fn(*conv)


def ref_constrain_point(p, p0, sigma):
    L = ref_lib()
    p, p0, sigma = (np.ascontiguousarray(a, dtype=np.float64) for a in (p, p0, sigma))
    r = np.zeros(3)
    L.ref_constrain_point(_dp(p), _dp(p0), _dp(sigma), _dp(r))
    return r
