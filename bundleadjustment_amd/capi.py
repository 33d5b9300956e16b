"""ctypes descriptions of the C ABI declared in include/ba_gpu.h.

The same struct layouts are used by the HIP solver library (libba_gpu.so) and, in tests only,
by the CPU oracle, so they live in one place.
"""
import ctypes as C
from dataclasses import dataclass, field

import numpy as np

c_double_p = C.POINTER(C.c_double)
c_int32_p = C.POINTER(C.c_int32)


class BaDims(C.Structure):
    _fields_ = [("n_obs", C.c_int32), ("n_sta", C.c_int32), ("n_pts", C.c_int32), ("n_cam", C.c_int32)]


class BaInputs(C.Structure):
    _fields_ = [
        ("obs_pt", c_int32_p), ("obs_sta", c_int32_p), ("sta_cam", c_int32_p),
        ("obs_x", c_double_p), ("obs_y", c_double_p),
        ("obs_sx", c_double_p), ("obs_sy", c_double_p),
        ("obs_cx", c_double_p), ("obs_cy", c_double_p),
        ("pt_prior", c_double_p), ("pt_sigma", c_double_p),
        ("iop_prior", c_double_p), ("iop_sigma", c_double_p),
        ("ap_free_mask", C.POINTER(C.c_uint16)), ("iop_free_mask", C.POINTER(C.c_uint8)),
        ("huber_a", C.c_double), ("model", C.c_int32),
        ("cam_in_use", C.POINTER(C.c_uint8)),
    ]


class BaOptions(C.Structure):
    _fields_ = [
        ("min_lm_diagonal", C.c_double), ("max_lm_diagonal", C.c_double),
        ("initial_trust_region_radius", C.c_double), ("max_trust_region_radius", C.c_double),
        ("min_trust_region_radius", C.c_double), ("min_relative_decrease", C.c_double),
        ("function_tolerance", C.c_double), ("gradient_tolerance", C.c_double),
        ("parameter_tolerance", C.c_double),
        ("max_num_iterations", C.c_int32), ("max_num_consecutive_invalid_steps", C.c_int32),
        ("jacobi_scaling", C.c_int32), ("linear_solver", C.c_int32),
        ("pcg_rtol", C.c_double), ("pcg_max_iters", C.c_int32), ("preconditioner", C.c_int32),
    ]


class BaIterRow(C.Structure):
    _fields_ = [
        ("iteration", C.c_int32), ("step_is_successful", C.c_int32), ("step_is_valid", C.c_int32),
        ("linear_iterations", C.c_int32),
        ("cost", C.c_double), ("cost_change", C.c_double), ("gradient_max_norm", C.c_double),
        ("step_norm", C.c_double), ("relative_decrease", C.c_double),
        ("trust_region_radius", C.c_double),
        ("t_linearize_s", C.c_double), ("t_solve_s", C.c_double), ("t_eval_s", C.c_double),
    ]


class BaSummary(C.Structure):
    _fields_ = [
        ("initial_cost", C.c_double), ("final_cost", C.c_double), ("fixed_cost", C.c_double),
        ("num_iterations", C.c_int32), ("termination", C.c_int32), ("stop_reason", C.c_int32),
        ("num_successful_steps", C.c_int32), ("num_unsuccessful_steps", C.c_int32),
        ("num_residuals_reduced", C.c_int32), ("num_parameters_reduced", C.c_int32),
        ("num_effective_parameters_reduced", C.c_int32), ("num_parameters", C.c_int32),
        ("num_residuals", C.c_int32), ("total_linear_iterations", C.c_int32), ("linear_solver_used", C.c_int32),
        ("t_total_s", C.c_double),
    ]


class BaLsSample(C.Structure):
    _fields_ = [("x", C.c_double), ("value", C.c_double), ("gradient", C.c_double),
                ("value_is_valid", C.c_int32), ("gradient_is_valid", C.c_int32)]


SOLVER_DENSE_SCHUR = 0
SOLVER_PCG_SCHUR = 1
SOLVER_PCG_IMPLICIT = 2
SOLVER_AUTO = 3
SOLVER_SPARSE_SCHUR = 4
AUTO_DENSE_ORDER = 4096
AUTO_SPARSE_ORDER = 250000
CONVERGENCE, NO_CONVERGENCE, FAILURE = 0, 1, 2
ERR_INVALID, ERR_DEVICE, ERR_NUMERIC, ERR_UNSUPPORTED, ERR_COMM = -1, -2, -3, -4, -5
STOP_REASONS = {0: "none", 1: "max_iterations", 2: "gradient_tol", 3: "parameter_tol",
                4: "function_tol", 5: "min_radius", 6: "invalid_steps", 7: "linear_solver"}


def default_options(linear_solver=SOLVER_AUTO, pcg_rtol=1e-12, pcg_max_iters=2000):
    """Option values of src/main.cpp:4705-4719 over the Ceres 1.13 defaults."""
    o = BaOptions()
    o.min_lm_diagonal = 1e-150
    o.max_lm_diagonal = 1e-150
    o.initial_trust_region_radius = 1e4
    o.max_trust_region_radius = 1e16
    o.min_trust_region_radius = 1e-32
    o.min_relative_decrease = 1e-3
    o.function_tolerance = 1e-6
    o.gradient_tolerance = 1e-10
    o.parameter_tolerance = 1e-8
    o.max_num_iterations = 100
    o.max_num_consecutive_invalid_steps = 5
    o.jacobi_scaling = 1
    o.linear_solver = linear_solver
    o.pcg_rtol = pcg_rtol
    o.pcg_max_iters = pcg_max_iters
    return o


def _f64(a):
    return np.ascontiguousarray(a, dtype=np.float64)


def _i32(a):
    return np.ascontiguousarray(a, dtype=np.int32)


@dataclass
class Network:
    """One bundle-adjustment problem as flat numpy arrays (the contents of the four input
    files after the index maps of src/main.cpp:3821-3831 have been resolved)."""
    obs_pt: np.ndarray
    obs_sta: np.ndarray
    sta_cam: np.ndarray
    obs_x: np.ndarray
    obs_y: np.ndarray
    obs_sx: np.ndarray
    obs_sy: np.ndarray
    obs_cx: np.ndarray
    obs_cy: np.ndarray
    pt_prior: np.ndarray   # [P,3]
    pt_sigma: np.ndarray   # [P,3]
    iop_prior: np.ndarray  # [C,3]
    iop_sigma: np.ndarray  # [C,3]
    ap_free_mask: np.ndarray   # [C] uint16
    iop_free_mask: np.ndarray  # [C] uint8
    eop: np.ndarray  # [S,6] omega,phi,kappa,Xo,Yo,Zo
    xyz: np.ndarray  # [P,3]
    iop: np.ndarray  # [C,3]
    ap: np.ndarray   # [C,16]
    huber_a: float = 1.0
    model: int = 0
    truth: dict = field(default_factory=dict)
    cam_in_use: np.ndarray = None  # a rank's share of a network only: [C] uint8, cameras that the whole network observes

    def __post_init__(self):
        for k in ("obs_pt", "obs_sta", "sta_cam"):
            setattr(self, k, _i32(getattr(self, k)))
        for k in ("obs_x", "obs_y", "obs_sx", "obs_sy", "obs_cx", "obs_cy", "pt_prior", "pt_sigma",
                  "iop_prior", "iop_sigma", "eop", "xyz", "iop", "ap"):
            setattr(self, k, _f64(getattr(self, k)))
        self.ap_free_mask = np.ascontiguousarray(self.ap_free_mask, dtype=np.uint16)
        self.iop_free_mask = np.ascontiguousarray(self.iop_free_mask, dtype=np.uint8)
        if self.cam_in_use is not None:
            self.cam_in_use = np.ascontiguousarray(self.cam_in_use, dtype=np.uint8)

    @property
    def n_obs(self):
        return int(self.obs_pt.shape[0])

    @property
    def n_sta(self):
        return int(self.sta_cam.shape[0])

    @property
    def n_pts(self):
        return int(self.pt_prior.reshape(-1, 3).shape[0])

    @property
    def n_cam(self):
        return int(self.iop_prior.reshape(-1, 3).shape[0])

    def dims(self):
        return BaDims(self.n_obs, self.n_sta, self.n_pts, self.n_cam)

    def inputs(self):
        """ctypes view of the constant data; the Network must outlive the struct."""
        s = BaInputs()

        def dp(a):
            return a.ctypes.data_as(c_double_p)

        s.obs_pt = self.obs_pt.ctypes.data_as(c_int32_p)
        s.obs_sta = self.obs_sta.ctypes.data_as(c_int32_p)
        s.sta_cam = self.sta_cam.ctypes.data_as(c_int32_p)
        s.obs_x, s.obs_y = dp(self.obs_x), dp(self.obs_y)
        s.obs_sx, s.obs_sy = dp(self.obs_sx), dp(self.obs_sy)
        s.obs_cx, s.obs_cy = dp(self.obs_cx), dp(self.obs_cy)
        s.pt_prior, s.pt_sigma = dp(self.pt_prior), dp(self.pt_sigma)
        s.iop_prior, s.iop_sigma = dp(self.iop_prior), dp(self.iop_sigma)
        s.ap_free_mask = self.ap_free_mask.ctypes.data_as(C.POINTER(C.c_uint16))
        s.iop_free_mask = self.iop_free_mask.ctypes.data_as(C.POINTER(C.c_uint8))
        s.huber_a = float(self.huber_a)
        s.model = int(self.model)
        s.cam_in_use = None if self.cam_in_use is None else self.cam_in_use.ctypes.data_as(C.POINTER(C.c_uint8))
        return s

    def params_copy(self):
        return self.eop.copy(), self.xyz.copy(), self.iop.copy(), self.ap.copy()


def rows_to_dicts(rows, n):
    out = []
    for i in range(n):
        r = rows[i]
        out.append({name: getattr(r, name) for name, _ in BaIterRow._fields_})
    return out


def summary_to_dict(s):
    d = {name: getattr(s, name) for name, _ in BaSummary._fields_}
    d["stop_reason_name"] = STOP_REASONS.get(d["stop_reason"], "?")
    return d
