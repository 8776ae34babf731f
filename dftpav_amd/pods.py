"""ctypes mirrors of the plain-C PODs declared in include/dftpav_hip.h.

Field order and types must match the header exactly; tests/test_abi.py checks
sizeof/offsets against the compiled library (dftpav_abi_sizeof_*).
"""
import ctypes as C

import numpy as np

c_double_p = C.POINTER(C.c_double)
c_int_p = C.POINTER(C.c_int)
c_ll_p = C.POINTER(C.c_longlong)


class Params(C.Structure):
    """dftpav_params (include/dftpav_hip.h) == OptCfg + VehicleParam + lbfgs_parameter_t."""
    _fields_ = [
        ("traj_resolution", C.c_int),
        ("des_traj_resolution", C.c_int),
        ("wei_obs", C.c_double),
        ("wei_surround", C.c_double),
        ("wei_feas", C.c_double),
        ("wei_sqrvar", C.c_double),
        ("wei_time", C.c_double),
        ("surround_clearance", C.c_double),
        ("half_margin", C.c_double),
        ("max_forward_vel", C.c_double),
        ("max_forward_acc", C.c_double),
        ("max_forward_cur", C.c_double),
        ("max_backward_vel", C.c_double),
        ("max_backward_acc", C.c_double),
        ("max_backward_cur", C.c_double),
        ("max_latacc", C.c_double),
        ("max_phidot", C.c_double),
        ("gear_opt", C.c_int),
        ("non_sinv", C.c_double),
        ("mini_T", C.c_double),
        ("fail_cost", C.c_double),
        ("veh_width", C.c_double),
        ("veh_length", C.c_double),
        ("veh_wheel_base", C.c_double),
        ("veh_d_cr", C.c_double),
        ("lbfgs_mem_size", C.c_int),
        ("lbfgs_past", C.c_int),
        ("lbfgs_delta", C.c_double),
        ("lbfgs_g_epsilon", C.c_double),
        ("lbfgs_max_iterations", C.c_int),
        ("lbfgs_max_linesearch", C.c_int),
        ("lbfgs_min_step", C.c_double),
        ("lbfgs_max_step", C.c_double),
        ("lbfgs_f_dec_coeff", C.c_double),
        ("lbfgs_s_curv_coeff", C.c_double),
        ("lbfgs_cautious_factor", C.c_double),
        ("lbfgs_machine_prec", C.c_double),
    ]

    def as_dict(self):
        return {n: getattr(self, n) for n, _ in self._fields_}


class Surround(C.Structure):
    _fields_ = [
        ("S", C.c_int),
        ("piece_offsets", c_int_p),
        ("durations", c_double_p),
        ("coeffs", c_double_p),
        ("total_duration", c_double_p),
        ("start_time", c_double_p),
    ]


class Layout(C.Structure):
    _fields_ = [
        ("M", C.c_int),
        ("piece_nums", c_int_p),
        ("singuls", c_int_p),
        ("H", C.c_int),
    ]


class BatchData(C.Structure):
    _fields_ = [
        ("ini_states", c_double_p),
        ("fin_states", c_double_p),
        ("inner_pts", c_double_p),
        ("init_Ts", c_double_p),
        ("corridor", c_double_p),
        ("t_now", C.c_double),
        ("help_eps", C.c_double),
    ]


def dptr(a):
    assert a.dtype == np.float64 and a.flags["C_CONTIGUOUS"]
    return a.ctypes.data_as(c_double_p)


def iptr(a):
    assert a.dtype == np.int32 and a.flags["C_CONTIGUOUS"]
    return a.ctypes.data_as(c_int_p)


def llptr(a):
    assert a.dtype == np.int64 and a.flags["C_CONTIGUOUS"]
    return a.ctypes.data_as(c_ll_p)


class SurroundSet:
    """Owner of the arrays behind a dftpav_surround (keeps numpy buffers alive)."""

    def __init__(self, piece_offsets, durations, coeffs, total_duration, start_time):
        self.piece_offsets = np.ascontiguousarray(piece_offsets, dtype=np.int32)
        self.durations = np.ascontiguousarray(durations, dtype=np.float64)
        self.coeffs = np.ascontiguousarray(coeffs, dtype=np.float64).reshape(-1, 12)
        self.total_duration = np.ascontiguousarray(total_duration, dtype=np.float64)
        self.start_time = np.ascontiguousarray(start_time, dtype=np.float64)
        self.S = len(self.total_duration)
        assert self.piece_offsets.shape == (self.S + 1,)
        assert self.durations.shape[0] == self.piece_offsets[-1] == self.coeffs.shape[0]

    def c_struct(self):
        s = Surround()
        s.S = self.S
        s.piece_offsets = iptr(self.piece_offsets)
        s.durations = dptr(self.durations)
        s.coeffs = dptr(self.coeffs)
        s.total_duration = dptr(self.total_duration)
        s.start_time = dptr(self.start_time)
        return s


class LayoutSpec:
    """Owner of a dftpav_layout."""

    def __init__(self, piece_nums, singuls, H=4):
        self.piece_nums = np.ascontiguousarray(piece_nums, dtype=np.int32)
        self.singuls = np.ascontiguousarray(singuls, dtype=np.int32)
        self.M = len(self.piece_nums)
        self.H = int(H)
        assert self.singuls.shape == (self.M,)

    @property
    def n_vars(self):
        M = self.M
        return int(2 * np.sum(self.piece_nums - 1) + M + 3 * (M - 1))

    @property
    def n_inner(self):
        return int(2 * np.sum(self.piece_nums - 1))

    @property
    def n_pieces(self):
        return int(np.sum(self.piece_nums))

    def n_points(self, K, Kd):
        return int(sum((int(N) - 2) * (K + 1) + 2 * (Kd + 1) for N in self.piece_nums))

    def c_struct(self):
        l = Layout()
        l.M = self.M
        l.piece_nums = iptr(self.piece_nums)
        l.singuls = iptr(self.singuls)
        l.H = self.H
        return l


class FrontendParams(C.Structure):
    """dftpav_frontend_params (include/dftpav_hip.h), defaults of minco_config.pb.txt:66-67,76-80."""
    _fields_ = [("max_forward_vel", C.c_double), ("max_forward_acc", C.c_double), ("max_backward_vel", C.c_double),
                ("max_backward_acc", C.c_double), ("non_siguav", C.c_double), ("wheel_base", C.c_double),
                ("piece_duration", C.c_double), ("traj_res", C.c_int), ("dense_traj_res", C.c_int)]

    @classmethod
    def default(cls, K=16, Kd=32):
        return cls(5.0, 8.0, 2.0, 4.0, 0.2, 2.85, 1.0, K, Kd)


class FrontendOutC(C.Structure):
    _fields_ = [("max_seg", C.c_int), ("max_pieces", C.c_int), ("max_states", C.c_int), ("n_seg", C.c_void_p),
                ("singul", C.c_void_p), ("piece_nums", C.c_void_p), ("piece_dt", C.c_void_p), ("ini_states", C.c_void_p),
                ("fin_states", C.c_void_p), ("inner_pts", C.c_void_p), ("n_states", C.c_void_p), ("states", C.c_void_p)]


class FrontendOut:
    """Owner of the padded output arrays of dftpav_frontend_resample."""

    def __init__(self, n_hyp, max_seg=8, max_pieces=64, max_states=2304):
        self.n_seg = np.zeros(n_hyp, dtype=np.int32)
        self.singul = np.zeros((n_hyp, max_seg), dtype=np.int32)
        self.piece_nums = np.zeros((n_hyp, max_seg), dtype=np.int32)
        self.piece_dt = np.zeros((n_hyp, max_seg))
        self.ini_states = np.zeros((n_hyp, max_seg, 6))
        self.fin_states = np.zeros((n_hyp, max_seg, 6))
        self.inner_pts = np.zeros((n_hyp, max_seg, max_pieces - 1, 2))
        self.n_states = np.zeros((n_hyp, max_seg), dtype=np.int32)
        self.states = np.zeros((n_hyp, max_seg, max_states, 3))
        self.c = FrontendOutC(max_seg, max_pieces, max_states, *[a.ctypes.data for a in (
            self.n_seg, self.singul, self.piece_nums, self.piece_dt, self.ini_states, self.fin_states, self.inner_pts,
            self.n_states, self.states)])

    def arrays(self):
        return dict(n_seg=self.n_seg, singul=self.singul, piece_nums=self.piece_nums, piece_dt=self.piece_dt,
                    ini_states=self.ini_states, fin_states=self.fin_states, inner_pts=self.inner_pts, n_states=self.n_states,
                    states=self.states)
