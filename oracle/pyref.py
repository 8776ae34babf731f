"""ctypes loader of oracle/_ref/*.so — TEST INFRASTRUCTURE, and (for libdftpav_ref*.so) RETIRED.

Rounds 3-5 compiled the reference's own solve-path sources (traj_optimizer.cpp, poly_traj_utils.hpp, lbfgs.hpp) from where they lie
under /root/reference against STAND-INS for Eigen, ROS and the protobuf config written in this repository (oracle/ref_shim; recipe
oracle/Makefile.ref) and used that build as the yardstick of the restatement.  The reference needs those external libraries and
generated code, so by this project's rules it is UNBUILDABLE here and a build against stand-ins is not a reference build: since
round 6 nothing builds it, nothing loads it, and no parity claim rests on it (parity UNPINNED -- the
reference holds no golden vectors and cannot be built).  The loader below only answers when DFTPAV_STANDIN_BUILD=1 is set by a
developer who wants to look at that historical cross-check (tests/test_ref_pin.py then runs instead of skipping).

What stays in use is the DROP-IN's own library (oracle/Makefile.dropin -> oracle/_ref/libdftpav_dropin.so): the PRODUCT's
implementation of the reference's class (dftpav_amd/csrc/host/dropin/traj_optimizer_hip.cpp) compiled against the reference's
unmodified header -- a compile / ABI check of the binding INTEGRATION.md describes, needing the same interface stand-ins to
compile, pinning nothing.
"""
import ctypes as C
import os
import subprocess

import numpy as np

from dftpav_amd.pods import Params, c_double_p, c_int_p, dptr, iptr
from oracle.pyoracle import EVAL_FN, Problem

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "_ref", "libdftpav_ref.so")
_LIB = None
# the same driver linked against the DROP-IN's implementation of the reference's class (oracle/Makefile.dropin:
# dftpav_amd/csrc/host/dropin/traj_optimizer_hip.cpp over libdftpav_hip.so) -- needs a GPU at run time
_SO_DROPIN = os.path.join(_HERE, "_ref", "libdftpav_dropin.so")
_LIB_DROPIN = None
# the reference's objects on a correctly rounded exp / log / pow / sin / cos (oracle/cr_libm.c): what oracle order 2 and the
# device's reference order are bit-equal to where the reference's loop calls libm
_SO_CR = os.path.join(_HERE, "_ref", "libdftpav_ref_cr.so")
_LIB_CR = None
# the reference over the stand-in Eigen with the reductions of dynamic vectors in Eigen 3.3's SSE2 order (DFTPAV_SHIM_EIGEN_REDUX=1)
_SO_EIGEN = os.path.join(_HERE, "_ref", "libdftpav_ref_eigen.so")
_LIB_EIGEN = None


def opted_in():
    return os.environ.get("DFTPAV_STANDIN_BUILD") == "1"


def build():
    """Retired (module docstring): runs the recipe only for a developer who opted in; otherwise nothing."""
    if opted_in():
        subprocess.check_call(["make", "-C", _HERE, "-f", "Makefile.ref", "-s"])
    return _SO


def available():
    return opted_in() and os.path.exists(_SO)


def cr_available():
    return opted_in() and os.path.exists(_SO_CR)


def cr_lib():
    global _LIB_CR
    if _LIB_CR is None:
        if not cr_available():
            build()
        L = C.CDLL(_SO_CR)
        _bind_common(L)
        _LIB_CR = L
    return _LIB_CR


def eigen_lib():
    global _LIB_EIGEN
    if _LIB_EIGEN is None:
        if not os.path.exists(_SO_EIGEN):
            build()
        L = C.CDLL(_SO_EIGEN)
        _bind_common(L)
        _LIB_EIGEN = L
    return _LIB_EIGEN


def dropin_available():
    return os.path.exists(_SO_DROPIN)


def build_dropin():
    subprocess.check_call(["make", "-C", _HERE, "-f", "Makefile.dropin", "-s"])
    return _SO_DROPIN


def dropin_lib():
    """The reference's PolyTrajOptimizer class with the drop-in's implementation behind it (GPU)."""
    global _LIB_DROPIN
    if _LIB_DROPIN is None:
        if not dropin_available():
            build_dropin()
        L = C.CDLL(_SO_DROPIN)
        _bind_common(L)
        assert L.ref_is_dropin() == 1
        _LIB_DROPIN = L
    return _LIB_DROPIN


def _bind_common(L):
    L.ref_prepare.restype = C.c_void_p
    L.ref_prepare.argtypes = [C.POINTER(Params), C.POINTER(Problem)]
    L.ref_free.argtypes = [C.c_void_p]
    L.ref_optimize.argtypes = [C.c_void_p, C.c_int, c_double_p, c_double_p, c_int_p, c_int_p, c_int_p]
    L.ref_num_vars.argtypes = [C.c_void_p]
    L.ref_trace_num_evals.argtypes = [C.c_void_p]
    L.ref_trace_num_iters.argtypes = [C.c_void_p]
    L.ref_trace_evals.argtypes = [C.c_void_p, c_double_p, c_double_p, c_double_p]
    L.ref_trace_iters.argtypes = [C.c_void_p, c_double_p, c_double_p, c_double_p, c_double_p, c_int_p, c_int_p]
    L.ref_eval.restype = C.c_double
    L.ref_eval.argtypes = [C.c_void_p, c_double_p, c_double_p]
    L.ref_last_coeffs.argtypes = [C.c_void_p, c_double_p, c_double_p]
    L.ref_surround_state.argtypes = [C.c_void_p, C.c_int, C.c_double, c_double_p]


def lib():
    global _LIB
    if _LIB is None:
        if not available():
            build()
        L = C.CDLL(_SO)
        L.ref_prepare.restype = C.c_void_p
        L.ref_prepare.argtypes = [C.POINTER(Params), C.POINTER(Problem)]
        L.ref_free.argtypes = [C.c_void_p]
        L.ref_optimize.argtypes = [C.c_void_p, C.c_int, c_double_p, c_double_p, c_int_p, c_int_p, c_int_p]
        L.ref_num_vars.argtypes = [C.c_void_p]
        L.ref_trace_num_evals.argtypes = [C.c_void_p]
        L.ref_trace_num_iters.argtypes = [C.c_void_p]
        L.ref_trace_evals.argtypes = [C.c_void_p, c_double_p, c_double_p, c_double_p]
        L.ref_trace_iters.argtypes = [C.c_void_p, c_double_p, c_double_p, c_double_p, c_double_p, c_int_p, c_int_p]
        L.ref_eval.restype = C.c_double
        L.ref_eval.argtypes = [C.c_void_p, c_double_p, c_double_p]
        L.ref_last_coeffs.argtypes = [C.c_void_p, c_double_p, c_double_p]
        L.ref_lbfgs.argtypes = [C.c_int, c_double_p, c_double_p, C.c_void_p, C.c_void_p, C.POINTER(Params), c_int_p, c_int_p]
        L.ref_banded_solve.argtypes = [C.c_int, C.c_int, C.c_int, c_double_p, C.c_int, c_double_p, C.c_int]
        L.ref_minco.restype = C.c_double
        L.ref_minco.argtypes = [C.c_int, c_double_p, C.c_double, c_double_p, c_double_p, c_double_p, c_double_p, c_double_p,
                                c_double_p, c_double_p, c_double_p]
        L.ref_surround_state.argtypes = [C.c_void_p, C.c_int, C.c_double, c_double_p]
        L.ref_smoothed_l1.argtypes = [C.POINTER(Params), C.c_double, c_double_p, c_double_p]
        L.ref_virtual_T_grad_cost.argtypes = [C.POINTER(Params), C.c_double, C.c_double, C.c_double, c_double_p, c_double_p]
        L.ref_log_sum_exp.restype = C.c_double
        L.ref_log_sum_exp.argtypes = [C.c_double, C.c_int, c_double_p, c_double_p]
        _LIB = L
    return _LIB


class RefProblem:
    """PolyTrajOptimizer of the reference on element b of a Scenario (same inputs as oracle.pyoracle.OracleProblem).
    dropin=True: the same class object with the drop-in's implementation behind it (libdftpav_dropin.so, GPU)."""

    def __init__(self, params, scen, b=0, dropin=False, cr=False, eigen_redux=False):
        """cr=True: the reference's objects linked against the correctly rounded libm of oracle/cr_libm.c;
        eigen_redux=True: the build whose dynamic-vector reductions add in Eigen 3.3's SSE2 order"""
        from oracle.pyoracle import OracleProblem
        self._L = dropin_lib() if dropin else (cr_lib() if cr else (eigen_lib() if eigen_redux else lib()))
        # reuse the flattening of the oracle's wrapper (plain arrays; nothing of the oracle's arithmetic is involved)
        self._flat = OracleProblem.__new__(OracleProblem)
        lay = scen.layout
        keep = dict(ini=np.ascontiguousarray(scen.ini_states[b]), fin=np.ascontiguousarray(scen.fin_states[b]),
                    inner=np.ascontiguousarray(scen.inner_pts[b]), Ts=np.ascontiguousarray(scen.init_Ts[b]),
                    cor=np.ascontiguousarray(scen.corridor[b]))
        pb = Problem()
        pb.M = lay.M
        pb.piece_nums = iptr(lay.piece_nums)
        pb.singuls = iptr(lay.singuls)
        pb.ini_states = dptr(keep["ini"])
        pb.fin_states = dptr(keep["fin"])
        pb.inner_pts = dptr(keep["inner"])
        pb.init_Ts = dptr(keep["Ts"])
        pb.H = lay.H
        pb.corridor = dptr(keep["cor"])
        pb.t_now = scen.t_now
        pb.help_eps = scen.help_eps
        self._sur = scen.surround.c_struct() if scen.surround is not None else None
        pb.surround = C.pointer(self._sur) if self._sur is not None else None
        self._keep, self.pb, self.scen = keep, pb, scen
        self.n = lay.n_vars
        self.ctx = self._L.ref_prepare(C.byref(params), C.byref(pb))
        self.result = None

    def __del__(self):
        if getattr(self, "ctx", None) and getattr(self, "_L", None) is not None:
            self._L.ref_free(self.ctx)
            self.ctx = None

    def optimize(self, trace=False):
        """OptimizeTrajectory.  Returns dict(ok, x, final_cost, status, iters, evals) (+ trace arrays)."""
        x = np.zeros(self.n)
        f, st, it, ev = C.c_double(0), C.c_int(0), C.c_int(0), C.c_int(0)
        ok = self._L.ref_optimize(self.ctx, int(trace), dptr(x), C.byref(f), C.byref(st), C.byref(it), C.byref(ev))
        r = dict(ok=bool(ok), x=x, final_cost=f.value, status=st.value, iters=it.value, evals=ev.value)
        if trace:
            ne, ni = self._L.ref_trace_num_evals(self.ctx), self._L.ref_trace_num_iters(self.ctx)
            ex, eg, ef = np.zeros((ne, self.n)), np.zeros((ne, self.n)), np.zeros(ne)
            self._L.ref_trace_evals(self.ctx, dptr(ex), dptr(eg), dptr(ef))
            ix, ig, ifx, istp = np.zeros((ni, self.n)), np.zeros((ni, self.n)), np.zeros(ni), np.zeros(ni)
            ik, ils = np.zeros(ni, dtype=np.int32), np.zeros(ni, dtype=np.int32)
            self._L.ref_trace_iters(self.ctx, dptr(ix), dptr(ig), dptr(ifx), dptr(istp), iptr(ik), iptr(ils))
            r.update(eval_x=ex, eval_g=eg, eval_f=ef, iter_x=ix, iter_g=ig, iter_fx=ifx, iter_step=istp, iter_k=ik, iter_ls=ils)
        self.result = r
        return r

    def last_choice(self):
        """drop-in only (DFTPAV_DROPIN_RESTARTS=K): dict(K, chosen, chosen_cost, n_success, n_colliding, solve_ms) of the last optimize()"""
        ch, ns, nc = C.c_int(0), C.c_int(0), C.c_int(0)
        cost, ms = C.c_double(0), C.c_double(0)
        K = self._L.ref_dropin_last_choice(C.c_void_p(self.ctx), C.byref(ch), C.byref(cost), C.byref(ns), C.byref(nc), C.byref(ms))
        return dict(K=K, chosen=ch.value, chosen_cost=cost.value, n_success=ns.value, n_colliding=nc.value, solve_ms=ms.value)

    def set_map(self, grid, resolution, origin):
        """drop-in only: the occupancy grid the restarts' candidates are re-checked on (uint8 [size_y][size_x], 80 = occupied)"""
        g = np.ascontiguousarray(grid, dtype=np.uint8)
        self._keep["grid"] = g
        return self._L.ref_dropin_set_map(C.c_void_p(self.ctx), g.ctypes.data_as(C.c_void_p), g.shape[1], g.shape[0], C.c_double(resolution),
                                          C.c_double(origin[0]), C.c_double(origin[1]))

    def eval(self, x):
        """costFunctionCallback at x (runs OptimizeTrajectory once first: its set-up lives in the object's members)."""
        if self.result is None:
            self.optimize()
        x = np.ascontiguousarray(x, dtype=np.float64)
        g = np.zeros(self.n)
        f = self._L.ref_eval(self.ctx, dptr(x), dptr(g))
        return f, g

    def coeffs(self):
        c = np.zeros((self.scen.layout.n_pieces, 6, 2))
        dt = np.zeros(self.scen.layout.M)
        self._L.ref_last_coeffs(self.ctx, dptr(c), dptr(dt))
        return c, dt

    def surround_state(self, o, t):
        out = np.zeros(14)
        self._L.ref_surround_state(self.ctx, int(o), float(t), dptr(out))
        return out


def lbfgs(fn, x0, params):
    """lbfgs::lbfgs_optimize of the reference with a Python callback fn(x) -> (f, g)."""
    x = np.ascontiguousarray(x0, dtype=np.float64).copy()
    n = len(x)

    def tramp(inst, xp, gp, nn):
        xv = np.ctypeslib.as_array(xp, shape=(nn,))
        f, g = fn(xv.copy())
        np.ctypeslib.as_array(gp, shape=(nn,))[:] = g
        return float(f)

    cb = EVAL_FN(tramp)
    f, it, ev = C.c_double(0), C.c_int(0), C.c_int(0)
    ret = lib().ref_lbfgs(n, dptr(x), C.byref(f), C.cast(cb, C.c_void_p), None, C.byref(params), C.byref(it), C.byref(ev))
    return dict(ret=ret, x=x, f=f.value, iters=it.value, evals=ev.value)


def banded_solve(A, p, q, b, adjoint=False):
    A = np.ascontiguousarray(A, dtype=np.float64)
    b = np.ascontiguousarray(b, dtype=np.float64).copy()
    if b.ndim == 1:
        b = b[:, None]
    lib().ref_banded_solve(A.shape[0], int(p), int(q), dptr(A), b.shape[1], dptr(b), int(adjoint))
    return b


def minco(inner, dT, head, tail, gdC_add=None, grads=True):
    """MinJerkOpt: returns dict(coeffs [N][6][2], energy, gdP [N-1][2], gdHead [3][2], gdTail [3][2], gdT)."""
    inner = np.ascontiguousarray(inner, dtype=np.float64)
    N = inner.shape[0] + 1
    c = np.zeros((N, 6, 2))
    gdP, gh, gt, gdT = np.zeros((N - 1, 2)), np.zeros((3, 2)), np.zeros((3, 2)), C.c_double(0)
    add = np.ascontiguousarray(gdC_add, dtype=np.float64) if gdC_add is not None else None
    e = lib().ref_minco(N, dptr(inner), float(dT), dptr(np.ascontiguousarray(head, dtype=np.float64)),
                        dptr(np.ascontiguousarray(tail, dtype=np.float64)), dptr(c), dptr(add) if add is not None else None,
                        dptr(gdP) if grads else None, dptr(gh), dptr(gt), C.byref(gdT))
    return dict(coeffs=c, energy=e, gdP=gdP, gdHead=gh, gdTail=gt, gdT=gdT.value)


def smoothed_l1(params, x):
    f, df = C.c_double(0), C.c_double(0)
    lib().ref_smoothed_l1(C.byref(params), float(x), C.byref(f), C.byref(df))
    return f.value, df.value


def log_sum_exp(alpha, dists):
    d = np.ascontiguousarray(dists, dtype=np.float64).copy()
    s = C.c_double(0)
    r = lib().ref_log_sum_exp(float(alpha), len(d), dptr(d), C.byref(s))
    return r, d, s.value
