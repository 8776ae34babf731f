"""ctypes loader of the CPU oracle — TEST INFRASTRUCTURE (see dftpav_oracle.h).

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg import this.
"""
import ctypes as C
import os
import subprocess

import numpy as np

from dftpav_amd.pods import (BatchData, Layout, Params, Surround, c_double_p, c_int_p, c_ll_p, dptr, iptr,
                             llptr)

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None


class Problem(C.Structure):
    _fields_ = [
        ("M", C.c_int),
        ("piece_nums", c_int_p),
        ("singuls", c_int_p),
        ("ini_states", c_double_p),
        ("fin_states", c_double_p),
        ("inner_pts", c_double_p),
        ("init_Ts", c_double_p),
        ("H", C.c_int),
        ("corridor", c_double_p),
        ("t_now", C.c_double),
        ("help_eps", C.c_double),
        ("surround", C.POINTER(Surround)),
    ]


class Result(C.Structure):
    _fields_ = [
        ("final_cost", C.c_double),
        ("status", C.c_int),
        ("success", C.c_int),
        ("iters", C.c_int),
        ("evals", C.c_int),
        ("hist_sum", C.c_longlong),
    ]


def build(force=False):
    so = os.path.join(_HERE, "libdftpav_oracle.so")
    srcs = [os.path.join(_HERE, f) for f in ("dftpav_oracle.c", "dftpav_oracle_dev.cpp", "dftpav_oracle.h",
                                             "oracle_internal.h")]
    srcs.append(os.path.join(_HERE, "..", "dftpav_amd", "csrc", "traj_math.h"))
    if force or not os.path.exists(so) or any(os.path.getmtime(so) < os.path.getmtime(s) for s in srcs):
        subprocess.check_call(["make", "-C", _HERE, "-s"])
    return so


def lib():
    global _LIB
    if _LIB is None:
        so = os.path.join(_HERE, "libdftpav_oracle.so")
        if not os.path.exists(so):
            build()
        L = C.CDLL(so)
        L.oracle_default_params.argtypes = [C.POINTER(Params)]
        L.oracle_num_vars.argtypes = [C.POINTER(Problem)]
        L.oracle_num_points.argtypes = [C.POINTER(Params), C.POINTER(Problem)]
        L.oracle_prepare.restype = C.c_void_p
        L.oracle_prepare.argtypes = [C.POINTER(Params), C.POINTER(Problem), c_int_p]
        L.oracle_free.argtypes = [C.c_void_p]
        L.oracle_set_order.argtypes = [C.c_void_p, C.c_int]
        L.oracle_pack_x0.argtypes = [C.c_void_p, c_double_p]
        L.oracle_eval.restype = C.c_double
        L.oracle_eval.argtypes = [C.c_void_p, c_double_p, c_double_p]
        L.oracle_last_cost_terms.argtypes = [C.c_void_p, c_double_p]
        L.oracle_last_coeffs.argtypes = [C.c_void_p, c_double_p, c_double_p]
        L.oracle_solve.argtypes = [C.c_void_p, c_double_p, C.POINTER(Result)]
        L.oracle_solve_batch.argtypes = [C.POINTER(Params), C.POINTER(Layout), C.c_int, C.POINTER(BatchData),
                                         C.POINTER(Surround), C.c_int, C.c_int, c_double_p, c_double_p, c_int_p, c_int_p,
                                         c_int_p, c_int_p, c_ll_p, c_double_p]
        L.oracle_minco_generate.restype = C.c_double
        L.oracle_minco_generate.argtypes = [C.c_int, c_double_p, C.c_double, c_double_p, c_double_p, c_double_p]
        L.oracle_minco_operator.argtypes = [C.c_int, c_double_p]
        L.oracle_smoothed_l1.argtypes = [C.c_double, c_double_p, c_double_p]
        L.oracle_virtual_to_real_T.restype = C.c_double
        L.oracle_virtual_to_real_T.argtypes = [C.POINTER(Params), C.c_double]
        L.oracle_real_to_virtual_T.restype = C.c_double
        L.oracle_real_to_virtual_T.argtypes = [C.POINTER(Params), C.c_double]
        L.oracle_lbfgs.argtypes = [C.c_int, c_double_p, c_double_p, C.c_void_p, C.c_void_p, C.POINTER(Params),
                                   c_int_p, c_int_p, c_ll_p]
        _LIB = L
    return _LIB


EVAL_FN = C.CFUNCTYPE(C.c_double, C.c_void_p, c_double_p, c_double_p, C.c_int)


def default_params():
    p = Params()
    lib().oracle_default_params(C.byref(p))
    return p


class OracleProblem:
    """One prepared trajectory (element b of a Scenario)."""

    def __init__(self, params, scen, b=0, order=0):
        self.params = params
        self.scen = scen
        lay = scen.layout
        self._keep = dict(
            ini=np.ascontiguousarray(scen.ini_states[b]), fin=np.ascontiguousarray(scen.fin_states[b]),
            inner=np.ascontiguousarray(scen.inner_pts[b]), Ts=np.ascontiguousarray(scen.init_Ts[b]),
            cor=np.ascontiguousarray(scen.corridor[b]))
        pb = Problem()
        pb.M = lay.M
        pb.piece_nums = iptr(lay.piece_nums)
        pb.singuls = iptr(lay.singuls)
        pb.ini_states = dptr(self._keep["ini"])
        pb.fin_states = dptr(self._keep["fin"])
        pb.inner_pts = dptr(self._keep["inner"])
        pb.init_Ts = dptr(self._keep["Ts"])
        pb.H = lay.H
        pb.corridor = dptr(self._keep["cor"])
        pb.t_now = scen.t_now
        pb.help_eps = scen.help_eps
        self._sur = scen.surround.c_struct() if scen.surround is not None else None
        pb.surround = C.pointer(self._sur) if self._sur is not None else None
        self.pb = pb
        self.n = lay.n_vars
        err = C.c_int(0)
        self.ctx = lib().oracle_prepare(C.byref(params), C.byref(pb), C.byref(err))
        self.err = err.value
        if not self.ctx:
            raise ValueError("oracle_prepare failed: %d" % err.value)
        if order:
            self.set_order(order)

    def set_order(self, order):
        """0 = literal reference statement order, 1 = device order, 2 = literal with correctly rounded cos / sin of the
        junction angles (see dftpav_oracle.h)."""
        lib().oracle_set_order(self.ctx, order)

    def __del__(self):
        if getattr(self, "ctx", None):
            lib().oracle_free(self.ctx)
            self.ctx = None

    def x0(self):
        x = np.zeros(self.n)
        lib().oracle_pack_x0(self.ctx, dptr(x))
        return x

    def eval(self, x):
        x = np.ascontiguousarray(x, dtype=np.float64)
        g = np.zeros(self.n)
        f = lib().oracle_eval(self.ctx, dptr(x), dptr(g))
        return f, g

    def cost_terms(self):
        t = np.zeros(5)
        lib().oracle_last_cost_terms(self.ctx, dptr(t))
        return t

    def coeffs(self):
        c = np.zeros((self.scen.layout.n_pieces, 6, 2))
        dt = np.zeros(self.scen.layout.M)
        lib().oracle_last_coeffs(self.ctx, dptr(c), dptr(dt))
        return c, dt

    def solve(self, x=None):
        x = self.x0() if x is None else np.ascontiguousarray(x, dtype=np.float64).copy()
        r = Result()
        lib().oracle_solve(self.ctx, dptr(x), C.byref(r))
        return x, r


LITERAL, DEVICE_ORDER = 0, 1


def solve_batch(params, scen, nthreads=1, order=0):
    """oracle_solve_batch over every element of a Scenario. Returns a dict of arrays."""
    B, n = scen.B, scen.layout.n_vars
    x = np.zeros((B, n))
    fc = np.zeros(B)
    status = np.zeros(B, dtype=np.int32)
    success = np.zeros(B, dtype=np.int32)
    iters = np.zeros(B, dtype=np.int32)
    evals = np.zeros(B, dtype=np.int32)
    hist = np.zeros(B, dtype=np.int64)
    secs = np.zeros(B)
    lay = scen.layout.c_struct()
    d = scen.batch_data()
    sur = scen.surround.c_struct() if scen.surround is not None else None
    rc = lib().oracle_solve_batch(C.byref(params), C.byref(lay), B, C.byref(d),
                                  C.byref(sur) if sur is not None else None, nthreads, order, dptr(x), dptr(fc),
                                  iptr(status), iptr(success), iptr(iters), iptr(evals), llptr(hist), dptr(secs))
    return dict(rc=rc, x=x, final_cost=fc, status=status, success=success, iters=iters, evals=evals,
                hist_sum=hist, seconds=secs)


def batch_op(params, scen, op, x, nthreads=1, order=0):
    """oracle_batch_op: op 'restart' = lbfgs_optimize from the rows of x, 'eval' = f, g at the rows of x."""
    B, n = scen.B, scen.layout.n_vars
    x = np.ascontiguousarray(x, dtype=np.float64).copy()
    assert x.shape == (B, n)
    g = np.zeros((B, n))
    fc = np.zeros(B)
    status = np.zeros(B, dtype=np.int32)
    success = np.zeros(B, dtype=np.int32)
    iters = np.zeros(B, dtype=np.int32)
    evals = np.zeros(B, dtype=np.int32)
    hist = np.zeros(B, dtype=np.int64)
    secs = np.zeros(B)
    lay = scen.layout.c_struct()
    d = scen.batch_data()
    sur = scen.surround.c_struct() if scen.surround is not None else None
    fn = lib().oracle_batch_op
    fn.argtypes = [C.POINTER(Params), C.POINTER(Layout), C.c_int, C.POINTER(BatchData), C.POINTER(Surround), C.c_int, C.c_int, C.c_int,
                   c_double_p, c_double_p, c_double_p, c_int_p, c_int_p, c_int_p, c_int_p, c_ll_p, c_double_p]
    rc = fn(C.byref(params), C.byref(lay), B, C.byref(d), C.byref(sur) if sur is not None else None, nthreads, order,
            {"solve": 0, "restart": 1, "eval": 2}[op], dptr(x), dptr(g), dptr(fc), iptr(status), iptr(success), iptr(iters),
            iptr(evals), llptr(hist), dptr(secs))
    return dict(rc=rc, x=x, g=g, final_cost=fc, f=fc, status=status, success=success, iters=iters, evals=evals, hist_sum=hist,
                seconds=secs)


def lbfgs(fn, x0, params=None):
    """oracle_lbfgs with a Python callback fn(x)->(f,g)."""
    params = params or default_params()
    x = np.ascontiguousarray(x0, dtype=np.float64).copy()
    n = len(x)

    def tramp(inst, xp, gp, nn):
        xv = np.ctypeslib.as_array(xp, shape=(nn,))
        f, g = fn(xv.copy())
        np.ctypeslib.as_array(gp, shape=(nn,))[:] = g
        return float(f)

    cb = EVAL_FN(tramp)
    f = C.c_double(0)
    it = C.c_int(0)
    ev = C.c_int(0)
    hs = C.c_longlong(0)
    ret = lib().oracle_lbfgs(n, dptr(x), C.byref(f), C.cast(cb, C.c_void_p), None, C.byref(params), C.byref(it),
                             C.byref(ev), C.byref(hs))
    return dict(ret=ret, x=x, f=f.value, iters=it.value, evals=ev.value, hist_sum=hs.value)


def minco_operator(N):
    out = np.zeros((6 * N, N + 5))
    lib().oracle_minco_operator(N, dptr(out))
    return out


def minco_generate(inner, dT, head, tail):
    inner = np.ascontiguousarray(inner, dtype=np.float64)
    N = inner.shape[0] + 1
    c = np.zeros((N, 6, 2))
    J = lib().oracle_minco_generate(N, dptr(inner), dT, dptr(np.ascontiguousarray(head, dtype=np.float64)),
                                    dptr(np.ascontiguousarray(tail, dtype=np.float64)), dptr(c))
    return c, J

_REF_NEXT = None


def ref_next_lib():
    """oracle/_ref/libdftpav_ref_next.so: the REFERENCE'S OWN code of the steps either side of the solve path (functions cut
    verbatim out of /root/reference by oracle/ref_slices.py, compiled by oracle/ref_next_driver.cpp; recipe oracle/Makefile.ref).
    The wrappers below take `ref=True` to run it in the restatement's place (order 0 only): what tests/test_ref_pin.py compares."""
    global _REF_NEXT
    if _REF_NEXT is None:
        so = os.path.join(_HERE, "_ref", "libdftpav_ref_next.so")
        if not os.path.exists(so):
            subprocess.check_call(["make", "-C", _HERE, "-f", "Makefile.ref", "-s"])
        _REF_NEXT = C.CDLL(so)
    return _REF_NEXT


def ref_next_available():
    return os.path.exists(os.path.join(_HERE, "_ref", "libdftpav_ref_next.so")) or os.path.exists("/root/reference/src/Plan/traj_planner/src/traj_manager.cpp")


def _step_fn(name, ref):
    """oracle_<name> of the restatement, or ref_<name> of the reference's own code (returns 0, or < 0 where the reference
    hard-codes what the restatement takes as an argument)"""
    if not ref:
        fn = getattr(lib(), "oracle_" + name)
        fn.restype = None
        return fn
    raw = getattr(ref_next_lib(), "ref_" + name)
    raw.restype = C.c_int

    def call(*a):
        rc = raw(*a)
        if rc != 0:
            raise ValueError("ref_%s: the reference's code does not take these arguments (rc %d)" % (name, rc))
    call.raw = raw
    return call


def corridor_rectangles(grid, resolution, origin, states, veh=(1.90, 4.88, 1.015), order=0, ref=False):
    """getRectangleConst (traj_manager.cpp:1213-1469) on an occupancy grid.

    grid: uint8 [size_y][size_x] (cell (ix, iy) at grid[iy, ix], 80 = occupied); states: [n][3] (x, y, yaw).
    Returns [n][4][4]: per state the four columns (n_x, n_y, p_x, p_y) of hPoly."""
    L = lib()
    g = np.ascontiguousarray(grid, dtype=np.uint8)
    st = np.ascontiguousarray(states, dtype=np.float64).reshape(-1, 3)
    out = np.zeros((st.shape[0], 4, 4), dtype=np.float64)
    fn = _step_fn("corridor_rectangles", ref)
    (fn.raw if ref else fn).argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_double, C.c_double, C.c_double, C.c_void_p, C.c_int,
                   C.c_double, C.c_double, C.c_double, C.c_int, C.c_void_p]
    fn(g.ctypes.data, g.shape[1], g.shape[0], float(resolution), float(origin[0]), float(origin[1]), st.ctypes.data,
       st.shape[0], float(veh[0]), float(veh[1]), float(veh[2]), int(order), out.ctypes.data)
    return out


def validate_trajectories(grid, resolution, origin, coeffs, piece_dt, piece_nums, singuls, veh=(1.90, 4.88, 1.015),
                          sample_dt=0.05, vertex_res=0.1, order=0, ref=False):
    """The collision re-check of CheckReplan (traj_server_ros.cpp:385-397) for B trajectories.

    coeffs: [B][Ntot][6][2] (entry [k][d] = coefficient of s^k), piece_dt: [B][M].  Returns (collision [B], first_sample [B])."""
    L = lib()
    g = np.ascontiguousarray(grid, dtype=np.uint8)
    co = np.ascontiguousarray(coeffs, dtype=np.float64)
    B = co.shape[0]
    dt = np.ascontiguousarray(piece_dt, dtype=np.float64).reshape(B, -1)
    pn = np.ascontiguousarray(piece_nums, dtype=np.int32)
    sg = np.ascontiguousarray(singuls, dtype=np.int32)
    col = np.zeros(B, dtype=np.int32)
    first = np.zeros(B, dtype=np.int32)
    fn = _step_fn("validate_trajectories", ref)
    (fn.raw if ref else fn).argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_double, C.c_double, C.c_double, C.c_void_p, C.c_void_p, C.c_void_p,
                   C.c_void_p, C.c_int, C.c_int, C.c_double, C.c_double, C.c_double, C.c_double, C.c_double, C.c_int,
                   C.c_void_p, C.c_void_p]
    fn(g.ctypes.data, g.shape[1], g.shape[0], float(resolution), float(origin[0]), float(origin[1]), co.ctypes.data,
       dt.ctypes.data, pn.ctypes.data, sg.ctypes.data, len(pn), B, float(veh[0]), float(veh[1]), float(veh[2]),
       float(sample_dt), float(vertex_res), int(order), col.ctypes.data, first.ctypes.data)
    return col, first


def sample_states(coeffs, piece_dt, piece_nums, singuls, t0=0.0, sample_dt=0.01, n_samples=100, filter_singularity=True,
                  wheel_base=2.85, order=0, ref=False):
    """Trajectory::GetState over a time grid, played back as the server does (states_oracle.cpp): returns
    (states [B][n_samples][8], n_valid [B])."""
    L = lib()
    co = np.ascontiguousarray(coeffs, dtype=np.float64)
    B = co.shape[0]
    dt = np.ascontiguousarray(piece_dt, dtype=np.float64).reshape(B, -1)
    pn = np.ascontiguousarray(piece_nums, dtype=np.int32)
    sg = np.ascontiguousarray(singuls, dtype=np.int32)
    st = np.zeros((B, int(n_samples), 8))
    nv = np.zeros(B, dtype=np.int32)
    fn = _step_fn("sample_states", ref)
    (fn.raw if ref else fn).argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_double, C.c_double, C.c_double,
                   C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p]
    fn(co.ctypes.data, dt.ctypes.data, pn.ctypes.data, sg.ctypes.data, len(pn), B, float(wheel_base), float(t0),
       float(sample_dt), int(n_samples), int(bool(filter_singularity)), int(order), st.ctypes.data, nv.ctypes.data)
    return st, nv


def reeds_shepp_shots(from_, to, max_cur=1.0, checkl=0.2, max_samples=512, grid=None, resolution=0.3, origin=(0.0, 0.0),
                      veh=(1.90, 4.88, 1.015), vertex_res=0.1, order=0):
    """KinoAstar::computeShotTraj / is_shot_sucess (kino_astar.cpp:304-345) for n pose pairs: dict(length [n], type [n],
    seg [n][5], samples [n][max_samples][3], n_samples [n], collides [n] or None)."""
    L = lib()
    f = np.ascontiguousarray(from_, dtype=np.float64).reshape(-1, 3)
    t = np.ascontiguousarray(to, dtype=np.float64).reshape(-1, 3)
    n = f.shape[0]
    out = dict(length=np.zeros(n), type=np.zeros(n, dtype=np.int32), seg=np.zeros((n, 5)),
               samples=np.zeros((n, int(max_samples), 3)), n_samples=np.zeros(n, dtype=np.int32),
               collides=np.zeros(n, dtype=np.int32) if grid is not None else None)
    g = np.ascontiguousarray(grid, dtype=np.uint8) if grid is not None else None
    fn = L.oracle_reeds_shepp_shots
    fn.restype = None
    fn.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_double, C.c_double, C.c_double, C.c_void_p, C.c_void_p, C.c_int, C.c_double,
                   C.c_double, C.c_int, C.c_double, C.c_double, C.c_double, C.c_double, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p,
                   C.c_void_p, C.c_void_p, C.c_void_p]
    fn(g.ctypes.data if g is not None else None, g.shape[1] if g is not None else 0, g.shape[0] if g is not None else 0,
       float(resolution), float(origin[0]), float(origin[1]), f.ctypes.data, t.ctypes.data, n, 1.0 / float(max_cur), float(checkl),
       int(max_samples), float(veh[0]), float(veh[1]), float(veh[2]), float(vertex_res), int(order), out["length"].ctypes.data,
       out["type"].ctypes.data, out["seg"].ctypes.data, out["samples"].ctypes.data, out["n_samples"].ctypes.data,
       out["collides"].ctypes.data if out["collides"] is not None else None)
    return out


def reeds_shepp_literal(from_, to, max_cur=1.0, checkl=0.2, max_samples=512):
    """The independent Reeds-Shepp restatement (oracle/shot_oracle_literal.cpp: the paper's base words x symmetries, libm,
    every candidate validated by integration): dict(length, kinds [n][5], seg [n][5], samples, n_samples, n_valid)."""
    L = lib()
    f = np.ascontiguousarray(from_, dtype=np.float64).reshape(-1, 3)
    t = np.ascontiguousarray(to, dtype=np.float64).reshape(-1, 3)
    n = f.shape[0]
    out = dict(length=np.zeros(n), kinds=np.zeros((n, 5), dtype=np.int32), seg=np.zeros((n, 5)),
               samples=np.zeros((n, int(max_samples), 3)), n_samples=np.zeros(n, dtype=np.int32), n_valid=np.zeros(n, dtype=np.int32))
    fn = L.oracle_reeds_shepp_literal
    fn.restype = None
    fn.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_double, C.c_double, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
                   C.c_void_p, C.c_void_p]
    fn(f.ctypes.data, t.ctypes.data, n, 1.0 / float(max_cur), float(checkl), int(max_samples), out["length"].ctypes.data,
       out["kinds"].ctypes.data, out["seg"].ctypes.data, out["samples"].ctypes.data, out["n_samples"].ctypes.data,
       out["n_valid"].ctypes.data)
    return out


# segment kinds of the 18 path types of dftpav_amd/csrc/rs_math.h (0 none, 1 left, 2 straight, 3 right)
RS_TYPE_KINDS = np.array([[1, 3, 1, 0, 0], [3, 1, 3, 0, 0], [1, 3, 1, 3, 0], [3, 1, 3, 1, 0], [1, 3, 2, 1, 0], [3, 1, 2, 3, 0],
                          [1, 2, 3, 1, 0], [3, 2, 1, 3, 0], [1, 3, 2, 3, 0], [3, 1, 2, 1, 0], [3, 2, 3, 1, 0], [1, 2, 1, 3, 0],
                          [1, 2, 3, 0, 0], [3, 2, 1, 0, 0], [1, 2, 1, 0, 0], [3, 2, 3, 0, 0], [1, 3, 2, 1, 3], [3, 1, 2, 3, 1]],
                         dtype=np.int32)


def fit_surround(states, order=0, ref=False):
    """ConverSurroundTrajFromPoints (traj_manager.cpp:743-789): states [S][n][7] (x, y, angle, velocity, acceleration,
    curvature, time_stamp) -> dict(durations [S][n-1], coeffs [S][n-1][12], total [S], start [S])."""
    L = lib()
    st = np.ascontiguousarray(states, dtype=np.float64)
    S, n = st.shape[0], st.shape[1]
    out = dict(durations=np.zeros((S, n - 1)), coeffs=np.zeros((S, n - 1, 12)), total=np.zeros(S), start=np.zeros(S))
    fn = _step_fn("fit_surround", ref)
    (fn.raw if ref else fn).argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
    fn(st.ctypes.data, S, n, int(order), out["durations"].ctypes.data, out["coeffs"].ctypes.data, out["total"].ctypes.data,
       out["start"].ctypes.data)
    return out


def frontend_resample(paths, path_len, start_states, end_states, start_ctrl, fparams=None, order=0, ref=False, **caps):
    """getKinoNode (from SampleTraj on) + the resampling of RunMINCOParking (kino_astar.cpp:606-795, traj_manager.cpp:531-568).
    paths [n_hyp][max_path][3]; returns the dict of padded arrays of dftpav_amd.pods.FrontendOut."""
    from dftpav_amd.pods import FrontendParams, FrontendOut
    L = lib()
    P = np.ascontiguousarray(paths, dtype=np.float64)
    n_hyp, max_path = P.shape[0], P.shape[1]
    pl = np.ascontiguousarray(path_len, dtype=np.int32)
    ss = np.ascontiguousarray(start_states, dtype=np.float64)
    es = np.ascontiguousarray(end_states, dtype=np.float64)
    sc_ = np.ascontiguousarray(start_ctrl, dtype=np.float64)
    fp = fparams if fparams is not None else FrontendParams.default()
    out = FrontendOut(n_hyp, **caps)
    fn = _step_fn("frontend_resample", ref)
    (fn.raw if ref else fn).argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_void_p]
    fn(C.byref(fp), P.ctypes.data, pl.ctypes.data, max_path, ss.ctypes.data, es.ctypes.data, sc_.ctypes.data, n_hyp, int(order),
       C.byref(out.c))
    return out.arrays()


def sample_restarts(inner, durs, n_restarts, sigma=0.3, lo=0.8, hi=1.25, seed=0):
    """The restart sampler of dftpav_amd/csrc/restart.hip on the CPU: inner [n_hyp][n_inner], durs [n_hyp][M] ->
    (inner [n_hyp * n_restarts][n_inner], durs [n_hyp * n_restarts][M]), trajectory b = hypothesis * n_restarts + restart."""
    L = lib()
    a = np.ascontiguousarray(inner, dtype=np.float64)
    d = np.ascontiguousarray(durs, dtype=np.float64)
    n_hyp, n_inner, M = a.shape[0], a.shape[1], d.shape[1]
    oi = np.zeros((n_hyp * n_restarts, n_inner))
    od = np.zeros((n_hyp * n_restarts, M))
    fn = L.oracle_sample_restarts
    fn.restype = None
    fn.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_double, C.c_double, C.c_double,
                   C.c_ulonglong, C.c_void_p, C.c_void_p]
    fn(a.ctypes.data, d.ctypes.data, n_hyp, n_restarts, n_inner, M, float(sigma), float(lo), float(hi), int(seed),
       oi.ctypes.data, od.ctypes.data)
    return oi, od
