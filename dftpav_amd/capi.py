"""ctypes binding of libdftpav_hip.so (the C-ABI of include/dftpav_hip.h).

This is the Python host side used by tests and bench.py; the mirror of the
reference's class interface lives in optimizer.py.  The library is built
in-tree by dftpav_amd/csrc/Makefile; if it is missing this module raises — there
is no fallback path.
"""
import ctypes as C
import os
import subprocess

import numpy as np

from .pods import (BatchData, Layout, Params, Surround, c_double_p, c_int_p, c_ll_p, dptr, iptr, llptr)

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("DFTPAV_LIB") or os.path.join(_HERE, "libdftpav_hip.so")  # DFTPAV_LIB: an experimental build (scripts/build_variant.sh)
_LIB = None

OK = 0
ORDER_DEVICE, ORDER_REFERENCE = 0, 1
E_INVALID, E_MINI_T, E_ONE_PIECE, E_NO_DEVICE, E_HIP, E_UNSUPPORTED, E_COMM = -1, -2, -3, -4, -5, -6, -7

# every symbol include/dftpav_hip.h declares
EXPORTS = [
    "dftpav_default_params", "dftpav_num_vars", "dftpav_num_points", "dftpav_create", "dftpav_destroy",
    "dftpav_last_error", "dftpav_set_surround", "dftpav_batch_create", "dftpav_batch_destroy",
    "dftpav_batch_upload", "dftpav_batch_get_x0", "dftpav_batch_eval", "dftpav_batch_solve_async",
    "dftpav_batch_sync", "dftpav_batch_results", "dftpav_batch_pack_results", "dftpav_batch_records", "dftpav_batch_coeffs", "dftpav_batch_last_solve_ms",
    "dftpav_solve_batch", "dftpav_stream", "dftpav_set_grid_map", "dftpav_corridor_rectangles",
    "dftpav_corridor_last_ms", "dftpav_batch_corridor_from_states", "dftpav_batch_validate",
    "dftpav_fit_surround", "dftpav_get_surround", "dftpav_frontend_resample",
    "dftpav_sample_restarts", "dftpav_batch_corridor_from_hypotheses", "dftpav_batch_sample_states",
    "dftpav_reeds_shepp_shots", "dftpav_mark", "dftpav_marks_elapsed_ms", "dftpav_batch_set_hand_over", "dftpav_batch_solve_chained", "dftpav_batch_finish", "dftpav_wire_size", "dftpav_wire_pack", "dftpav_wire_info", "dftpav_wire_unpack", "dftpav_set_surround_wire",
    "dftpav_batch_trace", "dftpav_batch_get_trace", "dftpav_plan_cycle", "dftpav_plan_cycle_fetch", "dftpav_batch_create_shaped",
    "dftpav_batch_set_order", "dftpav_batch_get_order", "dftpav_batch_trace_range", "dftpav_batch_get_trace_of",
    "dftpav_comm_available", "dftpav_comm_unique_id", "dftpav_comm_create", "dftpav_comm_destroy", "dftpav_comm_share", "dftpav_comm_layout", "dftpav_batch_allgather_results",
]


def comm_unique_id():
    """the 128-byte id of a new RCCL communicator (ncclGetUniqueId through the C-ABI): rank 0 makes it, the host hands it round"""
    fn = lib().dftpav_comm_unique_id
    fn.argtypes = [C.c_void_p]
    buf = np.zeros(128, dtype=np.uint8)
    rc = fn(buf.ctypes.data_as(C.c_void_p))
    if rc != OK:
        raise DftpavError(rc, "comm_unique_id: RCCL is not loadable")
    return buf


def comm_available():
    """is RCCL loadable behind the C-ABI?  (dftpav_comm_available: dlopen only -- no bootstrap root is started)"""
    fn = lib().dftpav_comm_available
    fn.argtypes = []
    fn.restype = C.c_int
    return bool(fn())


def comm_layout(global_B, nranks, rank):
    """(first, count, block) of dftpav_comm_layout"""
    fn = lib().dftpav_comm_layout
    fn.argtypes = [C.c_int, C.c_int, C.c_int, c_int_p, c_int_p, c_int_p]
    a, b, c = C.c_int(0), C.c_int(0), C.c_int(0)
    rc = fn(int(global_B), int(nranks), int(rank), C.byref(a), C.byref(b), C.byref(c))
    if rc != OK:
        raise DftpavError(rc, "comm_layout")
    return a.value, b.value, c.value


class DftpavError(RuntimeError):
    def __init__(self, code, msg=""):
        super().__init__("dftpav error %d %s" % (code, msg))
        self.code = code


def build(force=False):
    """Compile libdftpav_hip.so for gfx950 (hipcc cross-compiles without a GPU).  Always goes through make: the
    Makefile tracks every source and header, so a stale library cannot pass for a fresh one.  The build is serialised
    with a file lock: several ranks of one job (the gloo / RCCL tests, torchrun) call this at the same moment, and two
    makes rewriting the same objects could leave one of them loading a half-written library."""
    src_dir = os.path.join(_HERE, "csrc")
    if os.environ.get("DFTPAV_LIB"):   # a prebuilt library was named: nothing of the tree is built or cleaned
        return LIB_PATH
    import fcntl
    with open(os.path.join(src_dir, ".build.lock"), "w") as lock:
        fcntl.flock(lock, fcntl.LOCK_EX)
        try:
            if force:
                subprocess.check_call(["make", "-C", src_dir, "-s", "clean"])
            if os.path.exists("/opt/rocm/bin/hipcc") or not os.path.exists(LIB_PATH):
                subprocess.check_call(["make", "-C", src_dir, "-s"])
        finally:
            fcntl.flock(lock, fcntl.LOCK_UN)
    return LIB_PATH


def lib():
    global _LIB
    if _LIB is None:
        if not os.path.exists(LIB_PATH):
            raise ImportError("libdftpav_hip.so is not built (run python -c 'import __graft_entry__ as g; g.build()'); "
                              "there is no CPU fallback")
        L = C.CDLL(LIB_PATH)
        vp = C.c_void_p
        L.dftpav_default_params.argtypes = [C.POINTER(Params)]
        L.dftpav_default_params.restype = None
        L.dftpav_num_vars.argtypes = [C.POINTER(Layout)]
        L.dftpav_num_points.argtypes = [C.POINTER(Params), C.POINTER(Layout)]
        L.dftpav_create.argtypes = [C.POINTER(Params), C.c_int, C.POINTER(vp)]
        L.dftpav_destroy.argtypes = [vp]
        L.dftpav_destroy.restype = None
        L.dftpav_last_error.argtypes = [vp]
        L.dftpav_last_error.restype = C.c_char_p
        L.dftpav_set_surround.argtypes = [vp, C.POINTER(Surround)]
        L.dftpav_batch_create.argtypes = [vp, C.POINTER(Layout), C.c_int, C.POINTER(vp)]
        L.dftpav_batch_destroy.argtypes = [vp]
        L.dftpav_batch_destroy.restype = None
        L.dftpav_batch_upload.argtypes = [vp, C.POINTER(BatchData)]
        L.dftpav_batch_get_x0.argtypes = [vp, c_double_p]
        L.dftpav_batch_eval.argtypes = [vp, c_double_p, c_double_p, c_double_p]
        L.dftpav_batch_solve_async.argtypes = [vp]
        L.dftpav_batch_sync.argtypes = [vp]
        L.dftpav_batch_results.argtypes = [vp, c_double_p, c_double_p, c_int_p, c_int_p, c_int_p, c_int_p, c_ll_p,
                                           c_double_p]
        L.dftpav_batch_pack_results.argtypes = [vp, vp]
        L.dftpav_batch_records.argtypes = [vp, vp]
        L.dftpav_batch_coeffs.argtypes = [vp, c_double_p, c_double_p]
        L.dftpav_batch_last_solve_ms.argtypes = [vp, C.POINTER(C.c_float)]
        L.dftpav_solve_batch.argtypes = [vp, C.POINTER(Layout), C.c_int, C.POINTER(BatchData), c_double_p,
                                         c_double_p, c_int_p, c_int_p, c_int_p, c_int_p]
        L.dftpav_stream.argtypes = [vp]
        L.dftpav_stream.restype = vp
        for nm in ("params", "layout", "batch_data", "surround"):
            getattr(L, "dftpav_abi_sizeof_" + nm).restype = C.c_int
        _LIB = L
    return _LIB


def default_params():
    p = Params()
    lib().dftpav_default_params(C.byref(p))
    return p


class Handle:
    """dftpav_handle: one HIP stream + the optimiser constants (== a PolyTrajOptimizer after setParam)."""

    def __init__(self, params=None, device=0):
        self.params = params if params is not None else default_params()
        self._h = C.c_void_p()
        rc = lib().dftpav_create(C.byref(self.params), device, C.byref(self._h))
        if rc != OK:
            self._h = None
            raise DftpavError(rc, "dftpav_create (no usable HIP device?)" if rc == E_NO_DEVICE else "dftpav_create")
        self._sur_keep = None

    def _check(self, rc, what):
        if rc != OK:
            msg = lib().dftpav_last_error(self._h)
            raise DftpavError(rc, "%s: %s" % (what, msg.decode() if msg else ""))

    def set_surround(self, surround_set):
        if surround_set is None:
            self._check(lib().dftpav_set_surround(self._h, None), "set_surround")
            self._sur_keep = None
            return
        s = surround_set.c_struct()
        self._check(lib().dftpav_set_surround(self._h, C.byref(s)), "set_surround")
        self._sur_keep = surround_set

    def reeds_shepp_shots(self, from_, to, max_cur=1.0, checkl=0.2, max_samples=512, vertex_res=0.1, check_collision=False):
        """KinoAstar::computeShotTraj / is_shot_sucess on the device for n pose pairs (x, y, yaw): dict(length, type, seg,
        samples [n][max_samples][3], n_samples, collides or None)."""
        f = np.ascontiguousarray(from_, dtype=np.float64).reshape(-1, 3)
        t = np.ascontiguousarray(to, dtype=np.float64).reshape(-1, 3)
        n = f.shape[0]
        out = dict(length=np.zeros(n), type=np.zeros(n, dtype=np.int32), seg=np.zeros((n, 5)),
                   samples=np.zeros((n, int(max_samples), 3)), n_samples=np.zeros(n, dtype=np.int32),
                   collides=np.zeros(n, dtype=np.int32) if check_collision else None)
        fn = lib().dftpav_reeds_shepp_shots
        fn.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_double, C.c_double, C.c_int, C.c_double] + [C.c_void_p] * 6
        ptr = lambda a: a.ctypes.data_as(C.c_void_p) if a is not None else None
        self._check(fn(self._h, ptr(f), ptr(t), n, float(max_cur), float(checkl), int(max_samples), float(vertex_res),
                       ptr(out["length"]), ptr(out["type"]), ptr(out["seg"]), ptr(out["samples"]), ptr(out["n_samples"]),
                       ptr(out["collides"])), "reeds_shepp_shots")
        return out

    def comm_create(self, nranks, rank, unique_id):
        """ncclCommInitRank on this handle's device, collectively (dftpav_comm_create); unique_id: the 128 bytes of comm_unique_id()
        of rank 0, distributed by the host."""
        fn = lib().dftpav_comm_create
        fn.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_void_p]
        buf = np.ascontiguousarray(unique_id, dtype=np.uint8)
        assert buf.size == 128
        self._check(fn(self._h, int(nranks), int(rank), buf.ctypes.data_as(C.c_void_p)), "comm_create")

    def comm_share(self, owner):
        """this handle uses `owner`'s communicator (dftpav_comm_share): one communicator per rank for several streams"""
        fn = lib().dftpav_comm_share
        fn.argtypes = [C.c_void_p, C.c_void_p]
        self._check(fn(self._h, owner._h), "comm_share")

    def comm_destroy(self):
        fn = lib().dftpav_comm_destroy
        fn.argtypes = [C.c_void_p]
        self._check(fn(self._h), "comm_destroy")

    def mark(self, slot=0):
        """Records one of the handle's two marker events on its stream (dftpav_mark)."""
        fn = lib().dftpav_mark
        fn.argtypes = [C.c_void_p, C.c_int]
        self._check(fn(self._h, int(slot)), "mark")

    def elapsed_since(self, other, other_slot=0, slot=1):
        """Device time in ms from marker `other_slot` of handle `other` to marker `slot` of this handle."""
        fn = lib().dftpav_marks_elapsed_ms
        fn.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_void_p]
        ms = C.c_float(0.0)
        self._check(fn(other._h, int(other_slot), self._h, int(slot), C.byref(ms)), "marks_elapsed_ms")
        return float(ms.value)

    def set_surround_wire(self, blobs):
        """Installs serialised trajectories (wire_pack) as the moving obstacles; each blob becomes one obstacle."""
        self._sur_keep = None
        fn = lib().dftpav_set_surround_wire
        fn.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int]
        S = len(blobs)
        if S == 0:
            self._check(fn(self._h, None, None, 0), "set_surround_wire")
            return
        keep = [C.create_string_buffer(bytes(bl), len(bl)) for bl in blobs]
        ptrs = (C.c_void_p * S)(*[C.cast(k, C.c_void_p) for k in keep])
        sizes = (C.c_size_t * S)(*[len(bl) for bl in blobs])
        self._check(fn(self._h, ptrs, sizes, S), "set_surround_wire")

    def frontend_resample(self, paths, path_len, start_states, end_states, start_ctrl, fparams=None, **caps):
        """getKinoNode (from SampleTraj on) + the resampling of RunMINCOParking on the device; returns the dict of padded
        arrays of pods.FrontendOut."""
        from .pods import FrontendParams, FrontendOut
        P = np.ascontiguousarray(paths, dtype=np.float64)
        n_hyp, max_path = P.shape[0], P.shape[1]
        pl = np.ascontiguousarray(path_len, dtype=np.int32)
        ss = np.ascontiguousarray(start_states, dtype=np.float64)
        es = np.ascontiguousarray(end_states, dtype=np.float64)
        sc_ = np.ascontiguousarray(start_ctrl, dtype=np.float64)
        fp = fparams if fparams is not None else FrontendParams.default()
        out = FrontendOut(n_hyp, **caps)
        fn = lib().dftpav_frontend_resample
        fn.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int,
                       C.c_void_p]
        self._check(fn(self._h, C.byref(fp), P.ctypes.data_as(C.c_void_p), pl.ctypes.data_as(C.c_void_p), max_path,
                       ss.ctypes.data_as(C.c_void_p), es.ctypes.data_as(C.c_void_p), sc_.ctypes.data_as(C.c_void_p), n_hyp,
                       C.byref(out.c)), "frontend_resample")
        return out.arrays()

    def sample_restarts(self, inner, durs, n_restarts, sigma=0.3, lo=0.8, hi=1.25, seed=0):
        """Seeded restarts of hypotheses on the device: inner [n_hyp][n_inner], durs [n_hyp][M] ->
        (inner [n_hyp * n_restarts][n_inner], durs [n_hyp * n_restarts][M])."""
        a = np.ascontiguousarray(inner, dtype=np.float64)
        d = np.ascontiguousarray(durs, dtype=np.float64)
        n_hyp, n_inner, M = a.shape[0], a.shape[1], d.shape[1]
        oi = np.zeros((n_hyp * n_restarts, n_inner))
        od = np.zeros((n_hyp * n_restarts, M))
        fn = lib().dftpav_sample_restarts
        fn.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_double, C.c_double, C.c_double,
                       C.c_ulonglong, C.c_void_p, C.c_void_p]
        self._check(fn(self._h, a.ctypes.data_as(C.c_void_p), d.ctypes.data_as(C.c_void_p), n_hyp, n_restarts, n_inner, M,
                       float(sigma), float(lo), float(hi), int(seed), oi.ctypes.data_as(C.c_void_p),
                       od.ctypes.data_as(C.c_void_p)), "sample_restarts")
        return oi, od

    def fit_surround(self, states):
        """ConverSurroundTrajFromPoints + setSurroundTrajs on the device: states [S][n][7] (x, y, angle, velocity,
        acceleration, curvature, time_stamp)."""
        st = np.ascontiguousarray(states, dtype=np.float64)
        S, n = (st.shape[0], st.shape[1]) if st.size else (0, 0)
        fn = lib().dftpav_fit_surround
        fn.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int]
        self._check(fn(self._h, st.ctypes.data_as(C.c_void_p), S, n), "fit_surround")
        self._sur_keep = None

    def get_surround(self):
        """The installed moving obstacles as dict(offsets, durations, coeffs [np][12], total, start)."""
        fn = lib().dftpav_get_surround
        fn.argtypes = [C.c_void_p] + [C.c_void_p] * 7
        S, npc = C.c_int(0), C.c_int(0)
        self._check(fn(self._h, C.byref(S), C.byref(npc), None, None, None, None, None), "get_surround")
        out = dict(offsets=np.zeros(S.value + 1, dtype=np.int32), durations=np.zeros(npc.value), coeffs=np.zeros((npc.value, 12)),
                   total=np.zeros(S.value), start=np.zeros(S.value))
        if S.value:
            self._check(fn(self._h, None, None, out["offsets"].ctypes.data_as(C.c_void_p), out["durations"].ctypes.data_as(C.c_void_p),
                           out["coeffs"].ctypes.data_as(C.c_void_p), out["total"].ctypes.data_as(C.c_void_p),
                           out["start"].ctypes.data_as(C.c_void_p)), "get_surround")
        return out

    def set_grid_map(self, grid, resolution, origin):
        """Obstacle map of the corridor generator: uint8 [size_y][size_x], 80 = occupied (dftpav_grid_map)."""
        g = np.ascontiguousarray(grid, dtype=np.uint8)
        m = GridMap(g.ctypes.data_as(C.c_void_p), g.shape[1], g.shape[0], float(resolution), float(origin[0]), float(origin[1]))
        fn = lib().dftpav_set_grid_map
        fn.argtypes = [C.c_void_p, C.c_void_p]
        self._check(fn(self._h, C.byref(m)), "set_grid_map")

    def corridor_rectangles(self, states):
        """getRectangleConst on the device: states [n][3] (x, y, yaw) -> [n][4][4] columns (n_x, n_y, p_x, p_y)."""
        st = np.ascontiguousarray(states, dtype=np.float64).reshape(-1, 3)
        out = np.zeros((st.shape[0], 4, 4), dtype=np.float64)
        fn = lib().dftpav_corridor_rectangles
        fn.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_void_p]
        self._check(fn(self._h, st.ctypes.data_as(C.c_void_p), st.shape[0], out.ctypes.data_as(C.c_void_p)), "corridor_rectangles")
        return out

    def corridor_last_ms(self):
        ms = C.c_float(0.0)
        fn = lib().dftpav_corridor_last_ms
        fn.argtypes = [C.c_void_p, C.POINTER(C.c_float)]
        self._check(fn(self._h, C.byref(ms)), "corridor_last_ms")
        return float(ms.value)

    def close(self):
        if self._h:
            lib().dftpav_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class GridMap(C.Structure):
    """dftpav_grid_map (include/dftpav_hip.h)."""
    _fields_ = [("cells", C.c_void_p), ("size_x", C.c_int), ("size_y", C.c_int), ("resolution", C.c_double),
                ("origin_x", C.c_double), ("origin_y", C.c_double)]


class Batch:
    """dftpav_batch: B trajectories of one layout, resident in HBM."""

    def __init__(self, handle, layout_spec, B, residency=-1):
        """residency: -1 by B (dftpav_batch_create); 0 / 1 / 2 = one / two / four workgroups per CU (dftpav_batch_create_shaped)"""
        self.handle = handle
        self.layout = layout_spec
        self.B = B
        self.n = layout_spec.n_vars
        self._b = C.c_void_p()
        lay = layout_spec.c_struct()
        if residency < 0:
            rc = lib().dftpav_batch_create(handle._h, C.byref(lay), B, C.byref(self._b))
        else:
            fn = lib().dftpav_batch_create_shaped
            fn.argtypes = [C.c_void_p, C.POINTER(Layout), C.c_int, C.c_int, C.POINTER(C.c_void_p)]
            rc = fn(handle._h, C.byref(lay), B, int(residency), C.byref(self._b))
        if rc != OK:
            self._b = None
            handle._check(rc, "batch_create")

    def set_order(self, order):
        """ORDER_DEVICE (default, the throughput kernels) or ORDER_REFERENCE: every sum in the order PolyTrajOptimizer
        executes it, whole solves bit-equal to OptimizeTrajectory's (dftpav_batch_set_order)."""
        fn = lib().dftpav_batch_set_order
        fn.argtypes = [C.c_void_p, C.c_int]
        self.handle._check(fn(self._b, int(order)), "batch_set_order")

    def upload(self, scen_or_data, with_corridor=True):
        """with_corridor=False: everything but the half-planes (they come from corridor_from_states)."""
        d = scen_or_data.batch_data() if hasattr(scen_or_data, "batch_data") else scen_or_data
        self._keep = scen_or_data
        if not with_corridor:
            d.corridor = None
        rc = lib().dftpav_batch_upload(self._b, C.byref(d))
        self.handle._check(rc, "batch_upload")

    def validate(self, sample_dt=0.05, vertex_res=0.1):
        """Collision re-check of the solved trajectories against the handle's grid map (CheckReplan,
        traj_server_ros.cpp:385-397): returns (collision [B], first_sample [B])."""
        col = np.zeros(self.B, dtype=np.int32)
        first = np.zeros(self.B, dtype=np.int32)
        fn = lib().dftpav_batch_validate
        fn.argtypes = [C.c_void_p, C.c_double, C.c_double, C.c_void_p, C.c_void_p]
        self.handle._check(fn(self._b, float(sample_dt), float(vertex_res), col.ctypes.data_as(C.c_void_p),
                              first.ctypes.data_as(C.c_void_p)), "batch_validate")
        return col, first

    def sample_states(self, t0=0.0, sample_dt=0.01, n_samples=None, filter_singularity=True):
        """Trajectory::GetState over the grid t0 + k * sample_dt for every solved trajectory, played back as the
        server does (traj_server_ros.cpp:244-259): returns (states [B][n_samples][8] = time_stamp, x, y, angle,
        curvature, velocity, acceleration, steer; n_valid [B])."""
        if n_samples is None:
            _, dts = self.coeffs()
            total = float(np.max(np.sum(dts * self.layout.piece_nums[None, :], axis=1)))
            n_samples = int(np.ceil((total - t0) / sample_dt)) + 1
        st = np.zeros((self.B, int(n_samples), 8))
        nv = np.zeros(self.B, dtype=np.int32)
        fn = lib().dftpav_batch_sample_states
        fn.argtypes = [C.c_void_p, C.c_double, C.c_double, C.c_int, C.c_int, C.c_void_p, C.c_void_p]
        self.handle._check(fn(self._b, float(t0), float(sample_dt), int(n_samples), int(bool(filter_singularity)),
                              st.ctypes.data_as(C.c_void_p), nv.ctypes.data_as(C.c_void_p)), "batch_sample_states")
        return st, nv

    def corridor_from_states(self, states, n_restarts=1):
        """getRectangleConst for every constraint point, on the device, straight into the solver's layout: states
        [B / n_restarts][Npts][3] (x, y, yaw), the restarts of a hypothesis sharing its corridor; needs Handle.set_grid_map."""
        st = np.ascontiguousarray(states, dtype=np.float64)
        assert st.shape[0] * n_restarts == self.B and st.shape[-1] == 3
        fn = lib().dftpav_batch_corridor_from_hypotheses
        fn.argtypes = [C.c_void_p, C.c_void_p, C.c_int]
        self.handle._check(fn(self._b, st.ctypes.data_as(C.c_void_p), int(n_restarts)), "batch_corridor_from_hypotheses")

    def x0(self):
        x = np.zeros((self.B, self.n))
        self.handle._check(lib().dftpav_batch_get_x0(self._b, dptr(x)), "get_x0")
        return x

    def eval(self, x):
        x = np.ascontiguousarray(x, dtype=np.float64).reshape(self.B, self.n)
        f = np.zeros(self.B)
        g = np.zeros((self.B, self.n))
        self.handle._check(lib().dftpav_batch_eval(self._b, dptr(x), dptr(f), dptr(g)), "batch_eval")
        return f, g

    def solve_async(self):
        self.handle._check(lib().dftpav_batch_solve_async(self._b), "solve_async")

    def sync(self):
        self.handle._check(lib().dftpav_batch_sync(self._b), "sync")

    def last_solve_ms(self):
        ms = C.c_float(0)
        self.handle._check(lib().dftpav_batch_last_solve_ms(self._b, C.byref(ms)), "last_solve_ms")
        return ms.value

    def results(self):
        B, n = self.B, self.n
        r = dict(x=np.zeros((B, n)), final_cost=np.zeros(B), status=np.zeros(B, dtype=np.int32),
                 success=np.zeros(B, dtype=np.int32), iters=np.zeros(B, dtype=np.int32),
                 evals=np.zeros(B, dtype=np.int32), hist_sum=np.zeros(B, dtype=np.int64), latency_us=np.zeros(B))
        rc = lib().dftpav_batch_results(self._b, dptr(r["x"]), dptr(r["final_cost"]), iptr(r["status"]),
                                        iptr(r["success"]), iptr(r["iters"]), iptr(r["evals"]), llptr(r["hist_sum"]),
                                        dptr(r["latency_us"]))
        self.handle._check(rc, "results")
        return r

    def solve(self):
        self.solve_async()
        return self.results()

    def solve_chained(self, prev=None):
        """Throughput mode for a stream of equally shaped batches (dftpav_batch_solve_chained): this batch's last
        trajectories stay suspended for the next chained solve, `prev`'s are taken over and finished here."""
        fn = lib().dftpav_batch_solve_chained
        fn.argtypes = [C.c_void_p, C.c_void_p]
        self.handle._check(fn(self._b, prev._b if prev is not None else None), "solve_chained")

    def set_hand_over(self, hand_over):
        """End game of a scheduled solve (dftpav_batch_set_hand_over): 0 keeps every trajectory in the queue launch."""
        fn = lib().dftpav_batch_set_hand_over
        fn.argtypes = [C.c_void_p, C.c_int]
        self.handle._check(fn(self._b, int(hand_over)), "set_hand_over")

    def finish(self):
        fn = lib().dftpav_batch_finish
        fn.argtypes = [C.c_void_p]
        self.handle._check(fn(self._b), "finish")

    def plan_cycle(self, scen, states, n_restarts=1, check_dt=0.05, vertex_res=0.1, t0=0.0, state_dt=0.01, n_samples=100,
                   filter_singularity=True):
        """dftpav_plan_cycle: upload, corridor from the map, solve, collision re-check and read-out enqueued in one call."""
        d = scen.batch_data() if hasattr(scen, "batch_data") else scen
        st = np.ascontiguousarray(states, dtype=np.float64)
        fn = lib().dftpav_plan_cycle
        fn.argtypes = [C.c_void_p, C.POINTER(BatchData), C.c_void_p, C.c_int, C.c_double, C.c_double, C.c_double, C.c_double, C.c_int,
                       C.c_int]
        self._pc_keep = (d, st)
        self._pc_n = int(n_samples)
        self.handle._check(fn(self._b, C.byref(d), st.ctypes.data_as(C.c_void_p), int(n_restarts), float(check_dt), float(vertex_res),
                              float(t0), float(state_dt), int(n_samples), int(bool(filter_singularity))), "plan_cycle")

    def plan_cycle_fetch(self):
        """-> dict(x, final_cost, status, success, iters, collision, first_sample, states [B][n_samples][8], n_valid)"""
        B, n = self.B, self.layout.n_vars
        r = dict(x=np.zeros((B, n)), final_cost=np.zeros(B), status=np.zeros(B, dtype=np.int32), success=np.zeros(B, dtype=np.int32),
                 iters=np.zeros(B, dtype=np.int32), collision=np.zeros(B, dtype=np.int32), first_sample=np.zeros(B, dtype=np.int32),
                 states=np.zeros((B, self._pc_n, 8)), n_valid=np.zeros(B, dtype=np.int32))
        fn = lib().dftpav_plan_cycle_fetch
        fn.argtypes = [C.c_void_p, c_double_p, c_double_p, c_int_p, c_int_p, c_int_p, c_int_p, c_int_p, c_double_p, c_int_p]
        self.handle._check(fn(self._b, dptr(r["x"]), dptr(r["final_cost"]), iptr(r["status"]), iptr(r["success"]), iptr(r["iters"]),
                              iptr(r["collision"]), iptr(r["first_sample"]), dptr(r["states"]), iptr(r["n_valid"])), "plan_cycle_fetch")
        return r

    def trace(self, traj, max_evals=4096, count=1):
        """Record every evaluation of trajectories traj .. traj + count - 1 during the following solves
        (dftpav_batch_trace_range); max_evals = 0 switches it off."""
        fn = lib().dftpav_batch_trace_range
        fn.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int]
        self.handle._check(fn(self._b, int(traj), int(count), int(max_evals)), "batch_trace")
        self._trace_cap, self._trace_first = int(max_evals), int(traj)

    def get_trace(self, traj=None):
        """-> dict(x [E][n], g [E][n], d [E][n], f [E], stp [E], k [E], count [E]) of a traced trajectory (default: the first)"""
        fn = lib().dftpav_batch_get_trace_of
        fn.argtypes = [C.c_void_p, C.c_int, c_double_p, c_int_p]
        n = self.layout.n_vars
        rows = np.zeros((self._trace_cap, 3 * n + 4))
        cnt = C.c_int(0)
        self.handle._check(fn(self._b, self._trace_first if traj is None else int(traj), dptr(rows), C.byref(cnt)), "batch_get_trace")
        r = rows[:cnt.value]
        return dict(x=r[:, :n].copy(), g=r[:, n:2 * n].copy(), d=r[:, 2 * n:3 * n].copy(), f=r[:, 3 * n].copy(),
                    stp=r[:, 3 * n + 1].copy(), k=r[:, 3 * n + 2].astype(np.int64), count=r[:, 3 * n + 3].astype(np.int64))

    def profile(self, enable=True):
        """Debug: switch the in-kernel phase profiler (thread 0 shader-clock deltas) on or off."""
        fn = lib().dftpav_debug_profile
        fn.argtypes = [C.c_void_p, C.c_int, c_ll_p]
        self.handle._check(fn(self._b, int(enable), None), "profile")

    def read_profile(self):
        fn = lib().dftpav_debug_profile
        fn.argtypes = [C.c_void_p, C.c_int, c_ll_p]
        out = np.zeros((self.B, 12), dtype=np.int64)
        self.handle._check(fn(self._b, 1, llptr(out)), "read_profile")
        return out

    def allgather_results(self, global_B, device_ptr):
        """packs this rank's 16-byte records and all-gathers them over RCCL into device memory of nranks * block * 16 bytes
        (dftpav_batch_allgather_results; asynchronous on the handle's stream)"""
        fn = lib().dftpav_batch_allgather_results
        fn.argtypes = [C.c_void_p, C.c_int, C.c_void_p]
        self.handle._check(fn(self._b, int(global_B), C.c_void_p(device_ptr)), "batch_allgather_results")

    def records(self):
        """the 16-byte result records of the last solve, uint8 [B][16] on the host (waits for the solve; one DMA copy)"""
        out = np.zeros((self.B, 16), dtype=np.uint8)
        self.handle._check(lib().dftpav_batch_records(self._b, out.ctypes.data_as(C.c_void_p)), "records")
        return out

    def pack_results(self, device_ptr):
        """16-byte {f64 cost, i32 status, i32 iters} records into device memory (async on the handle's stream)."""
        self.handle._check(lib().dftpav_batch_pack_results(self._b, C.c_void_p(device_ptr)), "pack_results")

    def coeffs(self):
        c = np.zeros((self.B, self.layout.n_pieces, 6, 2))
        dt = np.zeros((self.B, self.layout.M))
        self.handle._check(lib().dftpav_batch_coeffs(self._b, dptr(c), dptr(dt)), "coeffs")
        return c, dt

    def close(self):
        if self._b:
            lib().dftpav_batch_destroy(self._b)
            self._b = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


# ---- serialised trajectories ("DPTJ" v1, include/dftpav_hip.h): pure host code, no device needed
def wire_size(piece_nums):
    pn = np.ascontiguousarray(piece_nums, dtype=np.int32)
    fn = lib().dftpav_wire_size
    fn.restype = C.c_size_t
    fn.argtypes = [C.c_int, C.c_void_p]
    return int(fn(len(pn), pn.ctypes.data_as(C.c_void_p)))


def wire_pack(layout, coeffs, piece_dt, drone_id=0, traj_id=0, start_time=0.0):
    """One trajectory (coeffs [Ntot][6][2], piece_dt [M] as Batch.coeffs returns them per trajectory) -> bytes."""
    co = np.ascontiguousarray(coeffs, dtype=np.float64)
    dt = np.ascontiguousarray(piece_dt, dtype=np.float64)
    assert co.shape == (layout.n_pieces, 6, 2) and dt.shape == (layout.M,)
    cap = wire_size(layout.piece_nums)
    buf = C.create_string_buffer(cap)
    wr = C.c_size_t(0)
    fn = lib().dftpav_wire_pack
    fn.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_double, C.c_void_p, C.c_size_t, C.c_void_p]
    ls = layout.c_struct()
    rc = fn(C.byref(ls), co.ctypes.data_as(C.c_void_p), dt.ctypes.data_as(C.c_void_p), int(drone_id), int(traj_id),
            float(start_time), buf, cap, C.byref(wr))
    if rc != 0:
        raise DftpavError(rc, "wire_pack")
    return buf.raw[:wr.value]


def wire_unpack(blob):
    """bytes -> dict(drone_id, traj_id, start_time, singuls, piece_nums, seg_start, seg_duration, durations [np],
    coeffs [np][12] in CoefficientMat order x5,y5,...,x0,y0)."""
    blob = bytes(blob)
    did, tid, M, npc = C.c_int(0), C.c_int(0), C.c_int(0), C.c_int(0)
    st = C.c_double(0.0)
    fi = lib().dftpav_wire_info
    fi.argtypes = [C.c_char_p, C.c_size_t] + [C.c_void_p] * 5
    rc = fi(blob, len(blob), C.byref(did), C.byref(tid), C.byref(st), C.byref(M), C.byref(npc))
    if rc != 0:
        raise DftpavError(rc, "wire_info")
    out = dict(drone_id=did.value, traj_id=tid.value, start_time=st.value, singuls=np.zeros(M.value, dtype=np.int32),
               piece_nums=np.zeros(M.value, dtype=np.int32), seg_start=np.zeros(M.value), seg_duration=np.zeros(M.value),
               durations=np.zeros(npc.value), coeffs=np.zeros((npc.value, 12)))
    fu = lib().dftpav_wire_unpack
    fu.argtypes = [C.c_char_p, C.c_size_t] + [C.c_void_p] * 6
    rc = fu(blob, len(blob), *[out[k].ctypes.data_as(C.c_void_p)
                               for k in ("singuls", "piece_nums", "seg_start", "seg_duration", "durations", "coeffs")])
    if rc != 0:
        raise DftpavError(rc, "wire_unpack")
    return out
