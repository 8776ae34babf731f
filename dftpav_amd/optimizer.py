"""Python mirror of the reference's class interface over the C-ABI.

plan_manage::PolyTrajOptimizer (traj_optimizer.h:24-250) keeps its entry points —
setParam, setSurroundTrajs, OptimizeTrajectory, getMinJerkOptPtr — and its error
behaviour (bool return, no exceptions on bad sizes: traj_optimizer.cpp:26-48).
The C++ twin for the ROS host is dftpav_amd/csrc/host/poly_traj_optimizer.hpp.
"""
import numpy as np

from . import capi
from .pods import BatchData, LayoutSpec, dptr


class MinJerkOptView:
    """What callers read from getMinJerkOptPtr()[i] (poly_traj_utils.hpp:987-997,1069-1074)."""

    def __init__(self, coeffs, dt):
        self._c = coeffs  # [N][6][2], row k multiplies s^k
        self._dt = dt

    def getCoeffs(self):
        """(6N)x2, row 6i+k = k-th power coefficient of piece i (poly_traj_utils.hpp:865,982-983)."""
        return self._c.reshape(-1, 2)

    def getDt(self):
        return self._dt

    def getTraj(self, singul):
        """[(duration, 2x6 coeffMat with column 0 = t^5, singul)] per piece (poly_traj_utils.hpp:987-997)."""
        return [(self._dt, self._c[i].T[:, ::-1].copy(), singul) for i in range(self._c.shape[0])]


class PolyTrajOptimizer:
    def __init__(self, device=0, reference_order=False):
        """reference_order: solve with every sum in the order the reference executes it (dftpav_batch_set_order): the bits of
        the reference's program with sequential reductions, no FMA and -- gear shifts, moving obstacles -- correctly rounded
        libm calls (include/dftpav_hip.h states the contract).  Where the layout is outside its limits (n > 64, H > 5,
        5 H + S + 4 > 64) the throughput order runs; `last["order"]` says which one did."""
        self._device = device
        self._reference_order = bool(reference_order)
        self._params = capi.default_params()
        self._handle = None
        self._surround = None
        self._mjo = []
        self.last = None  # results of the last solve (status, iters, evals, final_cost)

    # traj_optimizer.h:100 (the ros::NodeHandle argument only served debug publishers)
    def setParam(self, params):
        self._params = params
        if self._handle is not None:
            self._handle.close()
            self._handle = None

    # traj_optimizer.h:108
    def setSurroundTrajs(self, surround_set):
        self._surround = surround_set
        if self._handle is not None:
            self._handle.set_surround(surround_set)

    def get_traj_resolution_(self):  # traj_optimizer.h:113
        return self._params.traj_resolution

    def get_destraj_resolution_(self):  # traj_optimizer.h:114
        return self._params.des_traj_resolution

    def getMinJerkOptPtr(self):  # traj_optimizer.h:112
        return self._mjo

    def _ensure(self):
        if self._handle is None:
            self._handle = capi.Handle(self._params, self._device)  # raises without a GPU: no fallback
            self._handle.set_surround(self._surround)
        return self._handle

    # traj_optimizer.h:118-120
    def OptimizeTrajectory(self, iniStates, finStates, initInnerPts, initTs, hPoly_container, singuls, now=0.0,
                           help_eps=0.0):
        """One trajectory (B = 1), containers exactly as the reference passes them:
        iniStates/finStates: list of 2x3; initInnerPts: list of 2x(N_i-1); initTs: [M];
        hPoly_container: list (segment) of list (point) of 4xH; singuls: [M]."""
        p = self._params
        M = len(initInnerPts)
        if len(initTs) != M:  # traj_optimizer.cpp:26-29
            return False
        if np.min(initTs) < p.mini_T:  # traj_optimizer.cpp:30-33
            return False
        piece_nums = []
        for i in range(M):
            cols = np.asarray(initInnerPts[i]).shape[1] if np.asarray(initInnerPts[i]).ndim == 2 else 0
            if cols == 0:  # traj_optimizer.cpp:38-41
                return False
            N = cols + 1
            piece_nums.append(N)
            need = (N - 2) * (p.traj_resolution + 1) + 2 * (p.des_traj_resolution + 1)
            if len(hPoly_container[i]) != need:  # traj_optimizer.cpp:44-48
                return False
        H = max(np.asarray(h).shape[1] for seg in hPoly_container for h in seg)
        lay = LayoutSpec(piece_nums, list(singuls), H=H)
        ini = np.stack([np.asarray(s, dtype=np.float64).T.reshape(-1) for s in iniStates])[None]  # col-major 2x3
        fin = np.stack([np.asarray(s, dtype=np.float64).T.reshape(-1) for s in finStates])[None]
        inner = np.concatenate([np.asarray(w, dtype=np.float64).T.reshape(-1) for w in initInnerPts])[None]
        Ts = np.asarray(initTs, dtype=np.float64)[None]
        planes = []
        for seg in hPoly_container:
            for h in seg:
                h = np.asarray(h, dtype=np.float64)
                cols = [h[:, k] for k in range(h.shape[1])]
                while len(cols) < H:  # pad with a far-away plane that can never be violated
                    cols.append(np.array([1.0, 0.0, 1.0e9, 0.0]))
                planes.append(np.stack(cols))
        cor = np.ascontiguousarray(np.stack(planes)[None])
        r = self._solve(lay, 1, np.ascontiguousarray(ini), np.ascontiguousarray(fin), np.ascontiguousarray(inner),
                        np.ascontiguousarray(Ts), cor, now, help_eps)
        return bool(r["success"][0])

    def OptimizeTrajectoryBatch(self, scen):
        """The new axis: B random-restart / multi-hypothesis trajectories with a shared layout."""
        self.setSurroundTrajs(scen.surround)
        return self._solve(scen.layout, scen.B, scen.ini_states, scen.fin_states, scen.inner_pts, scen.init_Ts,
                           scen.corridor, scen.t_now, scen.help_eps)

    def _solve(self, lay, B, ini, fin, inner, Ts, cor, now, help_eps):
        h = self._ensure()
        bt = capi.Batch(h, lay, B)
        d = BatchData()
        d.ini_states, d.fin_states, d.inner_pts = dptr(ini), dptr(fin), dptr(inner)
        d.init_Ts, d.corridor = dptr(Ts), dptr(cor)
        d.t_now, d.help_eps = float(now), float(help_eps)
        bt._keep = (ini, fin, inner, Ts, cor)
        rc = capi.lib().dftpav_batch_upload(bt._b, d)
        if rc != capi.OK:
            bt.close()
            if rc in (capi.E_MINI_T, capi.E_INVALID, capi.E_ONE_PIECE):
                return dict(success=np.zeros(B, dtype=np.int32))
            h._check(rc, "upload")
        order = capi.ORDER_DEVICE
        if self._reference_order:
            try:
                bt.set_order(capi.ORDER_REFERENCE)
                order = capi.ORDER_REFERENCE
            except capi.DftpavError as ex:
                if ex.code != capi.E_UNSUPPORTED:
                    raise
        r = bt.solve()
        r["order"] = order
        c, dt = bt.coeffs()
        # results stay inside the optimiser until the next call (traj_optimizer.h:91,112); B=1 view for the ROS path
        off = 0
        self._mjo = []
        for i, N in enumerate(lay.piece_nums):
            self._mjo.append(MinJerkOptView(c[0, off:off + N].copy(), float(dt[0, i])))
            off += int(N)
        r["coeffs"], r["piece_dt"] = c, dt
        self.last = r
        bt.close()
        return r
