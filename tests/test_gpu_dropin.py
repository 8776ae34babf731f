"""The drop-in compiled INTO the reference's own class.

oracle/_ref/libdftpav_dropin.so (recipe: oracle/Makefile.dropin) is the reference's `plan_manage::PolyTrajOptimizer` -- its header
traj_optimizer.h unmodified, as are poly_traj_utils.hpp / traj_container.hpp behind it -- with the drop-in's implementation of the
live path (dftpav_amd/csrc/host/dropin/traj_optimizer_hip.cpp over libdftpav_hip.so) in place of traj_optimizer.cpp, driven through
oracle/ref_driver.cpp.  So the calls a planner makes on that class -- setParam, setSurroundTrajs, OptimizeTrajectory(iniStates,
finStates, innerPts, Ts, hPolys, singuls, now, eps) (traj_manager.cpp:604-610), getMinJerkOptPtr() (:618-625),
costFunctionCallback -- run on the GPU behind the reference's own interface, and the test compares what the object hands back with
the CPU restatement of the reference (oracle/dftpav_oracle.c).  (Compiling the class here takes interface stand-ins for Eigen / ROS
-- oracle/ref_shim: a compile / ABI check of the binding, not a build of the reference; rounds 4-5 also compared with a stand-in
build of the reference's traj_optimizer.cpp, retired in round 6, oracle/pyref.py.)

Bar: single-segment static problems (no libm call in the reference's loop): BIT-EQUAL to the restatement's order 0 -- the bool, the
solution vector, the final cost, the solver status, iterations, evaluations, and the coefficients / piece durations
getMinJerkOptPtr() exposes.  Gear shifts and moving obstacles: bit-equal to order 2 (the reference's program with correctly rounded
cos / sin / exp / log / pow).
"""
import numpy as np
import pytest

from dftpav_amd import scenarios as sc

pytestmark = pytest.mark.gpu


def _pyref():
    from oracle import pyref
    if not pyref.dropin_available():
        pytest.skip("oracle/_ref/libdftpav_dropin.so not built (it is built where /root/reference exists and travels with the tree)")
    return pyref


class _Restatement:
    """the CPU restatement on element b of a scenario, with the surface of pyref.RefProblem the tests use"""

    def __init__(self, oracle, p, s, b, order=0):
        self.o, self.p, self.s, self.b, self.order, self.r = oracle, p, s, b, order, None

    def optimize(self):
        r = self.o.solve_batch(self.p, self.s.subset([self.b]), nthreads=1, order=self.order)
        self.r = dict(final_cost=r["final_cost"][0], x=r["x"][0], status=int(r["status"][0]), iters=int(r["iters"][0]), evals=int(r["evals"][0]),
                      ok=bool(r["success"][0]))
        return self.r

    def eval(self, x):
        return self.o.OracleProblem(self.p, self.s, self.b, order=self.order).eval(x)

    def coeffs(self):
        lp = self.o.OracleProblem(self.p, self.s, self.b, order=self.order)
        lp.eval(self.r["x"])
        return lp.coeffs()


def _scenario(oracle, hiplib, name, B):
    p = hiplib.default_params()
    if name == "default_arena":
        from test_default_map import _default_map_scenario
        p.traj_resolution, p.des_traj_resolution = 16, 32
        return p, _default_map_scenario(oracle, p, 16, 32, B)
    s = sc.baseline_config({"cfg1": 1, "cfg3": 3}[name], B=B)
    s.apply_resolution(p)
    return p, s


@pytest.mark.parametrize("name,B", [("cfg1", 6), ("cfg3", 6), ("default_arena", 4)])
def test_reference_object_gpu_backed_returns_the_reference_bits(hiplib, oracle, name, B):
    pyref = _pyref()
    p, s = _scenario(oracle, hiplib, name, B)
    for b in range(B):
        ref = _Restatement(oracle, p, s, b)              # the restatement of traj_optimizer.cpp, CPU
        gpu = pyref.RefProblem(p, s, b, dropin=True)     # traj_optimizer_hip.cpp, GPU -- behind the reference's class
        rr, rg = ref.optimize(), gpu.optimize()
        assert bool(rg["ok"]) == rr["ok"], (name, b)
        assert rg["final_cost"] == rr["final_cost"] and np.array_equal(rg["x"], rr["x"]), (name, b, rg["final_cost"], rr["final_cost"])
        assert rg["status"] == rr["status"] and rg["iters"] == rr["iters"] and rg["evals"] == rr["evals"], (name, b)
        # getMinJerkOptPtr()[i].getCoeffs() / getDt(): what traj_manager.cpp:618-625 builds the published trajectory from.  The
        # reference's containers hold the LAST EVALUATED point; after a solve that ended with a successful line search that is the
        # solution (status >= 0), which is what the drop-in regenerates
        if rr["status"] >= 0:
            cr, dtr = ref.coeffs()
            cg, dtg = gpu.coeffs()
            assert np.array_equal(np.asarray(cg).reshape(np.asarray(cr).shape), cr) and np.array_equal(np.ravel(dtg), np.ravel(dtr)), (name, b)
        # costFunctionCallback on both objects at a common point
        x = rr["x"] + 0.05 * np.sin(np.arange(len(rr["x"])) + b)
        fr, gr = ref.eval(x)
        fg, gg = gpu.eval(x)
        assert fg == fr and np.array_equal(gg, gr), (name, b)


def test_reference_object_gpu_backed_live_case(hiplib, oracle):
    """gear shifts with moving obstacles through the class interface (setSurroundTrajs + multi-segment containers): the drop-in
    object against oracle order 2 bit for bit, whole solves and costFunctionCallback at a common point"""
    pyref = _pyref()
    from test_gpu_reference_order import _live_case
    p = hiplib.default_params()
    s = _live_case(([5, 4, 6], [1, -1, 1]), 4, 82)
    s.apply_resolution(p)
    want = oracle.solve_batch(p, s, nthreads=4, order=2)
    for b in range(s.B):
        gpu = pyref.RefProblem(p, s, b, dropin=True)
        rg = gpu.optimize()
        assert rg["final_cost"] == want["final_cost"][b] and np.array_equal(rg["x"], want["x"][b]), b
        assert rg["status"] == want["status"][b] and rg["iters"] == want["iters"][b] and rg["evals"] == want["evals"][b]
        assert bool(rg["ok"]) == bool(want["success"][b])
        x = want["x"][b] + 0.01 * np.cos(np.arange(len(want["x"][b])) + b)
        fo, go = oracle.OracleProblem(p, s, b, order=2).eval(x)
        fg, gg = gpu.eval(x)
        assert fg == fo and np.array_equal(gg, go), b


def test_restarts_behind_the_reference_entry_point(hiplib, oracle, monkeypatch):
    """DFTPAV_DROPIN_RESTARTS=64: the planner's one call becomes a batch of 64 in the same launch -- slot 0 the call's own problem,
    63 seeded restarts of it (north_star's batch dimension behind OptimizeTrajectory).  Slot 0 stays BIT-EQUAL to the restatement's
    solve (the restarts change nothing for it); the returned candidate is the cheapest successful one, collision-checked on
    the device when the map is given; getMinJerkOptPtr() hands out the chosen candidate's coefficients."""
    pyref = _pyref()
    p, s = _scenario(oracle, hiplib, "cfg3", 3)
    st = s.meta["states"]
    c = (0.5 * (st[..., 0].min() + st[..., 0].max()), 0.5 * (st[..., 1].min() + st[..., 1].max()))
    grid, origin = sc.occupancy_grid(s.meta["obstacles"], arena=140.0, centre=c)
    lay = s.layout
    for b in range(3):
        ref = _Restatement(oracle, p, s, b)
        rr = ref.optimize()
        monkeypatch.delenv("DFTPAV_DROPIN_RESTARTS", raising=False)
        gpu = pyref.RefProblem(p, s, b, dropin=True)
        r1 = gpu.optimize()
        ch1 = gpu.last_choice()
        assert ch1["K"] == 1 and ch1["chosen"] == 0 and ch1["chosen_cost"] == r1["final_cost"]
        monkeypatch.setenv("DFTPAV_DROPIN_RESTARTS", "64")
        assert gpu.set_map(grid, sc.MAP_RESL, origin) == 0
        rk = gpu.optimize()
        ch = gpu.last_choice()
        # slot 0: the reference's own solve, bit for bit
        assert rk["final_cost"] == rr["final_cost"] and np.array_equal(rk["x"], rr["x"])
        assert (rk["status"], rk["iters"], rk["evals"]) == (rr["status"], rr["iters"], rr["evals"])
        # the choice
        assert ch["K"] == 64 and 0 <= ch["chosen"] < 64 and ch["n_success"] >= 1 and ch["solve_ms"] > 0.0
        assert rk["ok"] is True
        if rr["ok"]:
            assert ch["chosen_cost"] <= rr["final_cost"]
        # what getMinJerkOptPtr() exposes is the chosen candidate: it passes the oracle's collision re-check on the same map
        cg, dtg = gpu.coeffs()
        col, _ = oracle.validate_trajectories(grid, sc.MAP_RESL, origin, cg[None], dtg[None], lay.piece_nums, lay.singuls, order=1)
        assert col[0] == 0
        if ch["chosen"] == 0 and rr["status"] >= 0:
            cr, dtr = ref.coeffs()
            assert np.array_equal(np.asarray(cg).reshape(np.asarray(cr).shape), cr) and np.array_equal(np.ravel(dtg), np.ravel(dtr))
        # the same seed gives the same choice; the cached batch is reused
        rk2 = gpu.optimize()
        assert gpu.last_choice()["chosen"] == ch["chosen"] and gpu.last_choice()["chosen_cost"] == ch["chosen_cost"] and np.array_equal(rk2["x"], rk["x"])
