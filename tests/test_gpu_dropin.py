"""The drop-in compiled INTO the reference's own class.

oracle/_ref/libdftpav_dropin.so (recipe: oracle/Makefile.dropin) is the reference's `plan_manage::PolyTrajOptimizer` -- its header
traj_optimizer.h unmodified, as are poly_traj_utils.hpp / traj_container.hpp behind it -- with the drop-in's implementation of the
live path (dftpav_amd/csrc/host/dropin/traj_optimizer_hip.cpp over libdftpav_hip.so) in place of traj_optimizer.cpp, driven by the
same oracle/ref_driver.cpp that drives the reference build oracle/_ref/libdftpav_ref.so.  So the same calls on the same class --
setParam, setSurroundTrajs, OptimizeTrajectory(iniStates, finStates, innerPts, Ts, hPolys, singuls, now, eps)
(traj_manager.cpp:604-610), getMinJerkOptPtr() (:618-625), costFunctionCallback -- run once on the CPU (the reference) and once on
the GPU (the drop-in), and the test compares what the two objects hand back.

Bar: single-segment static problems (no libm call in the reference's loop): BIT-EQUAL -- the bool, the solution vector, the final
cost, the solver status, iterations, evaluations, and the coefficients / piece durations getMinJerkOptPtr() exposes.  Gear shifts
and moving obstacles: bit-equal to oracle order 2 (the reference's program with correctly rounded cos / sin / exp / log / pow) and
within 1e-12 per evaluation of the reference build.
"""
import numpy as np
import pytest

from dftpav_amd import scenarios as sc

pytestmark = pytest.mark.gpu


def _pyref():
    from oracle import pyref
    if not (pyref.available() and pyref.dropin_available()):
        pytest.skip("oracle/_ref libraries not built (they are built where /root/reference exists and travel with the tree)")
    return pyref


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
        ref = pyref.RefProblem(p, s, b)                  # traj_optimizer.cpp, CPU
        gpu = pyref.RefProblem(p, s, b, dropin=True)     # traj_optimizer_hip.cpp, GPU -- the same class, the same driver
        rr, rg = ref.optimize(), gpu.optimize()
        assert rg["ok"] == rr["ok"], (name, b)
        assert rg["final_cost"] == rr["final_cost"] and np.array_equal(rg["x"], rr["x"]), (name, b, rg["final_cost"], rr["final_cost"])
        assert rg["status"] == rr["status"] and rg["iters"] == rr["iters"] and rg["evals"] == rr["evals"], (name, b)
        # getMinJerkOptPtr()[i].getCoeffs() / getDt(): what traj_manager.cpp:618-625 builds the published trajectory from.  The
        # reference's containers hold the LAST EVALUATED point; after a solve that ended with a successful line search that is the
        # solution (status >= 0), which is what the drop-in regenerates
        if rr["status"] >= 0:
            cr, dtr = ref.coeffs()
            cg, dtg = gpu.coeffs()
            assert np.array_equal(cg, cr) and np.array_equal(dtg, dtr), (name, b)
        # costFunctionCallback on both objects at a common point
        x = rr["x"] + 0.05 * np.sin(np.arange(len(rr["x"])) + b)
        fr, gr = ref.eval(x)
        fg, gg = gpu.eval(x)
        assert fg == fr and np.array_equal(gg, gr), (name, b)


def test_reference_object_gpu_backed_live_case(hiplib, oracle):
    """gear shifts with moving obstacles through the class interface (setSurroundTrajs + multi-segment containers): the drop-in
    object against oracle order 2 bit for bit, against the reference object to 1e-12 per evaluation"""
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
        ref = pyref.RefProblem(p, s, b)
        ref.optimize()
        x = want["x"][b]
        fr, gr = ref.eval(x)
        fg, gg = gpu.eval(x)
        assert abs(fg - fr) <= 1e-12 * abs(fr) and np.abs(gg - gr).max() <= 1e-12 * max(1.0, np.abs(gr).max()), b
        if pyref.cr_available():
            # the reference's own objects linked against a correctly rounded libm (oracle/cr_libm.c): THAT object and the GPU-backed
            # one hand back the same bits -- the whole solve and costFunctionCallback at a common point
            refc = pyref.RefProblem(p, s, b, cr=True)
            rc = refc.optimize()
            assert rg["final_cost"] == rc["final_cost"] and np.array_equal(rg["x"], rc["x"]), b
            assert (rg["status"], rg["iters"], rg["evals"], bool(rg["ok"])) == (rc["status"], rc["iters"], rc["evals"], bool(rc["ok"])), b
            fc, gc = refc.eval(x)
            assert fg == fc and np.array_equal(gg, gc), b


def test_restarts_behind_the_reference_entry_point(hiplib, oracle, monkeypatch):
    """DFTPAV_DROPIN_RESTARTS=64: the planner's one call becomes a batch of 64 in the same launch -- slot 0 the call's own problem,
    63 seeded restarts of it (north_star's batch dimension behind OptimizeTrajectory).  Slot 0 stays BIT-EQUAL to the reference
    build's solve (the restarts change nothing for it); the returned candidate is the cheapest successful one, collision-checked on
    the device when the map is given; getMinJerkOptPtr() hands out the chosen candidate's coefficients."""
    pyref = _pyref()
    p, s = _scenario(oracle, hiplib, "cfg3", 3)
    st = s.meta["states"]
    c = (0.5 * (st[..., 0].min() + st[..., 0].max()), 0.5 * (st[..., 1].min() + st[..., 1].max()))
    grid, origin = sc.occupancy_grid(s.meta["obstacles"], arena=140.0, centre=c)
    lay = s.layout
    for b in range(3):
        ref = pyref.RefProblem(p, s, b)
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
            assert np.array_equal(cg, cr) and np.array_equal(dtg, dtr)
        # the same seed gives the same choice; the cached batch is reused
        rk2 = gpu.optimize()
        assert gpu.last_choice()["chosen"] == ch["chosen"] and gpu.last_choice()["chosen_cost"] == ch["chosen_cost"] and np.array_equal(rk2["x"], rk["x"])
