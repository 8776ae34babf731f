"""GPU parity tests: the HIP path, called through the C-ABI, against the CPU oracle.

Bar (north_star: final cost within 1e-5 relative of the CPU L-BFGS): because the reference
solver is chaotic (tests/test_oracle_orders.py::test_reference_solver_is_chaotic) the kernel
is held to BIT-EXACT agreement with the oracle's device-order mode on whole solves — cost,
x, status, iterations, evaluations — and to <= 1e-11 relative against the literal mode on
single evaluations (the L1 cut, costFunctionCallback).
"""
import numpy as np
import pytest

from dftpav_amd import scenarios as sc
from golden_util import CASES, load

pytestmark = pytest.mark.gpu

REL_TOL_FINAL_COST = 1e-5  # north_star; met with 0.0 in device order


def _batch(hiplib, s, p):
    h = hiplib.Handle(p)
    h.set_surround(s.surround)
    bt = hiplib.Batch(h, s.layout, s.B)
    bt.upload(s)
    return h, bt


@pytest.mark.parametrize("cfg,B", [(1, 4), (2, 4), (3, 16), (5, 2)])
def test_eval_matches_oracle(hiplib, oracle, cfg, B):
    """L1 cut: costFunctionCallback (traj_optimizer.cpp:206-350) at x0 and at perturbed points."""
    p = hiplib.default_params()
    s = sc.baseline_config(cfg, B=B)
    s.apply_resolution(p)
    h, bt = _batch(hiplib, s, p)
    x0 = bt.x0()
    rng = np.random.default_rng(cfg)
    for scale in (0.0, 0.05, 0.5):
        x = x0 + rng.normal(0, scale, x0.shape) if scale else x0
        f, g = bt.eval(x)
        for b in range(B):
            dev = oracle.OracleProblem(p, s, b, order=1)
            lit = oracle.OracleProblem(p, s, b, order=0)
            if scale == 0.0:
                assert np.array_equal(dev.x0(), x0[b])  # packing of traj_optimizer.cpp:96-115
            fd, gd = dev.eval(x[b])
            assert f[b] == fd and np.array_equal(g[b], gd)  # bit-exact vs device order
            fl, gl = lit.eval(x[b])
            assert abs(f[b] - fl) <= 1e-11 * abs(fl)        # rounding level vs the literal restatement
            assert np.abs(g[b] - gl).max() <= 1e-10 * np.abs(gl).max()
    bt.close()
    h.close()


def test_far_trial_points(hiplib, oracle):
    """The first trial point of a line search (step 1 along a fresh direction, lbfgs.hpp:276-390) can lie 1e10 away late in a hard
    solve -- junction positions AND junction angles; the reference evaluates a finite cost there (8e52 in the case
    scripts/fuzz_reference_order.py found) and backs off.  The portable cos / sin of traj_math.h reduce such angles with the
    reduction of Payne and Hanek (up to 2^20: two-term Cody-Waite): evaluations stay finite, bit-equal to the device-order
    oracle and at rounding distance from the literal program."""
    p = hiplib.default_params()
    s = sc.make_scenario([8, 10, 5], [1, -1, 1], 21, 8, 2, seed=23150, n_obs=30)
    s.apply_resolution(p)
    h, bt = _batch(hiplib, s, p)
    x0 = bt.x0()
    rng = np.random.default_rng(5)
    for scale in (1.0e3, 1.0e6, 3.0e10, 1.0e15):
        x = x0.copy()
        x[:, :40] += rng.normal(0, min(scale, 1.0e10), (2, 40))   # inner points
        x[:, 40:43] = rng.uniform(-60, 60, (2, 3))                  # virtual times: durations of 1e-3 .. 1e3 s
        x[:, 43:47] += rng.normal(0, min(scale, 1.0e10), (2, 4))   # junction positions
        x[:, 47:] = rng.normal(0, scale, (2, 2))                    # junction angles
        f, g = bt.eval(x)
        for b in range(2):
            fd, gd = oracle.OracleProblem(p, s, b, order=1).eval(x[b])
            assert np.isfinite(fd) and f[b] == fd and np.array_equal(g[b], gd), (scale, b, f[b], fd)
            fl, gl = oracle.OracleProblem(p, s, b, order=0).eval(x[b])
            assert abs(f[b] - fl) <= 1e-10 * abs(fl) and np.abs(g[b] - gl).max() <= 1e-9 * np.abs(gl).max(), (scale, b)
    bt.close()
    h.close()


def test_configs4_sixty_four_solves_are_bit_exact(hiplib, oracle):
    """BASELINE configs[4] (32 pieces x 65 points, four moving cars), 64 trajectories in the device order: every field of every
    solve equal to the device-order oracle's -- with the round-4 bound that rejects a (point, obstacle) pair before any
    exponential when its penalty is certainly zero (traj_math.h: dynamic_pair; the oracle shares that code) -- and the literal
    cost at every final x within 1e-11 (the literal oracle evaluates every pair in full: the bound changes no value)."""
    p = hiplib.default_params()
    s = sc.baseline_config(5, B=64)
    s.apply_resolution(p)
    h, bt = _batch(hiplib, s, p)
    r = bt.solve()
    ro = oracle.solve_batch(p, s, nthreads=8, order=1)
    for k in ("final_cost", "x", "status", "iters", "evals", "hist_sum", "success"):
        assert np.array_equal(r[k], ro[k]), k
    f, _ = bt.eval(r["x"])
    for b in range(0, s.B, 4):
        fl, _ = oracle.OracleProblem(p, s, b, order=0).eval(r["x"][b])
        assert abs(f[b] - fl) <= 1e-11 * abs(fl), (b, f[b], fl)
    bt.close()
    h.close()


def test_moving_obstacles_that_start_after_t_now(hiplib, oracle):
    """An obstacle whose predicted trajectory starts later than the ego's clock (surround start_time > t_now, the normal
    swarm situation) is extrapolated BACKWARDS along its first piece (Trajectory::locatePieceIdx returns piece 0 with a
    negative local time, traj_optimizer.cpp:1374-1378): such a position lies outside the piece's hull box, so the box
    rejection of the gate must not be applied there.  The device-order oracle has no box table (it walks the pieces)."""
    p = hiplib.default_params()
    s = sc.baseline_config(5, B=3)
    s.apply_resolution(p)
    s.surround.start_time[:] = [4.0, 1.5, 0.0, 9.0]      # three of the four cars start after t_now = 0
    s.t_now = 0.25
    h, bt = _batch(hiplib, s, p)
    x0 = bt.x0()
    rng = np.random.default_rng(11)
    npairs = 0
    for scale in (0.0, 0.3):
        x = x0 + rng.normal(0, scale, x0.shape) if scale else x0
        f, g = bt.eval(x)
        for b in range(s.B):
            dev = oracle.OracleProblem(p, s, b, order=1)
            fd, gd = dev.eval(x[b])
            assert f[b] == fd and np.array_equal(g[b], gd)
            lit = oracle.OracleProblem(p, s, b, order=0)
            fl, gl = lit.eval(x[b])
            assert abs(f[b] - fl) <= 1e-11 * abs(fl)
            npairs += lit.cost_terms()[3] > 0.0
    assert npairs > 0        # the moving-obstacle term is active somewhere, or this test checks nothing
    r = bt.solve()
    ro = oracle.solve_batch(p, s, nthreads=3, order=1)
    assert np.array_equal(r["final_cost"], ro["final_cost"]) and np.array_equal(r["x"], ro["x"])
    assert np.array_equal(r["iters"], ro["iters"]) and np.array_equal(r["evals"], ro["evals"])
    bt.close()
    h.close()


@pytest.mark.parametrize("cfg,B", [(1, 4), (2, 4), (3, 32), (5, 2)])
def test_solve_matches_oracle(hiplib, oracle, cfg, B):
    """L2 cut: the whole lbfgs_optimize run (lbfgs.hpp:440-751), every trajectory of the batch."""
    p = hiplib.default_params()
    s = sc.baseline_config(cfg, B=B)
    s.apply_resolution(p)
    h, bt = _batch(hiplib, s, p)
    r = bt.solve()
    ro = oracle.solve_batch(p, s, nthreads=4, order=1)
    rel = np.abs(r["final_cost"] - ro["final_cost"]) / np.abs(ro["final_cost"])
    assert rel.max() <= REL_TOL_FINAL_COST
    assert np.array_equal(r["final_cost"], ro["final_cost"]) and np.array_equal(r["x"], ro["x"])
    assert np.array_equal(r["status"], ro["status"]) and np.array_equal(r["success"], ro["success"])
    assert np.array_equal(r["iters"], ro["iters"]) and np.array_equal(r["evals"], ro["evals"])
    assert np.array_equal(r["hist_sum"], ro["hist_sum"])
    assert r["success"].all()
    # coefficients regenerated from the returned x (getMinJerkOptPtr contract, traj_manager.cpp:618-625)
    c, dt = bt.coeffs()
    for b in range(min(B, 2)):
        dev = oracle.OracleProblem(p, s, b, order=1)
        dev.eval(r["x"][b])
        co, dto = dev.coeffs()
        assert np.array_equal(c[b], co) and np.array_equal(dt[b], dto)
    bt.close()
    h.close()


@pytest.mark.parametrize("cfg,B,slots,slice_,hand_over", [(3, 48, 8, 7, 3), (3, 48, 5, 40, 0), (5, 6, 2, 9, 1), (2, 12, 4, 1, 12)])
def test_time_sliced_schedule_is_bit_identical(hiplib, oracle, monkeypatch, cfg, B, slots, slice_, hand_over):
    """Batches larger than the device holds at once are solved by persistent workgroups that suspend and resume
    trajectories (solver.hip, SchedArgs): forced here on small batches, it must not change one bit."""
    monkeypatch.setenv("DFTPAV_SCHED", "1")
    monkeypatch.setenv("DFTPAV_SLOTS", str(slots))
    monkeypatch.setenv("DFTPAV_SLICE", str(slice_))
    monkeypatch.setenv("DFTPAV_HANDOVER", str(hand_over))
    p = hiplib.default_params()
    s = sc.baseline_config(cfg, B=B)
    s.apply_resolution(p)
    h, bt = _batch(hiplib, s, p)
    ro = oracle.solve_batch(p, s, nthreads=8, order=1)
    for rep in range(2):  # the second solve reuses queue, flags and state records
        r = bt.solve()
        assert np.array_equal(r["final_cost"], ro["final_cost"])
        assert np.array_equal(r["x"], ro["x"])
        for k in ("status", "iters", "evals", "success"):
            assert np.array_equal(r[k], ro[k]), k
        assert np.array_equal(r["hist_sum"], ro["hist_sum"])
        assert (r["latency_us"] > 0).all()
    bt.close()
    h.close()


@pytest.mark.parametrize("name", CASES)
def test_golden_fixtures(hiplib, name):
    """Against the committed golden vectors (no oracle at run time)."""
    s, z = load(name)
    p = hiplib.default_params()
    s.apply_resolution(p)
    h, bt = _batch(hiplib, s, p)
    assert np.array_equal(bt.x0(), z["x0"])
    f, g = bt.eval(z["x0"])
    assert np.array_equal(f, z["dev_f0"]) and np.array_equal(g, z["dev_g0"])
    assert np.allclose(f, z["lit_f0"], rtol=1e-11) and np.allclose(g, z["lit_g0"], rtol=1e-9, atol=1e-8)
    r = bt.solve()
    assert np.array_equal(r["final_cost"], z["dev_cost"]) and np.array_equal(r["x"], z["dev_x"])
    assert np.array_equal(r["iters"], z["dev_iters"]) and np.array_equal(r["evals"], z["dev_evals"])
    assert np.array_equal(r["status"], z["dev_status"]) and np.array_equal(r["hist_sum"], z["dev_hist"])
    bt.close()
    h.close()


def test_full_size_properties(hiplib, oracle):
    """BASELINE config 3 at full size (256 x 16 pieces x 33 pts): size-independent properties."""
    p = hiplib.default_params()
    s = sc.baseline_config(3, B=256)
    s.apply_resolution(p)
    h, bt = _batch(hiplib, s, p)
    r1 = bt.solve()
    r2 = bt.solve()
    # determinism: the resident batch solves to the same bits every launch
    for k in ("x", "final_cost", "status", "iters", "evals"):
        assert np.array_equal(r1[k], r2[k]), k
    assert r1["success"].all() and (r1["final_cost"] < 50000.0).all()
    # every solve decreased the cost it started from
    f0, _ = bt.eval(bt.x0())
    assert (r1["final_cost"] < f0).all()
    # permutation equivariance: trajectories are independent (no cross-batch coupling)
    perm = np.random.default_rng(0).permutation(256)
    sp = s.subset(perm)
    hb, bp = _batch(hiplib, sp, p)
    rp = bp.solve()
    assert np.array_equal(rp["final_cost"], r1["final_cost"][perm]) and np.array_equal(rp["x"], r1["x"][perm])
    # stationarity: the reference's L-BFGS (literal oracle) restarted from the kernel's solutions sees the same cost there
    # and stops at no higher cost, most of the time within the minimum of `past` = 3 iterations
    first = np.arange(16)
    ev = oracle.batch_op(p, s.subset(first), "eval", r1["x"][first], nthreads=4, order=0)
    assert np.max(np.abs(ev["f"] - r1["final_cost"][first]) / np.maximum(1.0, np.abs(ev["f"]))) <= 1e-11
    rst = oracle.batch_op(p, s.subset(first), "restart", r1["x"][first], nthreads=4, order=0)
    # (the line search accepts a step whose relative change is below delta / past whatever its sign, lbfgs.hpp:326-329)
    assert (rst["final_cost"] <= ev["f"] * (1.0 + p.lbfgs_delta / p.lbfgs_past)).all()
    assert np.median(rst["iters"]) <= 4 and rst["success"].all()
    # a sample of the full batch checked bit-for-bit against the oracle
    idx = np.array([0, 17, 101, 255])
    ro = oracle.solve_batch(p, s.subset(idx), nthreads=4, order=1)
    assert np.array_equal(ro["final_cost"], r1["final_cost"][idx]) and np.array_equal(ro["iters"], r1["iters"][idx])
    bp.close()
    hb.close()
    bt.close()
    h.close()


def test_edge_cases(hiplib, oracle):
    p = hiplib.default_params()
    # smallest legal layout: 2 pieces (traj_manager.cpp:543 max(...,2)), reference resolutions 16/32
    s = sc.make_scenario([2], [1], 16, 32, 3, seed=5, n_obs=10, name="two_piece")
    s.apply_resolution(p)
    h, bt = _batch(hiplib, s, p)
    r = bt.solve()
    ro = oracle.solve_batch(p, s, nthreads=1, order=1)
    assert np.array_equal(r["final_cost"], ro["final_cost"]) and np.array_equal(r["iters"], ro["iters"])
    bt.close()
    # three gear segments with uneven piece counts, reverse first
    s = sc.make_scenario([3, 5, 2], [-1, 1, -1], 8, 12, 2, seed=6, n_obs=10, name="three_seg")
    p2 = hiplib.default_params()
    s.apply_resolution(p2)
    h2, bt = _batch(hiplib, s, p2)
    f, g = bt.eval(bt.x0())
    for b in range(2):
        fd, gd = oracle.OracleProblem(p2, s, b, order=1).eval(bt.x0()[b])
        assert f[b] == fd and np.array_equal(g[b], gd)
    r = bt.solve()
    ro = oracle.solve_batch(p2, s, nthreads=1, order=1)
    assert np.array_equal(r["final_cost"], ro["final_cost"]) and np.array_equal(r["x"], ro["x"])
    bt.close()
    h2.close()
    # validation errors of traj_optimizer.cpp:26-48 surface as return codes, not crashes
    s = sc.baseline_config(1, B=2)
    p3 = hiplib.default_params()
    s.apply_resolution(p3)
    s.init_Ts[1, 0] = 0.01
    bt = hiplib.Batch(h, s.layout, 2)
    with pytest.raises(hiplib.DftpavError) as e:
        bt.upload(s)
    assert e.value.code == hiplib.E_MINI_T
    bt.close()
    # an obstacle without pieces is refused (Trajectory::locatePieceIdx has no empty case)
    from dftpav_amd.pods import SurroundSet
    good = sc.baseline_config(5, B=1).surround
    bad = SurroundSet(np.array([0, 3, 3, 5], dtype=np.int32), good.durations[:5], good.coeffs[:5], good.total_duration[:3], good.start_time[:3])
    with pytest.raises(hiplib.DftpavError) as e:
        h.set_surround(bad)
    assert e.value.code == hiplib.E_INVALID
    h.close()


def test_reference_class_interface(hiplib):
    """PolyTrajOptimizer mirror: containers as traj_manager.cpp:608-610 passes them, B = 1."""
    from dftpav_amd.optimizer import PolyTrajOptimizer
    s = sc.baseline_config(2, B=1)
    p = hiplib.default_params()
    s.apply_resolution(p)
    opt = PolyTrajOptimizer()
    opt.setParam(p)
    lay = s.layout
    ini = [s.ini_states[0, i].reshape(3, 2).T for i in range(lay.M)]
    fin = [s.fin_states[0, i].reshape(3, 2).T for i in range(lay.M)]
    inner, off = [], 0
    for N in lay.piece_nums:
        inner.append(s.inner_pts[0, off:off + 2 * (N - 1)].reshape(N - 1, 2).T)
        off += 2 * (N - 1)
    polys, pt = [], 0
    for N in lay.piece_nums:
        cnt = (N - 2) * (s.K + 1) + 2 * (s.Kd + 1)
        polys.append([s.corridor[0, pt + k].T for k in range(cnt)])  # 4xH each
        pt += cnt
    ok = opt.OptimizeTrajectory(ini, fin, inner, s.init_Ts[0], polys, list(lay.singuls), 0.0, 0.0)
    assert ok is True
    mjo = opt.getMinJerkOptPtr()
    assert len(mjo) == 2 and mjo[0].getCoeffs().shape == (48, 2) and mjo[0].getDt() > 0
    # same answer as the batched entry point
    rb = PolyTrajOptimizer()
    rb.setParam(p)
    r = rb.OptimizeTrajectoryBatch(s)
    assert r["final_cost"][0] == opt.last["final_cost"][0]
    # size mismatch -> False, as traj_optimizer.cpp:44-48
    assert opt.OptimizeTrajectory(ini, fin, inner, s.init_Ts[0], [polys[0][:-1], polys[1]], list(lay.singuls)) is False
    assert opt.OptimizeTrajectory(ini, fin, inner, [0.05, 8.0], polys, list(lay.singuls)) is False


def test_more_than_64_variables_generic_path(hiplib, oracle):
    """n = 79 decision variables: vectors span two elements per lane, the two-loop recursion takes the
    strided (n > 64) path, six butterfly levels."""
    p = hiplib.default_params()
    s = sc.make_scenario([40], [1], 4, 6, 2, seed=9, n_obs=10, name="forty_pieces")
    s.apply_resolution(p)
    assert s.layout.n_vars == 79
    p.lbfgs_max_iterations = 60  # keep the CPU side short; MAXIMUMITERATION counts as success (traj_optimizer.cpp:179)
    h, bt = _batch(hiplib, s, p)
    f, g = bt.eval(bt.x0())
    for b in range(2):
        fd, gd = oracle.OracleProblem(p, s, b, order=1).eval(bt.x0()[b])
        assert f[b] == fd and np.array_equal(g[b], gd)
    r = bt.solve()
    ro = oracle.solve_batch(p, s, nthreads=2, order=1)
    assert np.array_equal(r["final_cost"], ro["final_cost"]) and np.array_equal(r["x"], ro["x"])
    assert np.array_equal(r["iters"], ro["iters"]) and np.array_equal(r["status"], ro["status"])
    bt.close()
    h.close()


def test_six_half_planes_per_point(hiplib, oracle):
    """hPoly matrices with more than four columns (the reference takes any 4xH, traj_optimizer.cpp:600)."""
    p = hiplib.default_params()
    base = sc.baseline_config(1, B=2)
    base.apply_resolution(p)
    cor = np.zeros((2, base.n_points, 6, 4))
    cor[:, :, :4] = base.corridor
    # two extra planes: the first two pulled 0.2 m inwards (so that they are the active ones), un-normalised normals
    cor[:, :, 4:] = base.corridor[:, :, :2]
    cor[:, :, 4:, 2:] -= 0.2 * base.corridor[:, :, :2, :2]
    cor[:, :, 4:, :2] *= 3.0
    lay = type(base.layout)(base.layout.piece_nums, base.layout.singuls, H=6)
    s = sc.Scenario("six_planes", lay, base.K, base.Kd, 2, base.ini_states, base.fin_states, base.inner_pts,
                    base.init_Ts, np.ascontiguousarray(cor))
    h, bt = _batch(hiplib, s, p)
    x = bt.x0() + np.random.default_rng(3).normal(0, 0.3, bt.x0().shape)
    f, g = bt.eval(x)
    for b in range(2):
        fd, gd = oracle.OracleProblem(p, s, b, order=1).eval(x[b])
        fl, gl = oracle.OracleProblem(p, s, b, order=0).eval(x[b])
        assert f[b] == fd and np.array_equal(g[b], gd)
        assert abs(f[b] - fl) <= 1e-11 * abs(fl)
    r = bt.solve()
    ro = oracle.solve_batch(p, s, nthreads=2, order=1)
    assert np.array_equal(r["final_cost"], ro["final_cost"]) and np.array_equal(r["iters"], ro["iters"])
    bt.close()
    h.close()


def test_cpp_host_mirror_runs(hiplib):
    """The C++ PolyTrajOptimizer mirror (dftpav_amd/csrc/host) against the C-ABI on a real device."""
    import os
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    host = os.path.join(root, "dftpav_amd", "csrc", "host")
    subprocess.check_call(["make", "-C", host, "-s"])
    out = subprocess.run([os.path.join(host, "host_example")], capture_output=True, text=True, timeout=120)
    assert out.returncode == 0, out.stdout + out.stderr
    assert "OptimizeTrajectory -> 1" in out.stdout and "short corridor -> 0" in out.stdout
    # the TrajPlanner / KinoAstar steps around the solve (traj_planner_steps.hpp)
    assert "getKinoNode -> 1 segment(s)" in out.stdout and "surround fit -> 1" in out.stdout
    # the read-out (GetStates) and the plan as bytes, installed as an obstacle
    assert "GetStates -> " in out.stdout and "installed as an obstacle -> 1" in out.stdout
    # the Reeds-Shepp shot of the front end
    assert "is_shot_sucess: free 1, through the wall 0" in out.stdout


def _corridor_scene(seed):
    rng = np.random.default_rng(seed)
    obs = np.column_stack([rng.uniform(-25, 25, 60), rng.uniform(-25, 25, 60), rng.uniform(0.5, 1.5, 60)])
    grid, origin = sc.occupancy_grid(obs, arena=80.0)
    states = np.column_stack([rng.uniform(-22, 22, 1500), rng.uniform(-22, 22, 1500), rng.uniform(-7.0, 7.0, 1500)])
    return grid, origin, states


@pytest.mark.parametrize("seed", [1, 2])
def test_corridor_rectangles_match_oracle(hiplib, oracle, seed):
    """§8(f)-1: getRectangleConst (traj_manager.cpp:1213-1469) on the device: every rectangle BIT-EQUAL to the restatement of the
    reference's function with its libm calls (cos / sin of the pose's heading) correctly rounded -- oracle order 2, from binary128;
    the kernel's own double-double cos / sin (cr_trig.h) is a second, unrelated route to the same bits, and order 1 replays it on the
    host.  (Round 5 compared portable cos / sin with libm and tolerated corridor sides one step apart on 2 % of the poses.)"""
    grid, origin, states = _corridor_scene(seed)
    h = hiplib.Handle(hiplib.default_params())
    h.set_grid_map(grid, sc.MAP_RESL, origin)
    H = h.corridor_rectangles(states)
    H2 = oracle.corridor_rectangles(grid, sc.MAP_RESL, origin, states, order=2)
    assert np.array_equal(H, H2)
    assert np.array_equal(H, oracle.corridor_rectangles(grid, sc.MAP_RESL, origin, states, order=1))
    # edge cases: no states, a state outside the map (every sample out of range counts as free), an empty map
    assert h.corridor_rectangles(np.zeros((0, 3))).shape == (0, 4, 4)
    far = np.array([[500.0, -300.0, 0.7]])
    assert np.array_equal(h.corridor_rectangles(far), oracle.corridor_rectangles(grid, sc.MAP_RESL, origin, far, order=2))
    empty = np.full((50, 70), 127, dtype=np.uint8)
    h.set_grid_map(empty, 0.25, (-3.0, -4.0))
    st = states[:40]
    assert np.array_equal(h.corridor_rectangles(st), oracle.corridor_rectangles(empty, 0.25, (-3.0, -4.0), st, order=2))
    h.close()


def test_generated_corridor_feeds_the_solver(hiplib, oracle):
    """corridor generation -> solve, both on the device, against the same chain on the CPU oracle."""
    p = hiplib.default_params()
    s = sc.baseline_config(3, B=8)
    s.apply_resolution(p)
    st = s.meta["states"]
    c = (0.5 * (st[..., 0].min() + st[..., 0].max()), 0.5 * (st[..., 1].min() + st[..., 1].max()))
    grid, origin = sc.occupancy_grid(s.meta["obstacles"], arena=120.0, centre=c)
    h = hiplib.Handle(p)
    h.set_grid_map(grid, sc.MAP_RESL, origin)
    Hd = h.corridor_rectangles(st.reshape(-1, 3)).reshape(st.shape[0], st.shape[1], 4, 4)
    Ho = oracle.corridor_rectangles(grid, sc.MAP_RESL, origin, st.reshape(-1, 3), order=1).reshape(Hd.shape)
    assert np.array_equal(Hd, Ho)
    s.corridor = np.ascontiguousarray(Hd[s.meta["hyp_of"]])
    bt = hiplib.Batch(h, s.layout, s.B)
    bt.upload(s)
    r = bt.solve()
    ro = oracle.solve_batch(p, s, nthreads=4, order=1)
    assert np.array_equal(r["final_cost"], ro["final_cost"]) and np.array_equal(r["x"], ro["x"])
    assert r["success"].all()
    # the same without the rectangles leaving the device (dftpav_batch_corridor_from_states)
    b2 = hiplib.Batch(h, s.layout, s.B)
    b2.upload(s, with_corridor=False)
    with pytest.raises(hiplib.DftpavError):
        b2.solve_async()  # no corridor yet
    b2.corridor_from_states(st[s.meta["hyp_of"]])
    assert h.corridor_last_ms() > 0.0
    f1, g1 = bt.eval(bt.x0())
    f2, g2 = b2.eval(b2.x0())
    assert np.array_equal(f1, f2) and np.array_equal(g1, g2)
    r2 = b2.solve()
    assert np.array_equal(r2["final_cost"], r["final_cost"]) and np.array_equal(r2["x"], r["x"])
    # restarts sharing their hypothesis' corridor: one rectangle set per hypothesis, replicated on the device
    order = np.argsort(s.meta["hyp_of"], kind="stable")  # trajectory = hypothesis * n_restarts + restart
    nr = s.B // st.shape[0]
    s3 = s.subset(order)
    b3 = hiplib.Batch(h, s.layout, s.B)
    b3.upload(s3, with_corridor=False)
    b3.corridor_from_states(st, n_restarts=nr)
    f3, g3 = b3.eval(b3.x0())
    assert np.array_equal(f3, f1[order]) and np.array_equal(g3, g1[order])
    b3.close()
    b2.close()
    bt.close()
    h.close()


@pytest.mark.parametrize("cfg,B", [(3, 24), (2, 6)])
def test_validation_matches_oracle(hiplib, oracle, cfg, B):
    """§8(f)-2: the sampled collision re-check of CheckReplan (traj_server_ros.cpp:385-397) on the device: the collision flag and
    the first colliding sample of every trajectory EQUAL to the restatement with correctly rounded atan2 / cos / sin (oracle order 2,
    binary128) -- on the solved trajectories, on a clear map and on maps with obstacles dropped onto the paths, at two sampling /
    outline spacings.  (Round 5: portable functions against libm, 90 % of the verdicts.)"""
    p = hiplib.default_params()
    s = sc.baseline_config(cfg, B=B)
    s.apply_resolution(p)
    st = s.meta["states"]
    c = (0.5 * (st[..., 0].min() + st[..., 0].max()), 0.5 * (st[..., 1].min() + st[..., 1].max()))
    obs = s.meta["obstacles"]
    grid, origin = sc.occupancy_grid(obs, arena=140.0, centre=c)
    h, bt = _batch(hiplib, s, p)
    bt.solve()
    co, dts = bt.coeffs()
    lay = s.layout
    h.set_grid_map(grid, sc.MAP_RESL, origin)
    col, first = bt.validate()
    oc, of = oracle.validate_trajectories(grid, sc.MAP_RESL, origin, co, dts, lay.piece_nums, lay.singuls, order=2)
    assert np.array_equal(col, oc) and np.array_equal(first, of)
    assert h.corridor_last_ms() > 0.0
    # obstacles on the nominal paths: most trajectories must be flagged, with the same first sample
    rng = np.random.default_rng(cfg)
    extra = []
    for hyp in range(st.shape[0]):
        for k in rng.choice(st.shape[1], 3, replace=False):
            extra.append([st[hyp, k, 0], st[hyp, k, 1], 0.8])
    grid2, origin2 = sc.occupancy_grid(np.vstack([obs, np.array(extra)]), arena=140.0, centre=c)
    h.set_grid_map(grid2, sc.MAP_RESL, origin2)
    col2, first2 = bt.validate()
    oc2, of2 = oracle.validate_trajectories(grid2, sc.MAP_RESL, origin2, co, dts, lay.piece_nums, lay.singuls, order=2)
    assert np.array_equal(col2, oc2) and np.array_equal(first2, of2)
    assert col2.mean() > 0.5 and (first2[col2 == 1] >= 0).all() and (first2[col2 == 0] == -1).all()
    oc1, of1 = oracle.validate_trajectories(grid2, sc.MAP_RESL, origin2, co, dts, lay.piece_nums, lay.singuls, order=1)
    assert np.array_equal(col2, oc1) and np.array_equal(first2, of1)       # (the kernel's functions replayed on the host)
    # coarser sampling / spacing parameters
    col3, first3 = bt.validate(sample_dt=0.21, vertex_res=0.37)
    oc3, of3 = oracle.validate_trajectories(grid2, sc.MAP_RESL, origin2, co, dts, lay.piece_nums, lay.singuls,
                                            sample_dt=0.21, vertex_res=0.37, order=2)
    assert np.array_equal(col3, oc3) and np.array_equal(first3, of3)
    bt.close()
    h.close()


def test_fit_surround_matches_oracle_and_feeds_the_solver(hiplib, oracle):
    """§8(f)-4: ConverSurroundTrajFromPoints (traj_manager.cpp:743-789) on the device: bit-exact against the oracle's
    device-order mode, and the fitted obstacles drive the moving-obstacle penalty exactly as uploaded ones do."""
    st = sc.predicted_states()
    h = hiplib.Handle(hiplib.default_params())
    h.fit_surround(st)
    got = h.get_surround()
    want = oracle.fit_surround(st, order=2)   # (the kernel's algorithm -- the dense operator -- with cos / sin from binary128)
    S, n = st.shape[0], st.shape[1]
    assert np.array_equal(got["offsets"], np.arange(S + 1) * (n - 1))
    assert np.array_equal(got["coeffs"].reshape(want["coeffs"].shape), want["coeffs"])
    assert np.array_equal(got["durations"].reshape(S, n - 1), want["durations"])
    assert np.array_equal(got["total"], want["total"]) and np.array_equal(got["start"], want["start"])
    lit = oracle.fit_surround(st, order=0)
    assert np.abs(lit["coeffs"] - want["coeffs"]).max() < 1e-9
    # other sizes: 3 states (two pieces), many obstacles, uneven time stamps, a stationary first state
    rng = np.random.default_rng(5)
    st2 = sc.predicted_states(pre_time=9.0, deltatime=0.75, cars=[(rng.uniform(-30, 30), rng.uniform(-30, 30), rng.uniform(1, 6),
                                                                   rng.uniform(5, 20), rng.uniform(-3, 3)) for _ in range(37)])
    st2[:, :, 6] += rng.uniform(0, 0.01, st2.shape[:2]).cumsum(axis=1)
    st2[3, 0, 3] = 0.0
    for sub in (st2, st2[:, :3]):
        h.fit_surround(sub)
        g2, w2 = h.get_surround(), oracle.fit_surround(sub, order=2)
        assert np.array_equal(g2["coeffs"].reshape(w2["coeffs"].shape), w2["coeffs"])
        assert np.array_equal(g2["total"], w2["total"])
    # a ready-made set beyond the solver's limits is refused where it is installed, and the installed one stays
    g37 = h.get_surround()
    from dftpav_amd.pods import SurroundSet as _SS
    with pytest.raises(hiplib.DftpavError) as e37:
        h.set_surround(_SS(g37["offsets"], g37["durations"], g37["coeffs"], g37["total"], g37["start"]))
    assert e37.value.code == hiplib.E_UNSUPPORTED and h.get_surround()["total"].size == 37
    h.fit_surround(np.zeros((0, 0, 7)))
    assert h.get_surround()["total"].size == 0
    # cfg 5 with obstacles fitted on the device == cfg 5 with the same fit uploaded through dftpav_set_surround
    p = hiplib.default_params()
    s = sc.baseline_config(5, B=2)
    s.apply_resolution(p)
    h2 = hiplib.Handle(p)
    h2.fit_surround(st)
    bt = hiplib.Batch(h2, s.layout, s.B)
    bt.upload(s)
    f_dev, g_dev = bt.eval(bt.x0())
    from dftpav_amd.pods import SurroundSet
    ss = SurroundSet(got["offsets"], got["durations"], got["coeffs"], got["total"], got["start"])
    h3 = hiplib.Handle(p)
    h3.set_surround(ss)
    b3 = hiplib.Batch(h3, s.layout, s.B)
    b3.upload(s)
    f_up, g_up = b3.eval(b3.x0())
    assert np.array_equal(f_dev, f_up) and np.array_equal(g_dev, g_up)
    for x in (bt, b3):
        x.close()
    for x in (h, h2, h3):
        x.close()


@pytest.mark.parametrize("gears,K,Kd", [((1, -1), 16, 32), ((-1, 1, -1, 1), 32, 32), ((1,), 7, 11)])
def test_frontend_resampling_matches_oracle(hiplib, oracle, gears, K, Kd):
    """§8(f)-3: getKinoNode from SampleTraj on + the resampling of RunMINCOParking (kino_astar.cpp:606-795,
    traj_manager.cpp:531-568) on the device, bit for bit against the restatement with correctly rounded cos / sin / tan (order 2)."""
    from dftpav_amd.pods import FrontendParams
    P, pl, ss, es, ct = sc.searched_paths(40, seed=len(gears) + K, gears=gears, seg_duration=6.0)
    fp = FrontendParams.default(K=K, Kd=Kd)
    h = hiplib.Handle(hiplib.default_params())
    got = h.frontend_resample(P, pl, ss, es, ct, fp)
    want = oracle.frontend_resample(P, pl, ss, es, ct, fp, order=2)
    for k in want:
        assert np.array_equal(got[k], want[k]), k
    assert (got["n_seg"] == len(gears)).all()
    lit = oracle.frontend_resample(P, pl, ss, es, ct, fp, order=0)
    assert np.array_equal(lit["piece_nums"], got["piece_nums"]) and np.abs(lit["states"] - got["states"]).max() < 1e-12
    # capacity smaller than the number of gear changes: counts reported, nothing produced
    small = h.frontend_resample(P, pl, ss, es, ct, fp, max_seg=max(1, len(gears) - 1))
    ws = oracle.frontend_resample(P, pl, ss, es, ct, fp, order=2, max_seg=max(1, len(gears) - 1))
    for k in ws:
        assert np.array_equal(small[k], ws[k]), k
    h.close()


def test_frontend_to_validation_chain(hiplib, oracle):
    """searched path -> resampling -> corridor -> solve -> validation, every stage on the device, against the
    same chain on the CPU oracle (device-order modes)."""
    from dftpav_amd.pods import FrontendParams, LayoutSpec
    from dftpav_amd.scenarios import Scenario
    nh = 6
    P, pl, ss, es, ct = sc.searched_paths(nh, seed=4, gears=(1, -1), seg_duration=7.0)
    fp = FrontendParams.default(K=16, Kd=32)
    p = hiplib.default_params()
    p.traj_resolution, p.des_traj_resolution = 16, 32
    h = hiplib.Handle(p)
    fe = h.frontend_resample(P, pl, ss, es, ct, fp)
    assert np.array_equal(fe["piece_nums"], oracle.frontend_resample(P, pl, ss, es, ct, fp, order=1)["piece_nums"])
    rng = np.random.default_rng(0)
    obs = np.column_stack([rng.uniform(-40, 40, 60), rng.uniform(-40, 40, 60), rng.uniform(0.5, 1.5, 60)])
    # keep the obstacles off the paths
    d = np.hypot(obs[:, None, 0] - P[:, :, 0].reshape(1, -1), obs[:, None, 1] - P[:, :, 1].reshape(1, -1)).min(axis=1)
    obs = obs[d > 5.0]
    grid, origin = sc.occupancy_grid(obs, arena=120.0)
    h.set_grid_map(grid, sc.MAP_RESL, origin)
    done = 0
    for hyp in range(nh):  # hypotheses with the same layout could share a batch; here one batch each
        M = int(fe["n_seg"][hyp])
        pn = [int(x) for x in fe["piece_nums"][hyp, :M]]
        lay = LayoutSpec(pn, [int(x) for x in fe["singul"][hyp, :M]], 4)
        npts = lay.n_points(16, 32)
        states = np.concatenate([fe["states"][hyp, i, :fe["n_states"][hyp, i]] for i in range(M)])
        assert states.shape[0] == npts
        inner = np.concatenate([fe["inner_pts"][hyp, i, :pn[i] - 1].reshape(-1) for i in range(M)])
        s = Scenario("fe", lay, 16, 32, 1, fe["ini_states"][hyp:hyp + 1, :M].copy(), fe["fin_states"][hyp:hyp + 1, :M].copy(),
                     inner[None].copy(), (fe["piece_dt"][hyp, :M] * fe["piece_nums"][hyp, :M])[None].copy(),
                     np.zeros((1, npts, 4, 4)))
        bt = hiplib.Batch(h, lay, 1)
        bt.upload(s, with_corridor=False)
        bt.corridor_from_states(states[None])
        r = bt.solve()
        s.corridor = oracle.corridor_rectangles(grid, sc.MAP_RESL, origin, states, order=1)[None]
        ro = oracle.solve_batch(p, s, nthreads=1, order=1)
        assert np.array_equal(r["final_cost"], ro["final_cost"]) and np.array_equal(r["x"], ro["x"])
        col, first = bt.validate()
        co, dts = bt.coeffs()
        oc, of = oracle.validate_trajectories(grid, sc.MAP_RESL, origin, co, dts, lay.piece_nums, lay.singuls, order=1)
        assert np.array_equal(col, oc) and np.array_equal(first, of)
        done += int(r["success"][0])
        bt.close()
    assert done >= nh - 1
    h.close()


def test_restart_sampler_matches_oracle(hiplib, oracle):
    """The seeded restart sampler (restart.hip; SURVEY §8(d) "Restarts"): bit-exact against its oracle."""
    rng = np.random.default_rng(3)
    h = hiplib.Handle(hiplib.default_params())
    for n_hyp, n_inner, M, nr in ((7, 30, 1, 33), (3, 26, 3, 200), (1, 0, 2, 4)):
        inner = rng.uniform(-10, 10, (n_hyp, n_inner))
        durs = rng.uniform(5, 20, (n_hyp, M))
        gi, gd = h.sample_restarts(inner, durs, nr, sigma=0.3, lo=0.8, hi=1.25, seed=2024)
        oi, od = oracle.sample_restarts(inner, durs, nr, sigma=0.3, lo=0.8, hi=1.25, seed=2024)
        assert np.array_equal(gi, oi) and np.array_equal(gd, od)
    h.close()


@pytest.mark.parametrize("cfg,B", [(2, 6), (3, 16)])
def test_state_sampling_matches_oracle(hiplib, oracle, cfg, B):
    """§8(f)-2, the read-out half: Trajectory::GetState (poly_traj_utils.hpp:378-406) over a time grid for every
    solved trajectory, with the server's playback rule and singularity filter (traj_server_ros.cpp:244-259, 335-356):
    every state and every valid count bit for bit against the restatement with correctly rounded atan2 / atan / x^3 (order 2)."""
    p = hiplib.default_params()
    s = sc.baseline_config(cfg, B=B)
    s.apply_resolution(p)
    h, bt = _batch(hiplib, s, p)
    bt.solve()
    co, dts = bt.coeffs()
    lay = s.layout
    total = (dts * lay.piece_nums[None, :]).sum(axis=1)
    for t0, dt, n, filt in [(0.0, 0.01, int(total.max() / 0.01) + 40, True), (0.0, 0.01, 300, False),
                            (-0.3, 0.037, 600, True), (2.5, 0.2, 7, True)]:
        st, nv = bt.sample_states(t0=t0, sample_dt=dt, n_samples=n, filter_singularity=filt)
        so, no = oracle.sample_states(co, dts, lay.piece_nums, lay.singuls, t0=t0, sample_dt=dt, n_samples=n,
                                      filter_singularity=filt, wheel_base=p.veh_wheel_base, order=2)
        assert np.array_equal(nv, no)
        assert np.array_equal(st, so)
        sl, nl = oracle.sample_states(co, dts, lay.piece_nums, lay.singuls, t0=t0, sample_dt=dt, n_samples=n,
                                      filter_singularity=filt, wheel_base=p.veh_wheel_base, order=0)
        assert np.array_equal(nv, nl)
        # the libm order differs by rounding only (headings compared on the circle; curvature amplifies 1 / v^3)
        assert np.allclose(st[..., [0, 1, 2, 5]], sl[..., [0, 1, 2, 5]], rtol=1e-12, atol=1e-12)
        assert np.abs(np.angle(np.exp(1j * (st[..., 3] - sl[..., 3])))).max() < 1e-9
    # the first n_valid samples cover the plan, the rest is zero; the read-out starts at the initial state
    st, nv = bt.sample_states(t0=0.0, sample_dt=0.01, n_samples=int(total.max() / 0.01) + 40)
    for b in range(B):
        assert abs(nv[b] * 0.01 - total[b]) <= 0.011
        assert (st[b, nv[b]:] == 0.0).all()
        assert np.allclose(st[b, 0, 1:3], s.ini_states[b, 0, :2], atol=1e-9)
    assert h.corridor_last_ms() > 0.0
    # limits the solve enforces softly, seen on the read-out
    v = np.abs(st[..., 5])
    assert v.max() < p.max_forward_vel * 1.1
    bt.close()
    h.close()


def test_wire_round_trip_feeds_the_solver(hiplib, oracle):
    """§8(f)-4: trajectories serialised as "DPTJ" blobs and installed as the moving obstacles are the obstacles the
    scenario installs directly: same tables on the device, same solves bit for bit; a solved plan survives the
    trip into another handle's obstacle set."""
    p = hiplib.default_params()
    s = sc.baseline_config(5, B=3)
    s.apply_resolution(p)
    sur = s.surround
    from dftpav_amd import capi, pods
    blobs = []
    for k in range(sur.S):
        a, e = sur.piece_offsets[k], sur.piece_offsets[k + 1]
        assert np.all(sur.durations[a:e] == sur.durations[a])
        co = sur.coeffs[a:e].reshape(-1, 6, 2)[:, ::-1, :]  # CoefficientMat order -> [power][x/y]
        lay1 = pods.LayoutSpec([e - a], [1])
        blobs.append(capi.wire_pack(lay1, co, np.array([sur.durations[a]]), drone_id=k, traj_id=k + 1,
                                    start_time=float(sur.start_time[k])))
    h, bt = _batch(hiplib, s, p)
    bt.solve()
    ref = bt.results()
    direct = h.get_surround()
    bt.close()
    h2 = hiplib.Handle(p)
    h2.set_surround_wire(blobs)
    got = h2.get_surround()
    for key in ("offsets", "durations", "coeffs", "start"):
        assert np.array_equal(direct[key], got[key]), key
    run = np.array([np.cumsum(sur.durations[sur.piece_offsets[k]:sur.piece_offsets[k + 1]])[-1] for k in range(sur.S)])
    assert np.array_equal(got["total"], run)  # Trajectory::getTotalDuration: running sum of the pieces
    if np.array_equal(run, sur.total_duration):
        bt2 = hiplib.Batch(h2, s.layout, s.B)
        bt2.upload(s)
        bt2.solve()
        r2 = bt2.results()
        assert np.array_equal(ref["final_cost"], r2["final_cost"]) and np.array_equal(ref["x"], r2["x"])
        bt2.close()
    # a reverse segment cannot be an obstacle of the reference's model
    lay_r = pods.LayoutSpec([2], [-1])
    bad = capi.wire_pack(lay_r, np.zeros((2, 6, 2)), np.array([1.0]))
    with pytest.raises(hiplib.DftpavError):
        h2.set_surround_wire([bad])
    # a solved forward plan, serialised, becomes somebody else's obstacle
    s3 = sc.baseline_config(3, B=2)
    p3 = hiplib.default_params()
    s3.apply_resolution(p3)
    h3, bt3 = _batch(hiplib, s3, p3)
    bt3.solve()
    co3, dt3 = bt3.coeffs()
    blob = capi.wire_pack(s3.layout, co3[1], dt3[1], drone_id=3, traj_id=9, start_time=1.25)
    h2.set_surround_wire([blob])
    g = h2.get_surround()
    assert list(g["offsets"]) == [0, s3.layout.n_pieces] and g["start"][0] == 1.25
    assert np.array_equal(g["coeffs"].reshape(-1, 6, 2)[:, ::-1, :], co3[1])
    h2.set_surround_wire([])
    assert len(h2.get_surround()["total"]) == 0
    bt3.close()
    h3.close()
    h2.close()


def test_chained_solves_are_bit_identical(hiplib, oracle, monkeypatch):
    """Throughput mode (dftpav_batch_solve_chained): the stragglers of one batch are worked off inside the queue
    launch of the next.  Three different batches of the same shape through the chain, small slot / slice / hand-over
    settings so that every path runs (adoption, adopted trajectories caught by the next end game, a final finish):
    every result equals the plain solve bit for bit, and the oracle on a sample."""
    monkeypatch.setenv("DFTPAV_SCHED", "1")
    monkeypatch.setenv("DFTPAV_SLOTS", "8")
    monkeypatch.setenv("DFTPAV_SLICE", "9")
    monkeypatch.setenv("DFTPAV_HANDOVER", "6")
    p = hiplib.default_params()
    B = 40
    scen = [sc.baseline_config(3, B=B, seed=20240 + 31 * k) for k in range(3)]
    for s in scen:
        s.apply_resolution(p)
    h = hiplib.Handle(p)
    keys = ("final_cost", "x", "status", "iters", "evals", "hist_sum", "success")
    plain = []
    for s in scen:
        bt = hiplib.Batch(h, s.layout, B)
        bt.upload(s)
        plain.append(bt.solve())
        bt.close()
    bts = []
    for s in scen:
        bt = hiplib.Batch(h, s.layout, B)
        bt.upload(s)
        bts.append(bt)
    bts[0].solve_chained(None)
    bts[1].solve_chained(bts[0])
    r0 = bts[0].results()          # complete: its stragglers were finished inside batch 1's launches
    bts[2].solve_chained(bts[1])
    r1 = bts[1].results()
    r2 = bts[2].results()          # results() finishes the pending stragglers of the last batch
    for got, ref in zip((r0, r1, r2), plain):
        for k in keys:
            assert np.array_equal(got[k], ref[k]), k
    # the chain again over the same objects (their queues and lists are reused), ending with an explicit finish
    bts[0].solve_chained(bts[2])   # bts[2] is no longer pending: degrades to an unchained start of the chain
    bts[1].solve_chained(bts[0])
    bts[1].finish()
    for got, ref in zip((bts[0].results(), bts[1].results()), plain[:2]):
        for k in keys:
            assert np.array_equal(got[k], ref[k]), k
    # a reader on a pending batch finishes it by itself
    bts[2].solve_chained(None)
    co, dts = bts[2].coeffs()
    assert np.array_equal(bts[2].results()["x"], plain[2]["x"]) and np.isfinite(co).all()
    # and the oracle agrees with what came through the chain
    ro = oracle.solve_batch(p, scen[1].subset(np.arange(6)), order=1)
    assert np.array_equal(r1["final_cost"][:6], ro["final_cost"]) and np.array_equal(r1["x"][:6], ro["x"])
    # incompatible partner (another size): prev is finished first, then a plain chained start
    s4 = sc.baseline_config(3, B=24, seed=5)
    s4.apply_resolution(p)
    b4 = hiplib.Batch(h, s4.layout, 24)
    b4.upload(s4)
    bts[0].solve_chained(None)
    b4.solve_chained(bts[0])
    assert np.array_equal(bts[0].results()["final_cost"], plain[0]["final_cost"])
    ref4 = hiplib.Batch(h, s4.layout, 24)
    ref4.upload(s4)
    assert np.array_equal(b4.results()["x"], ref4.solve()["x"])
    for bt in bts + [b4, ref4]:
        bt.close()
    h.close()


@pytest.mark.parametrize("mode,cfg,B", [(2, 3, 12), (1, 3, 12), (2, 2, 6), (2, 5, 2), (1, 1, 4)])
def test_every_launch_shape_matches_oracle(hiplib, oracle, monkeypatch, mode, cfg, B):
    """The residency plan picks the launch shape from the batch size (0: one wide workgroup per CU with operators and
    corridor in LDS, 1: two per CU, 2: four 128-thread workgroups per CU with corridor and operators read from
    memory and the narrow form of the transposed reduction).  Small batches only ever get shape 0, so the other
    shapes are forced here and held to the same bar: bit-identical to the oracle's device order."""
    monkeypatch.setenv("DFTPAV_MODE", str(mode))
    p = hiplib.default_params()
    s = sc.baseline_config(cfg, B=B)
    s.apply_resolution(p)
    h, bt = _batch(hiplib, s, p)
    x0 = bt.x0()
    f, g = bt.eval(x0)
    r = bt.solve()
    ro = oracle.solve_batch(p, s, order=1)
    for b in range(B):
        fd, gd = oracle.OracleProblem(p, s, b, order=1).eval(x0[b])
        assert f[b] == fd and np.array_equal(g[b], gd)
    for k in ("final_cost", "x", "status", "iters", "evals"):
        assert np.array_equal(r[k], ro[k]), k
    bt.close()
    h.close()


def test_random_layouts_and_short_histories(hiplib, oracle, monkeypatch):
    """Randomly shaped problems (1-3 gear segments of 2-12 pieces, sample resolutions 3-24, with and without moving
    obstacles, every launch shape, L-BFGS memories down to 4 pairs -- shorter than a block of the two-loop
    recursion, which then runs in its plain form on both sides): every field bit-identical to the device-order
    oracle.  scripts/fuzz_parity.py is the long version of this test."""
    keys = ("final_cost", "x", "status", "iters", "evals", "hist_sum", "success")
    for c in range(24):
        rng = np.random.default_rng(7000 + c)
        M = int(rng.choice([1, 1, 2, 3]))
        pieces = [int(rng.integers(2, 13)) for _ in range(M)]
        sing = [int(rng.choice([1, -1]))]
        for _ in range(M - 1):
            sing.append(-sing[-1])
        K, Kd, B = int(rng.integers(3, 25)), int(rng.integers(3, 25)), int(rng.integers(1, 5))
        moving = c % 6 == 5 and sum(pieces) <= 12
        monkeypatch.setenv("DFTPAV_MODE", str(c % 3))
        p = hiplib.default_params()
        s = sc.make_scenario(pieces, sing, K, Kd, B, seed=9000 + c, with_moving=moving, n_obs=int(rng.integers(0, 60)))
        s.apply_resolution(p)
        if c % 4 == 0:
            p.lbfgs_mem_size = [4, 7, 8, 9, 17, 64][(c // 4) % 6]
        h, bt = _batch(hiplib, s, p)
        r = bt.solve()
        ro = oracle.solve_batch(p, s, order=1)
        for k in keys:
            assert np.array_equal(r[k], ro[k]), (c, pieces, sing, K, Kd, B, moving, p.lbfgs_mem_size, k)
        bt.close()
        h.close()


def test_reeds_shepp_shots_match_oracle(hiplib, oracle):
    """§8(f)-3, hypothesis generation: KinoAstar::computeShotTraj / is_shot_sucess (kino_astar.cpp:304-345) on the
    device -- shortest Reeds-Shepp word, its length, the sampled poses and the collision verdict -- bit for bit
    against the oracle's device-order mode, and to rounding against its libm mode."""
    rng = np.random.default_rng(3)
    n = 3000
    f = np.column_stack([rng.uniform(-20, 20, n), rng.uniform(-20, 20, n), rng.uniform(-4, 4, n)])
    t = np.column_stack([rng.uniform(-20, 20, n), rng.uniform(-20, 20, n), rng.uniform(-4, 4, n)])
    t[:50] = f[:50] + rng.normal(0, 0.05, (50, 3))  # nearly coincident poses
    t[50:60] = f[50:60]                              # coincident: zero length, one sample
    obs = np.column_stack([rng.uniform(-20, 20, 40), rng.uniform(-20, 20, 40), rng.uniform(0.5, 2.0, 40)])
    grid, origin = sc.occupancy_grid(obs, arena=60.0)
    h = hiplib.Handle(hiplib.default_params())
    h.set_grid_map(grid, sc.MAP_RESL, origin)
    for max_cur, checkl, ms in [(1.0, 0.2, 512), (0.4, 0.5, 64), (2.0, 0.05, 300)]:
        got = h.reeds_shepp_shots(f, t, max_cur=max_cur, checkl=checkl, max_samples=ms, check_collision=True)
        want = oracle.reeds_shepp_shots(f, t, max_cur=max_cur, checkl=checkl, max_samples=ms, grid=grid, resolution=sc.MAP_RESL,
                                        origin=origin, order=1)
        for k in ("length", "type", "seg", "samples", "n_samples", "collides"):
            assert np.array_equal(got[k], want[k]), (max_cur, k)
        lit = oracle.reeds_shepp_shots(f, t, max_cur=max_cur, checkl=checkl, max_samples=ms, grid=grid, resolution=sc.MAP_RESL,
                                       origin=origin, order=0)
        assert np.allclose(got["length"], lit["length"], rtol=0, atol=1e-9) and (got["type"] == lit["type"]).mean() > 0.99
        assert 0.05 < got["collides"].mean() < 0.995 and (got["collides"] == lit["collides"]).mean() > 0.98
        # and against the INDEPENDENT restatement (oracle/shot_oracle_literal.cpp: no shared header, libm, candidates
        # validated by integration): lengths to rounding, the same word except at ties, the same sampled poses
        ind = oracle.reeds_shepp_literal(f, t, max_cur=max_cur, checkl=checkl, max_samples=ms)
        rel = np.abs(got["length"] - ind["length"]) / np.maximum(1.0, ind["length"])
        assert rel.max() <= 1e-12
        same = (oracle.RS_TYPE_KINDS[got["type"]] == ind["kinds"]).all(axis=1) | (got["length"] == 0.0)
        assert same.mean() > 0.97
        assert np.abs(got["seg"][same & (got["length"] > 0)] - ind["seg"][same & (got["length"] > 0)]).max() <= 1e-9
        assert np.array_equal(got["n_samples"][same], ind["n_samples"][same])
        for i in np.where(same & (got["length"] > 0))[0][::17]:
            k = min(int(got["n_samples"][i]), ms)
            d = np.abs(got["samples"][i, :k] - ind["samples"][i, :k])
            d[:, 2] = np.abs((d[:, 2] + np.pi) % (2 * np.pi) - np.pi)
            assert d.max() <= 1e-9
    assert h.corridor_last_ms() > 0.0
    # without a collision output no map is needed; a zero-sized call is fine
    h2 = hiplib.Handle(hiplib.default_params())
    r = h2.reeds_shepp_shots(f[:5], t[:5])
    assert r["collides"] is None and np.array_equal(r["length"], got["length"][:0].tolist() or r["length"])
    assert h2.reeds_shepp_shots(np.zeros((0, 3)), np.zeros((0, 3)))["length"].shape == (0,)
    with pytest.raises(hiplib.DftpavError):
        h2.reeds_shepp_shots(f[:5], t[:5], check_collision=True)  # no map installed
    h2.close()
    # a shot is a searched path: resampling -> corridor -> solve -> validation accept it
    from dftpav_amd.pods import FrontendParams
    s0 = np.array([[-12.0, -12.0, 0.3]])
    s1 = np.array([[9.0, 6.0, 1.2]])
    empty = np.full_like(grid, 127)
    h.set_grid_map(empty, sc.MAP_RESL, origin)
    shot = h.reeds_shepp_shots(s0, s1, max_cur=0.5, checkl=0.2, max_samples=1024)
    npt = int(shot["n_samples"][0])
    path = np.zeros((1, npt + 1, 3))
    path[0, :npt] = shot["samples"][0, :npt]
    path[0, npt] = s1[0]  # the goal closes the list, kino_astar.cpp:599
    fp = FrontendParams.default(K=8, Kd=8)
    fe = h.frontend_resample(path, np.array([npt + 1], dtype=np.int32), np.array([[s0[0, 0], s0[0, 1], s0[0, 2], 0.5]]),
                             np.array([[s1[0, 0], s1[0, 1], s1[0, 2], 0.2]]), np.zeros((1, 2)), fp)
    fo = oracle.frontend_resample(path, np.array([npt + 1], dtype=np.int32), np.array([[s0[0, 0], s0[0, 1], s0[0, 2], 0.5]]),
                                  np.array([[s1[0, 0], s1[0, 1], s1[0, 2], 0.2]]), np.zeros((1, 2)), fp, order=1)
    for k in fo:
        assert np.array_equal(fe[k], fo[k]), k
    assert fe["n_seg"][0] >= 1
    h.close()


def test_golden_steps_on_device(hiplib):
    """The kernels of the steps around the solve against the committed vectors (tests/golden/steps.npz): no oracle in
    the loop, the frozen device-order outputs are the reference."""
    import os
    from dftpav_amd.pods import FrontendParams
    Z = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "steps.npz"))
    h = hiplib.Handle(hiplib.default_params())
    grid, origin = Z["grid"], tuple(Z["origin"])
    h.set_grid_map(grid, sc.MAP_RESL, origin)
    assert np.array_equal(h.corridor_rectangles(Z["cor_states"]), Z["cor_out"])
    sh = h.reeds_shepp_shots(Z["shot_from"], Z["shot_to"], max_cur=0.8, checkl=0.25, max_samples=96, check_collision=True)
    for k, v in sh.items():
        assert np.array_equal(v, Z["shot_out_" + k]), k
    fe = h.frontend_resample(Z["fe_paths"], Z["fe_len"], Z["fe_ss"], Z["fe_es"], Z["fe_ct"], FrontendParams.default(K=6, Kd=9))
    for k in fe:
        assert np.array_equal(fe[k], Z["fe_out_" + k]), k
    ri, rd = h.sample_restarts(Z["rs_inner"], Z["rs_durs"], 5, sigma=0.3, lo=0.8, hi=1.25, seed=77)
    assert np.array_equal(ri, Z["rs_out_inner"]) and np.array_equal(rd, Z["rs_out_durs"])
    h.fit_surround(Z["fit_states"])
    g = h.get_surround()
    S, ns = Z["fit_states"].shape[0], Z["fit_states"].shape[1]
    assert np.array_equal(g["durations"].reshape(S, -1), Z["fit_dur"]) and np.array_equal(g["coeffs"].reshape(S, ns - 1, 12), Z["fit_coef"])
    assert np.array_equal(g["total"], Z["fit_total"]) and np.array_equal(g["start"], Z["fit_start"])
    # CheckReplan's re-check and the GetState read-out on the committed trajectories (two gear segments, random quintic pieces; test
    # hook dftpav_debug_batch_set_coeffs): collision flag, first colliding sample, every state, every valid count -- array_equal
    import ctypes as C
    from dftpav_amd.pods import LayoutSpec
    p = hiplib.default_params()
    lay = LayoutSpec([int(v) for v in Z["traj_pn"]], [int(v) for v in Z["traj_sg"]], H=4)
    B = Z["traj_coeffs"].shape[0]
    bt = hiplib.Batch(h, lay, B)
    fn = hiplib.lib().dftpav_debug_batch_set_coeffs
    fn.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p]
    co, dts = np.ascontiguousarray(Z["traj_coeffs"], dtype=np.float64), np.ascontiguousarray(Z["traj_dt"], dtype=np.float64)
    assert fn(bt._b, co.ctypes.data_as(C.c_void_p), dts.ctypes.data_as(C.c_void_p)) == 0
    col, first = bt.validate(sample_dt=0.05, vertex_res=0.1)
    assert np.array_equal(col, Z["val_col"]) and np.array_equal(first, Z["val_first"])
    st, nv = bt.sample_states(t0=-0.1, sample_dt=0.03, n_samples=220, filter_singularity=True)
    assert np.array_equal(nv, Z["rd_valid"]) and np.array_equal(st, Z["rd_states"])
    bt.close()
    h.close()


def test_step_kernels_against_the_libm_written_vectors(hiplib):
    """The kernels of the steps around the solve against vectors computed with glibc's libm in place of the correctly rounded functions
    (tests/golden/ref_steps.npz; written in round 5 by slices of the reference's functions compiled against stand-in surroundings --
    a build that is retired, oracle/pyref.py -- and equal, on that host, to order 0 of the restatements).  What this shows is how far
    a host libm is from the correctly rounded functions on these inputs: discrete outputs identical, continuous ones to 1e-11."""
    import os
    from dftpav_amd.pods import FrontendParams
    g = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
    Z, R = np.load(os.path.join(g, "steps.npz")), np.load(os.path.join(g, "ref_steps.npz"))
    h = hiplib.Handle(hiplib.default_params())
    h.set_grid_map(Z["grid"], sc.MAP_RESL, tuple(Z["origin"]))
    H = h.corridor_rectangles(Z["cor_states"])
    assert np.abs(H - R["cor_out"]).max() < 1e-11
    fe = h.frontend_resample(Z["fe_paths"], Z["fe_len"], Z["fe_ss"], Z["fe_es"], Z["fe_ct"], FrontendParams.default(K=6, Kd=9))
    for k in ("n_seg", "singul", "piece_nums", "n_states"):
        assert np.array_equal(fe[k], R["fe_out_" + k]), k
    for k in ("piece_dt", "ini_states", "fin_states", "inner_pts", "states"):
        assert np.abs(fe[k] - R["fe_out_" + k]).max() < 1e-11, k
    h.fit_surround(Z["fit_states"])
    gs = h.get_surround()
    S, ns = Z["fit_states"].shape[0], Z["fit_states"].shape[1]
    assert np.array_equal(gs["durations"].reshape(S, -1), R["fit_dur"]) and np.array_equal(gs["total"], R["fit_total"])
    assert np.array_equal(gs["start"], R["fit_start"])
    scale = np.abs(R["fit_coef"]).max()
    assert np.abs(gs["coeffs"].reshape(S, ns - 1, 12) - R["fit_coef"]).max() < 1e-11 * scale
    h.close()


@pytest.mark.parametrize("D", [2, 4])
def test_overlapped_streams_and_hand_over_zero(hiplib, monkeypatch, D):
    """bench.py's schedule (--depth D; 4 is its default, 2 was until round 4): D handles (D HIP streams), hand-over 0 so that
    every trajectory finishes in its queue launch, batches launched in turn without waiting for the previous ones -- results
    equal the plain solves bit for bit; the marker events give the device time across the streams."""
    monkeypatch.setenv("DFTPAV_SCHED", "1")
    monkeypatch.setenv("DFTPAV_SLOTS", "8")
    monkeypatch.setenv("DFTPAV_SLICE", "7")
    p = hiplib.default_params()
    B = 36
    scen = [sc.baseline_config(3, B=B, seed=4242 + 17 * k) for k in range(D)]
    for s in scen:
        s.apply_resolution(p)
    hs = [hiplib.Handle(p) for _ in range(D)]
    bts = []
    for h, s in zip(hs, scen):
        bt = hiplib.Batch(h, s.layout, B)
        bt.upload(s)
        bts.append(bt)
    keys = ("final_cost", "x", "status", "iters", "evals", "hist_sum", "success")
    plain = [bt.solve() for bt in bts]
    for bt in bts:
        bt.set_hand_over(0)
    hs[0].mark(0)
    nb = 3 * D
    for k in range(nb):
        bts[k % D].solve_async()
        if k >= D - 1:
            bts[(k - D + 1) % D].sync()
    for k in range(nb - D + 1, nb):
        bts[k % D].sync()
    hs[(nb - 1) % D].mark(1)
    assert hs[(nb - 1) % D].elapsed_since(hs[0], 0, 1) > 0.0
    for bt, ref in zip(bts, plain):
        got = bt.results()
        for k in keys:
            assert np.array_equal(got[k], ref[k]), k
    # back to the default end game, and a hand-over larger than the batch is clamped
    bts[0].set_hand_over(-1)
    assert np.array_equal(bts[0].solve()["x"], plain[0]["x"])
    bts[0].set_hand_over(10 ** 6)
    assert np.array_equal(bts[0].solve()["x"], plain[0]["x"])
    for bt in bts:
        bt.close()
    for h in hs:
        h.close()


def test_plan_cycle_equals_the_separate_stages(hiplib, oracle):
    """dftpav_plan_cycle (upload -> rectangles from the map -> solve -> collision re-check -> read-out, one enqueue on the
    handle's stream) against the same stages called one by one, and the solve against the oracle chain; a second cycle on
    the same batch with other inputs reuses the work buffers."""
    p = hiplib.default_params()
    s = sc.baseline_config(3, B=8)
    s.apply_resolution(p)
    st = s.meta["states"]
    c = (0.5 * (st[..., 0].min() + st[..., 0].max()), 0.5 * (st[..., 1].min() + st[..., 1].max()))
    obs = np.vstack([s.meta["obstacles"], np.column_stack([st[0, 200:203, 0], st[0, 200:203, 1], [0.6, 0.6, 0.6]])])  # some on the path
    grid, origin = sc.occupancy_grid(obs, arena=120.0, centre=c)
    order = np.argsort(s.meta["hyp_of"], kind="stable")  # trajectory = hypothesis * n_restarts + restart
    nr = s.B // st.shape[0]
    s3 = s.subset(order)
    h = hiplib.Handle(p)
    h.set_grid_map(grid, sc.MAP_RESL, origin)
    n_rd = 400
    # the stages one by one
    b1 = hiplib.Batch(h, s.layout, s.B)
    b1.upload(s3, with_corridor=False)
    b1.corridor_from_states(st, n_restarts=nr)
    r1 = b1.solve()
    col1, first1 = b1.validate(sample_dt=0.05, vertex_res=0.1)
    rd1, nv1 = b1.sample_states(t0=0.0, sample_dt=0.05, n_samples=n_rd)
    # one enqueue
    b2 = hiplib.Batch(h, s.layout, s.B)
    b2.plan_cycle(s3, st, n_restarts=nr, check_dt=0.05, vertex_res=0.1, t0=0.0, state_dt=0.05, n_samples=n_rd)
    r2 = b2.plan_cycle_fetch()
    for k in ("x", "final_cost", "status", "success", "iters"):
        assert np.array_equal(r2[k], r1[k]), k
    assert np.array_equal(r2["collision"], col1) and np.array_equal(r2["first_sample"], first1)
    assert np.array_equal(r2["states"], rd1) and np.array_equal(r2["n_valid"], nv1)
    assert r2["success"].all()
    # against the oracle chain: rectangles -> solve
    Ho = oracle.corridor_rectangles(grid, sc.MAP_RESL, origin, st.reshape(-1, 3), order=1).reshape(st.shape[0], st.shape[1], 4, 4)
    so = s.subset(order)
    so.corridor = np.ascontiguousarray(np.repeat(Ho, nr, axis=0))
    ro = oracle.solve_batch(p, so, nthreads=4, order=1)
    assert np.array_equal(r2["final_cost"], ro["final_cost"]) and np.array_equal(r2["x"], ro["x"])
    # a second cycle: other restarts of the same hypotheses (reversed order inside each hypothesis)
    perm = np.concatenate([np.arange(i * nr, (i + 1) * nr)[::-1] for i in range(st.shape[0])])
    s4 = s3.subset(perm)
    b2.plan_cycle(s4, st, n_restarts=nr, check_dt=0.05, vertex_res=0.1, t0=0.0, state_dt=0.05, n_samples=n_rd)
    r4 = b2.plan_cycle_fetch()
    assert np.array_equal(r4["x"], r2["x"][perm]) and np.array_equal(r4["collision"], col1[perm])
    assert np.array_equal(r4["states"], rd1[perm])
    with pytest.raises(hiplib.DftpavError):
        b2.plan_cycle_fetch()  # nothing in flight
    b1.close()
    b2.close()
    h.close()
