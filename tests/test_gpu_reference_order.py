"""The reference-order device mode (dftpav_batch_set_order(DFTPAV_ORDER_REFERENCE), dftpav_amd/csrc/solver_ref.hip) against
the reference ITSELF: oracle/_ref is the reference's own traj_optimizer.cpp / poly_traj_utils.hpp / lbfgs.hpp compiled
unmodified (oracle/Makefile.ref; the .so travels to the GPU box), the literal oracle its statement-by-statement restatement
(bit-equal to it, tests/test_ref_pin.py).

Bar: BIT-EQUAL -- every evaluation (cost, gradient) and every whole solve (final x, final cost, status, iterations,
evaluations, flag_success) -- on the single-segment static configurations: BASELINE configs[0] (default arena), the tests'
cfg 1 (8 pieces forward) and cfg 3 (configs[2]: 16 pieces, 32 points per piece), 32 trajectories each.  north_star's
"final cost within 1e-5 relative of CPU" is met with 0.0.
"""
import numpy as np
import pytest

from dftpav_amd import scenarios as sc

pytestmark = pytest.mark.gpu


def _ref():
    from oracle import pyref
    return pyref if pyref.available() else None


def _scenario(oracle, hiplib, name, B):
    p = hiplib.default_params()
    if name == "default_arena":
        from test_default_map import _default_map_scenario
        p.traj_resolution, p.des_traj_resolution = 16, 32
        return p, _default_map_scenario(oracle, p, 16, 32, B)
    s = sc.baseline_config({"cfg1": 1, "cfg3": 3}[name], B=B)
    s.apply_resolution(p)
    return p, s


def _batch(hiplib, s, p):
    h = hiplib.Handle(p)
    bt = hiplib.Batch(h, s.layout, s.B)
    bt.upload(s)
    bt.set_order(hiplib.ORDER_REFERENCE)
    return h, bt


@pytest.mark.parametrize("name,B", [("cfg1", 6), ("cfg3", 6), ("default_arena", 4)])
def test_every_evaluation_has_the_reference_bits(hiplib, oracle, name, B):
    """costFunctionCallback (traj_optimizer.cpp:206-350) at x0, at perturbed points (many active penalty terms) and at
    the solution: cost and gradient bit for bit."""
    p, s = _scenario(oracle, hiplib, name, B)
    h, bt = _batch(hiplib, s, p)
    x0 = bt.x0()
    rng = np.random.default_rng(5)
    pyref = _ref()
    lits = [oracle.OracleProblem(p, s, b, order=0) for b in range(B)]
    refs = [pyref.RefProblem(p, s, b) for b in range(min(B, 2))] if pyref else []
    xs = [x0, x0 + rng.normal(0, 0.05, x0.shape), x0 + rng.normal(0, 0.7, x0.shape)]
    xs.append(np.stack([oracle.solve_batch(p, s, nthreads=4, order=0)["x"][b] for b in range(B)]))
    for x in xs:
        f, g = bt.eval(x)
        for b in range(B):
            fl, gl = lits[b].eval(x[b])
            assert f[b] == fl, (name, b, f[b], fl)
            assert np.array_equal(g[b], gl), (name, b, np.abs(g[b] - gl).max())
        for b, r in enumerate(refs):
            fr, gr = r.eval(x[b])
            assert f[b] == fr and np.array_equal(g[b], gr)
    bt.close()
    h.close()


@pytest.mark.parametrize("name,B", [("cfg1", 32), ("cfg3", 32), ("default_arena", 32)])
def test_whole_solves_are_bit_equal_to_the_reference(hiplib, oracle, name, B):
    """OptimizeTrajectory (traj_optimizer.cpp:7-202) of the reference build, trajectory by trajectory, against one
    reference-order solve of the batch on the device."""
    p, s = _scenario(oracle, hiplib, name, B)
    h, bt = _batch(hiplib, s, p)
    r = bt.solve()
    pyref = _ref()
    lit = oracle.solve_batch(p, s, nthreads=8, order=0)
    for k in ("final_cost", "x", "status", "iters", "evals", "success", "hist_sum"):
        assert np.array_equal(r[k], lit[k]), (name, k)
    if pyref:
        for b in range(B):
            rr = pyref.RefProblem(p, s, b).optimize()
            assert rr["final_cost"] == r["final_cost"][b] and np.array_equal(rr["x"], r["x"][b]), (name, b)
            assert rr["status"] == r["status"][b] and rr["iters"] == r["iters"][b] and rr["evals"] == r["evals"][b]
            assert bool(rr["ok"]) == bool(r["success"][b])
    assert r["success"].all()
    # the coefficients handed back for the solution (getMinJerkOptPtr()[i].getTraj(), traj_manager.cpp:618-625)
    c, dt = bt.coeffs()
    for b in range(min(B, 3)):
        lp = oracle.OracleProblem(p, s, b, order=0)
        lp.eval(r["x"][b])
        co, dto = lp.coeffs()
        assert np.array_equal(c[b], co) and np.array_equal(dt[b], dto)
    # and the device-order kernels on the same batch object still give their own (different, equally valid) answer
    bt.set_order(hiplib.ORDER_DEVICE)
    rd = bt.solve()
    ro = oracle.solve_batch(p, s, nthreads=8, order=1)
    assert np.array_equal(rd["final_cost"], ro["final_cost"]) and np.array_equal(rd["x"], ro["x"])
    bt.close()
    h.close()


def test_moving_obstacles_in_reference_order(hiplib, oracle):
    """With moving obstacles the reference calls libm's exp (40 times) and log (9 times) per (constraint point, obstacle) pair and
    pow(|v|, 3) in Piece::getRdot -- host-dependent bits, as with cos / sin.  The device runs the reference's program
    (dynamicObsGradCostP, traj_optimizer.cpp:1311-1684, statement by statement) with the CORRECTLY ROUNDED exp / log / x^3
    (cr_trig.h); oracle order 2 is that program on the CPU (through binary128).  Bar: bit-equal to order 2 -- evaluations
    and whole solves -- on BASELINE configs[4] (32 pieces x 65 points, four moving cars) and a smaller layout with the
    obstacles starting after t_now.  (Against libm's own bits no solve could agree: millions of exponentials per solve at
    one misrounding in a thousand.)"""
    keys = ("final_cost", "x", "status", "iters", "evals", "hist_sum", "success")
    for case in range(2):
        p = hiplib.default_params()
        if case == 0:
            s = sc.baseline_config(5, B=4)
        else:
            s = sc.make_scenario([24], [1], 10, 16, 3, seed=80, with_moving=True, n_obs=20, start_centre=(-38.0, 5.0))
            s.surround.start_time[:] = [2.0, 0.0, 5.5, 0.5]
            s.t_now = 0.75
        s.apply_resolution(p)
        h = hiplib.Handle(p)
        h.set_surround(s.surround)
        bt = hiplib.Batch(h, s.layout, s.B)
        bt.upload(s)
        bt.set_order(hiplib.ORDER_REFERENCE)
        x0 = bt.x0()
        rng = np.random.default_rng(9)
        active = 0
        for x in (x0, x0 + rng.normal(0, 0.3, x0.shape)):
            f, g = bt.eval(x)
            for b in range(s.B):
                o2 = oracle.OracleProblem(p, s, b, order=2)
                f2, g2 = o2.eval(x[b])
                assert f[b] == f2 and np.array_equal(g[b], g2), (case, b, f[b], f2, np.abs(g[b] - g2).max())
                active += o2.cost_terms()[3] > 0.0
                f0, g0 = oracle.OracleProblem(p, s, b, order=0).eval(x[b])      # libm: the same to rounding
                assert abs(f[b] - f0) <= 1e-12 * abs(f0)
        assert active > 0, case  # the moving-obstacle term is active somewhere, or this test checks nothing
        r = bt.solve()
        want = oracle.solve_batch(p, s, nthreads=4, order=2)
        for k in keys:
            assert np.array_equal(r[k], want[k]), (case, k)
        assert r["success"].all()
        bt.close()
        h.close()


def _live_case(layout, B, seed):
    """gear shifts AND moving obstacles in one call -- what the reference's only live caller passes (traj_manager.cpp:604-610):
    the four cars of dynamicObs.yaml around a multi-segment layout"""
    pieces, sing = layout
    s = sc.make_scenario(pieces, sing, 12, 16, B, seed=seed, with_moving=True, n_obs=25, start_centre=(-38.0, 5.0))
    s.surround.start_time[:] = [0.5, 0.0, 1.5, 0.25]     # obstacle clocks that differ from the ego's (OPT:1367-1369)
    s.t_now = 0.6
    return s


@pytest.mark.parametrize("layout,seed", [(([7, 6], [1, -1]), 81), (([5, 4, 6], [1, -1, 1]), 82)])
def test_gear_shifts_with_moving_obstacles_in_reference_order(hiplib, oracle, layout, seed):
    """The reference's live case: dynamicObsGradCostP inside a multi-segment solve.  This is where two quirks of the
    reference matter: trajtimes[i] is the duration of segment i - 1, not the time since the start (traj_optimizer.cpp:230-234,
    291, 1367-1369), and a moving-obstacle term of segment `trajid` adds `trajid` more addends to that segment's gdT
    (:1674-1676).  Bar: 64 trajectories, every evaluation and every whole solve bit-equal to oracle order 2 (the reference's
    program with correctly rounded cos / sin / exp / log / pow); every evaluation within 1e-12 of the reference BUILD
    (oracle/_ref, libm's bits) where it travelled; the moving-obstacle term active."""
    B = 64
    p = hiplib.default_params()
    s = _live_case(layout, B, seed)
    s.apply_resolution(p)
    h = hiplib.Handle(p)
    h.set_surround(s.surround)
    bt = hiplib.Batch(h, s.layout, s.B)
    bt.upload(s)
    bt.set_order(hiplib.ORDER_REFERENCE)
    pyref = _ref()
    x0 = bt.x0()
    rng = np.random.default_rng(seed)
    active = later_segment_active = 0
    for x in (x0, x0 + rng.normal(0, 0.25, x0.shape)):
        f, g = bt.eval(x)
        for b in range(B):
            o2 = oracle.OracleProblem(p, s, b, order=2)
            f2, g2 = o2.eval(x[b])
            assert f[b] == f2 and np.array_equal(g[b], g2), (layout, b, f[b], f2, np.abs(g[b] - g2).max())
            active += o2.cost_terms()[3] > 0.0
            if pyref and b < 16:
                fr, gr = pyref.RefProblem(p, s, b).eval(x[b])
                assert abs(f[b] - fr) <= 1e-12 * abs(fr), (layout, b, f[b], fr)
                assert np.abs(g[b] - gr).max() <= 1e-12 * max(1.0, np.abs(gr).max())
    assert active > B // 4, active   # the moving-obstacle term is active on a good part of the batch
    r = bt.solve()
    want = oracle.solve_batch(p, s, nthreads=8, order=2)
    for k in ("final_cost", "x", "status", "iters", "evals", "hist_sum", "success"):
        assert np.array_equal(r[k], want[k]), (layout, k)
    bt.close()
    h.close()


def test_configs4_in_reference_order_at_batch_64(hiplib, oracle):
    """BASELINE configs[4] (32 pieces x 65 points, four moving cars), 64 trajectories: whole solves bit-equal to order 2"""
    p = hiplib.default_params()
    s = sc.baseline_config(5, B=64)
    s.apply_resolution(p)
    h = hiplib.Handle(p)
    h.set_surround(s.surround)
    bt = hiplib.Batch(h, s.layout, s.B)
    bt.upload(s)
    bt.set_order(hiplib.ORDER_REFERENCE)
    r = bt.solve()
    want = oracle.solve_batch(p, s, nthreads=8, order=2)
    for k in ("final_cost", "x", "status", "iters", "evals", "hist_sum", "success"):
        assert np.array_equal(r[k], want[k]), k
    assert r["success"].all()
    bt.close()
    h.close()


def test_more_than_eight_moving_obstacles(hiplib, oracle):
    """5 H + S + 4 > 32 terms per constraint point: the 64-bit term mask (12 obstacles: the four cars three times, shifted)"""
    from dftpav_amd.pods import SurroundSet
    p = hiplib.default_params()
    s = sc.make_scenario([10], [1], 8, 8, 8, seed=90, with_moving=True, n_obs=10, start_centre=(-38.0, 5.0))
    sur, reps = s.surround, 3
    npieces = int(sur.piece_offsets[-1])
    offs = np.concatenate([[0]] + [sur.piece_offsets[1:] + k * npieces for k in range(reps)])
    coeffs = np.concatenate([sur.coeffs] * reps).copy()
    for k in range(1, reps):    # shift the copies sideways: the constant terms of x / y (Piece::coeffMat column 5: entries 10, 11)
        coeffs[k * npieces:(k + 1) * npieces, 10] += 0.7 * k
        coeffs[k * npieces:(k + 1) * npieces, 11] -= 0.4 * k
    s.surround = SurroundSet(offs, np.concatenate([sur.durations] * reps), coeffs, np.concatenate([sur.total_duration] * reps),
                             np.concatenate([sur.start_time + 0.3 * k for k in range(reps)]))
    assert s.surround.S == 12
    s.apply_resolution(p)
    h = hiplib.Handle(p)
    h.set_surround(s.surround)
    bt = hiplib.Batch(h, s.layout, s.B)
    bt.upload(s)
    bt.set_order(hiplib.ORDER_REFERENCE)
    x = bt.x0() + np.random.default_rng(1).normal(0, 0.2, bt.x0().shape)
    f, g = bt.eval(x)
    for b in range(s.B):
        f2, g2 = oracle.OracleProblem(p, s, b, order=2).eval(x[b])
        assert f[b] == f2 and np.array_equal(g[b], g2), b
    r = bt.solve()
    want = oracle.solve_batch(p, s, nthreads=8, order=2)
    for k in ("final_cost", "x", "status", "iters", "evals", "hist_sum", "success"):
        assert np.array_equal(r[k], want[k]), k
    bt.close()
    h.close()


@pytest.mark.parametrize("case", ["cfg3", "cfg2", "cfg5", "live"])
def test_wave_shape_and_ring_are_bit_identical(hiplib, oracle, monkeypatch, case):
    """The throughput shape of the reference order -- one WAVE per trajectory, the waves of a workgroup sharing the sweep tables,
    trajectories popped from the batch's ring and suspended after a slice of iterations (solver_ref.hip) -- forced on small
    batches with 2 persistent workgroups and slices of 3 and 17 iterations: every field of every solve equal to the latency
    shape's and to the oracle's (no sum depends on the shape; suspending and resuming moves no bit)."""
    p = hiplib.default_params()
    if case == "cfg3":
        s, order = sc.baseline_config(3, B=40), 0
    elif case == "cfg2":
        s, order = sc.baseline_config(2, B=24), 2
    elif case == "cfg5":
        s, order = sc.baseline_config(5, B=10), 2
    else:
        s, order = _live_case(([5, 4, 6], [1, -1, 1]), 20, 83), 2
    s.apply_resolution(p)
    want = oracle.solve_batch(p, s, nthreads=8, order=order)
    keys = ("final_cost", "x", "status", "iters", "evals", "hist_sum", "success")
    x = None
    for slice_ in (3, 17):
        monkeypatch.setenv("DFTPAV_REF_SHAPE", "wave")
        monkeypatch.setenv("DFTPAV_REF_SLOTS", "2")
        monkeypatch.setenv("DFTPAV_REF_SLICE", str(slice_))
        h = hiplib.Handle(p)
        if s.surround is not None:
            h.set_surround(s.surround)
        bt = hiplib.Batch(h, s.layout, s.B)
        bt.upload(s)
        bt.set_order(hiplib.ORDER_REFERENCE)
        if x is None:
            x = bt.x0() + np.random.default_rng(4).normal(0, 0.2, bt.x0().shape)
        f, g = bt.eval(x)                      # evaluations in the WAVE shape (static assignment of waves)
        for b in range(0, s.B, 5):
            fo, go = oracle.OracleProblem(p, s, b, order=order).eval(x[b])
            assert f[b] == fo and np.array_equal(g[b], go), (case, b)
        for rep in range(2):                   # twice: the ring is reset by every solve
            r = bt.solve()
            for k in keys:
                assert np.array_equal(r[k], want[k]), (case, slice_, rep, k)
        c, dt = bt.coeffs()
        lp = oracle.OracleProblem(p, s, s.B - 1, order=order)
        lp.eval(r["x"][s.B - 1])
        co, dto = lp.coeffs()
        assert np.array_equal(c[s.B - 1], co) and np.array_equal(dt[s.B - 1], dto)
        bt.close()
        h.close()
    monkeypatch.delenv("DFTPAV_REF_SHAPE")
    monkeypatch.delenv("DFTPAV_REF_SLOTS")
    monkeypatch.delenv("DFTPAV_REF_SLICE")


@pytest.mark.parametrize("cfg,B,slots,slice_", [(3, 40, 0, 0), (3, 41, 2, 5), (3, 23, 1, 64), (1, 37, 3, 2), (1, 9, 0, 0)])
def test_quad_shape_and_its_ring_are_bit_identical(hiplib, oracle, monkeypatch, cfg, B, slots, slice_):
    """The QUAD shape of the reference order (solver_ref4.hip): FOUR trajectories per wave, one per row of 16 lanes, a piece per lane,
    the band system / c / gdC / adjoint in registers, 32-step row chains for the dot products.  Forced on small batches -- cfg 3
    (16 pieces x 33 points, n = 31: BASELINE configs[2] / [3]) and cfg 1 (8 pieces, K = 16 / Kd = 32: the end pieces are longer than
    the inner ones, half the lanes of a row idle) -- once with every trajectory in a row of its own from the start, then with 1-3
    persistent waves whose rows pop from the batch's ring and suspend after a slice of 2 / 5 / 64 evaluations (batch sizes that leave
    rows empty): every evaluation and every field of every solve equal to the restatement's, the coefficient read-out (left to
    solver_ref.hip) too.  No sum depends on the shape; suspending and resuming moves no bit."""
    p = hiplib.default_params()
    s = sc.baseline_config(cfg, B=B)
    s.apply_resolution(p)
    want = oracle.solve_batch(p, s, nthreads=8, order=0)
    keys = ("final_cost", "x", "status", "iters", "evals", "hist_sum", "success")
    monkeypatch.setenv("DFTPAV_REF_SHAPE", "quad")
    monkeypatch.setenv("DFTPAV_REF_QUAD_WAVES", "1")
    if slots:
        monkeypatch.setenv("DFTPAV_REF_SLOTS", str(slots))
        monkeypatch.setenv("DFTPAV_REF_SLICE", str(slice_))
    h = hiplib.Handle(p)
    bt = hiplib.Batch(h, s.layout, s.B)
    bt.upload(s)
    bt.set_order(hiplib.ORDER_REFERENCE)
    rng = np.random.default_rng(4)
    for x in (bt.x0(), bt.x0() + rng.normal(0, 0.2, bt.x0().shape), bt.x0() + rng.normal(0, 0.7, bt.x0().shape)):
        f, g = bt.eval(x)
        for b in range(0, s.B, 4):
            fo, go = oracle.OracleProblem(p, s, b, order=0).eval(x[b])
            assert f[b] == fo and np.array_equal(g[b], go), (cfg, b)
    for rep in range(3):                   # the ring is reset by every solve
        # a batch that has the device to itself (the default) leaves its last B / 2 trajectories to a launch of the WAVE shape that
        # resumes them from their records; with other batches announced behind it (hand-over 0) the QUAD launch finishes them all
        bt.set_hand_over(0 if rep == 2 else -1)
        r = bt.solve()
        for k in keys:
            assert np.array_equal(r[k], want[k]), (cfg, slots, rep, k)
    bt.set_hand_over(-1)
    c, dt = bt.coeffs()
    lp = oracle.OracleProblem(p, s, s.B - 1, order=0)
    lp.eval(r["x"][s.B - 1])
    co, dto = lp.coeffs()
    assert np.array_equal(c[s.B - 1], co) and np.array_equal(dt[s.B - 1], dto)
    # a new upload (other half-planes) reaches the QUAD shape's own copy of the corridor
    s2 = sc.baseline_config(cfg, B=B, seed=977)
    s2.apply_resolution(p)
    bt.upload(s2)
    r2 = bt.solve()
    want2 = oracle.solve_batch(p, s2, nthreads=8, order=0)
    for k in keys:
        assert np.array_equal(r2[k], want2[k]), (cfg, "second upload", k)
    bt.close()
    h.close()


def test_quad_shape_is_what_a_full_batch_takes_and_equals_the_other_shapes(hiplib, oracle, monkeypatch):
    """BASELINE configs[3] at 4096 on one GPU: the plan picks the QUAD shape on its own (dftpav_debug_reference_plan), its scheduled
    launch (512 persistent waves for 4096 trajectories, slices of 64 evaluations) gives every trajectory the bits the WAVE shape gives
    it, and 24 sampled trajectories the restatement's."""
    p = hiplib.default_params()
    s = sc.baseline_config(3, B=4096)
    s.apply_resolution(p)
    h = hiplib.Handle(p)
    bt = hiplib.Batch(h, s.layout, s.B)
    bt.upload(s)
    bt.set_order(hiplib.ORDER_REFERENCE)
    rq = bt.solve()
    bt.set_order(hiplib.ORDER_DEVICE)
    monkeypatch.setenv("DFTPAV_REF_SHAPE", "wave")
    bt.set_order(hiplib.ORDER_REFERENCE)
    monkeypatch.delenv("DFTPAV_REF_SHAPE")
    rw = bt.solve()
    keys = ("final_cost", "x", "status", "iters", "evals", "hist_sum", "success")
    for k in keys:
        assert np.array_equal(rq[k], rw[k]), k
    pick = (np.arange(24) * 170 + 11) % s.B
    want = oracle.solve_batch(p, s.subset(pick), nthreads=8, order=0)
    for k in keys:
        assert np.array_equal(rq[k][pick], want[k]), k
    bt.close()
    h.close()


@pytest.mark.parametrize("case,B,slots,slice_", [("cfg2", 24, 0, 0), ("cfg2", 37, 3, 2), ("5-4-6", 21, 2, 5), ("3-2-4-3", 18, 1, 64), ("2-2", 9, 0, 0),
                                                 ("8-8-mem40", 13, 1, 7), ("6-5-5", 10, 2, 9), ("4-4-4-4", 7, 1, 33)])
def test_quad_shape_with_gear_shifts_is_bit_identical(hiplib, oracle, monkeypatch, case, B, slots, slice_):
    """The QUAD shape for several gear segments (solver_ref4m.hip): the pieces of up to four segments side by side on a row's sixteen
    lanes, the junction position / angle variables and their gradients, vectors of up to 48 variables in three registers per lane.
    BASELINE configs[1] (8 + 8 pieces, forward + reverse, n = 33), layouts of three and four segments of unequal length and two with 35 / 37
    variables (the third register of a vector holds more than one element: the 48-step chain), forced on
    small batches: every evaluation and every field of every solve equal to the restatement with correctly rounded cos / sin (oracle
    order 2), from rows of their own and through the ring with slices of 2 / 5 / 7 / 64 evaluations, with and without the hand-over of
    the last trajectories to the WAVE shape; the plan says which kernel ran."""
    p = hiplib.default_params()
    if case == "cfg2":
        s = sc.baseline_config(2, B=B)
    else:
        pieces = [int(t) for t in case.split("-")[:-1]] if case.endswith("mem40") else [int(t) for t in case.split("-")]
        sing = [1 if i % 2 == 0 else -1 for i in range(len(pieces))]
        s = sc.make_scenario(pieces, sing, 9, 14, B, seed=4242 + len(pieces), n_obs=30)
        if case.endswith("mem40"):
            p.lbfgs_mem_size = 40
    s.apply_resolution(p)
    want = oracle.solve_batch(p, s, nthreads=8, order=2)
    keys = ("final_cost", "x", "status", "iters", "evals", "hist_sum", "success")
    monkeypatch.setenv("DFTPAV_REF_SHAPE", "quad")
    if slots:
        monkeypatch.setenv("DFTPAV_REF_SLOTS", str(slots))
        monkeypatch.setenv("DFTPAV_REF_SLICE", str(slice_))
    h = hiplib.Handle(p)
    bt = hiplib.Batch(h, s.layout, s.B)
    bt.upload(s)
    bt.set_order(hiplib.ORDER_REFERENCE)
    import ctypes as C
    fn = hiplib.lib().dftpav_debug_reference_plan
    fn.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.POINTER(C.c_longlong)]
    out = (C.c_longlong * 8)()
    assert fn(C.byref(s.layout.c_struct()), C.byref(p), 0, s.B, 256, out) == 0
    assert int(out[1]) == 5, "the plan did not pick the QUAD shape for several segments"
    rng = np.random.default_rng(4)
    for x in (bt.x0(), bt.x0() + rng.normal(0, 0.2, bt.x0().shape), bt.x0() + rng.normal(0, 0.7, bt.x0().shape)):
        f, g = bt.eval(x)
        for b in range(0, s.B, 3):
            fo, go = oracle.OracleProblem(p, s, b, order=2).eval(x[b])
            assert f[b] == fo and np.array_equal(g[b], go), (case, b)
    for rep in range(3):
        bt.set_hand_over(0 if rep == 2 else -1)
        r = bt.solve()
        for k in keys:
            assert np.array_equal(r[k], want[k]), (case, slots, rep, k)
    bt.set_hand_over(-1)
    c, dt = bt.coeffs()
    lp = oracle.OracleProblem(p, s, s.B - 1, order=2)
    lp.eval(r["x"][s.B - 1])
    co, dto = lp.coeffs()
    assert np.array_equal(c[s.B - 1], co) and np.array_equal(dt[s.B - 1], dto)
    bt.close()
    h.close()


@pytest.mark.parametrize("shape", ["team", "wave", "quad"])
def test_recursion_with_true_divisions_gives_the_same_bits(hiplib, oracle, monkeypatch, shape):
    """The two-loop recursion divides by the stored y . s of a pair through its stored reciprocal (Markstein's correctly rounded
    quotient); a pair whose y . s lies beyond 2^+-500 switches the trajectory to true divisions for good (solver_ref.hip:
    div_by_rcp, iSLOWDIV).  That fallback forced from the first iteration on (DFTPAV_REF_EXACT_DIV=1): the same bits -- which
    is both the test of the fallback and the statement that the reciprocal route IS the division on real solves."""
    p = hiplib.default_params()
    s = sc.baseline_config(3, B=12)
    s.apply_resolution(p)
    want = oracle.solve_batch(p, s, nthreads=8, order=0)
    monkeypatch.setenv("DFTPAV_REF_EXACT_DIV", "1")
    monkeypatch.setenv("DFTPAV_REF_SHAPE", shape)
    if shape in ("wave", "quad"):
        monkeypatch.setenv("DFTPAV_REF_SLOTS", "1")
        monkeypatch.setenv("DFTPAV_REF_SLICE", "11")
    h, bt = _batch(hiplib, s, p)
    r = bt.solve()
    for k in ("final_cost", "x", "status", "iters", "evals", "hist_sum", "success"):
        assert np.array_equal(r[k], want[k]), (shape, k)
    bt.close()
    h.close()


def test_gear_shifts_in_reference_order(hiplib, oracle):
    """With gear shifts the reference calls libm's cos / sin of the junction angles in every evaluation: bits of the HOST (glibc's
    are not correctly rounded, and dispatched by CPU model).  The device runs the reference's program with the CORRECTLY ROUNDED
    cos / sin (cr_trig.h); oracle order 2 is that same program on the CPU (its cos / sin through binary128).  Bar: bit-equal
    to order 2 -- evaluations and whole solves -- on BASELINE configs[1] (8 + 8 pieces, forward + reverse) and on random
    layouts with up to four segments; against the reference build itself the solves agree exactly where this host's libm
    happened to round every angle correctly, which is reported, not asserted."""
    pyref = _ref()
    keys = ("final_cost", "x", "status", "iters", "evals", "hist_sum", "success")
    cases = [("cfg2", None)] + [("rand", c) for c in range(8)]
    same_as_build = total_build = 0
    for name, c in cases:
        p = hiplib.default_params()
        if name == "cfg2":
            s = sc.baseline_config(2, B=32)
        else:
            rng = np.random.default_rng(15000 + c)
            M = int(rng.integers(2, 5))
            pieces = [int(rng.integers(2, 9)) for _ in range(M)]
            while 2 * (sum(pieces) - M) + M + 3 * (M - 1) > 64:
                pieces[int(np.argmax(pieces))] -= 1
            sing = [int(rng.choice([1, -1]))]
            for _ in range(M - 1):
                sing.append(-sing[-1])
            s = sc.make_scenario(pieces, sing, int(rng.integers(4, 20)), int(rng.integers(4, 20)), int(rng.integers(1, 4)), seed=16000 + c,
                                 n_obs=int(rng.integers(0, 50)))
            if c % 3 == 0:
                p.lbfgs_mem_size = [6, 40, 11][c // 3]
        s.apply_resolution(p)
        h, bt = _batch(hiplib, s, p)
        x0 = bt.x0()
        rng2 = np.random.default_rng(3)
        for x in (x0, x0 + rng2.normal(0, 0.3, x0.shape)):
            f, g = bt.eval(x)
            for b in range(min(s.B, 4)):
                f2, g2 = oracle.OracleProblem(p, s, b, order=2).eval(x[b])
                assert f[b] == f2 and np.array_equal(g[b], g2), (name, c, b)
                f0, g0 = oracle.OracleProblem(p, s, b, order=0).eval(x[b])      # libm's cos / sin: the same to rounding
                assert abs(f[b] - f0) <= 1e-13 * abs(f0) and np.abs(g[b] - g0).max() <= 1e-11 * max(1.0, np.abs(g0).max())
        r = bt.solve()
        o2 = oracle.solve_batch(p, s, nthreads=8, order=2)
        for k in keys:
            assert np.array_equal(r[k], o2[k]), (name, c, k)
        assert r["success"].all()
        cf, dt = bt.coeffs()
        lp = oracle.OracleProblem(p, s, 0, order=2)
        lp.eval(r["x"][0])
        co, dto = lp.coeffs()
        assert np.array_equal(cf[0], co) and np.array_equal(dt[0], dto)
        if pyref and name == "cfg2":
            for b in range(s.B):
                rr = pyref.RefProblem(p, s, b).optimize()
                total_build += 1
                same_as_build += int(rr["final_cost"] == r["final_cost"][b] and np.array_equal(rr["x"], r["x"][b]) and rr["iters"] == r["iters"][b])
        bt.close()
        h.close()
    print("gear-shift solves of configs[1] bit-equal to the reference build on this host (its libm rounded every junction angle "
          "correctly): %d of %d" % (same_as_build, total_build))


def test_far_trial_points_of_a_line_search(hiplib, oracle):
    """A line search of lbfgs.hpp:276-390 starts an iteration at step 1 along the new direction; late in a hard solve that
    point can lie 1e10 away (found by scripts/fuzz_reference_order.py: 8 + 10 + 5 pieces, iteration 1056 -- the literal program
    returns a cost of 8e52 there and backs off 58 times).  The junction angle is 1e10 there too: cos / sin need the reduction of
    Payne and Hanek (cr_trig.h reduce_large).  Evaluations at such points -- finite, and bit-equal to the literal program."""
    p = hiplib.default_params()
    s = sc.make_scenario([8, 10, 5], [1, -1, 1], 21, 8, 2, seed=23150, n_obs=30)
    s.apply_resolution(p)
    h, bt = _batch(hiplib, s, p)
    x0 = bt.x0()
    n = x0.shape[1]
    rng = np.random.default_rng(5)
    for scale in (1.0e3, 1.0e6, 3.0e10, 1.0e15):
        x = x0.copy()
        x[:, :40] += rng.normal(0, min(scale, 1.0e10), (2, 40))   # inner points
        x[:, 40:43] = rng.uniform(-60, 60, (2, 3))                  # virtual times: durations of 1e-3 .. 1e3 s
        x[:, 43:47] += rng.normal(0, min(scale, 1.0e10), (2, 4))   # junction positions
        x[:, 47:] = rng.normal(0, scale, (2, 2))                    # junction angles
        f, g = bt.eval(x)
        for b in range(2):
            f2, g2 = oracle.OracleProblem(p, s, b, order=2).eval(x[b])
            assert np.isfinite(f2) and f[b] == f2 and np.array_equal(g[b], g2), (scale, b, f[b], f2)
    r = bt.solve()
    o2 = oracle.solve_batch(p, s, nthreads=2, order=2)
    for k in ("final_cost", "x", "status", "iters", "evals", "hist_sum", "success"):
        assert np.array_equal(r[k], o2[k]), k
    bt.close()
    h.close()


def test_random_layouts_in_reference_order(hiplib, oracle):
    """Randomly shaped single-segment problems -- 2 to 32 pieces (n = 3 .. 63: the three widths of the sequential sums), sample
    resolutions 3-24, forward and reverse gears, 0-60 obstacles, L-BFGS memories from 3 pairs (the ring wraps after three
    iterations) to 300, other `past` / `delta` settings: every field of every solve bit-equal to the reference's restatement,
    and to the reference build itself where it is there."""
    pyref = _ref()
    keys = ("final_cost", "x", "status", "iters", "evals", "hist_sum", "success")
    for c in range(20):
        rng = np.random.default_rng(12000 + c)
        N = int([2, 3, 5, 9, 12, 17, 24, 32, 8, 16][c % 10])
        K, Kd, B = int(rng.integers(3, 25)), int(rng.integers(3, 25)), int(rng.integers(1, 4))
        p = hiplib.default_params()
        s = sc.make_scenario([N], [int(rng.choice([1, -1]))], K, Kd, B, seed=13000 + c, n_obs=int(rng.integers(0, 60)))
        s.apply_resolution(p)
        if c % 2 == 0:
            p.lbfgs_mem_size = [3, 8, 5, 300, 17, 64, 9, 33, 4, 128][c // 2]
        if c % 5 == 1:
            p.lbfgs_past, p.lbfgs_delta = int(rng.integers(1, 7)), float(10.0 ** rng.uniform(-6, -3))
        if c % 7 == 3:
            s.help_eps = 1e-3       # the second reciprocal of the curvature term (traj_optimizer.cpp:556-558) differs from the first
        h, bt = _batch(hiplib, s, p)
        x = bt.x0() + rng.normal(0, 0.2, bt.x0().shape)
        f, g = bt.eval(x)
        for b in range(B):
            fl, gl = oracle.OracleProblem(p, s, b, order=0).eval(x[b])
            assert f[b] == fl and np.array_equal(g[b], gl), (c, N, K, Kd, b)
        r = bt.solve()
        lit = oracle.solve_batch(p, s, nthreads=2, order=0)
        for k in keys:
            assert np.array_equal(r[k], lit[k]), (c, N, K, Kd, B, p.lbfgs_mem_size, p.lbfgs_past, k)
        if pyref and c % 4 == 0:
            rr = pyref.RefProblem(p, s, 0).optimize()
            assert rr["final_cost"] == r["final_cost"][0] and np.array_equal(rr["x"], r["x"][0]) and rr["iters"] == r["iters"][0]
        bt.close()
        h.close()


def test_trace_is_a_device_order_facility(hiplib):
    """dftpav_batch_trace records nothing in the reference order: asking for it there, or choosing the reference order while a
    trace is on, is refused with DFTPAV_E_UNSUPPORTED (it used to return empty traces); the batch stays usable."""
    p = hiplib.default_params()
    s = sc.baseline_config(1, B=2)
    s.apply_resolution(p)
    h, bt = _batch(hiplib, s, p)
    with pytest.raises(hiplib.DftpavError) as e:
        bt.trace(0, 64)
    assert e.value.code == hiplib.E_UNSUPPORTED
    assert bt.solve()["success"].all()
    bt.set_order(hiplib.ORDER_DEVICE)
    bt.trace(0, 4096)
    with pytest.raises(hiplib.DftpavError) as e:
        bt.set_order(hiplib.ORDER_REFERENCE)
    assert e.value.code == hiplib.E_UNSUPPORTED
    r = bt.solve()
    assert len(bt.get_trace()["f"]) == r["evals"][0]
    bt.trace(0, 0)
    bt.set_order(hiplib.ORDER_REFERENCE)
    assert bt.solve()["success"].all()
    bt.close()
    h.close()


def _extra_planes(base, H):
    """the scenario `base` (H = 4 rectangles) with H half-planes per point: the extra ones are copies of the first ones pulled 0.2 m
    inwards (so that they are the active ones), their normals un-normalised"""
    cor = np.zeros((base.B, base.n_points, H, 4))
    cor[:, :, :4] = base.corridor
    for k in range(4, H):
        cor[:, :, k] = base.corridor[:, :, k % 4]
        cor[:, :, k, 2:] -= (0.2 + 0.05 * (k - 4)) * base.corridor[:, :, k % 4, :2]
        cor[:, :, k, :2] *= 3.0
    lay = type(base.layout)(base.layout.piece_nums, base.layout.singuls, H=H)
    return sc.Scenario("planes_%d" % H, lay, base.K, base.Kd, base.B, base.ini_states, base.fin_states, base.inner_pts, base.init_Ts, np.ascontiguousarray(cor))


@pytest.mark.parametrize("case", ["n79", "n77_two_segments", "H6", "H11", "n79_H7"])
def test_beyond_a_wave_of_variables_and_five_half_planes(hiplib, oracle, case):
    """What the reference accepts and the fast kernels do not -- more decision variables than a wave has lanes (N_i = max(round(dur / 1.0),
    2) is unbounded, traj_manager.cpp:543; lbfgs.hpp:512-513 sizes its history for any n) and more than four or five half-planes per point
    (H is the column count of hPoly, traj_optimizer.cpp:592-622) -- runs in the generic TEAM kernel (solver_ref.hip: lbfgs_advance_generic,
    twelve plane slots): every evaluation and every whole solve bit-equal to the restatement."""
    p = hiplib.default_params()
    order = 0
    if case == "n79":
        s = sc.make_scenario([40], [1], 8, 8, 3, seed=5)                 # n = 79
    elif case == "n77_two_segments":
        s, order = sc.make_scenario([20, 18], [1, -1], 6, 9, 3, seed=6), 2  # n = 2 * 36 + 2 + 2 + 1 = 77, a gear shift: cos / sin
    elif case == "H6":
        s = _extra_planes(sc.baseline_config(1, B=3), 6)
    elif case == "H11":
        s = _extra_planes(sc.baseline_config(1, B=2), 11)                # 5 H + 4 = 59 terms per point: the 64-bit masks
    else:
        s = _extra_planes(sc.make_scenario([40], [1], 8, 8, 2, seed=7), 7)
    s.apply_resolution(p)
    h = hiplib.Handle(p)
    bt = hiplib.Batch(h, s.layout, s.B)
    bt.upload(s)
    bt.set_order(hiplib.ORDER_REFERENCE)
    x0 = bt.x0()
    for x in (x0, x0 + np.random.default_rng(3).normal(0, 0.3, x0.shape)):
        f, g = bt.eval(x)
        for b in range(s.B):
            fo, go = oracle.OracleProblem(p, s, b, order=order).eval(x[b])
            assert f[b] == fo and np.array_equal(g[b], go), (case, b)
    r = bt.solve()
    want = oracle.solve_batch(p, s, nthreads=4, order=order)
    for k in ("final_cost", "x", "status", "iters", "evals", "hist_sum", "success"):
        assert np.array_equal(r[k], want[k]), (case, k)
    c, dt = bt.coeffs()
    lp = oracle.OracleProblem(p, s, 0, order=order)
    lp.eval(r["x"][0])
    co, dto = lp.coeffs()
    assert np.array_equal(c[0], co) and np.array_equal(dt[0], dto)
    bt.close()
    h.close()


@pytest.mark.parametrize("H", [3, 5])
def test_quad_shape_with_other_than_four_half_planes(hiplib, oracle, monkeypatch, H):
    """the QUAD shape's generic instantiation (H != 4: three half-planes -- a corridor open on one side -- and five): bit-equal to the
    restatement, as the FAST one (H = 4, help_eps = 0) is"""
    p = hiplib.default_params()
    base = sc.baseline_config(3, B=13)
    if H == 5:
        s = _extra_planes(base, 5)
    else:
        lay = type(base.layout)(base.layout.piece_nums, base.layout.singuls, H=3)
        s = sc.Scenario("planes_3", lay, base.K, base.Kd, base.B, base.ini_states, base.fin_states, base.inner_pts, base.init_Ts,
                        np.ascontiguousarray(base.corridor[:, :, :3]))
    s.apply_resolution(p)
    monkeypatch.setenv("DFTPAV_REF_SHAPE", "quad")
    monkeypatch.setenv("DFTPAV_REF_SLOTS", "2")
    monkeypatch.setenv("DFTPAV_REF_SLICE", "9")
    h = hiplib.Handle(p)
    bt = hiplib.Batch(h, s.layout, s.B)
    bt.upload(s)
    bt.set_order(hiplib.ORDER_REFERENCE)
    x = bt.x0() + np.random.default_rng(8).normal(0, 0.3, bt.x0().shape)
    f, g = bt.eval(x)
    for b in range(0, s.B, 3):
        fo, go = oracle.OracleProblem(p, s, b, order=0).eval(x[b])
        assert f[b] == fo and np.array_equal(g[b], go), (H, b)
    r = bt.solve()
    want = oracle.solve_batch(p, s, nthreads=8, order=0)
    for k in ("final_cost", "x", "status", "iters", "evals", "hist_sum", "success"):
        assert np.array_equal(r[k], want[k]), (H, k)
    bt.close()
    h.close()


def test_what_the_reference_order_still_refuses_is_refused_cleanly(hiplib):
    """more than 12 half-planes per point or more than 64 terms per point (the term mask), more than 256 variables: DFTPAV_E_UNSUPPORTED,
    the batch stays usable in device order"""
    p = hiplib.default_params()
    s = _extra_planes(sc.baseline_config(1, B=1), 13)
    s.apply_resolution(p)
    h = hiplib.Handle(p)
    bt = hiplib.Batch(h, s.layout, 1)
    bt.upload(s)
    with pytest.raises(hiplib.DftpavError) as e:
        bt.set_order(hiplib.ORDER_REFERENCE)
    assert e.value.code == hiplib.E_UNSUPPORTED
    assert bt.solve()["success"].all()
    bt.close()
    h.close()


def test_class_mirror_in_reference_order_returns_the_reference_answer(hiplib, oracle):
    """PolyTrajOptimizer(reference_order=True).OptimizeTrajectory with the containers traj_manager.cpp:608-610 passes: the
    reference's own answer for a single-segment problem (bit-equal to the reference build / its restatement), and the
    throughput order -- reported as such -- where the layout has a gear shift."""
    from dftpav_amd.optimizer import PolyTrajOptimizer

    def containers(s):
        lay = s.layout
        ini = [s.ini_states[0, i].reshape(3, 2).T for i in range(lay.M)]
        fin = [s.fin_states[0, i].reshape(3, 2).T for i in range(lay.M)]
        inner, off, polys, pt = [], 0, [], 0
        for N in lay.piece_nums:
            inner.append(s.inner_pts[0, off:off + 2 * (N - 1)].reshape(N - 1, 2).T)
            off += 2 * (N - 1)
            cnt = (N - 2) * (s.K + 1) + 2 * (s.Kd + 1)
            polys.append([s.corridor[0, pt + k].T for k in range(cnt)])
            pt += cnt
        return ini, fin, inner, polys

    p = hiplib.default_params()
    s = sc.baseline_config(1, B=1)
    s.apply_resolution(p)
    opt = PolyTrajOptimizer(reference_order=True)
    opt.setParam(p)
    ini, fin, inner, polys = containers(s)
    assert opt.OptimizeTrajectory(ini, fin, inner, s.init_Ts[0], polys, list(s.layout.singuls), 0.0, 0.0) is True
    assert opt.last["order"] == hiplib.ORDER_REFERENCE
    lit = oracle.solve_batch(p, s, nthreads=1, order=0)
    assert opt.last["final_cost"][0] == lit["final_cost"][0] and np.array_equal(opt.last["x"][0], lit["x"][0])
    assert opt.last["iters"][0] == lit["iters"][0] and opt.last["evals"][0] == lit["evals"][0]
    pyref = _ref()
    if pyref:
        rr = pyref.RefProblem(p, s, 0).optimize()
        assert rr["final_cost"] == opt.last["final_cost"][0] and np.array_equal(rr["x"], opt.last["x"][0])
    s2 = sc.baseline_config(2, B=1)
    p2 = hiplib.default_params()
    s2.apply_resolution(p2)
    opt2 = PolyTrajOptimizer(reference_order=True)
    opt2.setParam(p2)
    ini, fin, inner, polys = containers(s2)
    assert opt2.OptimizeTrajectory(ini, fin, inner, s2.init_Ts[0], polys, list(s2.layout.singuls), 0.0, 0.0) is True
    assert opt2.last["order"] == hiplib.ORDER_REFERENCE   # gear shift: the reference's program with correctly rounded cos / sin
    o2 = oracle.solve_batch(p2, s2, nthreads=1, order=2)
    assert opt2.last["final_cost"][0] == o2["final_cost"][0] and np.array_equal(opt2.last["x"][0], o2["x"][0])
    s5 = sc.baseline_config(5, B=1)
    p5 = hiplib.default_params()
    s5.apply_resolution(p5)
    opt5 = PolyTrajOptimizer(reference_order=True)
    opt5.setParam(p5)
    opt5.setSurroundTrajs(s5.surround)
    ini, fin, inner, polys = containers(s5)
    assert opt5.OptimizeTrajectory(ini, fin, inner, s5.init_Ts[0], polys, list(s5.layout.singuls), 0.0, 0.0) is True
    assert opt5.last["order"] == hiplib.ORDER_REFERENCE   # moving obstacles: the reference's program with correctly rounded exp / log / pow
    o5 = oracle.solve_batch(p5, s5, nthreads=1, order=2)
    assert opt5.last["final_cost"][0] == o5["final_cost"][0] and np.array_equal(opt5.last["x"][0], o5["x"][0])


def _ref_cr():
    from oracle import pyref
    return pyref if pyref.cr_available() else None


def _same_as_ref(r, b, rr):
    return bool(rr["final_cost"] == r["final_cost"][b] and np.array_equal(rr["x"], r["x"][b]) and rr["iters"] == r["iters"][b] and
                rr["evals"] == r["evals"][b] and rr["status"] == r["status"][b] and rr["ok"] == bool(r["success"][b]))


def test_reference_order_equals_the_reference_on_a_correctly_rounded_libm(hiplib, oracle):
    """The bit-level pin of the configurations whose loop calls libm to the reference's OWN CODE.  oracle/_ref/libdftpav_ref_cr.so
    is traj_optimizer.cpp compiled unmodified (the same two objects as oracle/_ref) linked against a correctly rounded exp /
    log / pow / sin / cos (oracle/cr_libm.c): the reference's program on the one libm every host agrees on -- the libm the
    device implements (cr_trig.h).  Whole solves of the device in reference order must be bit-equal to it: BASELINE configs[1]
    (gear shift, 32 trajectories), the reference's live case (gear shifts with moving obstacles, two layouts, 8 of 64
    trajectories each), BASELINE configs[4] (one trajectory: binary128 exp / log take seconds per solve on the host)."""
    pyref = _ref_cr()
    if pyref is None:
        pytest.skip("oracle/_ref/libdftpav_ref_cr.so did not travel")
    cases = [("configs[1]", lambda: sc.baseline_config(2, B=32), range(32)),
             ("live [7, 6]", lambda: _live_case(([7, 6], [1, -1]), 64, 81), range(0, 64, 8)),
             ("live [5, 4, 6]", lambda: _live_case(([5, 4, 6], [1, -1, 1]), 64, 82), range(0, 64, 8)),
             ("configs[4]", lambda: sc.baseline_config(5, B=2), range(1))]
    for name, mk, pick in cases:
        p = hiplib.default_params()
        s = mk()
        s.apply_resolution(p)
        h = hiplib.Handle(p)
        h.set_surround(s.surround)
        bt = hiplib.Batch(h, s.layout, s.B)
        bt.upload(s)
        bt.set_order(hiplib.ORDER_REFERENCE)
        r = bt.solve()
        for b in pick:
            rr = pyref.RefProblem(p, s, b, cr=True).optimize()
            assert _same_as_ref(r, b, rr), (name, b, r["final_cost"][b], rr["final_cost"], r["iters"][b], rr["iters"])
        cf, dt = bt.coeffs()           # getMinJerkOptPtr()'s coefficients of the solution (traj_manager.cpp:618-625)
        rp = pyref.RefProblem(p, s, 0, cr=True)
        rp.eval(r["x"][0])             # costFunctionCallback of the reference at the solution leaves them in its container
        co, dto = rp.coeffs()
        assert np.array_equal(cf[0], co) and np.array_equal(dt[0], dto), name
        bt.close()
        h.close()


@pytest.mark.parametrize("name,cfg,B", [("cfg5_b1024", 5, 1024), ("cfg2_b4096", 2, 4096)])
def test_reference_order_at_the_baseline_batch_sizes(hiplib, name, cfg, B):
    """BASELINE configs[4] at ITS batch (1024 trajectories among the four moving cars, 63 variables: the 64-term kernel) and
    configs[1]'s gear shift at 4096 (33 variables: the 40-term kernel), whole batches in reference order; 64 sampled trajectories
    of each against the reference's program with correctly rounded libm calls -- expected results computed where the cores are
    (tests/golden/make_golden_ref_order_batches.py: oracle order 2; the first 8 of each were also solved by the reference's OWN
    objects on the correctly rounded libm, oracle/_ref/libdftpav_ref_cr.so, and agreed bit for bit)."""
    import os
    Z = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "ref_order_batches.npz"))
    p = hiplib.default_params()
    s = sc.baseline_config(cfg, B=B, seed=int(Z["seed"]))
    s.apply_resolution(p)
    h = hiplib.Handle(p)
    h.set_surround(s.surround)
    bt = hiplib.Batch(h, s.layout, B)
    bt.upload(s)
    bt.set_order(hiplib.ORDER_REFERENCE)
    r = bt.solve()
    pk = Z[name + "_pick"]
    assert len(pk) == 64 and int(Z["n_checked_against_the_reference_objects_on_a_correctly_rounded_libm"]) >= 8
    for k in ("final_cost", "x", "status", "iters", "evals"):
        assert np.array_equal(r[k][pk], Z[name + "_" + k]), (name, k)
    assert r["success"].mean() > 0.95
    bt.close()
    h.close()
