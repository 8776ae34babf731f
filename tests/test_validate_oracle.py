"""CPU checks of the trajectory-validation oracle (oracle/validate_oracle.cpp): the collision
re-check of TrajPlannerServer::CheckReplan (traj_server_ros.cpp:385-397).  No golden vectors exist in the
reference; these are property checks -- the pin against the reference's own code is
tests/test_ref_pin.py::test_validate_oracle_is_bit_equal_to_CheckReplan."""
import numpy as np

from dftpav_amd import scenarios as sc

RES = sc.MAP_RESL


def _straight(x0, y0, yaw, speed, dt, N):
    """coefficients [N][6][2] of a constant-velocity trajectory in N pieces of duration dt"""
    co = np.zeros((N, 6, 2))
    for p in range(N):
        co[p, 0] = [x0 + speed * np.cos(yaw) * dt * p, y0 + speed * np.sin(yaw) * dt * p]
        co[p, 1] = [speed * np.cos(yaw), speed * np.sin(yaw)]
    return co


def test_free_and_blocked_straight_line(oracle):
    grid = np.full((400, 400), 127, dtype=np.uint8)
    origin = (-60.0, -60.0)
    co = _straight(-20.0, 0.0, 0.0, 2.0, 1.0, 10)[None]  # 20 m along +x in 10 s
    dts = np.array([[1.0]])
    col, first = oracle.validate_trajectories(grid, RES, origin, co, dts, [10], [1])
    assert col[0] == 0 and first[0] == -1
    # a wall across the path at x = 0
    ix = int(round((0.0 - origin[0]) / RES))
    grid[:, ix] = 80
    col, first = oracle.validate_trajectories(grid, RES, origin, co, dts, [10], [1])
    assert col[0] == 1
    # the front of the vehicle (rear axle + d_cr + L/2 = +3.455 m) reaches the wall's cell after (20 - 3.455 - 0.15) / 2 s
    t_hit = first[0] * 0.05
    assert abs(t_hit - (20.0 - 3.455 - 0.15) / 2.0) < 0.1
    # reversing along the same line (singul = -1, velocity pointing backwards): the heading stays 0
    co_r = _straight(20.0, 0.0, np.pi, 2.0, 1.0, 10)[None]
    col_r, first_r = oracle.validate_trajectories(grid, RES, origin, co_r, dts, [10], [-1])
    assert col_r[0] == 1
    t_r = first_r[0] * 0.05  # the rear (axle + d_cr - L/2 = -1.425 m) arrives first
    assert abs(t_r - (20.0 - 1.425 - 0.15) / 2.0) < 0.1


def test_sample_count_follows_the_running_sum(oracle):
    """t += 0.05 as a running sum: a 1 s trajectory gets the samples the reference's loop visits (0.05 * 20 summed is
    not exactly 1.0)."""
    grid = np.full((50, 50), 127, dtype=np.uint8)
    grid[:] = 80  # everything occupied: the first sample collides
    co = _straight(1.0, 1.0, 0.3, 1.0, 0.5, 2)[None]
    col, first = oracle.validate_trajectories(grid, RES, (0.0, 0.0), co, np.array([[0.5]]), [2], [1])
    assert col[0] == 1 and first[0] == 0
    t, n = 0.0, 0
    while t < 0.5 + 0.5:
        t += 0.05
        n += 1
    assert n in (20, 21)


def test_orders_agree_on_a_solved_batch(oracle, hiplib_or_none=None):
    from oracle import pyoracle as po
    p = po.default_params()
    s = sc.baseline_config(3, B=8)
    s.apply_resolution(p)
    r = po.solve_batch(p, s, nthreads=4, order=1)
    co, dts = [], []
    for b in range(s.B):
        pr = po.OracleProblem(p, s, b, order=1)
        pr.eval(r["x"][b])
        a, d = pr.coeffs()
        co.append(a)
        dts.append(d)
    st = s.meta["states"]
    c = (0.5 * (st[..., 0].min() + st[..., 0].max()), 0.5 * (st[..., 1].min() + st[..., 1].max()))
    grid, origin = sc.occupancy_grid(s.meta["obstacles"], arena=140.0, centre=c)
    c0 = oracle.validate_trajectories(grid, RES, origin, np.array(co), np.array(dts), s.layout.piece_nums, s.layout.singuls, order=0)
    c1 = oracle.validate_trajectories(grid, RES, origin, np.array(co), np.array(dts), s.layout.piece_nums, s.layout.singuls, order=1)
    assert np.array_equal(c0[0], c1[0]) and np.array_equal(c0[1], c1[1])
    assert c1[0].sum() == 0  # the optimised trajectories stay inside their corridors
