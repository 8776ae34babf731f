"""CPU checks of the trajectory read-out oracle (oracle/states_oracle.cpp): Trajectory::GetState
(poly_traj_utils.hpp:378-406) over a time grid, played back as TrajPlannerServer::PublishData does
(traj_server_ros.cpp:244-259) with FilterSingularityState (:335-356).  The reference holds no expected values for
this step; these are property checks -- the pin against the reference's own code is
tests/test_ref_pin.py::test_states_oracle_is_bit_equal_to_GetState_and_the_servers_playback."""
import numpy as np


def _arc(R, speed, dt, N, x0=0.0, y0=0.0):
    """[N][6][2]: quintic Taylor pieces of a circle of radius R driven at `speed` (exact enough for dt << R / speed)"""
    w = speed / R
    co = np.zeros((N, 6, 2))
    for p in range(N):
        a = w * dt * p
        for k in range(6):
            # d^k/dt^k of (R sin(wt), R (1 - cos(wt))) at t = a / w, over k!
            f = w ** k / float(np.prod(np.arange(1, k + 1))) if k else 1.0
            co[p, k, 0] = R * np.sin(a + k * np.pi / 2) * f
            co[p, k, 1] = -R * np.cos(a + k * np.pi / 2) * f
        co[p, 0, 0] += x0
        co[p, 0, 1] += y0 + R
    return co


def test_circle_states(oracle):
    R, v = 8.0, 2.0
    co = _arc(R, v, 0.5, 12)[None]
    st, nv = oracle.sample_states(co, [[0.5]], [12], [1], sample_dt=0.01, n_samples=700)
    assert nv[0] == 600  # t = 6.0 is the end_time of the only segment: `end_time <= t` moves past it
    s = st[0, :600]
    assert np.allclose(s[:, 0], 0.01 * np.arange(600), rtol=0, atol=1e-15)
    assert np.allclose(s[:, 5], v, atol=1e-6) and np.allclose(s[:, 4], 1.0 / R, atol=1e-6)
    assert np.allclose(s[:, 6], 0.0, atol=1e-5)
    assert np.allclose(s[:, 7], np.arctan(2.85 / R), atol=1e-6)
    assert np.allclose(s[:, 3], (v / R) * s[:, 0], atol=1e-6)
    assert np.allclose(s[:, 1], R * np.sin(s[:, 3]), atol=1e-6)
    assert (st[0, 600:] == 0.0).all()
    # reversing along the same curve: velocity negative, heading flipped by pi, curvature sign flips with v^3
    sr, _ = oracle.sample_states(co, [[0.5]], [12], [-1], sample_dt=0.01, n_samples=600)
    assert np.allclose(sr[0, :, 5], -v, atol=1e-6) and np.allclose(sr[0, :, 4], -1.0 / R, atol=1e-6)
    d = np.angle(np.exp(1j * (sr[0, :, 3] - s[:, 3])))
    assert np.allclose(np.abs(d), np.pi, atol=1e-6)


def test_orders_agree(oracle):
    co = _arc(5.0, 1.5, 1.0, 6)[None]
    a, na = oracle.sample_states(co, [[1.0]], [6], [1], sample_dt=0.013, n_samples=400, order=0)
    b, nb = oracle.sample_states(co, [[1.0]], [6], [1], sample_dt=0.013, n_samples=400, order=1)
    assert na[0] == nb[0] and np.allclose(a, b, rtol=1e-13, atol=1e-13)


def test_segments_are_chained_and_clamped(oracle):
    """two gear segments: forward 3 pieces of 1 s, then reverse 2 pieces of 1.5 s; the second starts at the first's
    end_time (traj_container.hpp:58-73) and is read with its own local time."""
    co = np.zeros((1, 5, 6, 2))
    for p in range(3):
        co[0, p, 0] = [1.0 * p, 0.0]
        co[0, p, 1] = [1.0, 0.0]
    for p in range(2):
        co[0, 3 + p, 0] = [3.0 - 0.5 * 1.5 * p, 0.0]
        co[0, 3 + p, 1] = [-0.5, 0.0]
    st, nv = oracle.sample_states(co, [[1.0, 1.5]], [3, 2], [1, -1], sample_dt=0.25, n_samples=30, filter_singularity=False)
    assert nv[0] == 24  # 6 s in total; t = 6.0 is past the end
    s = st[0]
    assert np.allclose(s[:12, 1], 0.25 * np.arange(12)) and np.allclose(s[:12, 5], 1.0)
    assert np.allclose(s[12:24, 1], 3.0 - 0.5 * (0.25 * np.arange(12, 24) - 3.0)) and np.allclose(s[12:24, 5], -0.5)
    assert np.allclose(s[:24, 3], 0.0)  # reversing along -x keeps the heading at 0
    # negative start: before the trajectory begins the first piece is extrapolated (locatePieceIdx with t < 0)
    st2, _ = oracle.sample_states(co, [[1.0, 1.5]], [3, 2], [1, -1], t0=-0.5, sample_dt=0.25, n_samples=4)
    assert np.allclose(st2[0, :, 1], [-0.5, -0.25, 0.0, 0.25])


def test_singularity_filter(oracle):
    """a trajectory that stops and leaves in another direction: without the filter the heading of the slow samples
    jumps, with it the heading is held at the last published value while |v| < 0.1 and the jump exceeds the
    steering-rate bound (traj_server_ros.cpp:340-353)."""
    co = np.zeros((1, 2, 6, 2))
    # piece 0 (1 s): decelerates along +x to a stop: x = t - t^2 / 2, v = 1 - t
    co[0, 0, 1] = [1.0, 0.0]
    co[0, 0, 2] = [-0.5, 0.0]
    # piece 1 (1 s): accelerates along +y from rest: y = t^2 / 2
    co[0, 1, 0] = [0.5, 0.0]
    co[0, 1, 2] = [0.0, 0.5]
    raw, _ = oracle.sample_states(co, [[1.0]], [2], [1], sample_dt=0.01, n_samples=200, filter_singularity=False)
    fil, _ = oracle.sample_states(co, [[1.0]], [2], [1], sample_dt=0.01, n_samples=200, filter_singularity=True)
    slow = np.abs(raw[0, :, 5]) < 0.1
    assert slow.sum() > 10
    assert np.array_equal(raw[0, ~slow], fil[0, ~slow])
    assert np.array_equal(raw[0, :, [0, 1, 2, 4, 5, 6, 7]], fil[0, :, [0, 1, 2, 4, 5, 6, 7]])  # only the heading is filtered
    k = np.nonzero(slow)[0]
    jumped = k[np.abs(raw[0, k, 3] - raw[0, k - 1, 3]) > 1.0]
    assert len(jumped) >= 1  # the raw heading flips from 0 to pi / 2 at the stop
    after = k[k >= jumped[0]]
    assert np.allclose(fil[0, after, 3], 0.0, atol=1e-9)  # held at the heading before the stop
    fast_after = np.nonzero(~slow & (np.arange(200) > after[-1]))[0]
    assert np.allclose(fil[0, fast_after, 3], np.pi / 2, atol=1e-9)
