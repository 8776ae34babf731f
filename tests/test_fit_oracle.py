"""CPU checks of the moving-obstacle fit oracle (oracle/fit_oracle.cpp): ConverSurroundTrajFromPoints
(traj_manager.cpp:743-789).  No golden vectors in the reference; these are the MINCO invariants -- the pin against the
reference's own code is tests/test_ref_pin.py::test_fit_oracle_is_bit_equal_to_ConverSurroundTrajFromPoints."""
import numpy as np

from dftpav_amd import scenarios as sc


def _eval(coef12, t, order=0):
    """value / derivative of one piece stored [x5,y5, ..., x0,y0] at local time t"""
    c = coef12.reshape(6, 2)[::-1]  # row k = coefficient of t^k
    k = np.arange(6)
    if order == 0:
        return (c * (t ** k)[:, None]).sum(0)
    if order == 1:
        return (c[1:] * (k[1:] * t ** (k[1:] - 1))[:, None]).sum(0)
    return (c[2:] * (k[2:] * (k[2:] - 1) * t ** (k[2:] - 2))[:, None]).sum(0)


def test_fit_interpolates_states_and_matches_boundary_conditions(oracle):
    st = sc.predicted_states()
    f = oracle.fit_surround(st, order=0)
    S, n = st.shape[0], st.shape[1]
    assert f["coeffs"].shape == (S, n - 1, 12)
    assert np.allclose(f["durations"], 1.0) and np.allclose(f["total"], 30.0) and np.allclose(f["start"], 0.0)
    for o in range(S):
        for p in range(n - 1):
            assert np.allclose(_eval(f["coeffs"][o, p], 0.0), st[o, p, :2], atol=1e-9)
            assert np.allclose(_eval(f["coeffs"][o, p], 1.0), st[o, p + 1, :2], atol=1e-8)
        # boundary velocity / acceleration = state_to_flat_output (traj_manager.cpp:139-158)
        for k, (p, t) in ((0, (0, 0.0)), (n - 1, (n - 2, 1.0))):
            ang, vel, cur = st[o, k, 2], st[o, k, 3], st[o, k, 5]
            v = np.array([np.cos(ang) * vel, np.sin(ang) * vel])
            a = np.array([-np.sin(ang), np.cos(ang)]) * cur * vel ** 2
            assert np.allclose(_eval(f["coeffs"][o, p], t, 1), v, atol=1e-8)
            assert np.allclose(_eval(f["coeffs"][o, p], t, 2), a, atol=1e-7)
        # C3 at the junctions (minimum jerk): value, velocity, acceleration continuous
        for p in range(n - 2):
            for order in (0, 1, 2):
                assert np.allclose(_eval(f["coeffs"][o, p], 1.0, order), _eval(f["coeffs"][o, p + 1], 0.0, order), atol=1e-6)


def test_orders_agree_and_match_the_numpy_generator(oracle):
    st = sc.predicted_states()
    f0, f1 = oracle.fit_surround(st, order=0), oracle.fit_surround(st, order=1)
    assert np.abs(f0["coeffs"] - f1["coeffs"]).max() < 1e-9
    ref = sc.moving_obstacles()  # independent NumPy fit of the same cars (dense solve)
    assert np.abs(ref.coeffs.reshape(f0["coeffs"].shape) - f0["coeffs"]).max() < 1e-7


def test_zero_velocity_state_uses_the_small_speed_rule(oracle):
    st = sc.predicted_states(pre_time=4.0)
    st[:, 0, 3] = 0.0  # vel == 0 -> 1e-5 (traj_manager.cpp:150-152)
    f = oracle.fit_surround(st, order=0)
    v0 = _eval(f["coeffs"][0, 0], 0.0, 1)
    assert np.isclose(np.hypot(*v0), 1e-5, rtol=1e-6)
