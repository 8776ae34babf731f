import numpy as np

from dftpav_amd import scenarios as sc


def test_generator_is_seeded():
    a = sc.baseline_config(3, B=8)
    b = sc.baseline_config(3, B=8)
    for f in ("ini_states", "fin_states", "inner_pts", "init_Ts", "corridor"):
        assert np.array_equal(getattr(a, f), getattr(b, f))
    c = sc.baseline_config(3, B=8, seed=7)
    assert not np.array_equal(a.inner_pts, c.inner_pts)


def test_shapes_follow_survey_table():
    # config -> (M, N, K/Kd, n, Npts)
    for cfg, M, Ntot, n, npts in ((1, 1, 8, 15, 168), (2, 2, 16, 33, 528), (3, 1, 16, 31, 528), (5, 1, 32, 63, 2080)):
        s = sc.baseline_config(cfg, B=2)
        assert s.layout.M == M and s.layout.n_pieces == Ntot and s.layout.n_vars == n and s.n_points == npts
        assert s.corridor.shape == (2, npts, 4, 4)


def test_corridor_contains_the_inflated_footprint_of_the_nominal_path():
    """Rectangles grown by getRectangleConst's rule must contain the vehicle at the state they were
    grown from (traj_manager.cpp:1296-1465)."""
    s = sc.baseline_config(3, B=1)
    H = s.corridor[0]  # restart 0 == nominal path
    n = H[:, :, :2]
    p = H[:, :, 2:]
    assert np.allclose(np.linalg.norm(n, axis=2), 1.0)
    # opposite planes are parallel, adjacent ones orthogonal
    assert np.allclose((n[:, 0] * n[:, 2]).sum(1), -1.0) and np.allclose((n[:, 0] * n[:, 1]).sum(1), 0.0, atol=1e-12)
    # rectangle extents at least the raw car + one 0.3 m step wherever obstacles allow; never below the raw car
    width = -((p[:, 2] - p[:, 0]) * n[:, 0]).sum(1)
    length = -((p[:, 3] - p[:, 1]) * n[:, 1]).sum(1)
    assert (width >= sc.VEH_W - 1e-9).all() and (length >= sc.VEH_L - 1e-9).all()


def test_moving_obstacles_follow_their_circles():
    sur = sc.moving_obstacles()
    assert sur.S == 4 and sur.piece_offsets[-1] == 120
    for u, (cx, cy, vel, rad, yaw0) in enumerate(sc.DYNAMIC_OBS_YAML):
        cm = sur.coeffs[sur.piece_offsets[u]:sur.piece_offsets[u + 1]].reshape(-1, 6, 2)  # [piece][t^5..t^0][xy]
        for p in (0, 7, 29):
            for t in (0.0, 0.5):
                pos = sum(cm[p, k] * t ** (5 - k) for k in range(6))
                assert abs(np.hypot(pos[0] - cx, pos[1] - cy) - rad) < 0.05
