"""CPU checks of the front-end resampling oracle (oracle/frontend_oracle.cpp): getKinoNode from
SampleTraj on (kino_astar.cpp:606-795) and the resampling of RunMINCOParking (traj_manager.cpp:531-568).  The
reference holds no golden vectors for it; these are property checks -- the pin against the reference's own code is
tests/test_ref_pin.py::test_frontend_oracle_is_bit_equal_to_getKinoNode_and_RunMINCOParking."""
import numpy as np
import pytest

from dftpav_amd import scenarios as sc
from dftpav_amd.pods import FrontendParams


@pytest.mark.parametrize("gears", [(1,), (1, -1), (-1, 1, -1)])
def test_segments_pieces_and_states(oracle, gears):
    P, pl, ss, es, ct = sc.searched_paths(5, seed=3, gears=gears)
    fp = FrontendParams.default(K=16, Kd=32)
    o = oracle.frontend_resample(P, pl, ss, es, ct, fp)
    for h in range(5):
        M = len(gears)
        assert o["n_seg"][h] == M and tuple(o["singul"][h, :M]) == gears
        for i in range(M):
            N = o["piece_nums"][h, i]
            assert N >= 2 and o["piece_dt"][h, i] > 0
            # piece duration close to traj_piece_duration (traj_manager.cpp:543-546)
            assert 0.5 < o["piece_dt"][h, i] < 1.5
            assert o["n_states"][h, i] == (N - 2) * 17 + 2 * 33
            st = o["states"][h, i, :o["n_states"][h, i]]
            # every pose lies on the searched path (distance to the polyline below the sample spacing)
            d = np.hypot(st[:, None, 0] - P[h, None, :pl[h], 0], st[:, None, 1] - P[h, None, :pl[h], 1]).min(axis=1)
            assert d.max() < 0.08
            # inner waypoints are the poses at the piece ends (k == resolution)
            ends = np.cumsum([33] + [17] * (N - 2))[: N - 1] - 1
            assert np.array_equal(o["inner_pts"][h, i, : N - 1], st[ends, :2])
            # boundary flat states: positions are the segment's end poses, |v| is the boundary speed
            assert np.array_equal(o["ini_states"][h, i, :2], st[0, :2]) or np.allclose(o["ini_states"][h, i, :2], st[0, :2], atol=1e-9)
            assert np.allclose(o["fin_states"][h, i, :2], st[-1, :2], atol=1e-9)
            vi = np.hypot(*o["ini_states"][h, i, 2:4])
            want = max(abs(ss[h, 3]), 0.2) if i == 0 else 0.2
            assert np.isclose(vi, want, rtol=1e-12)
        # the whole path is covered: first pose = path start, last pose = path end
        assert np.allclose(o["states"][h, 0, 0], P[h, 0], atol=1e-12)
        last = o["states"][h, M - 1, o["n_states"][h, M - 1] - 1]
        assert np.allclose(last, P[h, pl[h] - 1], atol=1e-9)


def test_time_allocation_is_a_trapezoid(oracle):
    """a straight 40 m forward path from rest to rest-ish: v_max 5, a_max 8 -> accelerate, cruise, brake"""
    n = 268
    P = np.zeros((1, 512, 3))
    P[0, :n, 0] = np.linspace(0.0, 40.05, n)
    ss, es, ct = np.array([[0, 0, 0, 0.0]]), np.array([[40.05, 0, 0, 0.0]]), np.zeros((1, 2))
    fp = FrontendParams.default(K=8, Kd=8)
    o = oracle.frontend_resample(P, [n], ss, es, ct, fp)
    assert o["n_seg"][0] == 1 and o["singul"][0, 0] == 1
    T = o["piece_dt"][0, 0] * o["piece_nums"][0, 0]
    L = 40.05
    crit = 2 * (25.0 - 0.0) / 16.0  # both ramps
    assert np.isclose(T, 2 * 5.0 / 8.0 + (L - crit) / 5.0, rtol=1e-6)
    # mid-course speed is the cruise speed: consecutive states of an inner piece are v_max * dt / K apart
    st = o["states"][0, 0, :o["n_states"][0, 0]]
    mid = st[9 + 9 * 3: 9 + 9 * 4, 0]
    assert np.allclose(np.diff(mid), 5.0 * o["piece_dt"][0, 0] / 8, rtol=1e-6)


def test_orders_agree(oracle):
    P, pl, ss, es, ct = sc.searched_paths(8, seed=9, gears=(1, -1, 1))
    a, b = oracle.frontend_resample(P, pl, ss, es, ct, order=0), oracle.frontend_resample(P, pl, ss, es, ct, order=1)
    assert np.array_equal(a["piece_nums"], b["piece_nums"]) and np.array_equal(a["singul"], b["singul"])
    assert np.abs(a["states"] - b["states"]).max() < 1e-12 and np.abs(a["ini_states"] - b["ini_states"]).max() < 1e-12


def test_more_gear_changes_than_capacity_produce_nothing(oracle):
    P, pl, ss, es, ct = sc.searched_paths(2, seed=1, gears=(1, -1, 1), seg_duration=4.0)
    o = oracle.frontend_resample(P, pl, ss, es, ct, max_seg=2)
    assert (o["n_seg"] == 3).all() and (o["piece_nums"] == 0).all() and (o["states"] == 0).all()
