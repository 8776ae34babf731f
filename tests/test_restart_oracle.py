"""CPU checks of the restart sampler's oracle (oracle/restart_oracle.cpp; definition in dftpav_amd/csrc/restart.hip)."""
import numpy as np


def test_restart_zero_is_the_hypothesis_and_statistics_are_right(oracle):
    rng = np.random.default_rng(0)
    inner = rng.uniform(-10, 10, (20, 30))
    durs = rng.uniform(5, 20, (20, 2))
    oi, od = oracle.sample_restarts(inner, durs, 400, sigma=0.3, lo=0.8, hi=1.25, seed=12345)
    oi, od = oi.reshape(20, 400, 30), od.reshape(20, 400, 2)
    assert np.array_equal(oi[:, 0], inner) and np.array_equal(od[:, 0], durs)
    dev = (oi[:, 1:] - inner[:, None]).reshape(-1)
    assert abs(dev.mean()) < 3e-3 and abs(dev.std() - 0.3) < 3e-3
    assert abs(np.mean(np.abs(dev) > 0.6) - 0.0455) < 3e-3  # two-sigma tail of a normal
    fac = (od[:, 1:] / durs[:, None]).reshape(-1)
    assert fac.min() >= 0.8 and fac.max() <= 1.25 and abs(fac.mean() - 1.025) < 3e-3
    # x / y of a waypoint come from one Box-Muller pair: uncorrelated
    d2 = (oi[:, 1:] - inner[:, None]).reshape(-1, 15, 2)
    assert abs(np.corrcoef(d2[..., 0].ravel(), d2[..., 1].ravel())[0, 1]) < 0.01


def test_streams_are_keyed_by_seed_hypothesis_and_restart(oracle):
    inner = np.zeros((3, 8))
    durs = np.ones((3, 1))
    a = oracle.sample_restarts(inner, durs, 5, seed=7)
    b = oracle.sample_restarts(inner, durs, 5, seed=7)
    c = oracle.sample_restarts(inner, durs, 5, seed=8)
    assert np.array_equal(a[0], b[0]) and np.array_equal(a[1], b[1])
    assert not np.array_equal(a[0], c[0])
    # a longer batch extends a shorter one: trajectory (h, r) does not depend on n_restarts or n_hyp
    d = oracle.sample_restarts(inner, durs, 9, seed=7)
    assert np.array_equal(d[0].reshape(3, 9, 8)[:, :5], a[0].reshape(3, 5, 8))
    e = oracle.sample_restarts(inner[:2], durs[:2], 5, seed=7)
    assert np.array_equal(e[0], a[0][:10])
    # all (hypothesis, restart > 0) streams differ
    flat = a[0].reshape(15, 8)[[i for i in range(15) if i % 5]]
    assert len({tuple(r) for r in flat}) == len(flat)
