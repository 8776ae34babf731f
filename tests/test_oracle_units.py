"""Self-made pins of the oracle (the reference ships no tests: SURVEY §4, §8c)."""
import numpy as np
import pytest

from dftpav_amd import scenarios as sc


def test_smoothed_l1_matches_formula_and_is_c1(oracle):
    import ctypes as C
    L = oracle.lib()
    pe = 1e-4
    for x in [1e-9, 3e-5, pe * (1 - 1e-12), pe, 2e-4, 1.0]:
        f, df = C.c_double(), C.c_double()
        L.oracle_smoothed_l1(x, C.byref(f), C.byref(df))
        if x < pe:  # traj_optimizer.cpp:793-796
            assert f.value == pytest.approx((-x / (2 * pe ** 3) + 1 / pe ** 2) * x ** 3, rel=1e-12)
            assert df.value == pytest.approx((-2 * x / pe ** 3 + 3 / pe ** 2) * x ** 2, rel=1e-12)
        else:
            assert f.value == x - pe / 2 and df.value == 1.0
    # continuity of value and slope at the joint
    fa, da, fb, db = C.c_double(), C.c_double(), C.c_double(), C.c_double()
    L.oracle_smoothed_l1(pe * (1 - 1e-9), C.byref(fa), C.byref(da))
    L.oracle_smoothed_l1(pe, C.byref(fb), C.byref(db))
    assert abs(fa.value - fb.value) < 1e-12 and abs(da.value - db.value) < 1e-7


def test_time_map_is_a_bijection(oracle):
    p = oracle.default_params()
    L = oracle.lib()
    for T in [0.1001, 0.5, 1.0, 1.1, 1.1001, 2.0, 16.0, 40.0]:
        vt = L.oracle_real_to_virtual_T(p, T)
        assert L.oracle_virtual_to_real_T(p, vt) == pytest.approx(T, rel=1e-12)
    # monotone
    vs = np.linspace(-5, 5, 101)
    Ts = [L.oracle_virtual_to_real_T(p, v) for v in vs]
    assert np.all(np.diff(Ts) > 0) and min(Ts) > p.mini_T


@pytest.mark.parametrize("N", [2, 5, 16])
def test_minco_invariants(oracle, N):
    """C^3 junctions, boundary states reproduced, waypoints interpolated, jerk energy = quadrature."""
    rng = np.random.default_rng(N)
    inner = rng.normal(0, 3, (N - 1, 2))
    head = rng.normal(0, 1, 6)
    tail = rng.normal(0, 1, 6)
    dT = 0.8
    c, J = oracle.minco_generate(inner, dT, head, tail)

    def ev(p, s, der):
        k = np.arange(6)
        coef = np.ones(6)
        for q in range(der):
            coef = coef * (k - q)
        pw = np.where(k - der >= 0, s ** np.clip(k - der, 0, None), 0.0)
        return (c[p] * (coef * pw)[:, None]).sum(axis=0)

    for der in range(3):
        assert np.allclose(ev(0, 0.0, der), head[2 * der:2 * der + 2], atol=1e-9)
        assert np.allclose(ev(N - 1, dT, der), tail[2 * der:2 * der + 2], atol=1e-9)
    for p in range(N - 1):
        assert np.allclose(ev(p, dT, 0), inner[p], atol=1e-9)
        for der in range(5):  # position .. snap continuous (poly_traj_utils.hpp:900-932)
            assert np.allclose(ev(p, dT, der), ev(p + 1, 0.0, der), atol=1e-7)
    # jerk energy by Gauss-Legendre quadrature
    xs, ws = np.polynomial.legendre.leggauss(8)
    Jq = 0.0
    for p in range(N):
        for x, w in zip(xs, ws):
            s = 0.5 * dT * (x + 1)
            Jq += 0.5 * dT * w * np.sum(ev(p, s, 3) ** 2)
    assert J == pytest.approx(Jq, rel=1e-9)
    # third implementation: dense numpy solve of the same matrix
    c2 = sc.minco_fit(inner, dT, head, tail)
    assert np.allclose(c, c2, rtol=1e-9, atol=1e-9)


def test_minco_operator_is_the_inverse_restricted(oracle):
    N = 6
    M = oracle.minco_operator(N)
    A = sc.minco_matrix(N)
    rows = [0, 1, 2] + [6 * i + 5 for i in range(N - 1)] + [6 * N - 3, 6 * N - 2, 6 * N - 1]
    assert np.allclose(A @ M, np.eye(6 * N)[:, rows], atol=1e-9)


def test_lbfgs_on_rosenbrock_and_quadratic(oracle):
    def rosen(x):
        f = np.sum(100.0 * (x[1:] - x[:-1] ** 2) ** 2 + (1 - x[:-1]) ** 2)
        g = np.zeros_like(x)
        g[:-1] = -400 * x[:-1] * (x[1:] - x[:-1] ** 2) - 2 * (1 - x[:-1])
        g[1:] += 200 * (x[1:] - x[:-1] ** 2)
        return f, g

    p = oracle.default_params()
    p.lbfgs_delta = 1e-12
    p.lbfgs_g_epsilon = 1e-8
    r = oracle.lbfgs(rosen, np.full(6, -1.2), p)
    assert r["ret"] in (0, 1) and np.allclose(r["x"], 1.0, atol=1e-4) and r["f"] < 1e-8

    Q = np.diag(np.arange(1.0, 11.0))

    def quad(x):
        return 0.5 * x @ Q @ x, Q @ x

    r = oracle.lbfgs(quad, np.ones(10), p)
    assert r["ret"] in (0, 1) and np.abs(r["x"]).max() < 1e-5
    # the reference's return codes survive: a stationary start converges in 0 iterations (lbfgs.hpp:542-547)
    r = oracle.lbfgs(quad, np.zeros(10), oracle.default_params())
    assert r["ret"] == 0 and r["iters"] == 0 and r["evals"] == 1


def test_lbfgs_early_exit_quirk(oracle):
    """lbfgs.hpp:326-329: the line search accepts any trial whose relative change is < delta/past."""
    w = np.array([1.0, 10.0, 100.0, 1000.0])

    def flat(x):
        return 1e6 + 1e-3 * float(w @ (x * x)), 2e-3 * w * x

    r = oracle.lbfgs(flat, np.ones(4))
    # relative changes are ~1e-6 < delta/past, so every first trial is accepted (one evaluation
    # per iteration) and the past-test stops after `past` iterations, far from the minimiser
    assert r["ret"] == 1 and r["iters"] == 3 and r["evals"] == 1 + r["iters"]
    assert np.abs(r["x"]).max() > 1e-2


@pytest.mark.parametrize("cfg,B", [(1, 2), (2, 2), (3, 2), (5, 2)])
def test_gradient_against_central_differences(oracle, cfg, B):
    """Every variable class (waypoints, tau, gear xy, gear angle), every penalty incl. moving obstacles."""
    p = oracle.default_params()
    s = sc.baseline_config(cfg, B=B)
    s.apply_resolution(p)
    rng = np.random.default_rng(cfg)
    for order in (0, 1):
        pr = oracle.OracleProblem(p, s, B - 1, order=order)
        x = pr.x0() + rng.normal(0, 0.05, pr.n)
        f, g = pr.eval(x)
        h = 1e-5
        gfd = np.zeros_like(g)
        for i in range(len(x)):
            xp, xm = x.copy(), x.copy()
            xp[i] += h
            xm[i] -= h
            gfd[i] = (pr.eval(xp)[0] - pr.eval(xm)[0]) / (2 * h)
        assert np.abs(g - gfd).max() / np.abs(g).max() < 2e-7, (cfg, order)


def test_validation_errors(oracle):
    p = oracle.default_params()
    s = sc.baseline_config(1, B=1)
    s.apply_resolution(p)
    s.init_Ts[0, 0] = 0.05  # < mini_T, traj_optimizer.cpp:30-33
    with pytest.raises(ValueError, match="-2"):
        oracle.OracleProblem(p, s, 0)
