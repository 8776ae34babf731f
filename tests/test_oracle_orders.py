"""Literal vs device-order oracle, and why whole solves need bit-exactness."""
import numpy as np
import pytest

from dftpav_amd import scenarios as sc


@pytest.mark.parametrize("cfg,B", [(1, 3), (2, 3), (3, 3), (5, 2)])
def test_device_order_is_a_reassociation_of_literal(oracle, cfg, B):
    """Same mathematics, different summation order: single evaluations agree to rounding level."""
    p = oracle.default_params()
    s = sc.baseline_config(cfg, B=B)
    s.apply_resolution(p)
    rng = np.random.default_rng(10 + cfg)
    for b in range(B):
        lit = oracle.OracleProblem(p, s, b)
        dev = oracle.OracleProblem(p, s, b, order=1)
        for scale in (0.0, 0.05, 0.3):
            x = lit.x0() + rng.normal(0, scale, lit.n) if scale else lit.x0()
            f0, g0 = lit.eval(x)
            f1, g1 = dev.eval(x)
            assert abs(f0 - f1) <= 1e-11 * abs(f0)
            assert np.abs(g0 - g1).max() <= 1e-10 * np.abs(g0).max()
            c0, dt0 = lit.coeffs()
            c1, dt1 = dev.coeffs()
            assert np.allclose(c0, c1, rtol=1e-11, atol=1e-11) and np.array_equal(dt0, dt1)


def test_reference_solver_is_chaotic(oracle):
    """One ulp on x0 changes the literal oracle's final cost by far more than the 1e-5 parity
    bar on some problems: the reason the GPU is held to the bit-exact device order."""
    p = oracle.default_params()
    s = sc.baseline_config(3, B=8)
    s.apply_resolution(p)
    worst = 0.0
    for b in range(8):
        pr = oracle.OracleProblem(p, s, b)
        x0 = pr.x0()
        _, r0 = pr.solve(x0)
        x1 = np.nextafter(x0, np.inf)
        _, r1 = pr.solve(x1)
        worst = max(worst, abs(r1.final_cost - r0.final_cost) / r0.final_cost)
    assert worst > 1e-5


def test_both_orders_solve_to_comparable_optima(oracle):
    """Different iterate sequences, same optimisation problem: final costs of the two orders are
    statistically equivalent (medians within a few percent) and every solve succeeds."""
    p = oracle.default_params()
    s = sc.baseline_config(3, B=24)
    s.apply_resolution(p)
    a = oracle.solve_batch(p, s, nthreads=4, order=0)
    b = oracle.solve_batch(p, s, nthreads=4, order=1)
    assert a["success"].all() and b["success"].all()
    assert abs(np.median(a["final_cost"]) - np.median(b["final_cost"])) < 0.05 * np.median(a["final_cost"])
    # both far below the initial cost
    f0 = np.array([oracle.OracleProblem(p, s, i).eval(oracle.OracleProblem(p, s, i).x0())[0] for i in range(4)])
    assert (a["final_cost"][:4] < f0).all() and (b["final_cost"][:4] < f0).all()


def test_solve_is_deterministic(oracle):
    p = oracle.default_params()
    s = sc.baseline_config(2, B=2)
    s.apply_resolution(p)
    for order in (0, 1):
        a = oracle.solve_batch(p, s, nthreads=1, order=order)
        b = oracle.solve_batch(p, s, nthreads=2, order=order)
        assert np.array_equal(a["x"], b["x"]) and np.array_equal(a["final_cost"], b["final_cost"])
