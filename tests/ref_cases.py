"""Unit cases run identically on the reference build (oracle/pyref.py) and on the restatement (oracle/pyoracle.py):
run_units(backend, params) -> dict of arrays.  A backend offers lbfgs(fn, x0, params) and the MinJerkOpt / BandedSystem /
scalar entry points below.  Used by tests/golden/make_golden_ref.py (reference build -> fixtures) and tests/test_ref_pin.py."""
import copy

import numpy as np


def rosenbrock(x):
    n = len(x)
    g = np.zeros(n)
    f = 0.0
    for i in range(n - 1):
        t1 = 1.0 - x[i]
        t2 = 10.0 * (x[i + 1] - x[i] * x[i])
        g[i + 1] += 20.0 * t2
        g[i] += -2.0 * (x[i] * 20.0 * t2 + t1)
        f += t1 * t1 + t2 * t2
    return f, g


def make_quadratic(n, seed):
    rng = np.random.default_rng(seed)
    Q = rng.standard_normal((n, n))
    A = Q @ np.diag(np.logspace(0, 4, n)) @ Q.T
    A = 0.5 * (A + A.T)
    b = rng.standard_normal(n)

    def fn(x):
        Ax = A @ x
        return 0.5 * float(x @ Ax) - float(b @ x), Ax - b
    return fn


def plateau(x):
    """tiny decreases: exercises the early exit of the line search (lbfgs.hpp:326-329) and the past/delta stop"""
    f = 1.0e3 + 1.0e-3 * float(x @ x)
    return f, 2.0e-3 * x


LBFGS_CASES = [
    ("rosen2", rosenbrock, np.array([-1.2, 1.0]), {}),
    ("rosen8", rosenbrock, np.array([-1.2, 1.0] * 4), dict(lbfgs_mem_size=8)),
    ("rosen31", rosenbrock, np.linspace(-1.5, 1.5, 31), dict(lbfgs_delta=1e-9)),
    ("quad12", make_quadratic(12, 3), np.ones(12), dict(lbfgs_mem_size=4, lbfgs_delta=1e-12)),
    ("quad40", make_quadratic(40, 5), np.linspace(-1, 1, 40), dict(lbfgs_delta=1e-10)),
    ("plateau", plateau, np.linspace(1, 2, 7), {}),
]


def with_fields(params, kw):
    p = copy.copy(params)
    q = type(params)()
    import ctypes as C
    C.memmove(C.byref(q), C.byref(params), C.sizeof(params))
    for k, v in kw.items():
        setattr(q, k, v)
    return q


def minco_inputs(N, seed):
    rng = np.random.default_rng(seed)
    inner = np.cumsum(rng.uniform(0.5, 2.0, (N - 1, 2)), axis=0)
    head = np.array([[0.0, 0.0], [rng.uniform(0.5, 2), rng.uniform(-1, 1)], [rng.uniform(-1, 1), rng.uniform(-1, 1)]])
    tail = np.array([inner[-1] + 1.5, [rng.uniform(0.5, 2), rng.uniform(-1, 1)], [rng.uniform(-1, 1), rng.uniform(-1, 1)]])
    return inner, float(rng.uniform(0.4, 1.8)), head, tail


def run_lbfgs(backend, params):
    out = {}
    for name, fn, x0, kw in LBFGS_CASES:
        r = backend.lbfgs(fn, x0, with_fields(params, kw))
        out["lbfgs_%s_x" % name] = r["x"]
        out["lbfgs_%s_meta" % name] = np.array([r["ret"], r["iters"], r["evals"]], dtype=np.int64)
        out["lbfgs_%s_f" % name] = np.array([r["f"]])
    return out


def run_units(ref, params):
    """the reference build's answers (pyref)"""
    out = run_lbfgs(ref, params)
    for N in (2, 3, 8, 16, 32):
        inner, dT, head, tail = minco_inputs(N, 100 + N)
        r = ref.minco(inner, dT, head, tail)
        out["minco_%d_coeffs" % N], out["minco_%d_energy" % N] = r["coeffs"], np.array([r["energy"]])
        out["minco_%d_gdP" % N], out["minco_%d_gdHT" % N] = r["gdP"], np.stack([r["gdHead"], r["gdTail"]])
        out["minco_%d_gdT" % N] = np.array([r["gdT"]])
    xs = np.concatenate([np.linspace(-1e-4, 3e-4, 41), np.array([1e-4, 0.99999e-4, 1.00001e-4, 1.0, 37.5])])
    out["l1_x"] = xs
    out["l1_f"] = np.array([ref.smoothed_l1(params, x) for x in xs])
    rng = np.random.default_rng(9)
    d8, d4 = rng.uniform(-2, 3, 8), rng.uniform(-2, 3, 4)
    r8, w8, s8 = ref.log_sum_exp(100.0, d8)
    r4, w4, s4 = ref.log_sum_exp(-100.0, d4)
    out["lse_in8"], out["lse_in4"] = d8, d4
    out["lse_out8"], out["lse_out4"] = np.concatenate([[r8, s8], w8]), np.concatenate([[r4, s4], w4])
    return out
