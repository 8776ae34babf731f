"""The oracle against the frozen golden vectors (tests/golden/*.npz, made by make_golden.py)."""
import numpy as np
import pytest

from dftpav_amd import scenarios as sc
from golden_util import CASES, load


@pytest.mark.parametrize("name", CASES)
def test_generator_still_produces_the_stored_inputs(name):
    s, z = load(name)
    g = sc.baseline_config(int(z["cfg"]), B=int(z["B"]))
    assert np.array_equal(g.corridor, s.corridor) and np.array_equal(g.inner_pts, s.inner_pts)
    assert np.array_equal(g.ini_states, s.ini_states) and np.array_equal(g.init_Ts, s.init_Ts)


@pytest.mark.parametrize("name", CASES)
@pytest.mark.parametrize("order,tag", [(0, "lit"), (1, "dev")])
def test_oracle_reproduces_golden(oracle, name, order, tag):
    s, z = load(name)
    p = oracle.default_params()
    s.apply_resolution(p)
    for b in range(s.B):
        pr = oracle.OracleProblem(p, s, b, order=order)
        x0 = pr.x0()
        assert np.array_equal(x0, z["x0"][b])
        f, g = pr.eval(x0)
        if order == 1:  # device order uses no libm: bit-stable across hosts
            assert f == z[tag + "_f0"][b] and np.array_equal(g, z[tag + "_g0"][b])
        else:           # literal order calls libm (cos/sin/exp/log/atan2): allow its last-bit freedom
            assert f == pytest.approx(z[tag + "_f0"][b], rel=1e-12)
            assert np.allclose(g, z[tag + "_g0"][b], rtol=1e-10, atol=1e-9)
    r = oracle.solve_batch(p, s, nthreads=1, order=order)
    if order == 1 or not int(z["has_surround"]) and s.layout.M == 1:
        assert np.array_equal(r["final_cost"], z[tag + "_cost"]) and np.array_equal(r["x"], z[tag + "_x"])
        assert np.array_equal(r["iters"], z[tag + "_iters"]) and np.array_equal(r["evals"], z[tag + "_evals"])
        assert np.array_equal(r["status"], z[tag + "_status"]) and np.array_equal(r["hist_sum"], z[tag + "_hist"])
    else:
        assert r["success"].all()
