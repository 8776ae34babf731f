"""GPU solve in lockstep with the reference's L-BFGS (see tests/lockstep.py): the kernel's evaluation trace of one
trajectory per BASELINE config is replayed against literal evaluations by the reference build oracle/_ref (or, where that
library is absent, the literal oracle that tests/test_ref_pin.py shows bit-equal to it)."""
import numpy as np
import pytest

import lockstep
from dftpav_amd import scenarios as sc

pytestmark = pytest.mark.gpu


def literal_evaluator(oracle, p, s, b):
    from oracle import pyref
    if pyref.available():
        r = pyref.RefProblem(p, s, b)
        r.optimize()
        return r.eval, "reference build"
    o = oracle.OracleProblem(p, s, b, order=0)
    return o.eval, "literal oracle"


# cfg 0: BASELINE configs[0] on the reference's default arena (tests/test_default_map.py), corridor from its map
@pytest.mark.parametrize("cfg,B,b", [(1, 2, 0), (2, 2, 1), (3, 4, 2), (5, 2, 0), (0, 2, 1)])
def test_solve_is_in_lockstep_with_the_reference_lbfgs(hiplib, oracle, cfg, B, b):
    capi = hiplib
    p = capi.default_params()
    if cfg == 0:
        from test_default_map import _default_map_scenario
        p.traj_resolution, p.des_traj_resolution = 16, 32
        s = _default_map_scenario(oracle, p, 16, 32, B)
    else:
        s = sc.baseline_config(cfg, B=B)
        s.apply_resolution(p)
    h = capi.Handle(p)
    h.set_surround(s.surround)
    bt = capi.Batch(h, s.layout, B)
    bt.upload(s)
    bt.trace(b, 4096)
    r = bt.solve()
    tr = bt.get_trace()
    assert len(tr["f"]) == r["evals"][b]            # every evaluation of the trajectory was recorded
    assert np.array_equal(tr["x"][0], bt.x0()[b])
    # tracing does not change the solve
    bt.trace(b, 0)
    r2 = bt.solve()
    assert np.array_equal(r["x"], r2["x"]) and np.array_equal(r["final_cost"], r2["final_cost"])
    lit, who = literal_evaluator(oracle, p, s, b)
    rep = lockstep.replay(tr, lit, p)
    print("cfg %d traj %d (%s): %d evaluations, %d iterations replayed, %d branches identical, first flip %s; "
          "rel f %.2e g %.2e d %.2e x %.2e, smallest branch margin %.2e" %
          (cfg, b, who, rep["evals"], rep["iterations"], rep["branches"], rep["flip"], rep["rel_f"], rep["rel_g"], rep["rel_d"],
           rep["rel_x"], rep["min_margin"]))
    # costs agree to rounding.  Gradients are compared relative to their largest component, which near the end of a solve is
    # 1e4..1e8 times smaller than the penalty terms that cancel inside it (1e8 curvature, traj_optimizer.cpp:783-806), and the
    # kernel applies the MINCO adjoint as a dense operator where the reference substitutes through the banded LU
    # (poly_traj_utils.hpp:831-852): two roundings of the same sum.  1e-11 / 1e-10 hold at x0 (test_eval_matches_oracle).
    assert rep["rel_f"] <= 1e-11 and rep["rel_g"] <= 1e-6
    assert rep["rel_x"] <= 1e-15
    assert rep["rel_d"] <= 1e-9
    # the replay covers the whole solve unless a branch sat within rounding of its threshold
    if rep["flip"] is None:
        assert abs(rep["iterations"] - r["iters"][b]) <= 1
    else:
        assert rep["iterations"] >= 10
    bt.close()
    h.close()
