"""Golden vectors written by the REFERENCE BUILD oracle/_ref (the reference's own traj_optimizer.cpp / poly_traj_utils.hpp /
lbfgs.hpp compiled unmodified, recipe oracle/Makefile.ref) — not by the restatement.  Runs only where /root/reference
exists (this container); the vectors travel as fixtures:  python tests/golden/make_golden_ref.py

Per BASELINE config (the inputs are those of tests/golden/cfg*.npz, stored there): x0 as the reference packs it, f and g at
x0, the whole solve (x, cost, status, iterations, evaluations) and the per-iteration trace (fx, step, evaluations of the line
search) that lbfgs_optimize reports through its progress callback.  Plus unit vectors of the pieces: lbfgs_optimize on
analytic functions, BandedSystem, MinJerkOpt, positiveSmoothedL1, log_sum_exp.
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from oracle import pyoracle as po, pyref as pr  # noqa: E402
from golden_util import CASES, load  # noqa: E402
import ref_cases  # noqa: E402


def main():
    out_dir = os.path.dirname(os.path.abspath(__file__))
    pr.build()
    for name in CASES:
        s, z = load(name)
        p = po.default_params()
        s.apply_resolution(p)
        n = s.layout.n_vars
        rec = dict(x0=np.zeros((s.B, n)), f0=np.zeros(s.B), g0=np.zeros((s.B, n)), x=np.zeros((s.B, n)), cost=np.zeros(s.B),
                   status=np.zeros(s.B, dtype=np.int32), iters=np.zeros(s.B, dtype=np.int32), evals=np.zeros(s.B, dtype=np.int32),
                   ok=np.zeros(s.B, dtype=np.int32))
        for b in range(s.B):
            r = pr.RefProblem(p, s, b)
            rr = r.optimize(trace=True)
            rec["x0"][b] = rr["eval_x"][0]
            rec["f0"][b], rec["g0"][b] = rr["eval_f"][0], rr["eval_g"][0]
            rec["x"][b], rec["cost"][b], rec["status"][b] = rr["x"], rr["final_cost"], rr["status"]
            rec["iters"][b], rec["evals"][b], rec["ok"][b] = rr["iters"], rr["evals"], int(rr["ok"])
            rec["iter_fx_%d" % b], rec["iter_step_%d" % b], rec["iter_ls_%d" % b] = rr["iter_fx"], rr["iter_step"], rr["iter_ls"]
            rec["iter_x_%d" % b] = rr["iter_x"]
        np.savez_compressed(os.path.join(out_dir, "ref_" + name + ".npz"), **rec)
        print(name, rec["cost"], rec["iters"], rec["evals"])
    np.savez_compressed(os.path.join(out_dir, "ref_units.npz"), **ref_cases.run_units(pr, po.default_params()))
    print("units written")


if __name__ == "__main__":
    main()
