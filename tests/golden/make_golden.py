"""Freezes golden vectors for the BASELINE configs (small batches).

The reference holds no golden vectors for this path, and cannot be run
here, so these are produced by the CPU oracle (oracle/, both summation orders) on
seeded inputs; the inputs themselves are stored so the fixtures do not depend on
the generator staying unchanged.  Run from the repo root:
    python tests/golden/make_golden.py
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from dftpav_amd import scenarios as sc  # noqa: E402
from oracle import pyoracle as po  # noqa: E402

CASES = {"cfg1": (1, 2), "cfg2": (2, 2), "cfg3": (3, 4), "cfg5": (5, 1)}


def main():
    out_dir = os.path.dirname(os.path.abspath(__file__))
    for name, (cfg, B) in CASES.items():
        p = po.default_params()
        s = sc.baseline_config(cfg, B=B)
        s.apply_resolution(p)
        rec = dict(cfg=cfg, B=B, K=s.K, Kd=s.Kd, piece_nums=s.layout.piece_nums, singuls=s.layout.singuls,
                   ini_states=s.ini_states, fin_states=s.fin_states, inner_pts=s.inner_pts, init_Ts=s.init_Ts,
                   corridor=s.corridor, t_now=s.t_now, has_surround=int(s.surround is not None))
        if s.surround is not None:
            rec.update(sur_off=s.surround.piece_offsets, sur_dur=s.surround.durations, sur_coef=s.surround.coeffs,
                       sur_total=s.surround.total_duration, sur_start=s.surround.start_time)
        n = s.layout.n_vars
        for order, tag in ((0, "lit"), (1, "dev")):
            x0 = np.zeros((B, n)); f0 = np.zeros(B); g0 = np.zeros((B, n))
            for b in range(B):
                pr = po.OracleProblem(p, s, b, order=order)
                x0[b] = pr.x0()
                f0[b], g0[b] = pr.eval(x0[b])
            r = po.solve_batch(p, s, nthreads=1, order=order)
            rec.update({"x0": x0, tag + "_f0": f0, tag + "_g0": g0, tag + "_x": r["x"], tag + "_cost": r["final_cost"],
                        tag + "_status": r["status"], tag + "_iters": r["iters"], tag + "_evals": r["evals"],
                        tag + "_hist": r["hist_sum"]})
        np.savez_compressed(os.path.join(out_dir, name + ".npz"), **rec)
        print(name, "lit", rec["lit_cost"], rec["lit_iters"], "dev", rec["dev_cost"], rec["dev_iters"])


if __name__ == "__main__":
    main()
