"""Developer script: the QUAD shape of the reference order (solver_ref4.hip) against the TEAM / WAVE shapes (solver_ref.hip) and
the restatement -- evaluations and whole solves bit for bit, then kernel times.  scripts/quad_check.py [B ...]"""
import os
import sys
import time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from dftpav_amd import capi, scenarios as sc
from oracle import pyoracle as po

KEYS = ("final_cost", "x", "status", "iters", "evals", "success", "hist_sum")


def shaped(bt, shape):
    bt.set_order(capi.ORDER_DEVICE)
    os.environ["DFTPAV_REF_SHAPE"] = shape
    bt.set_order(capi.ORDER_REFERENCE)
    os.environ.pop("DFTPAV_REF_SHAPE")


def main():
    Bs = [int(a) for a in sys.argv[1:]] or [8, 64, 4096]
    for cfg in (3, 1):
        for B in Bs:
            p = capi.default_params()
            s = sc.baseline_config(cfg, B=B)
            s.apply_resolution(p)
            h = capi.Handle(p)
            bt = capi.Batch(h, s.layout, B)
            bt.upload(s)
            x0 = bt.x0()
            rng = np.random.default_rng(5)
            xs = [x0, x0 + rng.normal(0, 0.05, x0.shape), x0 + rng.normal(0, 0.7, x0.shape)]
            ev = {}
            for shape in ("team", "quad"):
                shaped(bt, shape)
                ev[shape] = [bt.eval(x) for x in xs]
            bad = 0
            for i in range(len(xs)):
                ft, gt = ev["team"][i]
                fq, gq = ev["quad"][i]
                ok = np.array_equal(ft, fq) and np.array_equal(gt, gq)
                if not ok:
                    bad += 1
                    wf = np.flatnonzero(ft != fq)
                    wg = np.flatnonzero((gt != gq).any(axis=1))
                    print("  eval", i, "MISMATCH: f differs on", wf[:8], "g differs on", wg[:8])
                    if len(wg):
                        b = wg[0]
                        print("   traj", b, "f", ft[b], fq[b], "g idx", np.flatnonzero(gt[b] != gq[b]), "\n   team", gt[b], "\n   quad", gq[b])
            nb = min(B, 4)
            for b in range(nb):
                fl, gl = po.OracleProblem(p, s, b, order=0).eval(xs[1][b])
                if not (fl == ev["quad"][1][0][b] and np.array_equal(gl, ev["quad"][1][1][b])):
                    bad += 1
                    print("  eval vs restatement MISMATCH traj", b)
            print("cfg", cfg, "B", B, "evaluations: quad == team on", len(xs), "points,", nb, "also == restatement:", "OK" if not bad else "FAILED", flush=True)
            res, ms = {}, {}
            for shape in (("team", "wave", "quad") if B > 256 else ("team", "quad")):
                shaped(bt, shape)
                bt.solve_async(); bt.sync()
                t = []
                for _ in range(2):
                    bt.solve_async(); bt.sync(); t.append(bt.last_solve_ms())
                res[shape] = bt.results()
                ms[shape] = min(t)
            for shape in res:
                if shape == "team":
                    continue
                diff = [k for k in KEYS if not np.array_equal(res["team"][k], res[shape][k])]
                if diff:
                    w = np.flatnonzero(res["team"]["final_cost"] != res[shape]["final_cost"])
                    print("  solve", shape, "MISMATCH in", diff, "trajectories", w[:10], "of", len(w))
                    for b in w[:3]:
                        print("   ", b, "team", res["team"]["final_cost"][b], res["team"]["iters"][b], res["team"]["evals"][b], shape, res[shape]["final_cost"][b],
                              res[shape]["iters"][b], res[shape]["evals"][b])
                else:
                    print("  solve", shape, "== team on all of", KEYS)
            lit = po.solve_batch(p, s, nthreads=4, order=0) if B <= 64 else None
            if lit is not None:
                d2 = [k for k in KEYS if not np.array_equal(res["quad"][k][:len(lit[k])], lit[k])]
                print("  solve quad vs restatement:", "OK" if not d2 else ("MISMATCH " + str(d2)))
            print("  kernel ms", {k: round(v, 2) for k, v in ms.items()}, "solves/s", {k: round(B / (v * 1e-3)) for k, v in ms.items()},
                  "iters mean/max", res["quad"]["iters"].mean(), res["quad"]["iters"].max(), flush=True)
            bt.close()
            h.close()


if __name__ == "__main__":
    main()
