"""Developer script: the QUAD shape for several gear segments (solver_ref4m.hip) against the TEAM / WAVE shapes (solver_ref.hip):
evaluations and whole solves bit for bit, then times of configs[1] alone on the device.  scripts/quadm_check.py [B ...]"""
import os
import sys
import time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from dftpav_amd import capi, scenarios as sc

KEYS = ("final_cost", "x", "status", "iters", "evals", "success", "hist_sum")


def shaped(bt, shape):
    bt.set_order(capi.ORDER_DEVICE)
    os.environ["DFTPAV_REF_SHAPE"] = shape
    bt.set_order(capi.ORDER_REFERENCE)
    os.environ.pop("DFTPAV_REF_SHAPE")


def main():
    Bs = [int(a) for a in sys.argv[1:]] or [8, 64, 4096]
    cases = [("cfg2", None), ("5-4-6", [5, 4, 6]), ("3-2-4-3", [3, 2, 4, 3])]
    for name, pieces in cases:
        for B in Bs:
            if pieces is not None and B > 64:
                continue
            p = capi.default_params()
            if pieces is None:
                s = sc.baseline_config(2, B=B)
            else:
                s = sc.make_scenario(pieces, [1 if i % 2 == 0 else -1 for i in range(len(pieces))], 9, 14, B, seed=77, n_obs=30)
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
                if not (np.array_equal(ft, fq) and np.array_equal(gt, gq)):
                    bad += 1
                    wf = np.flatnonzero(ft != fq)
                    wg = np.flatnonzero((gt != gq).any(axis=1))
                    print("  eval", i, "MISMATCH: f differs on", len(wf), wf[:8], "g differs on", len(wg), wg[:8])
                    if len(wg):
                        b = wg[0]
                        print("   traj", b, "f", ft[b], fq[b], "g idx", np.flatnonzero(gt[b] != gq[b]), "\n   team", gt[b], "\n   quad", gq[b])
            print(name, "B", B, "evaluations: quad == team on", len(xs), "points:", "OK" if not bad else "FAILED", flush=True)
            res, ms = {}, {}
            for shape in (("team", "wave", "quad") if B > 256 else ("team", "quad")):
                shaped(bt, shape)
                bt.solve()
                t0 = time.perf_counter()
                res[shape] = bt.solve()
                ms[shape] = (time.perf_counter() - t0) * 1e3
            for shape in res:
                if shape == "team":
                    continue
                same = all(np.array_equal(res["team"][k], res[shape][k]) for k in KEYS)
                print("   solves: %s == team: %s" % (shape, "OK" if same else "FAILED"), flush=True)
                if not same:
                    for k in KEYS:
                        w = np.flatnonzero(res["team"][k] != res[shape][k]) if res["team"][k].ndim == 1 else np.flatnonzero((res["team"][k] != res[shape][k]).any(axis=1))
                        print("     ", k, len(w), w[:10])
            print("   ms per solve of the batch (alone):", {k: round(v, 1) for k, v in ms.items()}, "evals", int(res["team"]["evals"].sum()), flush=True)
            if B >= 1024:
                shaped(bt, "quad")
                bt.set_hand_over(0)
                bt.solve()
                t0 = time.perf_counter()
                r = bt.solve()
                print("   quad, hand-over off: %.1f ms, == team: %s" % ((time.perf_counter() - t0) * 1e3, all(np.array_equal(res["team"][k], r[k]) for k in KEYS)))
            bt.close()
            h.close()


if __name__ == "__main__":
    main()
