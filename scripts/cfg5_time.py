"""Developer script: BASELINE configs[4] (moving obstacles, 32 pieces x 65 points, batch 1024) — kernel time, phase profile, and a
bit check of a few trajectories against the device-order oracle."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from dftpav_amd import capi, scenarios as sc
from oracle import pyoracle as po
B = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
p = capi.default_params()
s = sc.baseline_config(5, B=B); s.apply_resolution(p)
h = capi.Handle(p); h.set_surround(s.surround)
bt = capi.Batch(h, s.layout, B); bt.upload(s)
bt.solve_async(); bt.sync()
ms = []
for _ in range(2):
    bt.solve_async(); bt.sync(); ms.append(bt.last_solve_ms())
r = bt.results()
print("cfg5 B", B, "kernel ms", np.round(ms, 2), "solves/s", B / (np.mean(ms) * 1e-3), "mean iters", r["iters"].mean(), "evals", r["evals"].mean(),
      "longest", int(r["iters"].max()), "iterations /", int(r["evals"].max()), "evaluations; the five longest:", np.sort(r["evals"])[-5:])
lat = r["latency_us"].astype(np.float64)
order = np.argsort(r["evals"])
print("time in service: sum %.1f s, max %.1f ms, mean us per evaluation %.1f; of the five longest: us per evaluation %s, in service ms %s, ids %s" %
      (lat.sum() * 1e-6, lat.max() * 1e-3, lat.sum() / r["evals"].sum(), np.round(lat[order[-5:]] / r["evals"][order[-5:]], 1), np.round(lat[order[-5:]] * 1e-3, 1), order[-5:]))
pick = np.array([0, B // 3, B - 1])
ro = po.solve_batch(p, s.subset(pick), nthreads=2, order=1)
print("bit-exact on 3 sampled:", all(np.array_equal(ro[k], r[k][pick]) for k in ("final_cost", "x", "iters", "evals")))
bt.profile(True); bt.solve_async(); bt.sync()
pr = bt.read_profile().astype(np.float64); tot = pr.sum()
names = ["E1", "E2", "E3+E4 samples", "E4 chain", "E5", "E6", "LS misc", "hist", "two-loop", "gate+scan", "pairs", "x"]
print(" ".join("%s %.1f%%" % (n, 100 * pr[:, i].sum() / tot) for i, n in enumerate(names)))
ev = r["evals"].astype(float)
print("per eval (median cycles): static samples", np.median(pr[:, 2] / ev), "gate+scan", np.median(pr[:, 9] / ev), "pairs", np.median(pr[:, 10] / ev), "chains", np.median(pr[:, 3] / ev))
