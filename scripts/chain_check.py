"""Developer script: chained solves with the default plan (no environment overrides) against plain solves.
  python scripts/chain_check.py [B] [n_batches]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from dftpav_amd import capi, scenarios as sc

B = int(sys.argv[1]) if len(sys.argv) > 1 else 1536
nb = int(sys.argv[2]) if len(sys.argv) > 2 else 4
p = capi.default_params()
scen = [sc.baseline_config(3, B=B, seed=777 + 13 * k) for k in range(nb)]
for s in scen:
    s.apply_resolution(p)
h = capi.Handle(p)
keys = ("final_cost", "x", "status", "iters", "evals", "hist_sum", "success")
plain = []
for s in scen:
    bt = capi.Batch(h, s.layout, B); bt.upload(s); plain.append(bt.solve()); bt.close()
objs = [capi.Batch(h, scen[0].layout, B) for _ in range(2)]
got = []
t0 = time.perf_counter()
for k, s in enumerate(scen):
    cur, prev = objs[k % 2], (objs[(k - 1) % 2] if k else None)
    cur.upload(s)
    cur.solve_chained(prev)
    if prev is not None:
        got.append(prev.results())
got.append(objs[(nb - 1) % 2].results())
t1 = time.perf_counter()
ok = all(np.array_equal(g[k], r[k]) for g, r in zip(got, plain) for k in keys)
print("B %d x %d batches through two chained objects: identical to plain solves: %s (%.0f ms, uploads included)" % (B, nb, ok, 1e3 * (t1 - t0)))
sys.exit(0 if ok else 1)
