"""Developer script: the bench's schedule (a step launches one batch and waits for the batch launched D - 1 steps earlier) at
pipeline depths D = 2, 3, 4 -- D handles, D resident batches of B trajectories (configs[2]).
  python scripts/depth_check.py [B] [n_batches] [depths...]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from dftpav_amd import capi, scenarios as sc

B = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
nb = int(sys.argv[2]) if len(sys.argv) > 2 else 24
depths = [int(a) for a in sys.argv[3:]] or [2, 3, 4]
p = capi.default_params()
Dmax = max(depths)
scen = [sc.baseline_config(3, B=B, seed=20240 + 104729 * i) for i in range(Dmax)]
for s in scen:
    s.apply_resolution(p)
hs = [capi.Handle(p) for _ in range(Dmax)]
bts = []
for h, s in zip(hs, scen):
    bt = capi.Batch(h, s.layout, B); bt.upload(s); bts.append(bt)
ref = [bt.solve() for bt in bts]
for D in depths:
    for bt in bts:
        bt.sync()
    for rep in range(2):
        t0 = time.perf_counter()
        for k in range(nb):
            bts[k % D].solve_async()
            if k >= D - 1:
                bts[(k - D + 1) % D].sync()
        for k in range(nb - D + 1, nb):
            bts[k % D].sync()
        el = time.perf_counter() - t0
    ok = all(np.array_equal(bt.results()["final_cost"], r["final_cost"]) for bt, r in zip(bts[:D], ref[:D]))
    print("depth %d: %d batches of %d: %.1f ms per batch, %.0f solves/s, results unchanged: %s" % (D, nb, B, 1e3 * el / nb, nb * B / el, ok), flush=True)
