"""Developer script: two handles (two HIP streams), batches solved alternately without chaining: does the next
batch's queue launch fill the slots the previous one frees while it thins out?
  DFTPAV_HANDOVER=0 python scripts/overlap_check.py [B] [n_batches]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from dftpav_amd import capi, scenarios as sc

B = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
nb = int(sys.argv[2]) if len(sys.argv) > 2 else 9
p = capi.default_params()
scen = [sc.baseline_config(3, B=B, seed=20240 + 104729 * i) for i in range(2)]
for s in scen:
    s.apply_resolution(p)
hs = [capi.Handle(p) for _ in range(2)]
bts = []
for h, s in zip(hs, scen):
    bt = capi.Batch(h, s.layout, B); bt.upload(s); bts.append(bt)
ref = [bt.solve() for bt in bts]
for bt in bts:
    bt.sync()
t0 = time.perf_counter()
for k in range(nb):
    bts[k % 2].solve_async()
    if k:
        bts[(k - 1) % 2].sync()
bts[(nb - 1) % 2].sync()
el = time.perf_counter() - t0
ok = all(np.array_equal(bt.results()["final_cost"], r["final_cost"]) for bt, r in zip(bts, ref))
print("hand-over %s: %d batches of %d on two streams: %.1f ms per batch, %.0f solves/s, results unchanged: %s" %
      (os.environ.get("DFTPAV_HANDOVER", "default"), nb, B, 1e3 * el / nb, nb * B / el, ok))
