import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from dftpav_amd import capi, scenarios as sc
from oracle import pyoracle as po
p = capi.default_params()
for cfg, B in ((1, 2),):
    s = sc.baseline_config(cfg, B=B); s.apply_resolution(p)
    for mi in (6,7,8,9,10,12,14,16,20,24,28,33,34,36):
        p.lbfgs_max_iterations = mi
        h = capi.Handle(p); bt = capi.Batch(h, s.layout, B); bt.upload(s)
        r = bt.solve(); ro = po.solve_batch(p, s, nthreads=1, order=1)
        print(cfg, "maxit", mi, "gpu", r["final_cost"], r["iters"], r["evals"], "cpu", ro["final_cost"], ro["iters"], ro["evals"],
              "xdiff", np.abs(r["x"] - ro["x"]).max())
        bt.close(); h.close()
