import sys, os
sys.path.insert(0, "/root/repo")
import numpy as np
from dftpav_amd import capi, scenarios as sc
B=4096
p = capi.default_params(); s = sc.baseline_config(3, B=B); s.apply_resolution(p)
h = capi.Handle(p); bt = capi.Batch(h, s.layout, B); bt.upload(s)
bt.solve_async(); bt.sync(); bt.profile(True); bt.solve_async(); bt.sync()
r = bt.results(); pr = bt.read_profile().astype(np.float64); ev = r["evals"].astype(float)
print("E1a (work before barrier)", np.median(pr[:,9]/ev), "E1b (barrier wait)", np.median(pr[:,10]/ev), "E1c (powers + to end)", np.median(pr[:,0]/ev))
