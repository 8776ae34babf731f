import os, sys
sys.path.insert(0, '/root/repo')
import numpy as np
from dftpav_amd import capi, scenarios as sc
os.environ["DFTPAV_REF_SHAPE"] = "quad"
for B in (8, 4096):
    p = capi.default_params(); s = sc.baseline_config(3, B=B); s.apply_resolution(p)
    h = capi.Handle(p); bt = capi.Batch(h, s.layout, B); bt.upload(s); bt.set_order(capi.ORDER_REFERENCE)
    bt.solve_async(); bt.sync(); bt.profile(True); bt.solve_async(); bt.sync()
    r = bt.results(); pr = bt.read_profile().astype(np.float64)
    print("B", B, "emit-loop trips of the wave per evaluation (mean over rows):", (pr[:, 11] / r["evals"]).mean(), "active terms per eval", (pr[:, 9] / r["evals"]).mean())
