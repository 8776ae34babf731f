"""Reference-order kernel: kernel time of isolated solves at a few batch sizes (developer script)."""
import sys
import os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from dftpav_amd import capi, scenarios as sc
cfg = int(sys.argv[1]) if len(sys.argv) > 1 else 3
for B in [int(a) for a in sys.argv[2:]] or [1, 32, 256, 2048]:
    p = capi.default_params()
    s = sc.baseline_config(cfg, B=B); s.apply_resolution(p)
    h = capi.Handle(p); h.set_surround(s.surround); bt = capi.Batch(h, s.layout, B); bt.upload(s)
    bt.set_order(capi.ORDER_REFERENCE)
    bt.solve_async(); bt.sync()
    ms = []
    for _ in range(2):
        bt.solve_async(); bt.sync(); ms.append(bt.last_solve_ms())
    r = bt.results()
    print("cfg", cfg, "B", B, "reference order: kernel ms", np.round(ms, 2), "solves/s", round(B / (min(ms) * 1e-3), 1), "max iters", int(r["iters"].max()),
          "us/iter of the longest", round(1e3 * min(ms) / r["iters"].max(), 1), flush=True)
    bt.set_order(capi.ORDER_DEVICE)
    bt.solve_async(); bt.sync(); bt.solve_async(); bt.sync()
    print("         device order: kernel ms", round(bt.last_solve_ms(), 2))
    bt.close(); h.close()
