import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from dftpav_amd import capi, scenarios as sc
p = capi.default_params()
for B in (256, 4096):
    s = sc.baseline_config(3, B=B); s.apply_resolution(p)
    h = capi.Handle(p); bt = capi.Batch(h, s.layout, B)
    t0 = time.perf_counter(); bt.upload(s); t1 = time.perf_counter()
    bt.solve_async(); bt.sync()
    t2 = time.perf_counter(); bt.solve_async(); bt.sync(); t3 = time.perf_counter()
    r = bt.results(); t4 = time.perf_counter()
    nbytes = s.corridor.nbytes + s.inner_pts.nbytes + s.ini_states.nbytes * 2 + s.init_Ts.nbytes
    print("B", B, "input MB", nbytes / 1e6, "upload s", t1 - t0, "solve s", t3 - t2, "download s", t4 - t3,
          "solves/s incl upload+download", B / ((t1 - t0) + (t3 - t2) + (t4 - t3)))
    bt.close(); h.close()
