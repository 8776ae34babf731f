"""Developer script: the in-kernel phase profile of the SLOWEST trajectory of a reference-order batch (the one an isolated batch waits for).
   scripts/slowest_profile.py cfg B"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from dftpav_amd import capi, scenarios as sc
cfg, B = int(sys.argv[1]), int(sys.argv[2])
p = capi.default_params()
s = sc.baseline_config(cfg, B=B); s.apply_resolution(p)
h = capi.Handle(p); h.set_surround(s.surround)
bt = capi.Batch(h, s.layout, B); bt.upload(s)
bt.set_order(capi.ORDER_REFERENCE)
bt.solve_async(); bt.sync()
bt.profile(True)
bt.solve_async(); bt.sync()
r = bt.results()
pr = bt.read_profile().astype(np.float64)
NAMES = ["E1 rhs", "E2 coeffs", "E3+E4 samples", "E4 reduce", "E5 adjoint", "E6 assemble", "line search misc", "history update", "two-loop", "(count) active terms", "numbering", "window list"]
for b in (int(np.argmax(r["latency_us"])), int(np.argsort(r["latency_us"])[B // 2])):
    print("trajectory", b, "latency ms", r["latency_us"][b] / 1e3, "iters", r["iters"][b], "evals", r["evals"][b], "active terms per eval", pr[b, 9] / r["evals"][b])
    for i, nm in enumerate(NAMES):
        if i == 9: continue
        per = pr[b, i] / (r["evals"][b] if i < 6 or i >= 10 else r["iters"][b])
        print("   %-18s %10.0f cycles per %s" % (nm, per, "eval" if i < 6 or i >= 10 else "iter"))
print("kernel ms", bt.last_solve_ms())
