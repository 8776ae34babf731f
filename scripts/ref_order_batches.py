"""Developer script (GPU box): the reference order on the BASELINE configurations AT THEIR OWN BATCH SIZES -- configs[4] (1024
trajectories among the moving cars) and configs[1]'s gear shift at 4096 -- kernel time, roofline on algorithmic bytes, and the
64 sampled trajectories of tests/golden/ref_order_batches.npz.  DFTPAV_LIB selects a library variant (e.g. the 40-term kernel
built for 256 registers: scripts/build_ref_variant.sh ... -DDFTPAV_REF_NARROW_CAP=40).
    python scripts/ref_order_batches.py [cfg5_b1024] [cfg2_b4096]"""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from dftpav_amd import capi, scenarios as sc

Z = np.load(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden", "ref_order_batches.npz"))
CASES = {"cfg5_b1024": (5, 1024), "cfg2_b4096": (2, 4096)}
for name in sys.argv[1:] or list(CASES):
    cfg, B = CASES[name]
    p = capi.default_params()
    s = sc.baseline_config(cfg, B=B, seed=int(Z["seed"])); s.apply_resolution(p)
    h = capi.Handle(p); h.set_surround(s.surround)
    bt = capi.Batch(h, s.layout, B); bt.upload(s)
    bt.set_order(capi.ORDER_REFERENCE)
    bt.solve_async(); bt.sync()
    ms = []
    for _ in range(2):
        bt.solve_async(); bt.sync(); ms.append(bt.last_solve_ms())
    r = bt.results()
    pk = Z[name + "_pick"]
    ok = {k: bool(np.array_equal(r[k][pk], Z[name + "_" + k])) for k in ("final_cost", "x", "status", "iters", "evals")}
    lay = s.layout
    n = lay.n_vars
    e_eval = (s.n_points * lay.H * 4 + 2 * n + 12 * lay.M + 1) * 8.0
    ab = float((r["evals"] * e_eval + (4.0 * r["hist_sum"] * n + 14.0 * n * r["iters"]) * 8.0).sum())
    print("%s [%s]: kernel ms %s -> %.0f solves/s; roofline (algorithmic bytes %.1f GB) %.3f of 8 TB/s; mean / max iters %.1f / %d; %d sampled bit-equal: %s" %
          (name, os.path.basename(os.environ.get("DFTPAV_LIB", "libdftpav_hip.so")), np.round(ms, 1), B / (min(ms) * 1e-3), ab / 1e9, ab / (min(ms) * 1e-3) / 8e12,
           r["iters"].mean(), int(r["iters"].max()), len(pk), all(ok.values()) or ok), flush=True)
    bt.close(); h.close()
