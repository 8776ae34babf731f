"""Diagnostic of the reference-order device mode: evaluations and solves against the literal oracle, with timings.
Run on a GPU box: python scripts/ref_order_diag.py [cfg ...]"""
import sys, time
import numpy as np
import os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from dftpav_amd import capi, scenarios as sc
from oracle import pyoracle as po

def run(cfg, B):
    p = capi.default_params()
    s = sc.baseline_config(cfg, B=B)
    s.apply_resolution(p)
    h = capi.Handle(p)
    bt = capi.Batch(h, s.layout, s.B)
    bt.upload(s)
    bt.set_order(capi.ORDER_REFERENCE)
    x0 = bt.x0()
    rng = np.random.default_rng(1)
    for scale in (0.0, 0.05, 0.7):
        x = x0 + rng.normal(0, scale, x0.shape) if scale else x0
        f, g = bt.eval(x)
        nb = 0
        for b in range(B):
            lit = po.OracleProblem(p, s, b, order=0)
            fl, gl = lit.eval(x[b])
            ok = (f[b] == fl) and np.array_equal(g[b], gl)
            nb += ok
            if not ok and b < 3:
                bad = np.nonzero(g[b] != gl)[0]
                print("  cfg", cfg, "scale", scale, "b", b, "f", f[b], fl, "rel", abs(f[b] - fl) / abs(fl), "g mismatches", len(bad), "of", len(gl),
                      "max rel", np.abs(g[b] - gl).max() / np.abs(gl).max(), "first", bad[:8], "terms", lit.cost_terms())
        print("cfg", cfg, "scale", scale, "evaluations bit-equal:", nb, "/", B, flush=True)
    t0 = time.time(); r = bt.solve(); t1 = time.time()
    ms_ref = bt.last_solve_ms()
    lit = po.solve_batch(p, s, nthreads=8, order=0)
    eq = [(r["final_cost"][b] == lit["final_cost"][b]) and np.array_equal(r["x"][b], lit["x"][b]) and r["iters"][b] == lit["iters"][b]
          and r["evals"][b] == lit["evals"][b] and r["status"][b] == lit["status"][b] for b in range(B)]
    print("cfg", cfg, "solves bit-equal:", sum(eq), "/", B, "iters", r["iters"][:6], lit["iters"][:6], "kernel ms", ms_ref,
          "us/iter of the longest", 1e3 * ms_ref / max(1, r["iters"].max()))
    bt.set_order(capi.ORDER_DEVICE)
    rd = bt.solve(); ms_dev = bt.last_solve_ms()
    print("cfg", cfg, "device order: kernel ms", ms_dev, "us/iter of the longest", 1e3 * ms_dev / max(1, rd["iters"].max()), flush=True)
    bt.close(); h.close()

if __name__ == "__main__":
    cfgs = [int(a) for a in sys.argv[1:]] or [1, 3]
    for c in cfgs:
        run(c, 8)
