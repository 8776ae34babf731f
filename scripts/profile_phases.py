"""Developer script: in-kernel phase breakdown of the solve kernel (shader clocks of thread 0)."""
import sys, os, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from dftpav_amd import capi, scenarios as sc

NAMES = ["E1 rhs", "E2 coeffs", "E3+E4 samples", "E4 reduce", "E5 adjoint", "E6 assemble", "line search misc",
         "history update", "two-loop", "x (experiments)", "init", "misc (experiments)"]
cfg = int(sys.argv[1]) if len(sys.argv) > 1 else 3
B = int(sys.argv[2]) if len(sys.argv) > 2 else 256
p = capi.default_params()
s = sc.baseline_config(cfg, B=B); s.apply_resolution(p)
h = capi.Handle(p); h.set_surround(s.surround)
bt = capi.Batch(h, s.layout, B); bt.upload(s)
if os.environ.get("ORDER") == "ref":   # the reference-order kernel (solver_ref.hip): slots 0..8 as its Prof says
    bt.set_order(capi.ORDER_REFERENCE)
bt.solve_async(); bt.sync()
ms0 = []
for _ in range(3):
    bt.solve_async(); bt.sync(); ms0.append(bt.last_solve_ms())
bt.profile(True)
bt.solve_async(); bt.sync(); ms1 = bt.last_solve_ms()
r = bt.results()
pr = bt.read_profile().astype(np.float64)
REF = os.environ.get("ORDER") == "ref"
# the reference-order kernel keeps COUNTS in some slots (solver_ref.hip): slot 9 = active terms (both launch shapes); in the WAVE
# shape slots 10 / 11 = evaluations beyond the LDS window and their terms, in the TEAM shape they are cycles (numbering, the
# window's list).  Counts are kept out of the cycle totals and printed under their own names.
wave_counts = REF and bool((pr[:, 10] <= r["evals"]).all())
count_slots = ([9, 10, 11] if wave_counts else [9]) if REF else []
if REF:
    NAMES[9], NAMES[10], NAMES[11] = "(count) active terms", "numbering" if not wave_counts else "(count)", "window list" if not wave_counts else "(count)"
cyc = pr.copy()
cyc[:, count_slots] = 0.0
tot = cyc.sum(axis=1)
print("cfg", cfg, "B", B, "kernel ms (no prof)", np.round(ms0, 3), "with prof", round(ms1, 3))
print("iters mean/max", r["iters"].mean(), r["iters"].max(), "evals mean/max", r["evals"].mean(), r["evals"].max(),
      "latency ms p50/max", np.median(r["latency_us"]) / 1e3, r["latency_us"].max() / 1e3)
ghz = tot / (r["latency_us"] * 1e3)
print("shader clock GHz (cycles/latency):", round(float(np.median(ghz)), 3))
ev, it = r["evals"].astype(float), r["iters"].astype(float)
for i, nm in enumerate(NAMES):
    if cyc[:, i].sum() == 0: continue
    per_eval = i < 6 or (REF and i in (10, 11)) or (not REF and i in (9, 11))
    per = cyc[:, i] / (ev if per_eval else it)
    print("%-18s %6.1f%%   %9.0f cycles per %s" % (nm, 100 * cyc[:, i].sum() / tot.sum(), np.median(per), "eval" if per_eval else "iter"))
hs = r["hist_sum"].astype(float)
if REF:
    print("reference order: active terms per evaluation, mean", round(float(pr[:, 9].sum() / ev.sum()), 1))
    if wave_counts:
        big = pr[:, 10].sum()
        print("   evaluations with more terms than the LDS window holds:", round(float(100 * big / ev.sum()), 1), "% of all, with",
              round(float(pr[:, 11].sum() / max(1.0, big)), 1), "active terms on average")
print("two-loop cycles per history step (2 per entry per iteration):", round(float(np.median(pr[:, 8] / (2 * hs))), 1),
      " mean depth", round(float((hs / it).mean()), 1))
print("solves/s (kernel)", B / (np.mean(ms0) * 1e-3))
