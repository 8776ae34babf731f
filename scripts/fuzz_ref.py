"""Developer script (CPU): the literal oracle against the reference build oracle/_ref on randomly shaped problems.

Random layouts (1-3 gear segments of 2-12 pieces, sample resolutions 3-24, moving obstacles, random limits / weights /
help_eps / obstacle clock / L-BFGS memory); x0, f, g at x0, and every field of the whole solve must be bit-identical.
  python scripts/fuzz_ref.py [n_cases] [first_seed] [cr]
cr: oracle order 2 (correctly rounded cos / sin / exp / log / x^3 through binary128) against oracle/_ref/libdftpav_ref_cr.so, the
reference's own objects linked against the same correctly rounded functions (oracle/cr_libm.c) -- the pin of the order the
device's reference-order kernel implements; half of the cases then carry moving obstacles."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from dftpav_amd import scenarios as sc
from oracle import pyoracle as po, pyref as pr

n_cases = int(sys.argv[1]) if len(sys.argv) > 1 else 60
seed0 = int(sys.argv[2]) if len(sys.argv) > 2 else 0
CR = len(sys.argv) > 3 and sys.argv[3] == "cr"
bad = 0
t0 = time.time()
stat = {}
for c in range(n_cases):
    rng = np.random.default_rng(1000 + seed0 + c)
    M = int(rng.choice([1, 1, 2, 3]))
    pieces = [int(rng.integers(2, 13)) for _ in range(M)]
    sing = [int(rng.choice([1, -1]))]
    for _ in range(M - 1):
        sing.append(-sing[-1])
    K = int(rng.integers(3, 25)); Kd = int(rng.integers(3, 25))
    B = int(rng.integers(1, 4))
    moving = bool(rng.uniform() < (0.5 if CR else 0.3)) and sum(pieces) <= 12
    p = po.default_params()
    s = sc.make_scenario(pieces, sing, K, Kd, B, seed=5000 + seed0 + c, with_moving=moving, n_obs=int(rng.integers(0, 60)))
    s.apply_resolution(p)
    if rng.uniform() < 0.3:
        p.lbfgs_mem_size = int(rng.choice([4, 8, 17, 64]))
    if rng.uniform() < 0.4:
        p.max_forward_vel *= float(rng.uniform(0.3, 1.0)); p.max_backward_vel *= float(rng.uniform(0.3, 1.0))
        p.max_forward_acc *= float(rng.uniform(0.2, 1.0)); p.max_backward_acc *= float(rng.uniform(0.2, 1.0))
        p.max_forward_cur *= float(rng.uniform(0.2, 1.0)); p.max_backward_cur *= float(rng.uniform(0.2, 1.0))
        p.wei_obs *= float(rng.uniform(0.1, 10)); p.wei_feas *= float(rng.uniform(0.1, 10)); p.wei_time *= float(rng.uniform(0.1, 10))
    if rng.uniform() < 0.3:
        s.help_eps = float(rng.choice([1e-3, 0.05]))
    if moving:
        s.t_now = float(rng.uniform(0.0, 5.0))
    for b in range(B):
        o = po.OracleProblem(p, s, b, order=2 if CR else 0)
        r = pr.RefProblem(p, s, b, cr=CR)
        rr = r.optimize(trace=True)
        xo, ro = o.solve()
        x0 = o.x0()
        fo, go = o.eval(x0)
        fr, gr = r.eval(x0)
        ok = (np.array_equal(rr["eval_x"][0], x0) and fo == fr and np.array_equal(go, gr) and np.array_equal(xo, rr["x"])
              and ro.final_cost == rr["final_cost"] and ro.status == rr["status"] and ro.iters == rr["iters"]
              and ro.evals == rr["evals"] and bool(ro.success) == rr["ok"])
        stat[rr["status"]] = stat.get(rr["status"], 0) + 1
        if not ok:
            bad += 1
            print("MISMATCH case %d b %d: pieces %s singuls %s K %d Kd %d moving %s mem %d eps %g | f0 %r %r | cost %r %r iters %d %d evals %d %d status %d %d" %
                  (c, b, pieces, sing, K, Kd, moving, p.lbfgs_mem_size, s.help_eps, fo, fr, ro.final_cost, rr["final_cost"], ro.iters, rr["iters"],
                   ro.evals, rr["evals"], ro.status, rr["status"]), flush=True)
print("%s: %d cases, %d mismatches, %.1f s; solver status counts %s" % ("order 2 vs _ref on the correctly rounded libm" if CR else "order 0 vs _ref", n_cases, bad, time.time() - t0, stat))
sys.exit(1 if bad else 0)
