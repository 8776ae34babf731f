"""Developer script: the QUAD shape for several gear segments (solver_ref4m.hip) against the restatement on randomly shaped problems (run
through gpurun).  Two to four gear segments of 2-8 pieces, 16 pieces and 48 variables at most, sample resolutions 3-24 (inner and end
pieces apart), 1-60 trajectories, random limits / weights / L-BFGS memories (3 ... 300) / stopping rules / help_eps (the generic
instantiation) / gear_opt, 1-3 persistent waves of 1, 2 or 4 per workgroup, slices of 2-80 evaluations, with and without the hand-over of
the last trajectories to the WAVE shape: every field of every solve and a random evaluation must be bit-identical to the program with
correctly rounded cos / sin (oracle order 2).
  python scripts/fuzz_quadm.py [n_cases] [first_seed]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from dftpav_amd import capi, scenarios as sc
from oracle import pyoracle as po

n_cases = int(sys.argv[1]) if len(sys.argv) > 1 else 60
seed0 = int(sys.argv[2]) if len(sys.argv) > 2 else 0
keys = ("final_cost", "x", "status", "iters", "evals", "hist_sum", "success")
bad = refused = solves = 0
t0 = time.time()
stat = {}
for c in range(n_cases):
    rng = np.random.default_rng(71000 + seed0 + c)
    M = int(rng.integers(2, 5))
    pieces = [int(rng.integers(2, 9)) for _ in range(M)]
    while sum(pieces) > 16 or 2 * (sum(pieces) - M) + M + 3 * (M - 1) > 48:
        pieces[int(np.argmax(pieces))] -= 1
    sing = [int(rng.choice([1, -1]))]
    for _ in range(M - 1):
        sing.append(-sing[-1])
    K = int(rng.integers(3, 25)); Kd = int(rng.integers(3, 25))
    B = int(rng.integers(1, 61))
    os.environ["DFTPAV_REF_SHAPE"] = "quad"
    os.environ["DFTPAV_REF_QUAD_WAVES"] = str(int(rng.choice([1, 2, 4])))
    os.environ["DFTPAV_REF_SLOTS"] = str(int(rng.integers(1, 4)))
    os.environ["DFTPAV_REF_SLICE"] = str(int(rng.integers(2, 81)))
    hand = int(rng.choice([-1, 0]))
    p = capi.default_params()
    s = sc.make_scenario(pieces, sing, K, Kd, B, seed=73000 + seed0 + c, with_moving=False, n_obs=int(rng.integers(0, 60)))
    s.apply_resolution(p)
    if rng.uniform() < 0.4:
        p.lbfgs_mem_size = int(rng.choice([3, 4, 8, 17, 64, 300]))
    if rng.uniform() < 0.4:
        p.max_forward_vel *= float(rng.uniform(0.3, 1.0)); p.max_backward_vel *= float(rng.uniform(0.3, 1.0))
        p.max_forward_acc *= float(rng.uniform(0.2, 1.0)); p.max_backward_acc *= float(rng.uniform(0.2, 1.0))
        p.max_forward_cur *= float(rng.uniform(0.2, 1.0)); p.max_backward_cur *= float(rng.uniform(0.2, 1.0))
        p.wei_obs *= float(rng.uniform(0.1, 10)); p.wei_feas *= float(rng.uniform(0.1, 10)); p.wei_time *= float(rng.uniform(0.1, 10))
    if rng.uniform() < 0.2:
        p.gear_opt = 0
    if rng.uniform() < 0.3:
        s.help_eps = float(rng.choice([1e-3, 0.05]))
    if rng.uniform() < 0.2:
        p.lbfgs_past, p.lbfgs_delta = int(rng.integers(1, 7)), float(10.0 ** rng.uniform(-6, -3))
    h = capi.Handle(p)
    bt = capi.Batch(h, s.layout, B); bt.upload(s)
    try:
        bt.set_order(capi.ORDER_REFERENCE)
    except capi.DftpavError:
        refused += 1
        bt.close(); h.close()
        continue
    bt.set_hand_over(hand)
    x = bt.x0() + rng.normal(0, float(rng.choice([0.02, 0.3])), bt.x0().shape)
    f, g = bt.eval(x)
    ok = True
    for b in range(0, B, 7):
        fo, go = po.OracleProblem(p, s, b, order=2).eval(x[b])
        ok = ok and f[b] == fo and np.array_equal(g[b], go)
    r = bt.solve()
    w = po.solve_batch(p, s, nthreads=4, order=2)
    ok = ok and all(np.array_equal(r[k], w[k]) for k in keys)
    solves += B
    for st in r["status"]:
        stat[int(st)] = stat.get(int(st), 0) + 1
    if not ok:
        bad += 1
        print("MISMATCH case %d: pieces %s singul %s K %d Kd %d B %d waves %s slots %s slice %s hand-over %d mem %d eps %g past %d" %
              (c, pieces, sing, K, Kd, B, os.environ["DFTPAV_REF_QUAD_WAVES"], os.environ["DFTPAV_REF_SLOTS"], os.environ["DFTPAV_REF_SLICE"], hand,
               p.lbfgs_mem_size, s.help_eps, p.lbfgs_past), flush=True)
    bt.close(); h.close()
print("%d cases (%d refused by the mode), %d solves, %d mismatches, %.1f s; solver status counts %s" % (n_cases, refused, solves, bad, time.time() - t0, stat))
sys.exit(1 if bad else 0)
