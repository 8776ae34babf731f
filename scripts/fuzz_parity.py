"""Developer script: GPU solve against the device-order oracle on randomly shaped problems (run through gpurun).

Random layouts (1-3 gear segments of 2-12 pieces, sample resolutions 3-24, with and without moving obstacles),
every launch shape, small batches; every field of the result must be bit-identical.
  python scripts/fuzz_parity.py [n_cases] [first_seed] [moving]     (a third argument: every case has moving obstacles)"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from dftpav_amd import capi, scenarios as sc
from oracle import pyoracle as po

n_cases = int(sys.argv[1]) if len(sys.argv) > 1 else 60
seed0 = int(sys.argv[2]) if len(sys.argv) > 2 else 0
force_moving = len(sys.argv) > 3
keys = ("final_cost", "x", "status", "iters", "evals", "hist_sum", "success")
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
    B = int(rng.integers(1, 7))
    moving = bool(rng.uniform() < 0.2) and sum(pieces) <= 12
    if force_moving:
        while sum(pieces) > 12:
            pieces[int(np.argmax(pieces))] -= 1
        moving = True
    mode = int(rng.choice([0, 1, 2]))
    os.environ["DFTPAV_MODE"] = str(mode)
    p = capi.default_params()
    s = sc.make_scenario(pieces, sing, K, Kd, B, seed=5000 + seed0 + c, with_moving=moving, n_obs=int(rng.integers(0, 60)))
    s.apply_resolution(p)
    if rng.uniform() < 0.3:
        p.lbfgs_mem_size = int(rng.choice([4, 8, 17, 64]))
    if rng.uniform() < 0.4:  # tighter limits and other weights: every penalty branch gets traffic
        p.max_forward_vel *= float(rng.uniform(0.3, 1.0)); p.max_backward_vel *= float(rng.uniform(0.3, 1.0))
        p.max_forward_acc *= float(rng.uniform(0.2, 1.0)); p.max_backward_acc *= float(rng.uniform(0.2, 1.0))
        p.max_forward_cur *= float(rng.uniform(0.2, 1.0)); p.max_backward_cur *= float(rng.uniform(0.2, 1.0))
        p.wei_obs *= float(rng.uniform(0.1, 10)); p.wei_feas *= float(rng.uniform(0.1, 10)); p.wei_time *= float(rng.uniform(0.1, 10))
    if rng.uniform() < 0.3:
        s.help_eps = float(rng.choice([1e-3, 0.05]))
    if moving:
        s.t_now = float(rng.uniform(0.0, 5.0))
    h = capi.Handle(p); h.set_surround(s.surround)
    bt = capi.Batch(h, s.layout, B); bt.upload(s)
    r = bt.solve()
    ro = po.solve_batch(p, s, order=1)
    ok = all(np.array_equal(r[k], ro[k]) for k in keys)
    for st in r["status"]:
        stat[int(st)] = stat.get(int(st), 0) + 1
    if not ok:
        bad += 1
        print("MISMATCH case %d: pieces %s singuls %s K %d Kd %d B %d moving %s mode %d mem %d" %
              (c, pieces, sing, K, Kd, B, moving, mode, p.lbfgs_mem_size), flush=True)
    bt.close(); h.close()
print("%d cases, %d mismatches, %.1f s; solver status counts %s" % (n_cases, bad, time.time() - t0, stat))
sys.exit(1 if bad else 0)
