"""Developer script (CPU): the dense direction (oracle order 3, csrc/dense_dir.h) on randomly shaped problems -- random layouts
(1-3 gear segments, 2-10 pieces each, n <= 64), resolutions, obstacle counts, moving obstacles, limits / weights, and memories
1 .. 300: every direction against the plain two-loop recursion over the same window in 80-bit arithmetic (oracle.dense_check; bar 1e-5:
the gate of dense_dir.h hands ill-conditioned windows to the plain recursion, whose own fp64 error reaches 2.5e-6 there),
every solve must succeed or end as the two-loop device order's does.
    python scripts/fuzz_dense_cpu.py [n_cases] [first_seed]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from dftpav_amd import scenarios as sc
from oracle import pyoracle as po
from dense_cases import make_case

n_cases = int(sys.argv[1]) if len(sys.argv) > 1 else 100
seed0 = int(sys.argv[2]) if len(sys.argv) > 2 else 0
po.build()
worst, ndir, nfront, nsolve, bad, nclosed, t0 = 0.0, 0, 0, 0, 0, 0, time.time()
status = {}
for c in range(n_cases):
    p, s, pieces = make_case(seed0 + c)
    B = s.B
    po.dense_check(True)
    for b in range(B):
        _, r3 = po.OracleProblem(p, s, b, order=3).solve()
        _, r1 = po.OracleProblem(p, s, b, order=1).solve()
        nsolve += 1
        status[r3.status] = status.get(r3.status, 0) + 1
        if bool(r3.success) != bool(r1.success) and r3.status < 0 and r1.status >= 0:
            bad += 1
            print("case %d b %d: order 3 ends with %d where order 1 ends with %d (pieces %s mem %d)" % (c, b, r3.status, r1.status, pieces, p.lbfgs_mem_size), flush=True)
    st = po.dense_stats()
    po.dense_check(False)
    worst = max(worst, st["max_rel_d"]); ndir += st["directions"]; nfront += st["with_front"]; nclosed += st["gate_closed"]
    if st["max_rel_d"] > 1e-5:
        bad += 1
        print("case %d: max rel d %.3e (pieces %s mem %d)" % (c, st["max_rel_d"], pieces, p.lbfgs_mem_size), flush=True)
print("%d cases, %d solves, %d directions (%d with a front aggregate, %d from the plain recursion: gate closed), largest relative difference to the 80-bit two-loop recursion %.3e, %d failures, %.0f s; status counts %s" %
      (n_cases, nsolve, ndir, nfront, nclosed, worst, bad, time.time() - t0, status))
sys.exit(1 if bad else 0)
