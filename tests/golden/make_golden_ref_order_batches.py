"""Golden vectors for the reference order AT THE BASELINE BATCH SIZES: BASELINE configs[4] (32 pieces x 65
points, 4 moving cars) at batch 1024 and configs[1] (gear shift, 8 + 8 pieces) at batch 4096, seed 20240 (bench.py's default).
The GPU box has two usable host cores and the order-2 restatement runs seconds per configs[4] solve, so the expected results of
the SAMPLED trajectories are computed here (8 cores) and travel as a fixture:
  * 64 sampled trajectories per batch by oracle order 2 (the reference's program with correctly rounded libm calls);
  * the first 8 of them ALSO by the reference's own objects on the correctly rounded libm (oracle/_ref/libdftpav_ref_cr.so):
    stored only if they agree with order 2 bit for bit (asserted here).
    python tests/golden/make_golden_ref_order_batches.py"""
import os
import sys
from multiprocessing import Pool

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
KEYS = ("final_cost", "x", "status", "iters", "evals")
CASES = {"cfg5_b1024": (5, 1024), "cfg2_b4096": (2, 4096)}
SEED, NS, NCR = 20240, 64, 8


def picks(B):
    return (np.arange(NS) * (B // NS) + (B // NS) // 2) % B


def solve(job):
    name, b, with_cr = job
    from dftpav_amd import scenarios as sc
    from oracle import pyoracle as po, pyref as pr
    cfg, B = CASES[name]
    p = po.default_params()
    s = sc.baseline_config(cfg, B=B, seed=SEED)
    s.apply_resolution(p)
    o = po.solve_batch(p, s.subset(np.array([b])), nthreads=1, order=2)
    rec = {k: o[k][0] for k in KEYS}
    if with_cr:
        r = pr.RefProblem(p, s, int(b), cr=True).optimize()
        assert r["final_cost"] == rec["final_cost"] and np.array_equal(r["x"], rec["x"]), (name, b)
        assert (r["status"], r["iters"], r["evals"]) == (rec["status"], rec["iters"], rec["evals"]), (name, b)
    return name, int(b), rec


def main():
    from oracle import pyoracle as po, pyref as pr
    po.build()
    pr.build()
    jobs = []
    for name, (cfg, B) in CASES.items():
        for i, b in enumerate(picks(B)):
            jobs.append((name, int(b), i < NCR))
    with Pool(8) as pool:
        res = pool.map(solve, jobs, chunksize=1)
    out = {"seed": np.array(SEED), "n_checked_against_the_reference_objects_on_a_correctly_rounded_libm": np.array(NCR)}
    for name, (cfg, B) in CASES.items():
        pk = picks(B)
        rows = {b: rec for nm, b, rec in res if nm == name}
        out[name + "_pick"] = pk
        for k in KEYS:
            out[name + "_" + k] = np.array([rows[int(b)][k] for b in pk])
        print(name, "iters", out[name + "_iters"].min(), out[name + "_iters"].max(), "median cost", np.median(out[name + "_final_cost"]))
    np.savez_compressed(os.path.join(os.path.dirname(os.path.abspath(__file__)), "ref_order_batches.npz"), **out)


if __name__ == "__main__":
    main()
