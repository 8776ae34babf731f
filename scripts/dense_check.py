"""Developer script (GPU box) for the EXPERIMENTAL dense direction (csrc/dense_dir.h, dftpav_debug_set_direction): the first
thing to run on a GPU in round 5 -- the device path was written and compiled in round 4 without one.
  1. parity: whole solves with the dense direction against oracle order 3, every field bit for bit, on small batches of
     BASELINE configs 1, 2, 3, 5 and with small memories (window sliding, rebuilds);
  2. time: the bench's batch (config 3, 4096) isolated, two-loop recursion against dense, with the phase profile.
python scripts/dense_check.py [parity|time|all]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import numpy as np
from dftpav_amd import capi, scenarios as sc
from oracle import pyoracle as po

what = sys.argv[1] if len(sys.argv) > 1 else "all"
KEYS = ("final_cost", "x", "status", "iters", "evals", "hist_sum", "success")

def batch(p, s, dense, residency=None):
    h = capi.Handle(p)
    h.set_surround(s.surround)
    bt = capi.Batch(h, s.layout, s.B) if residency is None else capi.Batch(h, s.layout, s.B, residency=residency)
    bt.upload(s)
    if dense:
        bt.debug_set_direction(True)
    return h, bt

if what in ("parity", "all"):
    po.build()
    bad = 0
    for cfg, B, mem in [(3, 16, 256), (3, 8, 32), (3, 8, 8), (3, 4, 1), (2, 8, 256), (1, 8, 256), (5, 4, 256), (2, 6, 17), (3, 64, 256)]:
        p = capi.default_params(); p.lbfgs_mem_size = mem
        s = sc.baseline_config(cfg, B=B); s.apply_resolution(p)
        h, bt = batch(p, s, True)
        r = bt.solve()
        want = po.solve_batch(p, s, nthreads=2, order=3)
        eq = {k: bool(np.array_equal(r[k], want[k])) for k in KEYS}
        nsame = int(sum(bool(r["final_cost"][i] == want["final_cost"][i] and np.array_equal(r["x"][i], want["x"][i])) for i in range(B)))
        print("cfg %d B %d mem %d: %d / %d solves bit-equal to oracle order 3; fields %s; iters max %d" % (cfg, B, mem, nsame, B, eq, int(r["iters"].max())), flush=True)
        bad += nsame != B
        bt.close(); h.close()
    # layouts whose windows close the conditioning gate (the plain recursion takes those iterations)
    from dense_cases import make_case
    for index in (1615, 2454, 3927, 3542):
        p, s, pieces = make_case(index)
        h, bt = batch(p, s, True)
        r = bt.solve()
        want = po.solve_batch(p, s, nthreads=2, order=3)
        ok = all(np.array_equal(r[k], want[k]) for k in KEYS)
        print("gate case %d (pieces %s, mem %d): %s" % (index, pieces, p.lbfgs_mem_size, "bit-equal" if ok else "DIFFERENT"), flush=True)
        bad += not ok
        bt.close(); h.close()
    # the stored vectors (tests/golden/dense.npz, written by the oracle in round 4)
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
    from golden_util import GOLDEN_DIR, load
    z = np.load(os.path.join(GOLDEN_DIR, "dense.npz"))
    for name, mem in (("cfg1", 256), ("cfg2", 256), ("cfg3", 256), ("cfg3", 8), ("cfg5", 256)):
        s, _ = load(name)
        p = capi.default_params(); p.lbfgs_mem_size = mem
        s.apply_resolution(p)
        h, bt = batch(p, s, True)
        r = bt.solve()
        ok = all(np.array_equal(r[k], z["%s_m%d_%s" % (name, mem, k)]) for k in KEYS)
        print("golden %s mem %d: %s" % (name, mem, "bit-equal" if ok else "DIFFERENT"), flush=True)
        bad += not ok
        bt.close(); h.close()
    print("PARITY", "OK" if not bad else "FAILED (%d cases)" % bad)

if what in ("time", "all"):
    p = capi.default_params()
    s = sc.baseline_config(3, B=4096, seed=20240); s.apply_resolution(p)
    for dense in (False, True):
        h, bt = batch(p, s, dense)
        bt.solve_async(); bt.sync()
        ms = []
        for _ in range(3):
            bt.solve_async(); bt.sync(); ms.append(bt.last_solve_ms())
        r = bt.results()
        print("%s: isolated 4096 x config 3: %s ms -> %.0f solves/s; mean iters %.1f evals %.1f success %.3f median cost %.2f" %
              ("dense   " if dense else "two-loop", np.round(ms, 1), 4096 / (np.mean(ms) * 1e-3), r["iters"].mean(), r["evals"].mean(), r["success"].mean(), np.median(r["final_cost"])), flush=True)
        bt.profile(True); bt.solve_async(); bt.sync()
        pr = bt.read_profile().astype(np.float64); tot = pr.sum()
        names = ["E1", "E2", "E3+E4 samples", "E4 reduce", "E5", "E6", "LS misc", "hist", "direction", "init"]
        print("   phases: " + " ".join("%s %.1f%%" % (n, 100 * pr[:, i].sum() / tot) for i, n in enumerate(names)), "| direction cycles per iteration %.0f" % (pr[:, 8].sum() / r["iters"].sum()))
        bt.close(); h.close()
    # one trajectory alone (configs[1]): the latency case
    for dense in (False, True):
        ms, its = [], []
        for sd in range(5):
            s1 = sc.baseline_config(2, B=1, seed=20240 + 17 * sd); s1.apply_resolution(p)
            h, bt = batch(p, s1, dense)
            bt.solve_async(); bt.sync(); bt.solve_async(); bt.sync()
            ms.append(bt.last_solve_ms()); its.append(int(bt.results()["iters"][0]))
            bt.close(); h.close()
        print("%s: one gear-shift trajectory: %.1f us per iteration" % ("dense   " if dense else "two-loop", 1e3 * sum(ms) / sum(its)))
