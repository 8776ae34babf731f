"""Developer script: reference-order GPU solves against the literal oracle on randomly shaped problems (run through gpurun).

Random layouts inside what the mode accepts (1-3 gear segments, n <= 64, sample resolutions 3-24, with and without moving
obstacles -- on multi-segment layouts too, the reference's live call -- and with up to 12 of them), random limits / weights /
L-BFGS memories, random launch shapes (a workgroup per trajectory of 128-256 threads, or a wave per trajectory with 1-2
persistent workgroups and slices of 2-40 iterations through the ring); every field of every
solve must be bit-identical to the literal program with correctly rounded libm functions (oracle order 2), and on static
single-segment layouts to the literal program with this host's libm (order 0, the one tests/test_ref_pin.py pins to the
reference build) as well.
  python scripts/fuzz_reference_order.py [n_cases] [first_seed]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from dftpav_amd import capi, scenarios as sc
from oracle import pyoracle as po

n_cases = int(sys.argv[1]) if len(sys.argv) > 1 else 60
seed0 = int(sys.argv[2]) if len(sys.argv) > 2 else 0
only = int(sys.argv[3]) if len(sys.argv) > 3 else None     # a third argument: that case alone, with a diagnosis of a mismatch
keys = ("final_cost", "x", "status", "iters", "evals", "hist_sum", "success")
bad = refused = 0
t0 = time.time()
stat = {}


def diagnose(p, s, bt, r, w):
    """where a solve leaves the literal program: the first evaluation of each differing trajectory whose bits differ"""
    for b in range(s.B):
        if all(np.array_equal(r[k][b], w[k][b]) for k in keys):
            continue
        print("  traj %d: status %d / %d, iters %d / %d, evals %d / %d, cost %r / %r" %
              (b, r["status"][b], w["status"][b], r["iters"][b], w["iters"][b], r["evals"][b], w["evals"][b], r["final_cost"][b], w["final_cost"][b]))
        # the literal L-BFGS driven from here over the literal evaluator records every point it evaluates; the device evaluates
        # the points around the one its solve stopped or turned at
        o = po.OracleProblem(p, s, b, order=2)
        xs = []
        def fn(x):
            xs.append(x.copy())
            return o.eval(x)
        lr = po.lbfgs(fn, o.x0(), p)
        assert lr["evals"] == w["evals"][b] and np.array_equal(lr["x"], w["x"][b])
        e0 = int(min(r["evals"][b], w["evals"][b]))
        for e in range(max(0, e0 - 3), min(len(xs), e0 + 3)):
            X = np.repeat(xs[e][None], s.B, axis=0)
            f, g = bt.eval(X)
            fo, go = o.eval(xs[e])
            nb = int((g[b] != go).sum())
            print("    evaluation %d: device f %r literal %r, %d gradient entries differ, |x| max %.3e, device g nan %s" %
                  (e, f[b], fo, nb, np.abs(xs[e]).max(), np.isnan(g[b]).any()))
            if f[b] != fo or nb:
                os.makedirs("gpurun_out/fz", exist_ok=True)
                np.save("gpurun_out/fz/x_case.npy", xs[e])
                print("      x =", repr(xs[e].tolist()))
                print("      literal terms", o.cost_terms())


for c in range(n_cases):
    if only is not None and c != only:
        continue
    rng = np.random.default_rng(21000 + seed0 + c)
    M = int(rng.choice([1, 1, 2, 3]))
    pieces = [int(rng.integers(2, 11 if M > 1 else 33)) for _ in range(M)]
    sing = [int(rng.choice([1, -1]))]
    for _ in range(M - 1):
        sing.append(-sing[-1])
    K = int(rng.integers(3, 25)); Kd = int(rng.integers(3, 25))
    B = int(rng.integers(1, 6))
    moving = bool(rng.uniform() < 0.35) and sum(pieces) <= 24
    os.environ["DFTPAV_REF_THREADS"] = str(int(rng.choice([128, 192, 256])))
    shape = str(rng.choice(["team", "wave", "wave"]))
    os.environ["DFTPAV_REF_SHAPE"] = shape
    os.environ["DFTPAV_REF_SLOTS"] = str(int(rng.integers(1, 3)))
    os.environ["DFTPAV_REF_SLICE"] = str(int(rng.integers(2, 41)))
    p = capi.default_params()
    if shape == "wave":
        B = int(rng.integers(1, 12 if moving else 40))  # more trajectories than the waves of the persistent workgroups: the ring is used
        os.environ["DFTPAV_REF_WAVES"] = str(int(rng.choice([1, 2, 4])))
    s = sc.make_scenario(pieces, sing, K, Kd, B, seed=23000 + seed0 + c, with_moving=moving, n_obs=int(rng.integers(0, 60)),
                         **({"start_centre": (-38.0, 5.0)} if moving and rng.uniform() < 0.5 else {}))
    if moving and rng.uniform() < 0.25:  # the four cars three times over, shifted: 12 obstacles, more than 32 terms per constraint point
        from dftpav_amd.pods import SurroundSet
        sur, reps = s.surround, 3
        npc = int(sur.piece_offsets[-1])
        offs = np.concatenate([[0]] + [sur.piece_offsets[1:] + k * npc for k in range(reps)])
        cf = np.concatenate([sur.coeffs] * reps).copy()
        for k in range(1, reps):
            cf[k * npc:(k + 1) * npc, 10] += 0.7 * k
            cf[k * npc:(k + 1) * npc, 11] -= 0.4 * k
        s.surround = SurroundSet(offs, np.concatenate([sur.durations] * reps), cf, np.concatenate([sur.total_duration] * reps),
                                 np.concatenate([sur.start_time + 0.3 * k for k in range(reps)]))
    if moving and rng.uniform() < 0.5:
        s.surround.start_time[:] = rng.uniform(0.0, 4.0, len(s.surround.start_time))
    s.apply_resolution(p)
    if rng.uniform() < 0.4:
        p.lbfgs_mem_size = int(rng.choice([3, 4, 8, 17, 64, 300]))
    if rng.uniform() < 0.4:
        p.max_forward_vel *= float(rng.uniform(0.3, 1.0)); p.max_backward_vel *= float(rng.uniform(0.3, 1.0))
        p.max_forward_acc *= float(rng.uniform(0.2, 1.0)); p.max_backward_acc *= float(rng.uniform(0.2, 1.0))
        p.max_forward_cur *= float(rng.uniform(0.2, 1.0)); p.max_backward_cur *= float(rng.uniform(0.2, 1.0))
        p.wei_obs *= float(rng.uniform(0.1, 10)); p.wei_feas *= float(rng.uniform(0.1, 10)); p.wei_time *= float(rng.uniform(0.1, 10))
    if rng.uniform() < 0.3:
        s.help_eps = float(rng.choice([1e-3, 0.05]))
    if rng.uniform() < 0.2:
        p.lbfgs_past, p.lbfgs_delta = int(rng.integers(1, 7)), float(10.0 ** rng.uniform(-6, -3))
    if moving:
        s.t_now = float(rng.uniform(0.0, 5.0))
    h = capi.Handle(p); h.set_surround(s.surround)
    bt = capi.Batch(h, s.layout, B); bt.upload(s)
    try:
        bt.set_order(capi.ORDER_REFERENCE)
    except capi.DftpavError:
        refused += 1
        bt.close(); h.close()
        continue
    r = bt.solve()
    want = [po.solve_batch(p, s, nthreads=4, order=2)]
    if M == 1 and not moving:
        want.append(po.solve_batch(p, s, nthreads=4, order=0))
    ok = all(np.array_equal(r[k], w[k]) for w in want for k in keys)
    for st in r["status"]:
        stat[int(st)] = stat.get(int(st), 0) + 1
    if not ok:
        bad += 1
        print("MISMATCH case %d: pieces %s singuls %s K %d Kd %d B %d moving %s (%d obstacles) shape %s threads %s slots %s slice %s mem %d" %
              (c, pieces, sing, K, Kd, B, moving, s.surround.S if s.surround is not None else 0, shape, os.environ["DFTPAV_REF_THREADS"],
               os.environ["DFTPAV_REF_SLOTS"], os.environ["DFTPAV_REF_SLICE"], p.lbfgs_mem_size), flush=True)
        if only is not None:
            diagnose(p, s, bt, r, want[0])
    bt.close(); h.close()
print("%d cases (%d refused by the mode), %d mismatches, %.1f s; solver status counts %s" % (n_cases, refused, bad, time.time() - t0, stat))
sys.exit(1 if bad else 0)
