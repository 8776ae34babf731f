"""Developer script: the kernels around the solve against their oracles on randomised inputs (run through gpurun).

corridor generation (random maps, resolutions, vehicle sizes, poses inside / at the edge of / outside the map),
front-end resampling (random gear patterns, durations, resolutions, path densities), restart sampler, obstacle
fit, and -- on solved batches -- validation and state read-out with random sampling parameters.
  python scripts/fuzz_steps.py [n_cases] [first_seed]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from dftpav_amd import capi, scenarios as sc
from dftpav_amd.pods import FrontendParams
from oracle import pyoracle as po

n_cases = int(sys.argv[1]) if len(sys.argv) > 1 else 20
seed0 = int(sys.argv[2]) if len(sys.argv) > 2 else 0
bad = []
t0 = time.time()


def check(name, c, ok):
    if not ok:
        bad.append((name, c))
        print("MISMATCH", name, "case", c, flush=True)


for c in range(n_cases):
    rng = np.random.default_rng(31000 + seed0 + c)
    p = capi.default_params()
    if c % 3 == 1:  # another vehicle
        p.veh_width, p.veh_length, p.veh_d_cr = float(rng.uniform(1.2, 2.6)), float(rng.uniform(3.0, 6.0)), float(rng.uniform(0.3, 1.6))
    h = capi.Handle(p)
    veh = (p.veh_width, p.veh_length, p.veh_d_cr)
    # ---- corridor
    res = float(rng.choice([0.1, 0.2, 0.3, 0.5]))
    nx, ny = int(rng.integers(40, 400)), int(rng.integers(40, 400))
    grid = np.where(rng.uniform(size=(ny, nx)) < rng.uniform(0.0, 0.03), 80, 127).astype(np.uint8)
    origin = (float(rng.uniform(-50, 0)), float(rng.uniform(-50, 0)))
    ex, ey = origin[0] + nx * res, origin[1] + ny * res
    n = int(rng.integers(1, 800))
    st = np.column_stack([rng.uniform(origin[0] - 8, ex + 8, n), rng.uniform(origin[1] - 8, ey + 8, n), rng.uniform(-10, 10, n)])
    h.set_grid_map(grid, res, origin)
    check("corridor", c, np.array_equal(h.corridor_rectangles(st), po.corridor_rectangles(grid, res, origin, st, veh=veh, order=2)))
    # ---- Reeds-Shepp shots (on the map above)
    m = int(rng.integers(1, 400))
    sp = float(rng.choice([1.0, 5.0, 40.0]))
    fr = np.column_stack([rng.uniform(origin[0], ex, m), rng.uniform(origin[1], ey, m), rng.uniform(-7, 7, m)])
    to = fr + np.column_stack([rng.uniform(-sp, sp, m), rng.uniform(-sp, sp, m), rng.uniform(-4, 4, m)])
    mc, cl, ms = float(rng.uniform(0.15, 3.0)), float(rng.uniform(0.03, 1.0)), int(rng.integers(1, 600))
    vr = float(rng.uniform(0.05, 0.5))
    got = h.reeds_shepp_shots(fr, to, max_cur=mc, checkl=cl, max_samples=ms, vertex_res=vr, check_collision=True)
    want = po.reeds_shepp_shots(fr, to, max_cur=mc, checkl=cl, max_samples=ms, grid=grid, resolution=res, origin=origin, veh=veh,
                                vertex_res=vr, order=1)
    check("shots", c, all(np.array_equal(got[k], want[k]) for k in want))
    # ---- front end
    ng = int(rng.integers(1, 5))
    g0 = int(rng.choice([1, -1]))
    gears = tuple(g0 * (-1) ** i for i in range(ng))
    K, Kd = int(rng.integers(2, 40)), int(rng.integers(2, 40))
    P, pl, ss, es, ct = sc.searched_paths(int(rng.integers(1, 30)), seed=int(rng.integers(0, 10 ** 6)), gears=gears,
                                          seg_duration=float(rng.uniform(1.5, 12.0)))
    fp = FrontendParams.default(K=K, Kd=Kd)
    got = h.frontend_resample(P, pl, ss, es, ct, fp)
    want = po.frontend_resample(P, pl, ss, es, ct, fp, order=2)
    check("frontend", c, all(np.array_equal(got[k], want[k]) for k in want))
    # ---- restarts
    nh, ni, M = int(rng.integers(1, 20)), int(rng.integers(1, 12)) * 2, int(rng.integers(1, 4))
    inner, durs = rng.normal(size=(nh, ni)), rng.uniform(1, 9, size=(nh, M))
    nr, sd = int(rng.integers(1, 40)), int(rng.integers(0, 2 ** 31))
    sg, lo, hi = float(rng.uniform(0, 1)), float(rng.uniform(0.5, 1.0)), float(rng.uniform(1.0, 2.0))
    a = h.sample_restarts(inner, durs, nr, sigma=sg, lo=lo, hi=hi, seed=sd)
    b = po.sample_restarts(inner, durs, nr, sigma=sg, lo=lo, hi=hi, seed=sd)
    check("restarts", c, np.array_equal(a[0], b[0]) and np.array_equal(a[1], b[1]))
    # ---- obstacle fit
    S, ns = int(rng.integers(1, 7)), int(rng.integers(3, 40))
    ps = np.zeros((S, ns, 7))
    ps[..., 0:2] = np.cumsum(rng.normal(0, 1.5, size=(S, ns, 2)), axis=1)
    ps[..., 2] = rng.uniform(-3.2, 3.2, size=(S, ns))
    ps[..., 3] = rng.uniform(0, 6, size=(S, ns))
    ps[..., 4] = rng.normal(0, 1, size=(S, ns))
    ps[..., 5] = rng.normal(0, 0.1, size=(S, ns))
    ps[..., 6] = rng.uniform(0, 5) + float(rng.uniform(0.2, 2.0)) * np.arange(ns)[None, :]
    h.fit_surround(ps)
    g = h.get_surround()
    w = po.fit_surround(ps, order=2)
    check("fit", c, np.array_equal(g["durations"].reshape(S, -1), w["durations"]) and
          np.array_equal(g["coeffs"].reshape(S, ns - 1, 12), w["coeffs"]) and np.array_equal(g["total"], w["total"]) and
          np.array_equal(g["start"], w["start"]))
    h.set_surround(None)
    # ---- validation and read-out of a solved batch (every fourth case: it needs a solve)
    if c % 4 == 0:
        cfg = int(rng.choice([2, 3]))
        B = int(rng.integers(2, 10))
        p2 = capi.default_params()
        s = sc.baseline_config(cfg, B=B, seed=int(rng.integers(0, 10 ** 6)))
        s.apply_resolution(p2)
        h2 = capi.Handle(p2)
        bt = capi.Batch(h2, s.layout, B); bt.upload(s); bt.solve()
        co, dts = bt.coeffs()
        stt = s.meta["states"]
        cen = (0.5 * (stt[..., 0].min() + stt[..., 0].max()), 0.5 * (stt[..., 1].min() + stt[..., 1].max()))
        obs = s.meta["obstacles"]
        extra = np.column_stack([rng.uniform(cen[0] - 30, cen[0] + 30, 40), rng.uniform(cen[1] - 30, cen[1] + 30, 40), rng.uniform(0.3, 2, 40)])
        g2, o2 = sc.occupancy_grid(np.vstack([obs, extra]), arena=140.0, centre=cen)
        h2.set_grid_map(g2, sc.MAP_RESL, o2)
        sdt, vres = float(rng.uniform(0.01, 0.4)), float(rng.uniform(0.03, 0.6))
        col, first = bt.validate(sample_dt=sdt, vertex_res=vres)
        oc, of = po.validate_trajectories(g2, sc.MAP_RESL, o2, co, dts, s.layout.piece_nums, s.layout.singuls, sample_dt=sdt,
                                          vertex_res=vres, order=2)
        check("validate", c, np.array_equal(col, oc) and np.array_equal(first, of))
        tt0, sd2, nsm, flt = float(rng.uniform(-1, 3)), float(rng.uniform(0.003, 0.5)), int(rng.integers(1, 3000)), bool(rng.integers(0, 2))
        sts, nv = bt.sample_states(t0=tt0, sample_dt=sd2, n_samples=nsm, filter_singularity=flt)
        so, no = po.sample_states(co, dts, s.layout.piece_nums, s.layout.singuls, t0=tt0, sample_dt=sd2, n_samples=nsm,
                                  filter_singularity=flt, wheel_base=p2.veh_wheel_base, order=2)
        check("states", c, np.array_equal(sts, so) and np.array_equal(nv, no))
        bt.close(); h2.close()
    h.close()
print("%d cases, %d mismatches %s, %.1f s" % (n_cases, len(bad), bad[:5], time.time() - t0))
sys.exit(1 if bad else 0)
