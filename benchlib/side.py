"""bench.py, part: the side runs of the one-GPU line: isolated batch, the exact BASELINE configs at their own sizes, the reference order on them, the steps either side of the solve."""
import os
import time

import numpy as np
import torch

from dftpav_amd import capi, scenarios as sc

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
from benchlib.common import (HBM_PEAK_GBS, DEV, SOLVE_FIELDS, Ctx, algorithmic_bytes, effective_cores, same_solve, same_as_ref_run,  # noqa: F401
                             bit_check)
from benchlib.stream import Stream, shard_schedule  # noqa: F401


def side_isolated(st, rs0=None):
    """the value line's batch as isolated solves (no chaining: its tail runs on a nearly empty device); rs0: what the stream's solve
    of the same batch returned (the schedule moves no bit)"""
    bt, iso = st.bts[0], []
    bt.set_hand_over(-1)  # the plan's default end game (the overlap schedule runs with 0)
    for _ in range(3):
        bt.solve_async(); bt.sync(); iso.append(bt.last_solve_ms())
    out = {"batch": int(st.shard.B), "kernel_ms": float(np.mean(iso)), "solves_per_s": st.shard.B / (float(np.mean(iso)) * 1e-3)}
    if rs0 is not None:
        ri = bt.results()
        out["same_bits_as_the_stream"] = bool(all(np.array_equal(ri[k_], rs0[k_]) for k_ in SOLVE_FIELDS))
    return out


def side_device_order(ctx, args, B_total):
    """The DEVICE order (solver.hip: reassociated sums) on the value line's stream of cycles -- same seeds, same batches -- at the value
    line's depth and at depth 2 (the schedule of rounds 3-4), so that a change of schedule cannot read as a change of kernel.  Its
    results of the first batch go to the parity legs ("_results", removed before the line is printed)."""
    out = {"note": "reassociated sums: faster, bit-equal only to its own CPU replay; against the reference see parity.bias"}
    k, w = max(4, min(args.steps, 20)), max(2, min(args.warmup, 5))
    for depth in sorted({2, max(2, args.depth)}):
        stD = Stream(ctx, B_total, args.config, args.seed, depth=depth)
        rD = stD.run(k, w)
        lay, sh, rs = stD.shard.layout, stD.shard, rD["rs"]
        eb = [float(algorithmic_bytes(lay, sh.n_points, lay.H, lay.M, q["iters"], q["evals"], q["hist_sum"]).sum()) if q is not None else 0.0 for q in rs]
        eb_steps = sum(eb[(stD.k - k + j) % stD.D] for j in range(k))
        kms = rD["gpu_ms"] / k
        ent = {"solves_per_s": rD["value"], "ms_per_step": rD["ms_per_step"], "steps": k, "warmup": w, "time_to_result_ms": rD["to_result_ms"],
               "roofline_frac": eb_steps / k / (kms * 1e-3) / 1e9 / HBM_PEAK_GBS, "kernel": "solver_kernel"}
        out["depth_%d" % depth] = ent
        if depth == max(2, args.depth):
            out.update(ent)
            out["steps_in_flight"] = depth
            out["_results"] = rs[0]
        stD.close()
    return out


def side_batch(ctx, args, po, cores, cfg, B, reps, n_check):
    """BASELINE config `cfg` at batch B: isolated solves (three draws), a stream of such batches on 8 HIP streams, sampled
    trajectories against the device-order oracle"""
    p2 = capi.default_params()
    s2 = sc.baseline_config(cfg, B=B, seed=args.seed)
    s2.apply_resolution(p2)
    h2 = capi.Handle(p2, device=ctx.local_rank)
    h2.set_surround(s2.surround)
    b2 = capi.Batch(h2, s2.layout, B)
    b2.upload(s2)
    b2.solve_async(); b2.sync()
    ms = []
    for _ in range(reps):
        b2.solve_async(); b2.sync(); ms.append(b2.last_solve_ms())
    r2 = b2.results()
    # An isolated batch is done when its LONGEST solve is, and which trajectory that is -- 800 or 870 iterations,
    # a cheap or an expensive one -- is a lottery of the last bit: two more draws of the same
    # batch with one waypoint coordinate of every x0 moved by one ulp
    draws, longest = [float(np.mean(ms))], [int(r2["iters"].max())]
    for k_ in (0, 1):
        s3 = s2.subset(np.arange(B))
        ip = np.ascontiguousarray(s3.inner_pts).copy()
        fl = ip.reshape(B, -1)
        fl[:, k_] = np.nextafter(fl[:, k_], np.inf)
        s3.inner_pts = ip
        b2.upload(s3)
        b2.solve_async(); b2.sync()
        draws.append(float(b2.last_solve_ms()))
        longest.append(int(b2.results()["iters"].max()))
    # a stream of such batches (planning cycles back to back on several planner threads): 8 resident batches on 8 HIP
    # streams in the throughput residency (four workgroups per CU, dftpav_batch_create_shaped), 3 rounds
    hx = [capi.Handle(p2, device=ctx.local_rank) for _ in range(8)]
    bx = []
    for hh in hx:
        hh.set_surround(s2.surround)
        bb = capi.Batch(hh, s2.layout, B, residency=2)
        bb.upload(s2)
        bx.append(bb)
    for bb in bx:
        bb.solve_async()
    for bb in bx:
        bb.sync()
    torch.cuda.synchronize()
    t1 = time.perf_counter()
    rounds = 3
    for _ in range(rounds):
        for bb in bx:
            bb.solve_async()
    for bb in bx:
        bb.sync()
    stream_s = time.perf_counter() - t1
    same = bool(all(np.array_equal(bb.results()["x"], r2["x"]) for bb in bx))
    pick = (np.arange(n_check) * max(1, B // n_check)) % B
    ok = bit_check(po, cores, p2, s2, r2, pick)
    for bb in bx:
        bb.close()
    b2.close(); h2.close()
    for hh in hx:
        hh.close()
    # the HBM roofline of this side run: algorithmic bytes of the first draw's solves (E_eval with this layout's n and
    # Npts) over the mean isolated kernel time
    lay2 = s2.layout
    npts2 = int(s2.corridor.shape[1])
    ab2 = float(algorithmic_bytes(lay2, npts2, lay2.H, lay2.M, r2["iters"], r2["evals"], r2["hist_sum"]).sum())
    gbs = ab2 / (float(np.mean(draws)) * 1e-3) / 1e9
    roof2 = {"bound": "hbm", "algorithmic_bytes_per_batch": ab2, "achieved": gbs, "peak": HBM_PEAK_GBS,
             "unit": "GB/s", "frac": gbs / HBM_PEAK_GBS, "n": int(lay2.n_vars), "Npts": npts2,
             "note": "isolated batch: its duration is that of its longest solve; algorithmic bytes as for the value line"}
    return {"batch": B, "solves_per_s": B * len(draws) / (sum(draws) * 1e-3), "kernel_ms": float(np.mean(draws)), "roofline": roof2,
            "draws": {"kernel_ms": draws, "longest_solve_iterations": longest,
                      "note": "the batch as generated, then with x0 moved by one ulp in one coordinate, twice: an isolated "
                              "batch lasts as long as its longest solve, which differs from draw to draw; solves_per_s is "
                              "over the three"},
            "p50_ms_per_solve": float(np.median(r2["latency_us"])) * 1e-3, "mean_iters": float(r2["iters"].mean()),
            "stream_of_batches": {"streams": len(bx), "batches": rounds * len(bx), "solves_per_s": rounds * len(bx) * B / stream_s,
                                  "results_identical": same},
            "device_order_oracle_bit_exact_on_%d_sampled" % n_check: ok}


def side_single(ctx, args, po, cores, cfg, seeds):
    """one gear-shift trajectory alone on the GPU: the solver is chaotic (an instance needs 90 or 340 iterations depending on
    the last bit), so the latency is quoted as the median over the seeded instances, with the per-iteration time beside it"""
    from oracle import pyref as _pr
    p2 = capi.default_params()
    ms, its, oks, ms_ref, its_ref, eq2, eqb, eqc, best64 = [], [], [], [], [], [], [], [], []
    for sd in seeds:
        s2 = sc.baseline_config(cfg, B=1, seed=args.seed + 17 * sd)
        s2.apply_resolution(p2)
        h2 = capi.Handle(p2, device=ctx.local_rank)
        b2 = capi.Batch(h2, s2.layout, 1)
        b2.upload(s2)
        b2.solve_async(); b2.sync()
        b2.solve_async(); b2.sync()
        r2 = b2.results()
        ms.append(b2.last_solve_ms()); its.append(int(r2["iters"][0]))
        oks.append(bit_check(po, cores, p2, s2, r2, np.array([0])))
        # the same instance in reference order: the reference's program with the correctly rounded cos / sin of the
        # junction angle (oracle order 2 is that program on the CPU); equal to the reference build itself whenever this
        # host's libm rounded every angle correctly
        b2.set_order(capi.ORDER_REFERENCE)
        b2.solve_async(); b2.sync()
        b2.solve_async(); b2.sync()
        r3 = b2.results()
        ms_ref.append(b2.last_solve_ms()); its_ref.append(int(r3["iters"][0]))
        o2 = po.solve_batch(p2, s2, nthreads=1, order=2)
        eq2.append(bool(o2["final_cost"][0] == r3["final_cost"][0] and np.array_equal(o2["x"][0], r3["x"][0]) and o2["iters"][0] == r3["iters"][0]))
        if _pr.available():
            rr_ = _pr.RefProblem(p2, s2, 0).optimize()
            eqb.append(bool(rr_["final_cost"] == r3["final_cost"][0] and np.array_equal(rr_["x"], r3["x"][0])))
        if _pr.cr_available():   # the reference's own objects on a correctly rounded libm (oracle/cr_libm.c): must agree on ALL
            eqc.append(same_as_ref_run(r3, 0, _pr.RefProblem(p2, s2, 0, cr=True).optimize()))
        # one trajectory leaves 255 CUs idle: the same call as slot 0 of a batch of 64 with 63 seeded restarts in the SAME launch
        # (what the drop-in does with DFTPAV_DROPIN_RESTARTS=64): time to the best of 64, slot 0's bits untouched
        sK = s2.with_restarts(h2, 64, seed=args.seed)
        bK = capi.Batch(h2, sK.layout, 64)
        bK.upload(sK)
        bK.set_order(capi.ORDER_REFERENCE)
        bK.solve_async(); bK.sync()
        bK.solve_async(); bK.sync()
        rK = bK.results()
        okK = rK["success"] != 0
        best64.append({"kernel_ms": bK.last_solve_ms(), "slot0_bit_equal_to_the_lone_solve": bool(all(np.array_equal(rK[k_][0], r3[k_][0]) for k_ in SOLVE_FIELDS)),
                       "slot0_cost": float(rK["final_cost"][0]), "best_cost": float(rK["final_cost"][okK].min()) if okK.any() else None,
                       "successes": int(okK.sum()), "lone_solve_ms": ms_ref[-1]})
        bK.close()
        b2.close(); h2.close()
    ms, its, ms_ref, its_ref = np.array(ms), np.array(its), np.array(ms_ref), np.array(its_ref)
    return {"batch": 1, "instances": len(seeds), "p50_ms_per_solve": float(np.median(ms)), "min_ms": float(ms.min()),
            "max_ms": float(ms.max()), "median_iters": float(np.median(its)), "us_per_iteration": float(1e3 * ms.sum() / its.sum()),
            "solves_per_s": float(1e3 / np.median(ms)), "device_order_oracle_bit_exact_on_all": bool(all(oks)),
            "reference_order": {"p50_ms_per_solve": float(np.median(ms_ref)), "us_per_iteration": float(1e3 * ms_ref.sum() / its_ref.sum()),
                                "median_iters": float(np.median(its_ref)),
                                "bit_equal_to_the_reference_program_with_correctly_rounded_cos_sin": int(sum(eq2)),
                                "bit_equal_to_the_retired_standin_build_on_this_host": (int(sum(eqb)) if eqb else None),
                                "bit_equal_to_the_retired_standin_build_on_a_correctly_rounded_libm": (int(sum(eqc)) if eqc else None),
                                "instances": len(seeds),
                                "best_of_64_restarts_in_one_launch": {
                                    "p50_ms": float(np.median([r_["kernel_ms"] for r_ in best64])), "p50_ms_of_the_lone_solve": float(np.median([r_["lone_solve_ms"] for r_ in best64])),
                                    "slot0_bit_equal_to_the_lone_solve_on_all": bool(all(r_["slot0_bit_equal_to_the_lone_solve"] for r_ in best64)),
                                    "median_cost_ratio_best_over_slot0": float(np.median([r_["best_cost"] / r_["slot0_cost"] for r_ in best64 if r_["best_cost"] is not None])),
                                    "mean_successes_of_64": float(np.mean([r_["successes"] for r_ in best64]))}}}


def side_reference_order_batch(ctx, args, po, cores, cfg, B, golden):
    """A BASELINE configuration in reference order AT ITS OWN BATCH SIZE (configs[4]: 1024, moving cars -- dynamicObsGradCostP
    statement by statement with the correctly rounded exp / log / x^3; configs[1]: the gear shift at 4096).  Checked against the
    reference's program with correctly rounded libm calls: 64 sampled trajectories whose expected results were computed where the
    cores are (tests/golden/ref_order_batches.npz: oracle order 2, the first 8 also by the reference's own objects on a correctly
    rounded libm), or, for other sizes and seeds, 4 sampled by oracle order 2 here."""
    try:
        p5 = capi.default_params()
        s5 = sc.baseline_config(cfg, B=B, seed=args.seed)
        s5.apply_resolution(p5)
        h5 = capi.Handle(p5, device=ctx.local_rank)
        h5.set_surround(s5.surround)
        b5 = capi.Batch(h5, s5.layout, B)
        b5.upload(s5)
        b5.set_order(capi.ORDER_REFERENCE)
        b5.solve_async(); b5.sync()
        b5.solve_async(); b5.sync()
        r5 = b5.results()
        ms5 = b5.last_solve_ms()
        lay5 = s5.layout
        ab5 = float(algorithmic_bytes(lay5, s5.n_points, lay5.H, lay5.M, r5["iters"], r5["evals"], r5["hist_sum"]).sum())
        row = {"batch": B, "kernel_ms": ms5, "solves_per_s": B / (ms5 * 1e-3), "mean_iters": float(r5["iters"].mean()), "success_rate": float(r5["success"].mean()),
               "us_per_iteration_of_the_longest": 1e3 * ms5 / max(1, int(r5["iters"].max())),
               "roofline": {"bound": "hbm", "achieved": ab5 / (ms5 * 1e-3) / 1e9, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": ab5 / (ms5 * 1e-3) / 1e9 / HBM_PEAK_GBS,
                            "isolated": True}}
        gz = os.path.join(ROOT, "tests", "golden", "ref_order_batches.npz")
        Z = np.load(gz) if os.path.exists(gz) else None
        if Z is not None and golden + "_pick" in Z.files and int(Z["seed"]) == args.seed and int(Z[golden + "_pick"].max()) < B and golden.endswith("_b%d" % B):
            pk = Z[golden + "_pick"]
            row["bit_equal_to_the_reference_program_with_correctly_rounded_libm_calls_on_%d_sampled" % len(pk)] = bool(
                all(np.array_equal(Z[golden + "_" + k_], r5[k_][pk]) for k_ in SOLVE_FIELDS))
            row["of_them_also_solved_by_the_retired_standin_build_in_round_5"] = int(Z["n_checked_against_the_reference_objects_on_a_correctly_rounded_libm"])
        else:
            pick5 = np.array([0, B // 3, 2 * B // 3, B - 1])
            o5 = po.solve_batch(p5, s5.subset(pick5), nthreads=min(4, cores), order=2)
            row["bit_equal_to_the_reference_program_with_correctly_rounded_libm_calls_on_4_sampled"] = bool(
                all(np.array_equal(o5[k_], r5[k_][pick5]) for k_ in SOLVE_FIELDS))
        b5.close(); h5.close()
        if cfg == 2:   # the gear shift also as the value line is measured: a stream of such batches, args.depth in flight (the isolated figure
            # above is the batch alone on the device: it lasts as long as its longest solve)
            stG = Stream(ctx, B, cfg, args.seed, depth=max(2, args.depth), order=capi.ORDER_REFERENCE)
            rG = stG.run(2 * max(2, args.depth), max(2, args.depth))
            rs = [q_ for q_ in rG["rs"] if q_ is not None]
            eb = sum(float(algorithmic_bytes(lay5, s5.n_points, lay5.H, lay5.M, q_["iters"], q_["evals"], q_["hist_sum"]).sum()) for q_ in rs) / len(rs)
            row["isolated_solves_per_s"] = row["solves_per_s"]
            row["overlapped"] = {"solves_per_s": rG["value"], "ms_per_step": rG["ms_per_step"], "steps": rG["steps"], "steps_in_flight": stG.D,
                                 "roofline_frac": eb / (rG["gpu_ms"] / rG["steps"] * 1e-3) / 1e9 / HBM_PEAK_GBS,
                                 "first_batch_equals_the_isolated_solve": bool(all(np.array_equal(rs[0][k_], r5[k_]) for k_ in SOLVE_FIELDS))}
            stG.close()
        return row
    except capi.DftpavError as ex:
        return {"unsupported": str(ex)}


def side_reference_order_other_configs(ctx, args, po, cores):
    """reference order on the remaining configurations: configs[0]'s layout (one forward segment, 8 pieces: no libm call in the
    reference's loop, so the reference BUILD itself is the yardstick) and the reference's live case (gear shifts together with
    moving obstacles, traj_manager.cpp:604-610: the reference's program with correctly rounded libm calls is the yardstick; how
    many solves the build on this host happens to share is reported beside it)"""
    try:
        from oracle import pyref as _pr2
        rows = {}
        for name_, mk in (("forward_8_pieces", lambda: sc.baseline_config(1, B=16, seed=args.seed)),
                          ("gear_shifts_with_moving_obstacles", lambda: sc.make_scenario([5, 4, 6], [1, -1, 1], 12, 16, 8, seed=args.seed + 82, with_moving=True,
                                                                                       n_obs=25, start_centre=(-38.0, 5.0)))):
            pz = capi.default_params()
            sz = mk()
            sz.apply_resolution(pz)
            hz = capi.Handle(pz, device=ctx.local_rank)
            hz.set_surround(sz.surround)
            bz = capi.Batch(hz, sz.layout, sz.B)
            bz.upload(sz)
            bz.set_order(capi.ORDER_REFERENCE)
            rz = bz.solve()
            libm = sz.layout.M > 1 or sz.surround is not None
            oz = po.solve_batch(pz, sz, nthreads=cores, order=2 if libm else 0)
            eq_prog = int(sum(same_solve(oz, i_, rz, i_) for i_ in range(sz.B)))
            row = {"trajectories": int(sz.B), "bit_equal_to_the_reference_program" + ("_with_correctly_rounded_libm_calls" if libm else ""): eq_prog,
                   "libm_calls_in_the_reference_loop": bool(libm)}
            if _pr2.available():
                eqb_ = 0
                for i_ in range(sz.B):
                    rr_ = _pr2.RefProblem(pz, sz, i_).optimize()
                    eqb_ += int(rr_["final_cost"] == rz["final_cost"][i_] and np.array_equal(rr_["x"], rz["x"][i_]) and rr_["iters"] == rz["iters"][i_])
                row["against_retired_standin_build"] = {"trajectories": int(sz.B), "bit_equal": eqb_,
                                                  "note": ("every solve must agree" if not libm else
                                                           "agrees where this host's libm rounded every call of the solve correctly")}
            if libm and _pr2.cr_available():
                row["against_retired_standin_build_on_a_correctly_rounded_libm"] = {
                    "trajectories": int(sz.B), "bit_equal": int(sum(same_as_ref_run(rz, i_, _pr2.RefProblem(pz, sz, i_, cr=True).optimize()) for i_ in range(sz.B))),
                    "note": "the reference's own objects linked against oracle/cr_libm.c: every solve must agree"}
            rows[name_] = row
            bz.close(); hz.close()
        return rows
    except capi.DftpavError as ex:
        return {"failed": str(ex)}


def side_neighbours(ctx, args, po, st, out):
    """the steps either side of the solve on the value line's shard: rectangle corridors before it; collision
    re-check, state read-out after it; Reeds-Shepp shots of the hypothesis generation"""
    shard, h, bt = st.shard, st.hs[0], st.bts[0]
    states = shard.meta["states"].reshape(-1, 3)
    cen = (0.5 * (states[:, 0].min() + states[:, 0].max()), 0.5 * (states[:, 1].min() + states[:, 1].max()))
    span = max(states[:, 0].max() - states[:, 0].min(), states[:, 1].max() - states[:, 1].min()) + 40.0
    grid, origin = sc.occupancy_grid(shard.meta["obstacles"], arena=span, centre=cen)
    h.set_grid_map(grid, sc.MAP_RESL, origin)
    Hc = h.corridor_rectangles(states)
    tcor = []
    for _ in range(3):
        t1 = time.perf_counter(); Hc = h.corridor_rectangles(states); tcor.append(time.perf_counter() - t1)
    cor_ms = h.corridor_last_ms()
    # ---- the step after the solve: collision re-check of all solved trajectories of the shard
    colv, _first = bt.validate()
    out["validate"] = {"trajectories": int(shard.B), "kernel_ms": h.corridor_last_ms(),
                       "trajectories_per_s": shard.B / (h.corridor_last_ms() * 1e-3),
                       "colliding": int(colv.sum())}
    # ---- the read-out of the result: GetState every 10 ms over every solved trajectory
    cor, dts = bt.coeffs()
    n_rd = int(float(np.max(np.sum(dts * shard.layout.piece_nums[None, :], axis=1))) / 0.01) + 2
    rd, nv = bt.sample_states(sample_dt=0.01, n_samples=n_rd)
    rd_ms = h.corridor_last_ms()
    nchk = min(64, shard.B)
    ord_, onv = po.sample_states(cor[:nchk], dts[:nchk], shard.layout.piece_nums, shard.layout.singuls, sample_dt=0.01,
                                 n_samples=n_rd, wheel_base=ctx.params.veh_wheel_base, order=1)
    out["readout"] = {"trajectories": int(shard.B), "samples_per_trajectory": n_rd, "kernel_ms": rd_ms,
                      "states_per_s": float(nv.sum()) / (rd_ms * 1e-3), "written_GB_per_s": rd.nbytes / (rd_ms * 1e-3) / 1e9,
                      "oracle_bit_exact_on_first_64": bool(np.array_equal(rd[:nchk], ord_) and np.array_equal(nv[:nchk], onv))}
    del rd
    # ---- hypothesis generation: Reeds-Shepp shots between random poses of the map, sampled and checked
    rng_s = np.random.default_rng(args.seed)
    n_sh = 8192
    lo_xy = np.array(origin); hi_xy = lo_xy + sc.MAP_RESL * np.array([grid.shape[1], grid.shape[0]])
    fr = np.column_stack([rng_s.uniform(lo_xy[0], hi_xy[0], n_sh), rng_s.uniform(lo_xy[1], hi_xy[1], n_sh), rng_s.uniform(-np.pi, np.pi, n_sh)])
    to = np.column_stack([rng_s.uniform(lo_xy[0], hi_xy[0], n_sh), rng_s.uniform(lo_xy[1], hi_xy[1], n_sh), rng_s.uniform(-np.pi, np.pi, n_sh)])
    sh = h.reeds_shepp_shots(fr, to, max_cur=1.0, checkl=0.2, max_samples=768, check_collision=True)
    sh_ms = h.corridor_last_ms()
    so_ = po.reeds_shepp_shots(fr[:256], to[:256], max_cur=1.0, checkl=0.2, max_samples=768, grid=grid, resolution=sc.MAP_RESL,
                               origin=origin, order=1)
    out["shots"] = {"pairs": n_sh, "poses": int(sh["n_samples"].sum()), "kernel_ms": sh_ms, "shots_per_s": n_sh / (sh_ms * 1e-3),
                    "free": float(1.0 - sh["collides"].mean()),
                    "oracle_bit_exact_on_first_256": bool(all(np.array_equal(sh[k][:256], so_[k]) for k in so_))}
    del sh
    nchk = min(2000, len(states))
    out["corridor"] = {"states": int(len(states)), "map_cells": [int(grid.shape[1]), int(grid.shape[0])],
                       "kernel_ms": cor_ms, "rectangles_per_s": len(states) / (cor_ms * 1e-3),
                       "rectangles_per_s_with_pcie": len(states) / min(tcor),
                       "oracle_bit_exact_on_first_%d" % nchk: bool(np.array_equal(
                           Hc[:nchk], po.corridor_rectangles(grid, sc.MAP_RESL, origin, states[:nchk], order=1)))}
