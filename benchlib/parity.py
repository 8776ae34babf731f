"""bench.py, part: the reference's CPU path beside the value line (cpu_baseline) and the parity legs that use its solves."""
import os
import sys
import time

import numpy as np

from dftpav_amd import capi

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
from benchlib.common import (HBM_PEAK_GBS, DEV, SOLVE_FIELDS, Ctx, algorithmic_bytes, effective_cores, same_solve, same_as_ref_run,  # noqa: F401
                             bit_check)
from benchlib.stream import Stream, shard_schedule  # noqa: F401


def cpu_baseline(ctx, args, po, pyref, shard, cpu, out):
    """The reference's CPU path (rank 0, N=1 only).  oracle/_ref IS that path: the reference's own traj_optimizer.cpp /
    poly_traj_utils.hpp / lbfgs.hpp compiled unmodified (oracle/Makefile.ref), OptimizeTrajectory with its per-evaluation
    corridor copy (traj_optimizer.cpp:445), run as the reference runs it: ONE planner thread (traj_server_ros.cpp:100).  Beside
    it the literal restatement (oracle/dftpav_oracle.c, bit-equal to that build) on the SAME trajectories, single-threaded and
    with OpenMP over trajectories on every core the process may use.  -> the sample: what the parity legs compare against"""
    params, cores = ctx.params, cpu["effective"]
    n_ref = min(64, shard.B)
    pick1 = (np.arange(n_ref) * max(1, shard.B // n_ref) + 17) % shard.B
    sub1 = shard.subset(pick1)
    r1 = po.solve_batch(params, sub1, nthreads=1, order=0)   # the restatement, one thread, trajectory after trajectory
    t1 = float(np.median(r1["seconds"]))
    ref_runs, t_ref = None, None
    if pyref.available():
        t_ref, ref_runs = [], []
        for b_ in range(n_ref):
            rp = pyref.RefProblem(params, sub1, b_)
            tq = time.perf_counter()
            rr_ = rp.optimize()
            t_ref.append(time.perf_counter() - tq)
            ref_runs.append(rr_)
        t_ref = np.array(t_ref)
    ns = args.cpu_sample if args.cpu_sample > 0 else int(min(max(4 * cores, 8.0 * cores / max(t1, 1e-3)), 8192))
    ns = max(ns, n_ref)
    sub_idx = np.concatenate([pick1, (np.arange(ns - n_ref) * 7 + 3) % shard.B]).astype(np.int64)  # the same 64 first
    tc = time.perf_counter()
    rc = po.solve_batch(params, shard.subset(sub_idx), nthreads=cores, order=0)
    wall = time.perf_counter() - tc
    restatement = {"kind": "port", "solves_per_s": ns / wall, "cores": cores, "trajectories": int(ns),
                   "wall_s": wall, "thread_seconds": float(rc["seconds"].sum()),
                   "p50_ms_per_solve_per_thread": float(np.median(rc["seconds"])) * 1e3,
                   "single_thread_p50_ms_per_solve": t1 * 1e3,
                   "single_thread_p95_ms_per_solve": float(np.percentile(r1["seconds"], 95)) * 1e3,
                   "single_thread_solves_per_s": float(n_ref / r1["seconds"].sum()),
                   # like for like: the same 64 trajectories, per-solve time alone over per-solve time with every core busy
                   "parallel_efficiency_same_trajectories": float(r1["seconds"].sum() / rc["seconds"][:n_ref].sum()),
                   "identical_results_single_vs_openmp": bool(np.array_equal(r1["final_cost"], rc["final_cost"][:n_ref]))}
    common = {"unit": "solves/s", "cores_logical": cpu["logical"], "cores_affinity": cpu["affinity"],
              "cgroup_cpu_quota": cpu["cgroup_quota"], "mean_iters": float(rc["iters"].mean())}
    if ref_runs is not None:
        same = all(ref_runs[b_]["final_cost"] == r1["final_cost"][b_] and np.array_equal(ref_runs[b_]["x"], r1["x"][b_])
                   and ref_runs[b_]["iters"] == r1["iters"][b_] for b_ in range(n_ref))
    else:
        same = None
    # The reference needs Eigen, ROS and protobuf-generated code: it cannot be built here, so the baseline is the PORT -- the CPU
    # restatement (oracle/dftpav_oracle.c), one thread, trajectory after trajectory, as the reference runs its planner
    # (traj_server_ros.cpp:100).  Where an earlier round's stand-in build of the reference's sources is present on the box (opt-in,
    # oracle/Makefile.ref: NOT a reference build, its Eigen / ROS / protobuf are stand-ins written in this repository) its time is
    # reported beside it for scale, never as the baseline.
    out["cpu_baseline"] = dict(common, value=restatement["single_thread_solves_per_s"], cores=1, kind="port",
        sample="%d trajectories of the same batch (strided), the CPU restatement of OptimizeTrajectory (oracle/dftpav_oracle.c, literal order) on ONE "
               "thread, %.1f s; the same restatement with OpenMP over %d trajectories on %d core(s) under `restatement`" %
               (n_ref, float(r1["seconds"].sum()), ns, cores),
        p50_ms_per_solve=t1 * 1e3, p95_ms_per_solve=float(np.percentile(r1["seconds"], 95)) * 1e3,
        us_per_iteration=float(1e6 * r1["seconds"].sum() / max(1, int(r1["iters"].sum()))), restatement=restatement)
    if ref_runs is not None:
        out["cpu_baseline"]["standin_build_of_the_reference_sources"] = {
            "solves_per_s": float(n_ref / t_ref.sum()), "cores": 1, "bit_equal_to_restatement_on_all": bool(same),
            "note": "the reference's traj_optimizer.cpp / poly_traj_utils.hpp / lbfgs.hpp compiled against Eigen / ROS / protobuf STAND-INS written in "
                    "this repository (oracle/ref_shim): not the reference build (its eager Eigen stand-in makes it slow), reported for scale only"}
    return dict(n_ref=n_ref, pick1=pick1, sub_idx=sub_idx, rc=rc, ref_runs=ref_runs)


def parity_device_order(ctx, po, shard, r, cores):
    """(1) bit-for-bit against the device-order oracle on sampled trajectories"""
    nd = min(max(32, cores), shard.B)
    pick = (np.arange(nd) * max(1, shard.B // nd)) % shard.B  # strided through the batch (restarts of all hypotheses)
    rd = po.solve_batch(ctx.params, shard.subset(pick), nthreads=cores, order=1)
    match = bool(np.array_equal(rd["final_cost"], r["final_cost"][pick]) and np.array_equal(rd["x"], r["x"][pick]) and
                 np.array_equal(rd["iters"], r["iters"][pick]))
    return {"device_order_oracle_bit_exact_on_%d_sampled" % nd: match}


def reference_order_batch(ctx, shard):
    hR = capi.Handle(ctx.params, device=ctx.local_rank)
    bR = capi.Batch(hR, shard.layout, shard.B)
    bR.upload(shard)
    bR.set_order(capi.ORDER_REFERENCE)
    return hR, bR


def paired(a_, b_, seed_):
    """paired comparison of the final costs of two solvers over the same trajectories"""
    # NB: the mean of (a - b) / b is positive for two exchangeable positive samples (E[a / b] = E[a] E[1 / b] > 1): that
    # figure is kept because earlier rounds quoted it, but the symmetric ones decide -- the log ratio, the plain
    # difference, the median and the sign test
    from scipy import stats
    rel = (a_ - b_) / np.maximum(1.0, np.abs(b_))
    rng_ = np.random.default_rng(seed_)
    boot = np.array([rel[rng_.integers(0, len(rel), len(rel))].mean() for _ in range(2000)])
    lr = np.log(a_ / b_)
    df = a_ - b_
    idx_ = [rng_.integers(0, len(rel), len(rel)) for _ in range(2000)]
    lr_boot = np.array([lr[i_].mean() for i_ in idx_])
    df_boot = np.array([df[i_].mean() for i_ in idx_])
    npos, nneg = int((rel > 0).sum()), int((rel < 0).sum())
    pv = float(stats.binomtest(npos, npos + nneg, 0.5).pvalue) if npos + nneg > 0 else 1.0
    med_boot = np.array([np.median(rel[rng_.integers(0, len(rel), len(rel))]) for _ in range(500)])
    return {"trajectories": int(len(rel)),
            "log_ratio_mean": float(lr.mean()),
            "log_ratio_mean_ci95": [float(np.percentile(lr_boot, 2.5)), float(np.percentile(lr_boot, 97.5))],
            "diff_mean": float(df.mean()), "diff_mean_ci95": [float(np.percentile(df_boot, 2.5)), float(np.percentile(df_boot, 97.5))],
            "rel_diff_signed_mean": float(rel.mean()),
            "rel_diff_signed_mean_ci95": [float(np.percentile(boot, 2.5)), float(np.percentile(boot, 97.5))],
            "rel_diff_signed_median": float(np.median(rel)),
            "rel_diff_signed_median_ci95": [float(np.percentile(med_boot, 2.5)), float(np.percentile(med_boot, 97.5))],
            "n_first_higher": npos, "n_first_lower": nneg, "sign_test_p": pv,
            "rel_diff_abs_p50": float(np.median(np.abs(rel))), "rel_diff_abs_p95": float(np.percentile(np.abs(rel), 95)),
            "frac_within_1e-5": float((np.abs(rel) <= 1e-5).mean()),
            "mean_cost_first": float(a_.mean()), "mean_cost_second": float(b_.mean()),
            "median_cost_first": float(np.median(a_)), "median_cost_second": float(np.median(b_))}


def parity_reference_order(ctx, args, po, st, r_ref, r_dev, sample, out, B_total):
    """(2) the REFERENCE-ORDER device mode (dftpav_batch_set_order: solver_ref.hip / solver_ref4.hip): every sum in the reference's
    order, so its solves must equal the CPU restatement's bit for bit -- checked on every trajectory cpu_baseline solved (and, where
    the stand-in build of the reference's sources exists on this box, against that on its 64).  r_ref: the value line's first batch
    when the value line runs in this order (its throughput IS the value); with --order device the batch is solved here.  (3) the bias
    of the device order r_dev against it, with the one-ulp control.  -> the reference-order results of the batch, or None"""
    shard, rc, ref_runs = st.shard, sample["rc"], sample["ref_runs"]
    try:
        ro = {}
        if r_ref is None:   # --order device: one reference-order batch beside the value line, and the same stream of cycles in that order
            hR, bR = reference_order_batch(ctx, shard)
            bR.solve_async(); bR.sync()
            bR.solve_async(); bR.sync()
            ro["isolated_kernel_ms"] = bR.last_solve_ms()
            r_ref = bR.results()
            bR.close(); hR.close()
            try:
                stR = Stream(ctx, B_total, args.config, args.seed, depth=args.depth, order=capi.ORDER_REFERENCE)
                k_ref, w_ref = max(4, min(args.steps, 20)), max(2, min(args.warmup, 5))
                rR = stR.run(k_ref, w_ref)
                ro["overlapped"] = {"solves_per_s": rR["value"], "ms_per_step": rR["ms_per_step"], "steps": k_ref, "warmup": w_ref, "schedule": ctx.schedule,
                                    "first_batch_equals_the_isolated_solve": bool(np.array_equal(rR["rs"][0]["final_cost"], r_ref["final_cost"]))}
                ro["solves_per_s"] = rR["value"]
                stR.close()
            except capi.DftpavError as ex:
                ro["overlapped"] = {"failed": str(ex)}
        else:
            ro["solves_per_s"] = out["value"]
            ro["solves_per_s_is"] = "the value line (this order IS the value line)"
            if "isolated" in out:
                ro["isolated_solves_per_s"], ro["isolated_kernel_ms"] = out["isolated"]["solves_per_s"], out["isolated"]["kernel_ms"]
                # the isolated solves of the value line's first batch (side_isolated) gave the bits of its overlapped solve
                ro["first_batch_equals_the_isolated_solve"] = out["isolated"].get("same_bits_as_the_stream")
        eq_port = [same_solve(r_ref, g_, rc, i_) for i_, g_ in enumerate(sample["sub_idx"])]
        ro.update({"trajectories": int(len(sample["sub_idx"])), "bit_equal": int(sum(eq_port)),
                   "against": "the CPU restatement of the reference (oracle/dftpav_oracle.c, order 0): final x, cost, status, iterations, evaluations",
                   "batch_solved_on_device": int(shard.B),
                   "frac_within_1e-5_of_cpu": float(np.mean([abs(r_ref["final_cost"][g_] - rc["final_cost"][i_]) <= 1e-5 * max(1.0, abs(rc["final_cost"][i_]))
                                                             for i_, g_ in enumerate(sample["sub_idx"])]))})
        if ref_runs is not None:
            eq_ref = [bool(r_ref["final_cost"][g_] == ref_runs[i_]["final_cost"] and np.array_equal(r_ref["x"][g_], ref_runs[i_]["x"]) and
                           r_ref["iters"][g_] == ref_runs[i_]["iters"] and r_ref["evals"][g_] == ref_runs[i_]["evals"] and
                           r_ref["status"][g_] == ref_runs[i_]["status"]) for i_, g_ in enumerate(sample["pick1"])]
            ro["against_standin_build_of_the_reference_sources"] = {"trajectories": int(sample["n_ref"]), "bit_equal": int(sum(eq_ref))}
        out["parity"]["reference_order"] = ro
        # (3) device order against the reference over the WHOLE batch, with the reference-order solves standing for the
        # reference (they are its restatement, bit for bit): the solver is chaotic, so the two follow different
        # iterate sequences after the first rounding difference; the question is whether the device order is BIASED.  Control: the
        # reference order against itself with one waypoint coordinate of x0 moved by one ulp -- same size of effect, no bias possible.
        if r_dev is not None:
            hR, bR = reference_order_batch(ctx, shard)
            sh1 = shard.subset(np.arange(shard.B))
            sh1.inner_pts = np.ascontiguousarray(sh1.inner_pts).copy()
            sh1.inner_pts[:, 0] = np.nextafter(sh1.inner_pts[:, 0], np.inf)
            bR.upload(sh1)
            bR.solve_async(); bR.sync()
            ulp_gpu = bR.results()
            bR.close(); hR.close()
            out["parity"]["bias"] = parity_bias(r_dev, r_ref, ulp_gpu)
        return r_ref
    except capi.DftpavError as ex:
        out["parity"]["reference_order"] = {"unsupported": str(ex)}
        return None


def parity_bias(r, ref_gpu, ulp_gpu):
    bias = {"device_order_vs_reference": paired(r["final_cost"], ref_gpu["final_cost"], 1),
            "control_reference_with_x0_moved_one_ulp_vs_reference": paired(ulp_gpu["final_cost"], ref_gpu["final_cost"], 2),
            "mean_iters": {"device_order": float(r["iters"].mean()), "reference": float(ref_gpu["iters"].mean()),
                           "reference_x0_one_ulp": float(ulp_gpu["iters"].mean())},
            "success_rate": {"device_order": float(r["success"].mean()), "reference": float(ref_gpu["success"].mean())}}
    d_, c_ = bias["device_order_vs_reference"], bias["control_reference_with_x0_moved_one_ulp_vs_reference"]

    def cov(q_, k_):
        return bool(q_[k_][0] <= 0.0 <= q_[k_][1])
    bias["verdict"] = {"log_ratio_ci_covers_0": cov(d_, "log_ratio_mean_ci95"), "diff_ci_covers_0": cov(d_, "diff_mean_ci95"),
                       "sign_test_p": d_["sign_test_p"],
                       "control_log_ratio_ci_covers_0": cov(c_, "log_ratio_mean_ci95"), "control_diff_ci_covers_0": cov(c_, "diff_mean_ci95"),
                       "mean_of_relative_difference_ci_covers_0": cov(d_, "rel_diff_signed_mean_ci95"),
                       "control_mean_of_relative_difference_ci_covers_0": cov(c_, "rel_diff_signed_mean_ci95"),
                       "note": "the mean of (a - b) / b is positive by construction for exchangeable samples with this spread "
                               "(the control shows the same offset); the symmetric statistics decide"}
    return bias


def restart_stats(rst):
    return {"iters_p50": float(np.median(rst["iters"])), "iters_p95": float(np.percentile(rst["iters"], 95)),
            "iters_max": int(rst["iters"].max()), "frac_stopping_within_3": float((rst["iters"] <= 3).mean()),
            "frac_stopping_within_5": float((rst["iters"] <= 5).mean())}


def parity_literal(ctx, po, shard, r, ref_gpu, cores):
    """(4) against the LITERAL oracle per evaluation over the whole batch:
      a. the literal cost at every final x of the kernel            (same function, rounding-level agreement)
      b. lbfgs_optimize restarted by the literal oracle from every final x of the kernel stops at once
         (past = 3 iterations is the minimum, lbfgs.hpp:642-659): the kernel's x is a stopping point of the reference"""
    ev = po.batch_op(ctx.params, shard, "eval", r["x"], nthreads=cores, order=0)
    rel_f = np.abs(ev["f"] - r["final_cost"]) / np.maximum(1.0, np.abs(ev["f"]))
    rst = po.batch_op(ctx.params, shard, "restart", r["x"], nthreads=cores, order=0)
    drop = (ev["f"] - rst["final_cost"]) / np.maximum(1.0, np.abs(ev["f"]))
    lit = {"trajectories": int(shard.B),
           "literal_cost_at_kernel_x_max_rel_diff": float(rel_f.max()),
           "literal_restart_from_kernel_x": dict(restart_stats(rst), rel_cost_decrease_p50=float(np.median(drop)),
                                                 rel_cost_decrease_p95=float(np.percentile(drop, 95)), rel_cost_decrease_max=float(drop.max()))}
    if ref_gpu is not None:  # for scale: the reference restarted from its own final points
        lit["literal_restart_from_reference_x"] = restart_stats(po.batch_op(ctx.params, shard, "restart", ref_gpu["x"], nthreads=cores, order=0))
    return lit


def parity_lockstep(ctx, po, shard):
    """(5) 256 trajectories in lockstep with the reference's line search and two-loop recursion (tests/lockstep.py): the
    device-order kernel's evaluation trace replayed branch for branch against literal evaluations"""
    try:
        sys.path.insert(0, os.path.join(ROOT, "tests"))
        import lockstep
        from test_gpu_lockstep import summarize
        nls = min(256, shard.B)
        subL = shard.subset(np.arange(nls))
        hL = capi.Handle(ctx.params, device=ctx.local_rank)
        bL = capi.Batch(hL, subL.layout, nls)
        bL.upload(subL)
        bL.trace(0, 4096, count=nls)
        bL.solve_async(); bL.sync()
        rL = bL.results()
        reps = []
        tls = time.perf_counter()
        for tb in range(nls):
            tr = bL.get_trace(tb)
            lp = po.OracleProblem(ctx.params, subL, tb, order=0)
            reps.append(lockstep.replay(tr, lp.eval, ctx.params, direction_every=1 if tb < 4 else 16))
            if time.perf_counter() - tls > 90.0 and tb >= 63:  # a slow host: at least 64, then stop at the time box
                break
        sm = summarize(reps)
        sm["whole_solve_replayed"] = int(sum(1 for q_, rp_ in enumerate(reps) if rp_["flip"] is None and abs(rp_["iterations"] - rL["iters"][q_]) <= 1))
        sm["seconds"] = time.perf_counter() - tls
        bL.close(); hL.close()
        return sm
    except (AssertionError, capi.DftpavError) as ex:
        return {"failed": str(ex)}


def with_upload(st):
    """PCIe-inclusive rate (never `value`): upload of the whole batch, isolated solve, results back"""
    bt, shard = st.bts[0], st.shard
    tu = time.perf_counter()
    bt.upload(shard)
    t_up = time.perf_counter() - tu
    bt.solve_async(); bt.sync()
    t_sv = bt.last_solve_ms() * 1e-3
    tdn = time.perf_counter()
    bt.results()
    t_dn = time.perf_counter() - tdn
    return {"upload_ms": 1e3 * t_up, "solve_ms": 1e3 * t_sv, "download_ms": 1e3 * t_dn, "solves_per_s": shard.B / (t_up + t_sv + t_dn)}
