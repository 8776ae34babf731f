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
        out["cpu_baseline"] = dict(common, value=float(n_ref / t_ref.sum()), cores=1, kind="reference",
            sample="%d trajectories of the same batch (strided), OptimizeTrajectory of oracle/_ref = the reference's own solve-path "
                   "sources compiled here against interface stand-ins, one thread as the reference runs its planner; %.1f s.  Its "
                   "Eigen is a stand-in that evaluates every expression eagerly into a heap temporary, so this build is SLOWER than "
                   "one against real Eigen would be; the restatement beside it (same bits, no temporaries) bounds it from the other "
                   "side" % (n_ref, float(t_ref.sum())),
            p50_ms_per_solve=float(np.median(t_ref)) * 1e3, p95_ms_per_solve=float(np.percentile(t_ref, 95)) * 1e3,
            us_per_iteration=float(1e6 * t_ref.sum() / max(1, sum(q_["iters"] for q_ in ref_runs))),
            bit_equal_to_restatement_on_all=bool(same), restatement=restatement)
    else:
        out["cpu_baseline"] = dict(common, value=restatement["solves_per_s"], cores=cores, kind="port",
            sample="%d trajectories of the same batch, literal-order oracle (oracle/_ref is not built on this box)" % ns,
            restatement=restatement)
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


def parity_reference_order(ctx, args, po, st, r, sample, out, B_total):
    """(2) the REFERENCE-ORDER device mode (dftpav_batch_set_order, solver_ref.hip) on the whole batch: every sum in the
    reference's order, so its solves must equal OptimizeTrajectory's bit for bit -- checked against the reference build on the 64
    trajectories timed by cpu_baseline and against the restatement on all it solved; the same stream of planning cycles as the
    value line in that order; (3) the bias of the device order against it.  -> the reference-order results of the batch, or None"""
    shard, rc, ref_runs = st.shard, sample["rc"], sample["ref_runs"]
    try:
        hR, bR = reference_order_batch(ctx, shard)
        bR.solve_async(); bR.sync()
        bR.solve_async(); bR.sync()
        ref_ms = bR.last_solve_ms()
        ref_gpu = bR.results()
        eq_port = [same_solve(ref_gpu, g_, rc, i_) for i_, g_ in enumerate(sample["sub_idx"])]
        ro = {"trajectories": int(len(sample["sub_idx"])), "bit_equal": int(sum(eq_port)),
              "against": "the literal restatement (bit-equal to oracle/_ref): final x, cost, status, iterations, evaluations",
              "batch_solved_on_device": int(shard.B), "kernel_ms": ref_ms, "solves_per_s": shard.B / (ref_ms * 1e-3),
              "us_per_iteration_of_the_longest": 1e3 * ref_ms / max(1, int(ref_gpu["iters"].max())),
              "slowdown_vs_device_order_isolated": None}
        if ref_runs is not None:
            eq_ref = [bool(ref_gpu["final_cost"][g_] == ref_runs[i_]["final_cost"] and np.array_equal(ref_gpu["x"][g_], ref_runs[i_]["x"]) and
                           ref_gpu["iters"][g_] == ref_runs[i_]["iters"] and ref_gpu["evals"][g_] == ref_runs[i_]["evals"] and
                           ref_gpu["status"][g_] == ref_runs[i_]["status"]) for i_, g_ in enumerate(sample["pick1"])]
            ro["against_reference_build"] = {"trajectories": int(sample["n_ref"]), "bit_equal": int(sum(eq_ref))}
        if "isolated" in out:
            ro["slowdown_vs_device_order_isolated"] = ref_ms / out["isolated"]["kernel_ms"]
        ro["isolated_solves_per_s"] = ro["solves_per_s"]
        out["parity"]["reference_order"] = ro
        # the same stream of planning cycles as the value line -- --depth resident batches on as many HIP streams, one launched
        # while the other thins out -- in the REFERENCE'S order: the throughput of the bit-equal mode
        bR.close(); hR.close()
        try:
            stR = Stream(ctx, B_total, args.config, args.seed, depth=args.depth, order=capi.ORDER_REFERENCE)
            k_ref, w_ref = max(4, min(args.steps, 20)), max(2, min(args.warmup, 5))   # the value line's K and W at the driver's settings
            rR = stR.run(k_ref, w_ref)
            same = bool(np.array_equal(rR["rs"][0]["final_cost"], ref_gpu["final_cost"])) if stR.shards[0].B == shard.B else None
            ro["overlapped"] = {"solves_per_s": rR["value"], "ms_per_step": rR["ms_per_step"], "steps": k_ref, "warmup": w_ref,
                                "schedule": ctx.schedule, "first_batch_equals_the_isolated_solve": same}
            ro["solves_per_s"] = rR["value"]
            ro["solves_per_s_is"] = "the overlapped stream of %d steps (as the value line); isolated_solves_per_s: one batch alone" % k_ref
            stR.close()
        except capi.DftpavError as ex:
            ro["overlapped"] = {"failed": str(ex)}
        # (3) device order against the reference over the WHOLE batch, with the reference-order solves standing for the
        # reference (they are it, bit for bit): the solver is chaotic, so the two follow different iterate
        # sequences after the first rounding difference; the question is whether the device order is BIASED.  Control: the
        # reference against itself with one waypoint coordinate of x0 moved by one ulp -- same size of effect, no bias possible.
        hR, bR = reference_order_batch(ctx, shard)
        sh1 = shard.subset(np.arange(shard.B))
        sh1.inner_pts = np.ascontiguousarray(sh1.inner_pts).copy()
        sh1.inner_pts[:, 0] = np.nextafter(sh1.inner_pts[:, 0], np.inf)
        bR.upload(sh1)
        bR.solve_async(); bR.sync()
        ulp_gpu = bR.results()
        bR.close(); hR.close()
        out["parity"]["bias"] = parity_bias(r, ref_gpu, ulp_gpu)
        return ref_gpu
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
