#!/usr/bin/env python
"""bench.py — batched trajectory solves/sec of the MINCO/L-BFGS solve path on MI355X.

A "step" = one pass of the hot path over one resident batch: every rank launches the
persistent solve kernel on its shard (inputs already in HBM), packs {cost,status,iters}
records on the device and — for N>1 — joins the single RCCL all-gather of SURVEY §8(e).

Workload (config.workload): the BASELINE.json configs[2]/[3] problem — random-restart /
multi-hypothesis trajectories of 16 MINCO pieces x 32 pts/piece (33 samples), 50 static
obstacles, every trajectory with its own H=4 rectangle corridor — at 4096 trajectories per GPU
(weak scaling: per-GPU batch fixed; the solver is latency-bound per trajectory, so throughput is
quoted on a batch that fills the chip).  At N=1 the line also carries the exact configs[2] case
(batch 256) and the configs[1] case (one gear-shift trajectory) under "batch256" / "single".
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402
import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

from dftpav_amd import capi, distributed as dd, scenarios as sc  # noqa: E402

HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: HBM3E peak BW 8.0 TB/s


def algorithmic_bytes(lay, npts, H, M, iters, evals, hist_sum, w=8):
    """BASELINE.md §4 / SURVEY §8(d): E_solve = evals*E_eval + (4*sum_k h_k*n + 14*n*iters)*w."""
    n = lay.n_vars
    e_eval = (npts * H * 4 + 2 * n + 12 * M + 1) * w
    return evals.astype(np.float64) * e_eval + (4.0 * hist_sum * n + 14.0 * n * iters) * w


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--batch-per-gpu", type=int, default=4096)
    ap.add_argument("--config", type=int, default=3, help="BASELINE config (1-based) used as the workload")
    ap.add_argument("--cpu-sample", type=int, default=-1,
                    help="trajectories timed on the host cores (-1 = 4 per core, 0 = skip)")
    ap.add_argument("--seed", type=int, default=20240)
    ap.add_argument("--no-extras", action="store_true", help="skip the batch-256 / single-trajectory side runs (profiling)")
    ap.add_argument("--schedule", choices=["overlap", "chain", "plain"], default="overlap",
                    help="overlap: two batches on two HIP streams, every trajectory stays in its queue launch, the next batch's "
                         "launch fills the slots the previous one frees (default); chain: one stream, the stragglers of a batch are "
                         "adopted by the next batch's launch; plain: isolated solves (the tail of each batch runs on a nearly empty device)")
    ap.add_argument("--no-chain", action="store_true", help="same as --schedule plain")
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            raise SystemExit("launch with torch.distributed.run --nproc-per-node %d" % args.gpus)
    # DFTPAV_BENCH_FORCE_DIST=1: take the RCCL path (init, barrier, all-gather, max-reduce) at world size 1 too
    distributed = world > 1 or os.environ.get("DFTPAV_BENCH_FORCE_DIST") == "1"
    torch.cuda.set_device(local_rank)
    if distributed:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local_rank))

    # ---- inputs: every rank generates its own shard from (seed, rank) — trajectories are independent,
    # nothing is scattered; rank r owns global trajectories [r*B/G, (r+1)*B/G)
    B_total = args.batch_per_gpu * world
    params = capi.default_params()
    lo, hi = dd.shard_range(B_total, rank, world)
    # Two resident batches of different problems, solved alternately: a stream of planning cycles.  A step launches
    # one batch and delivers the records of the batch that this completes (pack + all-gather); after the last step
    # the outstanding batch is completed and delivered INSIDE the timed region, so K steps deliver K batches.
    #   overlap (default): each batch on its own handle = HIP stream, hand-over 0: every trajectory finishes in its
    #     queue launch, and while that launch thins out the other stream's launch takes the freed workgroup slots.
    #   chain: one stream; the last trajectories of a batch are adopted by the next batch's queue launch
    #     (dftpav_batch_solve_chained), the last batch is flushed in the latency shape.
    #   plain: isolated solves, a step waits for its own batch.
    schedule = "plain" if args.no_chain else args.schedule
    shards = [sc.baseline_config(args.config, B=hi - lo, seed=args.seed + 7919 * rank + 104729 * i) for i in range(2)]
    for sh in shards:
        sh.apply_resolution(params)
    shard = scen = shards[0]
    h = capi.Handle(params, device=local_rank)
    h.set_surround(shard.surround)
    hs = [h, h]
    if schedule == "overlap":
        hs = [h, capi.Handle(params, device=local_rank)]
        hs[1].set_surround(shard.surround)
    bts = []
    for hh, sh in zip(hs, shards):
        b_ = capi.Batch(hh, sh.layout, sh.B)
        b_.upload(sh)  # resident in HBM from here on
        if schedule == "overlap":
            b_.set_hand_over(0)
        bts.append(b_)
    bt = bts[0]
    rec_dev = [torch.zeros((shard.B, dd.RECORD_BYTES), dtype=torch.uint8, device="cuda") for _ in range(2)]
    state = {"k": 0, "prev": None, "rec": None}

    def deliver(i):
        bts[i].pack_results(rec_dev[i].data_ptr())
        bts[i].sync()
        state["rec"] = (dd.allgather_records(rec_dev[i], B_total) if distributed else rec_dev[i], i)

    def step(last=False):
        i = state["k"] % 2
        cur, prev = bts[i], state["prev"]
        state["k"] += 1
        if schedule == "plain":
            cur.solve_async()
            deliver(i)
            return
        if schedule == "chain":
            cur.solve_chained(bts[prev] if prev is not None else None)  # prev is complete when this call's launches are
        else:
            # prev keeps running on the other stream.  Nothing follows the last launch of a run, so it ends with the
            # default end game (its stragglers in the latency shape) instead of thinning out alone.
            cur.set_hand_over(-1 if last else 0)
            cur.solve_async()
        if prev is not None:
            deliver(prev)
        state["prev"] = i

    def flush():
        """the outstanding batch: its stragglers in the latency shape (chain) / the rest of its launch (overlap)"""
        if state["prev"] is not None:
            if schedule == "chain":
                bts[state["prev"]].finish()
            deliver(state["prev"])
            state["prev"] = None

    for j in range(args.warmup):
        step(last=(j == args.warmup - 1))
    flush()  # the warm-up leaves nothing in flight: the timed region starts on an idle device
    if distributed:
        dist.barrier()
    torch.cuda.synchronize()
    first = state["k"] % 2
    hs[first].mark(0)  # HIP events on the library's own streams around the timed region
    t0 = time.perf_counter()
    for j in range(args.steps):
        step(last=(j == args.steps - 1))
    flush()
    last_h = hs[state["rec"][1]]
    last_h.mark(1)
    if distributed:
        dist.barrier()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    gpu_ms = last_h.elapsed_since(hs[first], 0, 1)
    if distributed:
        tmax = torch.tensor([elapsed], dtype=torch.float64, device="cuda")
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        elapsed = float(tmax.item())

    allrec, last = state["rec"]
    rs = [b_.results() for b_ in bts]
    r = rs[0]
    cost_all, status_all, iters_all = dd.unpack_records(allrec.cpu().numpy())
    assert len(cost_all) == B_total and np.array_equal(cost_all[lo:hi], rs[last]["final_cost"])

    if rank == 0:
        ms_per_step = 1e3 * elapsed / args.steps
        value = B_total * args.steps / elapsed
        lay = shard.layout
        eb = [float(algorithmic_bytes(lay, shard.n_points, lay.H, lay.M, q["iters"], q["evals"], q["hist_sum"]).sum()) for q in rs]
        ebytes_steps = sum(eb[(state["k"] - args.steps + j) % 2] for j in range(args.steps))  # the batches the timed steps solved
        kms = gpu_ms / args.steps  # device time of the timed region (marker events on the library's streams) per step
        achieved = ebytes_steps / args.steps / (kms * 1e-3) / 1e9
        traffic = None
        pmc = os.path.join(ROOT, "profiles", "pmc_latest.json")
        if os.path.exists(pmc):
            try:
                traffic = json.load(open(pmc)).get("hbm_bytes_per_launch")
            except Exception:
                traffic = None
        out = {
            "metric": "trajectory solves/sec (batched), 16-piece MINCO",
            "value": value, "unit": "solves/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": ms_per_step, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f64", "data": "synthetic",
            "config": {"workload": "BASELINE configs[2]/[3] problem (%s): %d trajectories/GPU x %d pieces x %d pts/piece, "
                                   "50 static obstacles, H=4 rectangle corridor per trajectory, fp64 bit-exact mode" %
                                   (scen.name, args.batch_per_gpu, lay.n_pieces, scen.K + 1),
                       "global_batch": B_total, "pieces": lay.n_pieces, "pts_per_piece": scen.K + 1,
                       "n_vars": lay.n_vars, "parallelism": "batch-sharded x%d, 1 all-gather of 16B records" % world},
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBS, "traffic": traffic,
                         "kernel": "solver_kernel", "kernel_ms": kms,
                         "launches_per_step": 1 if (schedule == "overlap" or shard.B < 4 * torch.cuda.get_device_properties(local_rank).multi_processor_count) else 2,
                         "algorithmic_bytes_per_launch": ebytes_steps / args.steps},
            "schedule": {"overlap": "overlap: two batches alternate on two HIP streams, every trajectory finishes in its queue launch, "
                                    "the next launch takes the slots the previous one frees; the last batch completes inside the timed region",
                         "chain": "chain: one stream, the stragglers of a batch finish inside the next batch's queue launch, the last "
                                  "batch is flushed inside the timed region",
                         "plain": "plain: every batch finishes on its own"}[schedule],
            "p50_ms_per_solve": float(np.median(r["latency_us"])) * 1e-3,
            "p95_ms_per_solve": float(np.percentile(r["latency_us"], 95)) * 1e-3,
            "mean_iters": float(r["iters"].mean()), "mean_evals": float(r["evals"].mean()),
            "mean_hist_depth": float(r["hist_sum"].sum() / max(1, r["iters"].sum())),
            "success_rate": float(r["success"].mean()),
        }
        if world == 1 and not args.no_extras:
            # ---- the exact BASELINE configs[2] case (batch 256) and configs[1] (one gear-shift trajectory)
            def side(cfg, B, reps):
                p2 = capi.default_params()
                s2 = sc.baseline_config(cfg, B=B, seed=args.seed)
                s2.apply_resolution(p2)
                h2 = capi.Handle(p2, device=local_rank)
                b2 = capi.Batch(h2, s2.layout, B)
                b2.upload(s2)
                b2.solve_async(); b2.sync()
                ms = []
                for _ in range(reps):
                    b2.solve_async(); b2.sync(); ms.append(b2.last_solve_ms())
                r2 = b2.results()
                b2.close(); h2.close()
                return {"batch": B, "solves_per_s": B / (float(np.mean(ms)) * 1e-3), "kernel_ms": float(np.mean(ms)),
                        "p50_ms_per_solve": float(np.median(r2["latency_us"])) * 1e-3, "mean_iters": float(r2["iters"].mean())}
            # the same batch as isolated solves (no chaining: its tail runs on a nearly empty device)
            iso = []
            bt.set_hand_over(-1)  # the plan's default end game (the overlap schedule runs with 0)
            for _ in range(3):
                bt.solve_async(); bt.sync(); iso.append(bt.last_solve_ms())
            out["isolated"] = {"batch": int(shard.B), "kernel_ms": float(np.mean(iso)), "solves_per_s": shard.B / (float(np.mean(iso)) * 1e-3)}
            out["batch256"] = side(3, 256, 3)
            # one gear-shift trajectory alone on the GPU: the solver is chaotic (an instance needs 90 or 340 iterations
            # depending on the last bit), so the latency is quoted as the median over 9 seeded instances, with the
            # per-iteration time beside it
            def single(cfg, seeds):
                p2 = capi.default_params()
                ms, its = [], []
                for sd in seeds:
                    s2 = sc.baseline_config(cfg, B=1, seed=args.seed + 17 * sd)
                    s2.apply_resolution(p2)
                    h2 = capi.Handle(p2, device=local_rank)
                    b2 = capi.Batch(h2, s2.layout, 1)
                    b2.upload(s2)
                    b2.solve_async(); b2.sync()
                    b2.solve_async(); b2.sync()
                    ms.append(b2.last_solve_ms()); its.append(int(b2.results()["iters"][0]))
                    b2.close(); h2.close()
                ms, its = np.array(ms), np.array(its)
                return {"batch": 1, "instances": len(seeds), "p50_ms_per_solve": float(np.median(ms)), "min_ms": float(ms.min()),
                        "max_ms": float(ms.max()), "median_iters": float(np.median(its)), "us_per_iteration": float(1e3 * ms.sum() / its.sum()),
                        "solves_per_s": float(1e3 / np.median(ms))}
            out["single"] = single(2, range(9))
            out["moving_obstacles_1024"] = side(5, 1024, 1)  # BASELINE configs[4]: 32 pieces x 65 pts, 4 moving cars
            # ---- the step before the solve: rectangle corridors of the shard's hypotheses on the device
            st = shard.meta["states"].reshape(-1, 3)
            cen = (0.5 * (st[:, 0].min() + st[:, 0].max()), 0.5 * (st[:, 1].min() + st[:, 1].max()))
            span = max(st[:, 0].max() - st[:, 0].min(), st[:, 1].max() - st[:, 1].min()) + 40.0
            grid, origin = sc.occupancy_grid(shard.meta["obstacles"], arena=span, centre=cen)
            h.set_grid_map(grid, sc.MAP_RESL, origin)
            Hc = h.corridor_rectangles(st)
            tcor = []
            for _ in range(3):
                t1 = time.perf_counter(); Hc = h.corridor_rectangles(st); tcor.append(time.perf_counter() - t1)
            cor_ms = h.corridor_last_ms()
            # ---- the step after the solve: collision re-check of all solved trajectories of the shard
            colv, firstv = bt.validate()
            out["validate"] = {"trajectories": int(shard.B), "kernel_ms": h.corridor_last_ms(),
                               "trajectories_per_s": shard.B / (h.corridor_last_ms() * 1e-3),
                               "colliding": int(colv.sum())}
            from oracle import pyoracle as po  # the checker, never the thing measured
            po.build()
            # ---- the read-out of the result: GetState every 10 ms over every solved trajectory
            cor, dts = bt.coeffs()
            n_rd = int(float(np.max(np.sum(dts * shard.layout.piece_nums[None, :], axis=1))) / 0.01) + 2
            rd, nv = bt.sample_states(sample_dt=0.01, n_samples=n_rd)
            rd_ms = h.corridor_last_ms()
            ord_, onv = po.sample_states(cor[:64], dts[:64], shard.layout.piece_nums, shard.layout.singuls, sample_dt=0.01,
                                         n_samples=n_rd, wheel_base=params.veh_wheel_base, order=1)
            out["readout"] = {"trajectories": int(shard.B), "samples_per_trajectory": n_rd, "kernel_ms": rd_ms,
                              "states_per_s": float(nv.sum()) / (rd_ms * 1e-3), "written_GB_per_s": rd.nbytes / (rd_ms * 1e-3) / 1e9,
                              "oracle_bit_exact_on_first_64": bool(np.array_equal(rd[:64], ord_) and np.array_equal(nv[:64], onv))}
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
            nchk = min(2000, len(st))
            out["corridor"] = {"states": int(len(st)), "map_cells": [int(grid.shape[1]), int(grid.shape[0])],
                               "kernel_ms": cor_ms, "rectangles_per_s": len(st) / (cor_ms * 1e-3),
                               "rectangles_per_s_with_pcie": len(st) / min(tcor),
                               "oracle_bit_exact_on_first_%d" % nchk: bool(np.array_equal(
                                   Hc[:nchk], po.corridor_rectangles(grid, sc.MAP_RESL, origin, st[:nchk], order=1)))}
        # ---- reference CPU path beside it (rank 0, N=1 only): the oracle's literal restatement on the host cores
        if world == 1 and args.cpu_sample != 0:
            from oracle import pyoracle as po
            po.build()
            cores = os.cpu_count() or 1
            ns = args.cpu_sample if args.cpu_sample > 0 else min(4 * cores, 1024)
            sub = shard.subset(np.arange(ns) % shard.B)
            tc = time.perf_counter()
            rc = po.solve_batch(params, sub, nthreads=cores, order=0)
            wall = time.perf_counter() - tc
            nd = min(max(32, cores), shard.B)  # one trajectory per core: about one solve time of wall clock
            pick = (np.arange(nd) * max(1, shard.B // nd)) % shard.B  # strided through the batch (restarts of all hypotheses)
            rd = po.solve_batch(params, shard.subset(pick), nthreads=cores, order=1)
            match = bool(np.array_equal(rd["final_cost"], r["final_cost"][pick]) and np.array_equal(rd["x"], r["x"][pick]) and
                         np.array_equal(rd["iters"], r["iters"][pick]))
            out["cpu_baseline"] = {"value": ns / wall, "unit": "solves/s", "cores": cores, "kind": "port",
                                   "sample": "%d trajectories of the same batch (4 per core), literal-order oracle "
                                             "(fp64 restatement of traj_optimizer.cpp/lbfgs.hpp), OpenMP over "
                                             "trajectories, %.1f core-seconds" % (ns, float(rc["seconds"].sum())),
                                   "p50_ms_per_solve_per_thread": float(np.median(rc["seconds"])) * 1e3,
                                   "mean_iters": float(rc["iters"].mean())}
            # how the reference runs it: one planner thread, the other cores idle
            r1 = po.solve_batch(params, shard.subset((np.arange(16) * max(1, shard.B // 16) + 17) % shard.B), nthreads=1, order=0)
            out["cpu_baseline"]["single_thread_p50_ms_per_solve"] = float(np.median(r1["seconds"])) * 1e3
            out["cpu_baseline"]["single_thread_p95_ms_per_solve"] = float(np.percentile(r1["seconds"], 95)) * 1e3
            out["parity"] = {"device_order_oracle_bit_exact_on_%d_sampled" % nd: match}
        print(json.dumps(out), flush=True)
    for b_ in bts:
        b_.close()
    for hh in set(hs):
        hh.close()
    if distributed:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
