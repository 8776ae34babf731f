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

# the cpu_baseline leg runs the oracle with OpenMP over trajectories: pin its threads to cores, and do it before torch / numpy
# load their OpenMP runtime (libgomp reads these once)
os.environ.setdefault("OMP_PROC_BIND", "close")
os.environ.setdefault("OMP_PLACES", "cores")
os.environ.setdefault("OMP_WAIT_POLICY", "passive")

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


def effective_cores():
    """host cores this process may actually use: the affinity mask, capped by the cgroup CPU quota"""
    n_aff = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    quota = None
    for path in ("/sys/fs/cgroup/cpu.max", "/sys/fs/cgroup/cpu/cpu.cfs_quota_us"):
        try:
            txt = open(path).read().split()
            if path.endswith("cpu.max"):
                if txt[0] != "max":
                    quota = float(txt[0]) / float(txt[1])
            else:
                q = float(txt[0])
                if q > 0:
                    quota = q / float(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
            break
        except Exception:
            continue
    eff = n_aff if quota is None else max(1, min(n_aff, int(quota + 0.5)))
    return dict(logical=os.cpu_count() or 1, affinity=n_aff, cgroup_quota=quota, effective=eff)


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
    ap.add_argument("--scaling", choices=["weak", "strong"], default="weak",
                    help="weak: --batch-per-gpu trajectories on every GPU (the value line); strong: --batch-per-gpu trajectories in all, "
                         "sharded over the GPUs (BASELINE configs[3]: 4096 over 8 = 512 per GPU).  At N > 1 the other mode is timed as "
                         "well and reported beside the value line")
    ap.add_argument("--literal-sample", type=int, default=-1,
                    help="trajectories of the batch solved by the literal oracle for parity.literal (-1 = as many as ~20 s of the host cores allow)")
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

    params = capi.default_params()
    schedule = "plain" if args.no_chain else args.schedule
    n_cu = torch.cuda.get_device_properties(local_rank).multi_processor_count

    class Stream:
        """A stream of planning cycles on this rank: two resident batches of different problems, solved alternately.  A step
        launches one batch and delivers the records of the batch that this completes (pack + all-gather); after the last step
        the outstanding batch is completed and delivered INSIDE the timed region, so K steps deliver K batches.
          overlap (default): each batch on its own handle = HIP stream, hand-over 0: every trajectory finishes in its queue
            launch, and while that launch thins out the other stream's launch takes the freed workgroup slots.
          chain: one stream; the last trajectories of a batch are adopted by the next batch's queue launch
            (dftpav_batch_solve_chained), the last batch is flushed in the latency shape.
          plain: isolated solves, a step waits for its own batch."""

        def __init__(self, B_total, config, seed):
            # every rank generates its own shard from (seed, rank) — trajectories are independent, nothing is scattered
            # (SURVEY §8e); rank r owns global trajectories [r*B/G, (r+1)*B/G)
            self.B_total = B_total
            self.lo, self.hi = dd.shard_range(B_total, rank, world)
            self.shards = [sc.baseline_config(config, B=self.hi - self.lo, seed=seed + 7919 * rank + 104729 * i) for i in range(2)]
            for sh in self.shards:
                sh.apply_resolution(params)
            self.shard = self.shards[0]
            h = capi.Handle(params, device=local_rank)
            h.set_surround(self.shard.surround)
            self.hs = [h, h]
            if schedule == "overlap":
                self.hs = [h, capi.Handle(params, device=local_rank)]
                self.hs[1].set_surround(self.shard.surround)
            self.bts = []
            for hh, sh in zip(self.hs, self.shards):
                b_ = capi.Batch(hh, sh.layout, sh.B)
                b_.upload(sh)  # resident in HBM from here on
                if schedule == "overlap":
                    b_.set_hand_over(0)
                self.bts.append(b_)
            self.rec_dev = [torch.zeros((self.shard.B, dd.RECORD_BYTES), dtype=torch.uint8, device="cuda") for _ in range(2)]
            self.k, self.prev, self.rec = 0, None, None
            self.t_launch = [0.0, 0.0]
            self.to_result, self.in_deliver = [], []

        def deliver(self, i):
            t1 = time.perf_counter()
            self.bts[i].pack_results(self.rec_dev[i].data_ptr())
            self.bts[i].sync()
            self.rec = (dd.allgather_records(self.rec_dev[i], self.B_total) if distributed else self.rec_dev[i], i)
            t2 = time.perf_counter()
            self.in_deliver.append(t2 - t1)
            self.to_result.append(t2 - self.t_launch[i])

        def step(self, last=False):
            i = self.k % 2
            cur, prev = self.bts[i], self.prev
            self.k += 1
            self.t_launch[i] = time.perf_counter()
            if schedule == "plain":
                cur.solve_async()
                self.deliver(i)
                return
            if schedule == "chain":
                cur.solve_chained(self.bts[prev] if prev is not None else None)  # prev is complete when this call's launches are
            else:
                # prev keeps running on the other stream.  Nothing follows the last launch of a run, so it ends with the
                # default end game (its stragglers in the latency shape) instead of thinning out alone.
                cur.set_hand_over(-1 if last else 0)
                cur.solve_async()
            if prev is not None:
                self.deliver(prev)
            self.prev = i

        def flush(self):
            """the outstanding batch: its stragglers in the latency shape (chain) / the rest of its launch (overlap)"""
            if self.prev is not None:
                if schedule == "chain":
                    self.bts[self.prev].finish()
                self.deliver(self.prev)
                self.prev = None

        def run(self, steps, warmup):
            for j in range(warmup):
                self.step(last=(j == warmup - 1))
            self.flush()  # the warm-up leaves nothing in flight: the timed region starts on an idle device
            self.to_result, self.in_deliver = [], []
            if distributed:
                dist.barrier()
            torch.cuda.synchronize()
            first = self.k % 2
            self.hs[first].mark(0)  # HIP events on the library's own streams around the timed region
            t0 = time.perf_counter()
            for j in range(steps):
                self.step(last=(j == steps - 1))
            self.flush()
            last_h = self.hs[self.rec[1]]
            last_h.mark(1)
            if distributed:
                dist.barrier()
            torch.cuda.synchronize()
            elapsed = time.perf_counter() - t0
            gpu_ms = last_h.elapsed_since(self.hs[first], 0, 1)
            if distributed:
                tmax = torch.tensor([elapsed], dtype=torch.float64, device="cuda")
                dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
                elapsed = float(tmax.item())
            allrec, last = self.rec
            rs = [b_.results() for b_ in self.bts]
            cost_all, status_all, iters_all = dd.unpack_records(allrec.cpu().numpy())
            assert len(cost_all) == self.B_total and np.array_equal(cost_all[self.lo:self.hi], rs[last]["final_cost"])
            return dict(elapsed=elapsed, gpu_ms=gpu_ms, rs=rs, steps=steps, value=self.B_total * steps / elapsed,
                        ms_per_step=1e3 * elapsed / steps, to_result_ms=1e3 * float(np.mean(self.to_result)),
                        deliver_ms=1e3 * float(np.mean(self.in_deliver)))

        def close(self):
            for b_ in self.bts:
                b_.close()
            for hh in set(self.hs):
                hh.close()

    # the value line: weak = --batch-per-gpu on every GPU, strong = --batch-per-gpu in all (BASELINE configs[3] as written)
    B_main = args.batch_per_gpu * world if args.scaling == "weak" else args.batch_per_gpu
    main_stream = Stream(B_main, args.config, args.seed)
    res = main_stream.run(args.steps, args.warmup)
    other = None
    if world > 1:  # the other scaling mode beside it
        B_other = args.batch_per_gpu if args.scaling == "weak" else args.batch_per_gpu * world
        o_stream = Stream(B_other, args.config, args.seed + 1)
        o = o_stream.run(args.steps, args.warmup)
        other = {"scaling": "strong" if args.scaling == "weak" else "weak", "global_batch": B_other, "per_gpu": B_other // world,
                 "value": o["value"], "ms_per_step": o["ms_per_step"], "unit": "solves/s"}
        o_stream.close()
    B_total = B_main
    shard = scen = main_stream.shard
    bts, hs, bt, h = main_stream.bts, main_stream.hs, main_stream.bts[0], main_stream.hs[0]
    rs = res["rs"]
    r = rs[0]
    elapsed, gpu_ms = res["elapsed"], res["gpu_ms"]
    state = {"k": main_stream.k}

    if rank == 0:
        ms_per_step = res["ms_per_step"]
        value = res["value"]
        lay = shard.layout
        eb = [float(algorithmic_bytes(lay, shard.n_points, lay.H, lay.M, q["iters"], q["evals"], q["hist_sum"]).sum()) for q in rs]
        ebytes_steps = sum(eb[(state["k"] - args.steps + j) % 2] for j in range(args.steps))  # the batches the timed steps solved
        kms = gpu_ms / args.steps  # device time of the timed region (marker events on the library's streams) per step
        achieved = ebytes_steps / args.steps / (kms * 1e-3) / 1e9
        traffic, traffic_source = None, None
        pmc = os.path.join(ROOT, "profiles", "pmc_latest.json")
        if os.path.exists(pmc):
            try:
                pj = json.load(open(pmc))
                traffic = pj.get("hbm_bytes_per_launch")
                traffic_source = "profiles/pmc_latest.json: separate rocprofv3 --pmc passes of `%s` (%s)" % (
                    pj.get("command", "bench.py --steps 3 --no-extras"), pj.get("collected", "round 1"))
            except Exception:
                traffic = None
        out = {
            "metric": "trajectory solves/sec (batched), 16-piece MINCO",
            "value": value, "unit": "solves/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": ms_per_step, "higher_is_better": True, "scaling": args.scaling, "vs_baseline": None,
            "dtype": "f64", "data": "synthetic",
            "config": {"workload": "BASELINE configs[2]/[3] problem (%s): %d trajectories%s x %d pieces x %d pts/piece, "
                                   "50 static obstacles, H=4 rectangle corridor per trajectory, fp64 bit-exact mode" %
                                   (scen.name, args.batch_per_gpu, "/GPU" if args.scaling == "weak" else " in all", lay.n_pieces, scen.K + 1),
                       "global_batch": B_total, "pieces": lay.n_pieces, "pts_per_piece": scen.K + 1,
                       "n_vars": lay.n_vars, "parallelism": "batch-sharded x%d, 1 all-gather of 16B records" % world},
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBS, "traffic": traffic, "traffic_source": traffic_source,
                         "kernel": "solver_kernel", "kernel_ms": kms,
                         "launches_per_step": 1 if (schedule == "overlap" or shard.B < 4 * n_cu) else 2,
                         "algorithmic_bytes_per_launch": ebytes_steps / args.steps},
            "schedule": {"overlap": "overlap: two batches alternate on two HIP streams, every trajectory finishes in its queue launch, "
                                    "the next launch takes the slots the previous one frees; the last batch completes inside the timed region",
                         "chain": "chain: one stream, the stragglers of a batch finish inside the next batch's queue launch, the last "
                                  "batch is flushed inside the timed region",
                         "plain": "plain: every batch finishes on its own"}[schedule],
            "p50_ms_per_solve": float(np.median(r["latency_us"])) * 1e-3,
            "p95_ms_per_solve": float(np.percentile(r["latency_us"], 95)) * 1e-3,
            # host clock, per batch of the timed steps: launch -> its records delivered (the pack kernel of a batch waits for
            # workgroup slots behind the other stream's persistent workgroups), and the part of it spent inside deliver()
            "time_to_result_ms": res["to_result_ms"], "deliver_ms": res["deliver_ms"],
            "mean_iters": float(r["iters"].mean()), "mean_evals": float(r["evals"].mean()),
            "mean_hist_depth": float(r["hist_sum"].sum() / max(1, r["iters"].sum())),
            "success_rate": float(r["success"].mean()),
        }
        if other is not None:
            out["other_scaling"] = other
        cpu = effective_cores()
        if world == 1 and not args.no_extras:
            # ---- the exact BASELINE configs[2] case (batch 256) and configs[1] (one gear-shift trajectory)
            from oracle import pyoracle as po  # the checker, never the thing measured
            po.build()

            def bit_check(p2, s2, r2, pick):
                """sampled trajectories of a side run against the device-order oracle: every field bit for bit"""
                ro = po.solve_batch(p2, s2.subset(pick), nthreads=min(len(pick), cpu["effective"]), order=1)
                return bool(all(np.array_equal(ro[k_], r2[k_][pick]) for k_ in ("final_cost", "x", "iters", "evals", "status")))

            def side(cfg, B, reps, n_check):
                p2 = capi.default_params()
                s2 = sc.baseline_config(cfg, B=B, seed=args.seed)
                s2.apply_resolution(p2)
                h2 = capi.Handle(p2, device=local_rank)
                h2.set_surround(s2.surround)
                b2 = capi.Batch(h2, s2.layout, B)
                b2.upload(s2)
                b2.solve_async(); b2.sync()
                ms = []
                for _ in range(reps):
                    b2.solve_async(); b2.sync(); ms.append(b2.last_solve_ms())
                r2 = b2.results()
                # a stream of such batches (planning cycles back to back on several planner threads): 8 resident batches on 8 HIP
                # streams in the throughput residency (four workgroups per CU, dftpav_batch_create_shaped), 3 rounds
                hx = [capi.Handle(p2, device=local_rank) for _ in range(8)]
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
                ok = bit_check(p2, s2, r2, pick)
                for bb in bx:
                    bb.close()
                b2.close(); h2.close()
                for hh in hx:
                    hh.close()
                return {"batch": B, "solves_per_s": B / (float(np.mean(ms)) * 1e-3), "kernel_ms": float(np.mean(ms)),
                        "p50_ms_per_solve": float(np.median(r2["latency_us"])) * 1e-3, "mean_iters": float(r2["iters"].mean()),
                        "stream_of_batches": {"streams": len(bx), "batches": rounds * len(bx), "solves_per_s": rounds * len(bx) * B / stream_s,
                                              "results_identical": same},
                        "device_order_oracle_bit_exact_on_%d_sampled" % n_check: ok}
            # the same batch as isolated solves (no chaining: its tail runs on a nearly empty device)
            iso = []
            bt.set_hand_over(-1)  # the plan's default end game (the overlap schedule runs with 0)
            for _ in range(3):
                bt.solve_async(); bt.sync(); iso.append(bt.last_solve_ms())
            out["isolated"] = {"batch": int(shard.B), "kernel_ms": float(np.mean(iso)), "solves_per_s": shard.B / (float(np.mean(iso)) * 1e-3)}
            out["batch256"] = side(3, 256, 3, 8)
            # one gear-shift trajectory alone on the GPU: the solver is chaotic (an instance needs 90 or 340 iterations
            # depending on the last bit), so the latency is quoted as the median over 9 seeded instances, with the
            # per-iteration time beside it
            def single(cfg, seeds):
                p2 = capi.default_params()
                ms, its, oks = [], [], []
                for sd in seeds:
                    s2 = sc.baseline_config(cfg, B=1, seed=args.seed + 17 * sd)
                    s2.apply_resolution(p2)
                    h2 = capi.Handle(p2, device=local_rank)
                    b2 = capi.Batch(h2, s2.layout, 1)
                    b2.upload(s2)
                    b2.solve_async(); b2.sync()
                    b2.solve_async(); b2.sync()
                    r2 = b2.results()
                    ms.append(b2.last_solve_ms()); its.append(int(r2["iters"][0]))
                    oks.append(bit_check(p2, s2, r2, np.array([0])))
                    b2.close(); h2.close()
                ms, its = np.array(ms), np.array(its)
                return {"batch": 1, "instances": len(seeds), "p50_ms_per_solve": float(np.median(ms)), "min_ms": float(ms.min()),
                        "max_ms": float(ms.max()), "median_iters": float(np.median(its)), "us_per_iteration": float(1e3 * ms.sum() / its.sum()),
                        "solves_per_s": float(1e3 / np.median(ms)), "device_order_oracle_bit_exact_on_all": bool(all(oks))}
            out["single"] = single(2, range(9))
            out["moving_obstacles_1024"] = side(5, 1024, 1, 4)  # BASELINE configs[4]: 32 pieces x 65 pts, 4 moving cars
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
        # ---- reference CPU path beside it (rank 0, N=1 only): the literal oracle on the host cores.  It is bit-equal to the
        # reference's own code compiled against the interface stand-ins (oracle/_ref, tests/test_ref_pin.py), without that
        # build's per-expression heap temporaries.
        if world == 1 and args.cpu_sample != 0:
            from oracle import pyoracle as po
            po.build()
            torch.set_num_threads(1)
            cores = cpu["effective"]
            # how the reference runs it: one planner thread (traj_server_ros.cpp:100), the other cores idle
            pick1 = (np.arange(16) * max(1, shard.B // 16) + 17) % shard.B
            r1 = po.solve_batch(params, shard.subset(pick1), nthreads=1, order=0)
            t1 = float(np.median(r1["seconds"]))
            ns = args.cpu_sample if args.cpu_sample > 0 else int(min(max(4 * cores, 15.0 * cores / max(t1, 1e-3)), 8192))
            sub_idx = np.arange(ns) % shard.B
            tc = time.perf_counter()
            rc = po.solve_batch(params, shard.subset(sub_idx), nthreads=cores, order=0)
            wall = time.perf_counter() - tc
            out["cpu_baseline"] = {"value": ns / wall, "unit": "solves/s", "cores": cores, "kind": "port",
                                   "cores_logical": cpu["logical"], "cores_affinity": cpu["affinity"], "cgroup_cpu_quota": cpu["cgroup_quota"],
                                   "sample": "%d trajectories of the same batch, literal-order oracle (fp64 restatement of traj_optimizer.cpp / "
                                             "lbfgs.hpp, bit-equal to the reference build oracle/_ref), OpenMP over trajectories on %d pinned "
                                             "threads, %.1f s wall, %.1f thread-seconds" % (ns, cores, wall, float(rc["seconds"].sum())),
                                   "p50_ms_per_solve_per_thread": float(np.median(rc["seconds"])) * 1e3,
                                   "single_thread_p50_ms_per_solve": t1 * 1e3,
                                   "single_thread_p95_ms_per_solve": float(np.percentile(r1["seconds"], 95)) * 1e3,
                                   "parallel_efficiency": (ns / wall) / (cores / max(float(np.mean(r1["seconds"])), 1e-9)),
                                   "mean_iters": float(rc["iters"].mean())}
            # ---- parity: (1) bit-for-bit against the device-order oracle on sampled trajectories
            nd = min(max(32, cores), shard.B)
            pick = (np.arange(nd) * max(1, shard.B // nd)) % shard.B  # strided through the batch (restarts of all hypotheses)
            rd = po.solve_batch(params, shard.subset(pick), nthreads=cores, order=1)
            match = bool(np.array_equal(rd["final_cost"], r["final_cost"][pick]) and np.array_equal(rd["x"], r["x"][pick]) and
                         np.array_equal(rd["iters"], r["iters"][pick]))
            out["parity"] = {"device_order_oracle_bit_exact_on_%d_sampled" % nd: match}
            # (2) against the LITERAL oracle (== the reference build, bit for bit) over the whole batch:
            #   a. the literal cost at every final x of the kernel            (same function, rounding-level agreement)
            #   b. lbfgs_optimize restarted by the literal oracle from every final x of the kernel stops at once
            #      (past = 3 iterations is the minimum, lbfgs.hpp:642-659): the kernel's x is a stopping point of the reference
            #   c. the literal solve from the same x0 next to the kernel's: the solver is chaotic, the two follow
            #      different iterate sequences after the first rounding difference, so this is a distribution, not an identity
            ev = po.batch_op(params, shard, "eval", r["x"], nthreads=cores, order=0)
            rel_f = np.abs(ev["f"] - r["final_cost"]) / np.maximum(1.0, np.abs(ev["f"]))
            rst = po.batch_op(params, shard, "restart", r["x"], nthreads=cores, order=0)
            drop = (ev["f"] - rst["final_cost"]) / np.maximum(1.0, np.abs(ev["f"]))
            nl = args.literal_sample if args.literal_sample >= 0 else int(min(shard.B, max(64, 8.0 * cores / max(t1, 1e-3))))
            lit = {"trajectories": int(shard.B),
                   "literal_cost_at_kernel_x_max_rel_diff": float(rel_f.max()),
                   "literal_restart_from_kernel_x": {"iters_p50": float(np.median(rst["iters"])), "iters_p95": float(np.percentile(rst["iters"], 95)),
                                                     "iters_max": int(rst["iters"].max()), "frac_stopping_within_3": float((rst["iters"] <= 3).mean()),
                                                     "frac_stopping_within_5": float((rst["iters"] <= 5).mean()),
                                                     "rel_cost_decrease_p50": float(np.median(drop)), "rel_cost_decrease_p95": float(np.percentile(drop, 95)),
                                                     "rel_cost_decrease_max": float(drop.max())}}
            if nl > 0:
                pl = (np.arange(nl) * max(1, shard.B // nl)) % shard.B
                subl = shard.subset(pl)
                ls = po.solve_batch(params, subl, nthreads=cores, order=0)
                # the reference's own sensitivity, for scale: the same literal solves with one waypoint coordinate moved by one ulp,
                # and the literal solver restarted from its own final points
                sub1 = shard.subset(pl)
                sub1.inner_pts = np.ascontiguousarray(sub1.inner_pts).copy()
                sub1.inner_pts[:, 0] = np.nextafter(sub1.inner_pts[:, 0], np.inf)
                l1 = po.solve_batch(params, sub1, nthreads=cores, order=0)
                rel_1 = np.abs(l1["final_cost"] - ls["final_cost"]) / np.maximum(1.0, np.abs(ls["final_cost"]))
                rs2 = po.batch_op(params, subl, "restart", ls["x"], nthreads=cores, order=0)
                rel_c = (r["final_cost"][pl] - ls["final_cost"]) / np.maximum(1.0, np.abs(ls["final_cost"]))
                lit["literal_solve_from_same_x0"] = {
                    "trajectories": int(nl), "success_rate_kernel": float(r["success"][pl].mean()), "success_rate_literal": float(ls["success"].mean()),
                    "rel_final_cost_diff_abs_p50": float(np.median(np.abs(rel_c))), "rel_final_cost_diff_abs_p95": float(np.percentile(np.abs(rel_c), 95)),
                    "rel_final_cost_diff_signed_mean": float(rel_c.mean()),
                    "frac_within_1e-5": float((np.abs(rel_c) <= 1e-5).mean()), "frac_kernel_cost_not_worse_by_1e-3": float((rel_c <= 1e-3).mean()),
                    "mean_iters_kernel": float(r["iters"][pl].mean()), "mean_iters_literal": float(ls["iters"].mean()),
                    "median_cost_kernel": float(np.median(r["final_cost"][pl])), "median_cost_literal": float(np.median(ls["final_cost"])),
                    "mean_cost_kernel": float(r["final_cost"][pl].mean()), "mean_cost_literal": float(ls["final_cost"].mean())}
                lit["literal_vs_literal_with_x0_moved_by_one_ulp"] = {
                    "trajectories": int(nl), "rel_final_cost_diff_abs_p50": float(np.median(rel_1)),
                    "rel_final_cost_diff_abs_p95": float(np.percentile(rel_1, 95)), "frac_within_1e-5": float((rel_1 <= 1e-5).mean())}
                lit["literal_restart_from_literal_x"] = {"iters_p50": float(np.median(rs2["iters"])), "iters_p95": float(np.percentile(rs2["iters"], 95)),
                                                         "iters_max": int(rs2["iters"].max()), "frac_stopping_within_3": float((rs2["iters"] <= 3).mean()),
                                                         "frac_stopping_within_5": float((rs2["iters"] <= 5).mean())}
            out["parity"]["literal"] = lit
            # (3) one trajectory in lockstep with the reference's line search and two-loop recursion (tests/lockstep.py)
            try:
                sys.path.insert(0, os.path.join(ROOT, "tests"))
                import lockstep
                tb = 0
                bt.trace(tb, 4096)
                bt.set_hand_over(-1)
                bt.solve_async(); bt.sync()
                tr = bt.get_trace()
                bt.trace(tb, 0)
                lp = po.OracleProblem(params, shard, tb, order=0)
                rep = lockstep.replay(tr, lp.eval, params)
                out["parity"]["lockstep"] = {"trajectory": tb, "evaluations": rep["evals"], "iterations_replayed": rep["iterations"],
                                             "branches_identical": rep["branches"], "first_flip": rep["flip"], "rel_f": rep["rel_f"],
                                             "rel_g": rep["rel_g"], "rel_d": rep["rel_d"], "min_branch_margin": float(rep["min_margin"])}
            except AssertionError as ex:
                out["parity"]["lockstep"] = {"failed": str(ex)}
            # ---- PCIe-inclusive rate (never `value`): upload of the whole batch, isolated solve, results back
            tu = time.perf_counter()
            bt.upload(shard)
            t_up = time.perf_counter() - tu
            bt.solve_async(); bt.sync()
            t_sv = bt.last_solve_ms() * 1e-3
            tdn = time.perf_counter()
            bt.results()
            t_dn = time.perf_counter() - tdn
            out["with_upload"] = {"upload_ms": 1e3 * t_up, "solve_ms": 1e3 * t_sv, "download_ms": 1e3 * t_dn,
                                  "solves_per_s": shard.B / (t_up + t_sv + t_dn)}
        print(json.dumps(out), flush=True)
    main_stream.close()
    if distributed:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
