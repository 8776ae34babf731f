#!/usr/bin/env python
"""bench.py — batched trajectory solves/sec of the MINCO/L-BFGS solve path on MI355X.

The value line runs in the REFERENCE ORDER (dftpav_batch_set_order(DFTPAV_ORDER_REFERENCE): every sum in the order the reference
executes it, whole solves bit-equal to the CPU restatement -- the mode that meets north_star's "final cost within 1e-5 relative of
CPU", with 0.0); the device order (reassociated sums, faster, equal to the reference only statistically) is timed beside it under
"device_order" (--order device makes it the value line).

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
# The HIP runtime multiplexes a process's streams onto 4 hardware queues by default: the fifth batch in flight waits for one of
# the first four to END, however empty the device is.  The side runs that keep 8..16 small batches in flight (a 512-trajectory
# shard of a strong-scaled 4096, 8 planner threads of 256) want as many queues as streams: 19 k -> 33 k solves/s on 512-
# trajectory steps.  Read once, when the runtime initialises; no effect on the value line (--depth = 4 streams).  INTEGRATION.md §5.
os.environ.setdefault("GPU_MAX_HW_QUEUES", "16")

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402
import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

from dftpav_amd import capi, distributed as dd, scenarios as sc  # noqa: E402
# the parts (benchlib/): what is timed, the counters, the side runs, the parity legs.  Their names are bound HERE as well: main() and
# side_runs() call them through this module (tests/bench_standin_main.py replaces some of them by name)
from benchlib.common import (HBM_PEAK_GBS, DEV, SOLVE_FIELDS, Ctx, algorithmic_bytes, effective_cores, same_solve, same_as_ref_run,  # noqa: E402,F401
                             bit_check)
from benchlib.stream import Stream, shard_schedule  # noqa: E402,F401
from benchlib.counters import live_pmc, hbm_traffic  # noqa: E402,F401
from benchlib.side import (side_isolated, side_batch, side_single, side_reference_order_batch, side_reference_order_other_configs,  # noqa: E402,F401
                           side_neighbours, side_device_order)
from benchlib.parity import (cpu_baseline, parity_device_order, reference_order_batch, paired, parity_reference_order, parity_bias,  # noqa: E402,F401
                             restart_stats, parity_literal, parity_lockstep, with_upload)


def parse_args(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    # (defaults = the driver's settings: a run of 5 steps ends with one batch's tail in five and reads 15-20 % lower; the timed region of 20
    # steps is 2.3 s of a run that takes minutes for its side measurements)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--batch-per-gpu", type=int, default=4096)
    ap.add_argument("--depth", type=int, default=4,
                    help="overlap schedule: steps in flight = resident batches = HIP streams per GPU (a step launches one batch and "
                         "takes delivery of the one launched DEPTH - 1 steps earlier); 2 was the value line until round 4")
    ap.add_argument("--config", type=int, default=3, help="BASELINE config (1-based) used as the workload")
    ap.add_argument("--order", choices=["reference", "device"], default="reference",
                    help="floating-point order of the value line: reference (bit-equal to the CPU restatement of the reference; default) or "
                         "device (reassociated sums; statistically equal only)")
    ap.add_argument("--cpu-sample", type=int, default=-1,
                    help="trajectories timed on the host cores (-1 = 4 per core, 0 = skip)")
    ap.add_argument("--seed", type=int, default=20240)
    ap.add_argument("--no-extras", action="store_true", help="skip the batch-256 / single-trajectory side runs (profiling)")
    ap.add_argument("--schedule", choices=["overlap", "chain", "plain"], default="overlap",
                    help="overlap: --depth batches resident, each on its own HIP stream, every trajectory stays in its queue launch, the next "
                         "launches fill the slots an earlier one frees (default); chain: one stream, the stragglers of a batch are "
                         "adopted by the next batch's launch; plain: isolated solves (the tail of each batch runs on a nearly empty device)")
    ap.add_argument("--no-chain", action="store_true", help="same as --schedule plain")
    ap.add_argument("--scaling", choices=["weak", "strong"], default="weak",
                    help="weak: --batch-per-gpu trajectories on every GPU (the value line); strong: --batch-per-gpu trajectories in all, "
                         "sharded over the GPUs (BASELINE configs[3]: 4096 over 8 = 512 per GPU).  At N > 1 the other mode is timed as "
                         "well and reported beside the value line")
    return ap.parse_args(argv)


def init_job(args):
    """this rank's place in the job (RANK / LOCAL_RANK / WORLD_SIZE of torch.distributed.run), the device, the process group"""
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
    ctx = Ctx("plain" if args.no_chain else args.schedule, rank, world, local_rank, distributed, capi.default_params(),
              n_cu=torch.cuda.get_device_properties(local_rank).multi_processor_count)
    ctx.order = capi.ORDER_REFERENCE if args.order == "reference" else None   # what every Stream of the value line's order is created with
    return ctx


def run_strong_shard(ctx, args):
    """BASELINE configs[3] as written is 4096 trajectories over 8 GPUs = 512 per GPU: that shard on this GPU, so that the
    strong-scaling expectation is on record before the driver measures it -- two steps in flight (the value line's
    schedule: the device is a quarter full) and as many as hold the value line's 8192 trajectories (16 steps of 512)"""
    per = max(1, args.batch_per_gpu // 8)
    s_stream = Stream(ctx, per, args.config, args.seed + 2, order=ctx.order)
    sr = s_stream.run(max(args.steps, 8), max(args.warmup, 2))
    s_stream.close()
    d_s, r_s = shard_schedule(args, ctx.schedule, per)
    s_stream = Stream(ctx, per, args.config, args.seed + 2, depth=d_s, residency=r_s, order=ctx.order)
    sd = s_stream.run(max(args.steps, 4 * d_s), max(args.warmup, d_s))
    s_stream.close()
    return {"per_gpu": per, "solves_per_s": sd["value"], "ms_per_step": sd["ms_per_step"], "steps": sd["steps"],
            "steps_in_flight": d_s, "time_to_result_ms": sd["to_result_ms"],
            "two_steps_in_flight": {"solves_per_s": sr["value"], "ms_per_step": sr["ms_per_step"], "steps": sr["steps"],
                                    "time_to_result_ms": sr["to_result_ms"]},
            "of": "configs[3]: %d trajectories over 8 GPUs" % args.batch_per_gpu}


SCHEDULE_NOTE = {
    "overlap": "overlap: --depth batches resident, each on its own HIP stream, launched in turn; every trajectory finishes in its "
               "queue launch, the next launches take the slots an earlier one frees; the timed region starts on an idle device and "
               "ends when the last batch is delivered",
    "chain": "chain: one stream, the stragglers of a batch finish inside the next batch's queue launch, the last "
             "batch is flushed inside the timed region",
    "plain": "plain: every batch finishes on its own"}


def value_line(ctx, args, st, res, B_total):
    """the JSON line of the contract from the timed run `res` of the main stream `st` (rank 0)"""
    shard, rs = st.shard, res["rs"]
    r, lay, world = rs[0], st.shard.layout, ctx.world
    value = res["value"]
    eb = [float(algorithmic_bytes(lay, shard.n_points, lay.H, lay.M, q["iters"], q["evals"], q["hist_sum"]).sum()) if q is not None else 0.0 for q in rs]
    ebytes_steps = sum(eb[(st.k - args.steps + j) % st.D] for j in range(args.steps))  # the batches the timed steps solved
    kms = res["gpu_ms"] / args.steps  # device time of the timed region (marker events on the library's streams) per step
    achieved = ebytes_steps / args.steps / (kms * 1e-3) / 1e9
    traffic, traffic_raw, traffic_source, valu_per_solve = hbm_traffic(ctx, args, shard.B)
    ref = ctx.order is not None
    out = {
        "order": ("reference: every sum in the order the reference executes it; whole solves bit-equal to the CPU restatement "
                  "(parity.reference_order)" if ref else "device: reassociated sums, bit-equal to its own CPU replay only (parity.bias: the distance to the reference)"),
        "metric": "trajectory solves/sec (batched), 16-piece MINCO",
        "value": value, "unit": "solves/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": res["ms_per_step"], "higher_is_better": True, "scaling": args.scaling, "vs_baseline": None,
        "dtype": "f64", "data": "synthetic",
        "config": {"workload": "BASELINE configs[2]/[3] problem (%s): %d trajectories%s x %d pieces x %d pts/piece, "
                               "50 static obstacles, H=4 rectangle corridor per trajectory, fp64" %
                               (shard.name, args.batch_per_gpu, "/GPU" if args.scaling == "weak" else " in all", lay.n_pieces, shard.K + 1),
                   "global_batch": B_total, "pieces": lay.n_pieces, "pts_per_piece": shard.K + 1,
                   "n_vars": lay.n_vars, "parallelism": "batch-sharded x%d, 1 all-gather of 16B records" % world,
                   "steps_in_flight": st.D, "allgather_via": st.via},
        "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                     "frac": achieved / HBM_PEAK_GBS, "traffic": traffic, "traffic_uncorrected": traffic_raw,
                     "traffic_source": traffic_source,
                     "kernel": ("ref4_kernel (reference order, QUAD shape: four trajectories per wave)" if ref else "solver_kernel"), "kernel_ms": kms,
                     "launches_per_step": 1 if (ctx.schedule == "overlap" or shard.B < 4 * ctx.n_cu) else 2,
                     "algorithmic_bytes_per_launch": ebytes_steps / args.steps},
        "schedule": SCHEDULE_NOTE[ctx.schedule],
        "p50_ms_per_solve": float(np.median(r["latency_us"])) * 1e-3,
        "p95_ms_per_solve": float(np.percentile(r["latency_us"], 95)) * 1e-3,
        # host clock, per batch of the timed steps.  time_to_result_ms: launch -> its records delivered; deliver_ms: from the
        # completion of a batch's solve to its records in the caller's hands (the records are written by the solve kernels'
        # epilogues; one DMA copy at N = 1, the all-gather at N > 1 -- with the C-ABI collective the figure includes the wait for
        # the solve, which is enqueued behind it on the same stream); wait_for_solve_ms: the host blocked on the solve
        "time_to_result_ms": res["to_result_ms"], "deliver_ms": res["deliver_ms"], "wait_for_solve_ms": res["wait_for_solve_ms"],
        "mean_iters": float(r["iters"].mean()), "mean_evals": float(r["evals"].mean()),
        "mean_hist_depth": float(r["hist_sum"].sum() / max(1, r["iters"].sum())),
        "success_rate": float(r["success"].mean()),
    }
    if valu_per_solve:
        # the roof that actually binds: VALU issue.  SQ_INSTS_VALU per solve (the --pmc pass named in traffic_source) x solves/s
        # against one wave64 VALU instruction per SIMD every 4 cycles (fp64 FMA / add / mul issue at that rate on CDNA4)
        clk = 2.4e9  # MI355X engine clock (MI355X_MICROARCH.md; the in-kernel phase timer measures 2.39-2.40 GHz under this load)
        peak_i = ctx.n_cu * 4 * clk / 4.0
        out["roofline"]["valu"] = {"instructions_per_solve": valu_per_solve, "achieved_instr_per_s": valu_per_solve * value / world,
                                   "peak_instr_per_s": peak_i, "frac": valu_per_solve * value / world / peak_i,
                                   "clock_hz": clk, "simds": ctx.n_cu * 4,
                                   "note": "wave64 VALU instructions issued per second over (SIMDs x clock / 4); the kernel is bound "
                                           "here and by dependent latency (the QUAD shape runs one wave per SIMD), not by bytes"}
    return out


def strong_shard_entry(strong_shard, value):
    strong_shard["eight_gpu_expectation_solves_per_s"] = 8 * strong_shard["solves_per_s"]
    strong_shard["ratio_to_one_gpu_value"] = 8 * strong_shard["solves_per_s"] / value
    strong_shard["note"] = ("8 x the 512-trajectory shard rate over this line's 4096-per-GPU rate: what strong scaling of configs[3] "
                            "can reach at best.  Two 512-trajectory steps in flight fill a quarter of the device's one-wave slots; "
                            "steps_in_flight of them hold what the value line holds, at a longer time to result")
    return strong_shard


# ---------------------------------------------------------------------------------------------------------------------------
# side runs of the N = 1 line (a failure in one of them costs its entries, not the line).  `po` / `pyref` are the checkers
# (oracle/pyoracle.py, oracle/pyref.py): never the thing measured.


def side_runs(ctx, args, st, out, cores, B_total, rs0=None):
    """the exact BASELINE configs[2] case (batch 256), configs[1] (one gear-shift trajectory), configs[4] (moving cars), the
    reference order on the other configurations, the neighbouring steps of the solve"""
    from oracle import pyoracle as po  # the checker, never the thing measured
    po.build()
    out["isolated"] = side_isolated(st, rs0)
    if ctx.order is not None:   # the device order on the same stream of cycles, at the value line's depth and at depth 2
        out["device_order"] = side_device_order(ctx, args, B_total)
    out["batch256"] = side_batch(ctx, args, po, cores, 3, 256, 3, 8)
    out["single"] = side_single(ctx, args, po, cores, 2, range(9))
    out["moving_obstacles_1024"] = side_batch(ctx, args, po, cores, 5, 1024, 1, 8)  # BASELINE configs[4]: 32 pieces x 65 pts, 4 moving cars
    out["moving_obstacles_1024"]["reference_order"] = side_reference_order_batch(ctx, args, po, cores, 5, 1024, "cfg5_b1024")
    out["gear_shift_4096_reference_order"] = side_reference_order_batch(ctx, args, po, cores, 2, 4096, "cfg2_b4096")
    out.setdefault("parity", {})["reference_order_other_configs"] = side_reference_order_other_configs(ctx, args, po, cores)
    side_neighbours(ctx, args, po, st, out)


# ---------------------------------------------------------------------------------------------------------------------------
# the reference's CPU path beside the value line, and the parity legs that use its solves


def cpu_baseline_and_parity(ctx, args, st, r, out, cpu, B_total):
    """r: the results of the value line's first batch (in the value line's order)"""
    from oracle import pyoracle as po
    from oracle import pyref
    po.build()
    torch.set_num_threads(1)
    cores, shard = cpu["effective"], st.shard
    sample = cpu_baseline(ctx, args, po, pyref, shard, cpu, out)
    out.setdefault("parity", {})
    # the device order's solves of the same batch: the value line's when it runs in that order, else the side run's
    r_dev = r if ctx.order is None else (out.get("device_order") or {}).pop("_results", None)
    if r_dev is not None:
        out["parity"].update(parity_device_order(ctx, po, shard, r_dev, cores))
    ref_gpu = parity_reference_order(ctx, args, po, st, r if ctx.order is not None else None, r_dev, sample, out, B_total)
    if r_dev is not None:
        out["parity"]["literal"] = parity_literal(ctx, po, shard, r_dev, ref_gpu, cores)
        out["parity"]["lockstep"] = parity_lockstep(ctx, po, shard)
    out["with_upload"] = with_upload(st)


def guarded(out, name, fn, *a):
    """a failure in a side run costs its entries, not the line"""
    try:
        fn(*a)
    except Exception as ex:  # noqa: BLE001
        import traceback
        out.setdefault("side_run_errors", {})[name] = "%s: %s | %s" % (type(ex).__name__, ex, traceback.format_exc(limit=3).replace("\n", " / "))


def other_scaling(ctx, args, out):
    """The other scaling mode beside the value line (weak <-> strong), AFTER the value line is complete and under a watchdog: a
    side run that hangs or fails on some rank costs its own entry, not the line."""
    import threading

    def bail():
        if ctx.rank == 0:
            out["other_scaling"] = {"error": "the side run did not finish within its time limit"}
            print(json.dumps(out), flush=True)
        os._exit(0)
    wd = threading.Timer(float(os.environ.get("DFTPAV_BENCH_SIDE_LIMIT_S", "300")), bail)
    wd.daemon = True
    wd.start()
    try:
        B_other = args.batch_per_gpu if args.scaling == "weak" else args.batch_per_gpu * ctx.world
        d_o, r_o = shard_schedule(args, ctx.schedule, B_other // ctx.world)
        d_o = min(d_o, 8)
        o_stream = Stream(ctx, B_other, args.config, args.seed + 1, depth=d_o, residency=r_o, order=ctx.order)
        o = o_stream.run(max(args.steps, 2 * d_o), max(args.warmup, d_o))
        o_stream.close()
        if ctx.rank == 0:
            out["other_scaling"] = {"scaling": "strong" if args.scaling == "weak" else "weak", "global_batch": B_other,
                                    "per_gpu": B_other // ctx.world, "value": o["value"], "ms_per_step": o["ms_per_step"], "unit": "solves/s",
                                    "steps_in_flight": d_o, "steps": o["steps"]}
    except Exception as ex:  # noqa: BLE001
        if ctx.rank == 0:
            out["other_scaling"] = {"error": "%s: %s" % (type(ex).__name__, ex)}
    wd.cancel()


def main(argv=None):
    args = parse_args(argv)
    ctx = init_job(args)
    world = ctx.world
    # the value line: weak = --batch-per-gpu on every GPU, strong = --batch-per-gpu in all (BASELINE configs[3] as written)
    B_total = args.batch_per_gpu * world if args.scaling == "weak" else args.batch_per_gpu
    d_main, r_main = shard_schedule(args, ctx.schedule, B_total // world)
    st = Stream(ctx, B_total, args.config, args.seed, depth=d_main, residency=r_main, order=ctx.order)
    res = st.run(args.steps, args.warmup)
    side = world == 1 and not args.no_extras
    strong_shard = run_strong_shard(ctx, args) if side and args.scaling == "weak" else None
    out = None
    if ctx.rank == 0:
        out = value_line(ctx, args, st, res, B_total)
        if strong_shard is not None:
            out["strong_shard"] = strong_shard_entry(strong_shard, res["value"])
        cpu = effective_cores()
        if side:
            guarded(out, "extras", side_runs, ctx, args, st, out, cpu["effective"], B_total, res["rs"][0])
        if world == 1 and args.cpu_sample != 0:
            guarded(out, "cpu_baseline_and_parity", cpu_baseline_and_parity, ctx, args, st, res["rs"][0], out, cpu, B_total)
    if world > 1 or (ctx.distributed and os.environ.get("DFTPAV_BENCH_FORCE_OTHER") == "1"):  # (forced: the one-GPU test of this code)
        other_scaling(ctx, args, out)
    if ctx.rank == 0:
        (out.get("device_order") or {}).pop("_results", None)
        print(json.dumps(out), flush=True)
    st.close()
    if ctx.distributed:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
