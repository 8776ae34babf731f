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
# The HIP runtime multiplexes a process's streams onto 4 hardware queues by default: the fifth batch in flight waits for one of
# the first four to END, however empty the device is.  The side runs that keep 8..16 small batches in flight (a 512-trajectory
# shard of a strong-scaled 4096, 8 planner threads of 256) want as many queues as streams: 19 k -> 33 k solves/s on 512-
# trajectory steps.  Read once, when the runtime initialises; no effect on the value line (two streams).  INTEGRATION.md §5.
os.environ.setdefault("GPU_MAX_HW_QUEUES", "16")

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402
import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

from dftpav_amd import capi, distributed as dd, scenarios as sc  # noqa: E402

HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: HBM3E peak BW 8.0 TB/s
DEV = "cuda"            # where the record tensors and the flags of the collectives live


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


def live_pmc(args, schedule):
    """HBM traffic and VALU instruction count of the dominant kernel, collected NOW: separate `rocprofv3 --pmc <one counter>`
    passes (nothing else enabled: no trace domain, no --stats) of a short run of this same bench -- same batch, same seed, same
    schedule, 1 warm-up + 2 timed steps -- each its own process, as MI355X_MICROARCH.md's HBM section prescribes.  Returns
    (dict or None, note).  Per batch = summed over the kernel's dispatches (queue launch + straggler launch) / queue launches."""
    import csv
    import glob
    import shutil
    import subprocess
    import tempfile
    exe = shutil.which("rocprofv3") or ("/opt/rocm/bin/rocprofv3" if os.path.exists("/opt/rocm/bin/rocprofv3") else None)
    if exe is None:
        return None, "rocprofv3 not found"
    if any(k.startswith(("ROCPROF", "ROCP_", "ROCPROFILER")) for k in os.environ):
        return None, "this run is itself profiled: no nested collection"
    child = [sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "2", "--warmup", "1", "--no-extras", "--cpu-sample", "0",
             "--batch-per-gpu", str(args.batch_per_gpu), "--config", str(args.config), "--seed", str(args.seed), "--schedule", schedule]
    env = dict(os.environ, TMPDIR="/tmp")
    got, t0 = {}, time.perf_counter()
    for ctr in ("FETCH_SIZE", "WRITE_SIZE", "SQ_INSTS_VALU"):
        if time.perf_counter() - t0 > 150.0:
            return (got or None), "time limit reached after %s" % ", ".join(got)
        d = tempfile.mkdtemp(prefix="dftpav_pmc_", dir="/tmp")
        try:
            subprocess.run([exe, "--pmc", ctr, "--output-format", "csv", "-d", d, "--"] + child, cwd="/tmp", env=env,
                           stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL, timeout=120, check=True)
            rows = []
            for f in glob.glob(os.path.join(d, "**", "*_counter_collection.csv"), recursive=True):
                for r in csv.DictReader(open(f)):
                    if "solver_kernel" in r["Kernel_Name"] and r["Counter_Name"] == ctr:
                        rows.append((int(r["Grid_Size"]), float(r["Counter_Value"])))
            if not rows:
                return (got or None), "no %s rows for solver_kernel" % ctr
            gmax = max(g for g, _ in rows)
            got[ctr] = sum(v for _, v in rows) / sum(1 for g, _ in rows if g == gmax)
        except Exception as ex:  # noqa: BLE001
            return (got or None), "%s pass failed: %s" % (ctr, type(ex).__name__)
        finally:
            shutil.rmtree(d, ignore_errors=True)
    return got, "`rocprofv3 --pmc <counter> -- python bench.py %s`, one pass per counter, %.0f s" % (" ".join(child[2:]), time.perf_counter() - t0)


class Ctx:
    """what a Stream needs to know about the job: the schedule, this rank's place in it, the solver parameters"""

    def __init__(self, schedule, rank, world, local_rank, distributed, params, n_cu=256):
        self.schedule, self.rank, self.world, self.local_rank, self.distributed, self.params = schedule, rank, world, local_rank, distributed, params
        self.n_cu = n_cu   # compute units of this rank's device


class Stream:
    """A stream of planning cycles on this rank: two resident batches of different problems, solved alternately.  A step
    launches one batch and delivers the records of the batch that this completes (pack + all-gather); after the last step
    the outstanding batch is completed and delivered INSIDE the timed region, so K steps deliver K batches.
      overlap (default): each batch on its own handle = HIP stream, hand-over 0: every trajectory finishes in its queue
        launch, and while that launch thins out the other stream's launch takes the freed workgroup slots.
      chain: one stream; the last trajectories of a batch are adopted by the next batch's queue launch
        (dftpav_batch_solve_chained), the last batch is flushed in the latency shape.
      plain: isolated solves, a step waits for its own batch."""

    def __init__(self, ctx, B_total, config, seed, depth=2, residency=None, order=None):
        self.c = ctx
        # every rank generates its own shard from (seed, rank) — trajectories are independent, nothing is scattered
        # (SURVEY §8e); rank r owns global trajectories [r*B/G, (r+1)*B/G)
        # depth: resident batches = steps in flight (overlap schedule only; 2 for the value line).  A shard too small to
        # fill the device alone (512 of a strong-scaled 4096) is run deeper, in the throughput residency, so that a GPU
        # holds as many trajectories as it does at 4096 per step.
        self.D = D = depth if self.c.schedule == "overlap" else 2
        self.B_total = B_total
        self.lo, self.hi = dd.shard_range(B_total, self.c.rank, self.c.world)
        self.shards = [sc.baseline_config(config, B=self.hi - self.lo, seed=seed + 7919 * self.c.rank + 104729 * i) for i in range(D)]
        for sh in self.shards:
            sh.apply_resolution(self.c.params)
        self.shard = self.shards[0]
        h = capi.Handle(self.c.params, device=self.c.local_rank)
        h.set_surround(self.shard.surround)
        self.hs = [h] * D
        if self.c.schedule == "overlap":
            self.hs = [h] + [capi.Handle(self.c.params, device=self.c.local_rank) for _ in range(D - 1)]
            for hh in self.hs[1:]:
                hh.set_surround(self.shard.surround)
        self.bts = []
        for hh, sh in zip(self.hs, self.shards):
            b_ = capi.Batch(hh, sh.layout, sh.B) if residency is None else capi.Batch(hh, sh.layout, sh.B, residency=residency)
            b_.upload(sh)  # resident in HBM from here on
            if order is not None:   # capi.ORDER_REFERENCE: the same stream of cycles in the reference's floating-point order
                b_.set_order(order)
            if self.c.schedule == "overlap":
                b_.set_hand_over(0)
            self.bts.append(b_)
        self.rec_dev = [torch.zeros((self.shard.B, dd.RECORD_BYTES), dtype=torch.uint8, device=DEV) for _ in range(D)]
        # the collective: RCCL behind the C-ABI (dftpav_comm_create / dftpav_batch_allgather_results, one communicator per
        # handle = per HIP stream), torch.distributed only carries the 128-byte id; DFTPAV_BENCH_COMM=torch (or a failure to
        # set the communicators up) takes torch.distributed's all_gather_into_tensor instead
        self.comms, self.via = None, "none (one rank)"
        if self.c.distributed:
            self.via = "torch.distributed all_gather_into_tensor (RCCL)"
            if os.environ.get("DFTPAV_BENCH_COMM", "capi") == "capi":
                try:
                    # ONE communicator per rank: the first handle owns it, the others borrow it (dftpav_comm_share) -- 16 steps in
                    # flight would otherwise mean 16 ncclCommInitRank rendezvous and 16 sets of RCCL buffers per rank
                    # (DFTPAV_BENCH_COMM_PER_HANDLE=1: a communicator per handle, as in round 3)
                    cm, owner = {}, None
                    for hh in self.hs:
                        if id(hh) not in cm:
                            if owner is None or os.environ.get("DFTPAV_BENCH_COMM_PER_HANDLE") == "1":
                                cm[id(hh)] = dd.RcclComm(hh)
                                owner = owner or cm[id(hh)]
                            else:
                                cm[id(hh)] = dd.RcclComm(hh, share=owner)
                    self.comms = [cm[id(hh)] for hh in self.hs]
                    self.via = "dftpav_batch_allgather_results (ncclAllGather behind the C-ABI, on the solve's stream)"
                except Exception as ex:  # noqa: BLE001  (RcclComm decides collectively: it raises on every rank or on none)
                    self.via += "; C-ABI communicator not set up: %s" % ex
                    self.comms = None
                # belt and braces: the path is the same on every rank or the job would hang in the first collective
                flag = torch.tensor([1 if self.comms is not None else 0], dtype=torch.int32, device=DEV)
                dist.all_reduce(flag, op=dist.ReduceOp.MIN)
                if int(flag.item()) == 0 and self.comms is not None:
                    for c_ in set(self.comms):
                        c_.close()
                    self.comms = None
                    self.via += "; C-ABI communicator not set up on another rank"
        self.k, self.out, self.rec = 0, [], None   # out: the batches in flight, oldest first
        self.t_launch = [0.0] * D
        self.to_result, self.in_deliver, self.wait_solve = [], [], []

    def deliver(self, i):
        t1 = time.perf_counter()
        if self.comms is not None:
            # (a failure here is fatal, not a reason to change path: the other ranks are inside the same ncclAllGather)
            self.rec = (self.comms[i].allgather(self.bts[i], self.B_total), i)
        t_mid = t1
        if self.comms is None:
            # the records are written by the solve kernels' epilogues: nothing of ours runs between the solve and their delivery
            self.bts[i].sync()
            t_mid = time.perf_counter()          # the batch's solve is complete here
            if self.c.distributed:
                self.bts[i].pack_results(self.rec_dev[i].data_ptr())
                self.bts[i].sync()
                self.rec = (dd.allgather_records(self.rec_dev[i], self.B_total), i)
            else:
                self.rec = (self.bts[i].records(), i)   # one DMA copy to the host
        t2 = time.perf_counter()
        self.in_deliver.append(t2 - t_mid)       # delivery proper (with the C-ABI collective: the wait for the solve included)
        self.wait_solve.append(t_mid - t1)
        self.to_result.append(t2 - self.t_launch[i])

    def step(self, last=False):
        i = self.k % self.D
        cur, prev = self.bts[i], (self.out[-1] if self.out else None)
        self.k += 1
        self.t_launch[i] = time.perf_counter()
        if self.c.schedule == "plain":
            cur.solve_async()
            self.deliver(i)
            return
        if self.c.schedule == "chain":
            cur.solve_chained(self.bts[prev] if prev is not None else None)  # prev is complete when this call's launches are
        else:
            # the earlier batches keep running on the other streams.  Nothing follows the last launch of a run, so it ends
            # with the default end game (its stragglers in the latency shape) instead of thinning out alone.
            cur.set_hand_over(-1 if last else 0)
            cur.solve_async()
        self.out.append(i)
        if len(self.out) >= self.D:
            self.deliver(self.out.pop(0))

    def flush(self):
        """the outstanding batches, oldest first: the stragglers in the latency shape (chain) / the rest of their launches
        (overlap)"""
        while self.out:
            if self.c.schedule == "chain" and len(self.out) == 1:
                self.bts[self.out[0]].finish()
            self.deliver(self.out.pop(0))

    def run(self, steps, warmup):
        for j in range(warmup):
            self.step(last=(j == warmup - 1))
        self.flush()  # the warm-up leaves nothing in flight: the timed region starts on an idle device
        self.to_result, self.in_deliver, self.wait_solve = [], [], []
        if self.c.distributed:
            dist.barrier()
        torch.cuda.synchronize()
        first = self.k % self.D
        self.hs[first].mark(0)  # HIP events on the library's own streams around the timed region
        t0 = time.perf_counter()
        for j in range(steps):
            self.step(last=(j == steps - 1))
        self.flush()
        last_h = self.hs[self.rec[1]]
        last_h.mark(1)
        if self.c.distributed:
            dist.barrier()
        torch.cuda.synchronize()
        elapsed = time.perf_counter() - t0
        gpu_ms = last_h.elapsed_since(self.hs[first], 0, 1)
        if self.c.distributed:
            tmax = torch.tensor([elapsed], dtype=torch.float64, device=DEV)
            dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
            elapsed = float(tmax.item())
        allrec, last = self.rec
        # a stream deeper than warmup + steps (a strong-scaled shard, 16 batches resident) leaves batches that never ran: None
        rs = [b_.results() if i_ < self.k else None for i_, b_ in enumerate(self.bts)]
        cost_all, status_all, iters_all = dd.unpack_records(allrec if isinstance(allrec, np.ndarray) else allrec.cpu().numpy())
        assert len(cost_all) == self.B_total and np.array_equal(cost_all[self.lo:self.hi], rs[last]["final_cost"])
        return dict(elapsed=elapsed, gpu_ms=gpu_ms, rs=rs, steps=steps, value=self.B_total * steps / elapsed,
                    ms_per_step=1e3 * elapsed / steps, to_result_ms=1e3 * float(np.mean(self.to_result)),
                    deliver_ms=1e3 * float(np.mean(self.in_deliver)), wait_for_solve_ms=1e3 * float(np.mean(self.wait_solve)))

    def close(self):
        cs = list(dict.fromkeys(self.comms or []))
        for c_ in reversed(cs):   # borrowers before the owner of the communicator
            c_.close()
        for b_ in self.bts:
            b_.close()
        for hh in set(self.hs):
            hh.close()



def parse_args(argv=None):
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
    return Ctx("plain" if args.no_chain else args.schedule, rank, world, local_rank, distributed, capi.default_params(),
               n_cu=torch.cuda.get_device_properties(local_rank).multi_processor_count)


def shard_schedule(args, schedule, per_gpu):
    """steps in flight and residency for a per-GPU shard: a shard that is a fraction of --batch-per-gpu runs as many steps
    deep as it takes to hold 2 x --batch-per-gpu trajectories per GPU (what the value line holds), at most 16, in the
    throughput residency (several workgroups per CU)"""
    if os.environ.get("DFTPAV_BENCH_DEPTH"):   # developer knob: "depth[,residency]"
        v = os.environ["DFTPAV_BENCH_DEPTH"].split(",")
        return int(v[0]), (int(v[1]) if len(v) > 1 else None)
    if per_gpu >= args.batch_per_gpu or schedule != "overlap":
        return 2, None
    return max(2, min(16, 2 * args.batch_per_gpu // max(1, per_gpu))), 2


def run_strong_shard(ctx, args):
    """BASELINE configs[3] as written is 4096 trajectories over 8 GPUs = 512 per GPU: that shard on this GPU, so that the
    strong-scaling expectation is on record before the driver measures it -- two steps in flight (the value line's
    schedule: the device is a quarter full) and as many as hold the value line's 8192 trajectories (16 steps of 512)"""
    per = max(1, args.batch_per_gpu // 8)
    s_stream = Stream(ctx, per, args.config, args.seed + 2)
    sr = s_stream.run(max(args.steps, 8), max(args.warmup, 2))
    s_stream.close()
    d_s, r_s = shard_schedule(args, ctx.schedule, per)
    s_stream = Stream(ctx, per, args.config, args.seed + 2, depth=d_s, residency=r_s)
    sd = s_stream.run(max(args.steps, 4 * d_s), max(args.warmup, d_s))
    s_stream.close()
    return {"per_gpu": per, "solves_per_s": sd["value"], "ms_per_step": sd["ms_per_step"], "steps": sd["steps"],
            "steps_in_flight": d_s, "time_to_result_ms": sd["to_result_ms"],
            "two_steps_in_flight": {"solves_per_s": sr["value"], "ms_per_step": sr["ms_per_step"], "steps": sr["steps"],
                                    "time_to_result_ms": sr["to_result_ms"]},
            "of": "configs[3]: %d trajectories over 8 GPUs" % args.batch_per_gpu}


def hbm_traffic(ctx, args, shard_B):
    """HBM bytes per launch and VALU instructions per solve of the dominant kernel: the counters of THIS tree on THIS box
    (live_pmc) when the line carries its side runs at N = 1, else the last collection committed (profiles/pmc_latest.json).
    -> (traffic, traffic_uncorrected, traffic_source, valu_per_solve)"""
    traffic, traffic_source, valu_per_solve, traffic_raw = None, None, None, None
    pmc = os.path.join(ROOT, "profiles", "pmc_latest.json")
    if os.path.exists(pmc):
        try:
            pj = json.load(open(pmc))
            traffic = pj.get("hbm_bytes_per_launch")
            valu_per_solve = pj.get("valu_instructions_per_solve")
            traffic_source = "profiles/pmc_latest.json: separate rocprofv3 --pmc passes of `%s` (%s)" % (
                pj.get("command", "bench.py --steps 3 --no-extras"), pj.get("collected", "round 1"))
        except Exception:
            traffic = None
    if ctx.world == 1 and not args.no_extras and os.environ.get("DFTPAV_BENCH_PMC", "1") != "0":
        try:
            lp, note = live_pmc(args, ctx.schedule)
        except Exception as ex:  # noqa: BLE001
            lp, note = None, "failed: %s" % type(ex).__name__
        if lp and "FETCH_SIZE" in lp and "WRITE_SIZE" in lp:
            # KB units; gfx950 counts a 128-byte read request as 64 bytes (MI355X_MICROARCH.md, HBM section): FETCH_SIZE x 2
            traffic = (2.0 * lp["FETCH_SIZE"] + lp["WRITE_SIZE"]) * 1024.0
            traffic_raw = (lp["FETCH_SIZE"] + lp["WRITE_SIZE"]) * 1024.0
            traffic_source = "live: " + note
        else:
            traffic_source = "%s [live collection: %s]" % (traffic_source, note)
        if lp and "SQ_INSTS_VALU" in lp:
            valu_per_solve = lp["SQ_INSTS_VALU"] / float(shard_B)
    return traffic, traffic_raw, traffic_source, valu_per_solve


SCHEDULE_NOTE = {
    "overlap": "overlap: two batches alternate on two HIP streams, every trajectory finishes in its queue launch, "
               "the next launch takes the slots the previous one frees; the last batch completes inside the timed region",
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
    out = {
        "metric": "trajectory solves/sec (batched), 16-piece MINCO",
        "value": value, "unit": "solves/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": res["ms_per_step"], "higher_is_better": True, "scaling": args.scaling, "vs_baseline": None,
        "dtype": "f64", "data": "synthetic",
        "config": {"workload": "BASELINE configs[2]/[3] problem (%s): %d trajectories%s x %d pieces x %d pts/piece, "
                               "50 static obstacles, H=4 rectangle corridor per trajectory, fp64 bit-exact mode" %
                               (shard.name, args.batch_per_gpu, "/GPU" if args.scaling == "weak" else " in all", lay.n_pieces, shard.K + 1),
                   "global_batch": B_total, "pieces": lay.n_pieces, "pts_per_piece": shard.K + 1,
                   "n_vars": lay.n_vars, "parallelism": "batch-sharded x%d, 1 all-gather of 16B records" % world,
                   "allgather_via": st.via},
        "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                     "frac": achieved / HBM_PEAK_GBS, "traffic": traffic, "traffic_uncorrected": traffic_raw,
                     "traffic_source": traffic_source,
                     "kernel": "solver_kernel", "kernel_ms": kms,
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
                                           "here and by dependent latency at two waves per SIMD, not by bytes"}
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

SOLVE_FIELDS = ("final_cost", "x", "iters", "evals", "status")


def same_solve(a, i, b, j):
    """trajectory i of result set a and j of b: final x, cost, status, iterations, evaluations, bit for bit"""
    return bool(a["final_cost"][i] == b["final_cost"][j] and np.array_equal(a["x"][i], b["x"][j]) and a["iters"][i] == b["iters"][j] and
                a["evals"][i] == b["evals"][j] and a["status"][i] == b["status"][j])


def same_as_ref_run(r, b, rr):
    """trajectory b of a device result set against one OptimizeTrajectory run of a reference build (oracle/pyref.py)"""
    return bool(rr["final_cost"] == r["final_cost"][b] and np.array_equal(rr["x"], r["x"][b]) and rr["iters"] == r["iters"][b] and
                rr["evals"] == r["evals"][b] and rr["status"] == r["status"][b])


def bit_check(po, cores, p2, s2, r2, pick):
    """sampled trajectories of a side run against the device-order oracle: every field bit for bit"""
    ro = po.solve_batch(p2, s2.subset(pick), nthreads=min(len(pick), cores), order=1)
    return bool(all(np.array_equal(ro[k_], r2[k_][pick]) for k_ in SOLVE_FIELDS))


def side_isolated(st):
    """the value line's batch as isolated solves (no chaining: its tail runs on a nearly empty device)"""
    bt, iso = st.bts[0], []
    bt.set_hand_over(-1)  # the plan's default end game (the overlap schedule runs with 0)
    for _ in range(3):
        bt.solve_async(); bt.sync(); iso.append(bt.last_solve_ms())
    return {"batch": int(st.shard.B), "kernel_ms": float(np.mean(iso)), "solves_per_s": st.shard.B / (float(np.mean(iso)) * 1e-3)}


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
                                "bit_equal_to_the_reference_build_on_this_host": (int(sum(eqb)) if eqb else None),
                                "bit_equal_to_the_reference_build_on_a_correctly_rounded_libm": (int(sum(eqc)) if eqc else None),
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
        gz = os.path.join(os.path.dirname(os.path.abspath(__file__)), "tests", "golden", "ref_order_batches.npz")
        Z = np.load(gz) if os.path.exists(gz) else None
        if Z is not None and golden + "_pick" in Z.files and int(Z["seed"]) == args.seed and int(Z[golden + "_pick"].max()) < B and golden.endswith("_b%d" % B):
            pk = Z[golden + "_pick"]
            row["bit_equal_to_the_reference_program_with_correctly_rounded_libm_calls_on_%d_sampled" % len(pk)] = bool(
                all(np.array_equal(Z[golden + "_" + k_], r5[k_][pk]) for k_ in SOLVE_FIELDS))
            row["of_them_solved_by_the_reference_build_on_a_correctly_rounded_libm"] = int(Z["n_checked_against_the_reference_objects_on_a_correctly_rounded_libm"])
        else:
            pick5 = np.array([0, B // 3, 2 * B // 3, B - 1])
            o5 = po.solve_batch(p5, s5.subset(pick5), nthreads=min(4, cores), order=2)
            row["bit_equal_to_the_reference_program_with_correctly_rounded_libm_calls_on_4_sampled"] = bool(
                all(np.array_equal(o5[k_], r5[k_][pick5]) for k_ in SOLVE_FIELDS))
        b5.close(); h5.close()
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
                row["against_reference_build"] = {"trajectories": int(sz.B), "bit_equal": eqb_,
                                                  "note": ("every solve must agree" if not libm else
                                                           "agrees where this host's libm rounded every call of the solve correctly")}
            if libm and _pr2.cr_available():
                row["against_reference_build_on_a_correctly_rounded_libm"] = {
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


def side_runs(ctx, args, st, out, cores):
    """the exact BASELINE configs[2] case (batch 256), configs[1] (one gear-shift trajectory), configs[4] (moving cars), the
    reference order on the other configurations, the neighbouring steps of the solve"""
    from oracle import pyoracle as po  # the checker, never the thing measured
    po.build()
    out["isolated"] = side_isolated(st)
    out["batch256"] = side_batch(ctx, args, po, cores, 3, 256, 3, 8)
    out["single"] = side_single(ctx, args, po, cores, 2, range(9))
    out["moving_obstacles_1024"] = side_batch(ctx, args, po, cores, 5, 1024, 1, 8)  # BASELINE configs[4]: 32 pieces x 65 pts, 4 moving cars
    out["moving_obstacles_1024"]["reference_order"] = side_reference_order_batch(ctx, args, po, cores, 5, 1024, "cfg5_b1024")
    out["gear_shift_4096_reference_order"] = side_reference_order_batch(ctx, args, po, cores, 2, 4096, "cfg2_b4096")
    out.setdefault("parity", {})["reference_order_other_configs"] = side_reference_order_other_configs(ctx, args, po, cores)
    side_neighbours(ctx, args, po, st, out)


# ---------------------------------------------------------------------------------------------------------------------------
# the reference's CPU path beside the value line, and the parity legs that use its solves

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
        # the same stream of planning cycles as the value line -- two resident batches on two HIP streams, one launched
        # while the other thins out -- in the REFERENCE'S order: the throughput of the bit-equal mode
        bR.close(); hR.close()
        try:
            stR = Stream(ctx, B_total, args.config, args.seed, depth=2, order=capi.ORDER_REFERENCE)
            k_ref = max(4, min(args.steps, 8))
            rR = stR.run(k_ref, 2)
            same = bool(np.array_equal(rR["rs"][0]["final_cost"], ref_gpu["final_cost"])) if stR.shards[0].B == shard.B else None
            ro["overlapped"] = {"solves_per_s": rR["value"], "ms_per_step": rR["ms_per_step"], "steps": k_ref, "warmup": 2,
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


def cpu_baseline_and_parity(ctx, args, st, r, out, cpu, B_total):
    from oracle import pyoracle as po
    from oracle import pyref
    po.build()
    torch.set_num_threads(1)
    cores, shard = cpu["effective"], st.shard
    sample = cpu_baseline(ctx, args, po, pyref, shard, cpu, out)
    out["parity"] = dict(out.get("parity", {}), **parity_device_order(ctx, po, shard, r, cores))
    ref_gpu = parity_reference_order(ctx, args, po, st, r, sample, out, B_total)
    out["parity"]["literal"] = parity_literal(ctx, po, shard, r, ref_gpu, cores)
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
        o_stream = Stream(ctx, B_other, args.config, args.seed + 1, depth=d_o, residency=r_o)
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
    st = Stream(ctx, B_total, args.config, args.seed, depth=d_main, residency=r_main)
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
            guarded(out, "extras", side_runs, ctx, args, st, out, cpu["effective"])
        if world == 1 and args.cpu_sample != 0:
            guarded(out, "cpu_baseline_and_parity", cpu_baseline_and_parity, ctx, args, st, res["rs"][0], out, cpu, B_total)
    if world > 1 or (ctx.distributed and os.environ.get("DFTPAV_BENCH_FORCE_OTHER") == "1"):  # (forced: the one-GPU test of this code)
        other_scaling(ctx, args, out)
    if ctx.rank == 0:
        print(json.dumps(out), flush=True)
    st.close()
    if ctx.distributed:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
