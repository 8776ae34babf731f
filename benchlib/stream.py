"""bench.py, part: the timed object: a stream of planning cycles on this rank (--depth resident batches, each on its own HIP stream, launched in turn) and the shard schedules."""
import os
import time

import numpy as np
import torch
import torch.distributed as dist

from dftpav_amd import capi, distributed as dd, scenarios as sc

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
import benchlib.common as common  # (DEV is read through the module: the tests' stand-in moves it to the CPU)
from benchlib.common import (HBM_PEAK_GBS, SOLVE_FIELDS, Ctx, algorithmic_bytes, effective_cores, same_solve, same_as_ref_run,  # noqa: F401
                             bit_check)


class Stream:
    """A stream of planning cycles on this rank: `depth` resident batches of different problems, launched in turn.  A step
    launches one batch and delivers the records of the batch launched depth - 1 steps earlier (pack + all-gather); after the
    last step the outstanding batches are completed and delivered INSIDE the timed region, so K steps deliver K batches.
      overlap (default): each batch on its own handle = HIP stream, hand-over 0: every trajectory finishes in its queue
        launch, and while that launch thins out the other stream's launch takes the freed workgroup slots.
      chain: one stream; the last trajectories of a batch are adopted by the next batch's queue launch
        (dftpav_batch_solve_chained), the last batch is flushed in the latency shape.
      plain: isolated solves, a step waits for its own batch."""

    def __init__(self, ctx, B_total, config, seed, depth=2, residency=None, order=None):
        self.c = ctx
        # every rank generates its own shard from (seed, rank) — trajectories are independent, nothing is scattered
        # (SURVEY §8e); rank r owns global trajectories [r*B/G, (r+1)*B/G)
        # depth: resident batches = steps in flight (overlap schedule only; --depth for the value line).  A shard too small to
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
                b_.set_hand_over(int(os.environ.get("DFTPAV_STREAM_HAND_OVER", "0")))   # (developer knob; 0: other batches follow on other streams)
            self.bts.append(b_)
        self.rec_dev = [torch.zeros((self.shard.B, dd.RECORD_BYTES), dtype=torch.uint8, device=common.DEV) for _ in range(D)]
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
                flag = torch.tensor([1 if self.comms is not None else 0], dtype=torch.int32, device=common.DEV)
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
            tmax = torch.tensor([elapsed], dtype=torch.float64, device=common.DEV)
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


def shard_schedule(args, schedule, per_gpu):
    """steps in flight and residency for a per-GPU shard: --depth for a full shard; a shard that is a fraction of --batch-per-gpu runs
    as many steps deep as it takes to hold --depth x --batch-per-gpu trajectories per GPU (what the value line holds), at most 16, in the
    throughput residency (several workgroups per CU)"""
    if os.environ.get("DFTPAV_BENCH_DEPTH"):   # developer knob: "depth[,residency]"
        v = os.environ["DFTPAV_BENCH_DEPTH"].split(",")
        return int(v[0]), (int(v[1]) if len(v) > 1 else None)
    depth = max(2, getattr(args, "depth", 2))
    if per_gpu >= args.batch_per_gpu or schedule != "overlap":
        return depth, None
    return max(2, min(16, depth * args.batch_per_gpu // max(1, per_gpu))), 2
