"""TEST INFRASTRUCTURE -- stands where the device library stands so that the HOST LOGIC of bench.py can be executed without a GPU.

bench.py cannot run here (no GPU) and the driver launches its N > 1 form only at round end: rank arithmetic, the order of
collectives, the shape of the JSON line, the side runs' bookkeeping would meet their first execution there.  This module offers
the part of `dftpav_amd.capi`'s Python surface that bench.py uses -- Handle, Batch, the communicator -- with the ORACLE computing
the solves (order 1 for the device order, order 2 for the reference order), so that tests/bench_standin_main.py can run
bench.main() under torch.distributed.run with the gloo backend.  Nothing outside tests/ imports it; nothing it produces is a
measurement (its "kernel" times are host seconds of the oracle); bench.py itself has no switch that selects it -- the launcher
replaces names in the imported module.
"""
import ctypes
import hashlib
import os
import time

import numpy as np
import torch
import torch.distributed as dist

from dftpav_amd import capi as real
from dftpav_amd import distributed as dd
from oracle import pyoracle as po

ORDER_DEVICE, ORDER_REFERENCE = real.ORDER_DEVICE, real.ORDER_REFERENCE
DftpavError = real.DftpavError
default_params = real.default_params
comm_layout = real.comm_layout
_CACHE = {}
THREADS = 4
CALLS = []   # (what, detail): the order of the calls a run made, for the tests


def _key(scen, order):
    h = hashlib.sha1()
    for a in (scen.inner_pts, scen.init_Ts, scen.ini_states, scen.fin_states, scen.corridor, scen.layout.piece_nums, scen.layout.singuls):
        h.update(np.ascontiguousarray(a).tobytes())
    h.update(repr((order, scen.t_now, scen.help_eps, scen.surround is not None)).encode())
    return h.hexdigest()


class Handle:
    def __init__(self, params=None, device=0):
        self.params = params if params is not None else default_params()
        self.device, self.surround, self.grid = device, None, None
        self.marks, self.last_ms, self.comm = {}, 0.0, None

    def set_surround(self, surround_set):
        self.surround = surround_set

    def mark(self, slot=0):
        self.marks[slot] = time.perf_counter()

    def elapsed_since(self, other, other_slot=0, slot=1):
        return 1e3 * (self.marks[slot] - other.marks[other_slot])

    def set_grid_map(self, grid, resolution, origin):
        self.grid = (np.ascontiguousarray(grid, dtype=np.uint8), float(resolution), tuple(origin))

    def _timed(self, fn, *a, **k):
        t0 = time.perf_counter()
        r = fn(*a, **k)
        self.last_ms = 1e3 * (time.perf_counter() - t0)
        return r

    def corridor_rectangles(self, states):
        g, res, org = self.grid
        return self._timed(po.corridor_rectangles, g, res, org, states, order=1)

    def corridor_last_ms(self):
        return max(self.last_ms, 1e-3)

    def reeds_shepp_shots(self, from_, to, max_cur=1.0, checkl=0.2, max_samples=512, vertex_res=0.1, check_collision=False):
        g, res, org = self.grid if check_collision else (None, 0.3, (0.0, 0.0))
        return self._timed(po.reeds_shepp_shots, from_, to, max_cur=max_cur, checkl=checkl, max_samples=max_samples, grid=g,
                           resolution=res, origin=org, vertex_res=vertex_res, order=1)

    def sample_restarts(self, inner, durs, n_restarts, sigma=0.3, lo=0.8, hi=1.25, seed=0):
        return po.sample_restarts(inner, durs, n_restarts, sigma=sigma, lo=lo, hi=hi, seed=seed)

    def comm_create(self, nranks, rank, unique_id):
        if os.environ.get("STANDIN_COMM") == "fail_rank1" and rank == 1:   # one rank cannot set its communicator up
            raise DftpavError(real.E_COMM, "stand-in: no communicator on rank 1")
        self.comm = (nranks, rank)

    def comm_share(self, owner):
        assert owner.comm is not None
        self.comm = owner.comm

    def comm_destroy(self):
        self.comm = None

    def close(self):
        CALLS.append(("handle_close", id(self)))


class Batch:
    def __init__(self, handle, layout_spec, B, residency=-1):
        self.handle, self.layout, self.B, self.n = handle, layout_spec, B, layout_spec.n_vars
        self.order, self.scen, self.r, self.ms, self.closed = 1, None, None, 0.0, False

    def set_order(self, order):
        self.order = 2 if order == ORDER_REFERENCE else 1

    def upload(self, scen, with_corridor=True):
        assert scen.B == self.B and scen.layout.n_vars == self.n
        self.scen, self.r = scen, None

    def set_hand_over(self, hand_over):
        pass

    def solve_async(self):
        assert not self.closed and self.scen is not None
        k = _key(self.scen, self.order)
        t0 = time.perf_counter()
        if k not in _CACHE:
            _CACHE[k] = po.solve_batch(self.handle.params, self.scen, nthreads=THREADS, order=self.order)
        self.r = _CACHE[k]
        self.ms = max(1e3 * (time.perf_counter() - t0), 1e3 * float(np.max(self.r["seconds"])))
        CALLS.append(("solve", self.B))

    def solve_chained(self, prev=None):
        self.solve_async()

    def finish(self):
        pass

    def sync(self):
        pass

    def solve(self):
        self.solve_async()
        return self.results()

    def last_solve_ms(self):
        return self.ms

    def results(self):
        if self.r is None:
            raise DftpavError(real.E_INVALID, "nothing solved since the upload")     # as dftpav_batch_results
        r = self.r
        return dict(x=r["x"].copy(), final_cost=r["final_cost"].copy(), status=r["status"].copy(), success=r["success"].copy(),
                    iters=r["iters"].copy(), evals=r["evals"].copy(), hist_sum=r["hist_sum"].copy(), latency_us=1e6 * r["seconds"])

    def records(self):
        return dd.pack_records(self.r["final_cost"], self.r["status"], self.r["iters"])

    def pack_results(self, device_ptr):
        rec = self.records()
        ctypes.memmove(device_ptr, rec.ctypes.data, rec.nbytes)   # "device" memory is host memory here

    def coeffs(self):
        c = np.zeros((self.B, self.layout.n_pieces, 6, 2))
        dt = np.zeros((self.B, self.layout.M))
        for b in range(self.B):
            o = po.OracleProblem(self.handle.params, self.scen, b, order=self.order)
            o.eval(self.r["x"][b])
            c[b], dt[b] = o.coeffs()
        return c, dt

    def validate(self, sample_dt=0.05, vertex_res=0.1):
        g, res, org = self.handle.grid
        c, dt = self.coeffs()
        return self.handle._timed(po.validate_trajectories, g, res, org, c, dt, self.layout.piece_nums, self.layout.singuls,
                                  sample_dt=sample_dt, vertex_res=vertex_res, order=1)

    def sample_states(self, t0=0.0, sample_dt=0.01, n_samples=None, filter_singularity=True):
        c, dt = self.coeffs()
        return self.handle._timed(po.sample_states, c, dt, self.layout.piece_nums, self.layout.singuls, t0=t0, sample_dt=sample_dt,
                                  n_samples=n_samples, filter_singularity=filter_singularity, wheel_base=self.handle.params.veh_wheel_base, order=1)

    def trace(self, traj, max_evals=4096, count=1):
        raise DftpavError(real.E_UNSUPPORTED, "the stand-in keeps no evaluation trace")

    def close(self):
        self.closed = True


class Comm:
    """where dd.RcclComm stands: the same constructor arguments and collective behaviour (every rank constructs it or none), the
    all-gather itself over the job's gloo group"""

    def __init__(self, handle, group=None, share=None):
        self.handle = handle
        if share is not None:
            handle.comm_share(share.handle)
            self.world, self.rank = share.world, share.rank
            return
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        self.rank = dist.get_rank(group) if dist.is_initialized() else 0
        if self.world > 1:
            t = torch.zeros(129, dtype=torch.uint8)
            dist.broadcast(t, src=0, group=group)          # the id's journey
        handle.comm_create(self.world, self.rank, None)
        CALLS.append(("comm_create", self.world))

    def allgather(self, batch, B):
        _, count, _block = comm_layout(B, self.world, self.rank)
        assert count == batch.B and batch.handle.comm is not None
        CALLS.append(("allgather", B))
        return dd.allgather_records(torch.from_numpy(batch.records()), B)

    def close(self):
        self.handle.comm_destroy()
