"""bench.py, part: what every part of the bench shares: the roofline constants, the bytes model of SURVEY 8(d), the job context, the bit-equality helpers."""
import os

import numpy as np


ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


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


class Ctx:
    """what a Stream needs to know about the job: the schedule, this rank's place in it, the solver parameters"""

    def __init__(self, schedule, rank, world, local_rank, distributed, params, n_cu=256):
        self.schedule, self.rank, self.world, self.local_rank, self.distributed, self.params = schedule, rank, world, local_rank, distributed, params
        self.n_cu = n_cu   # compute units of this rank's device
        self.order = None  # floating-point order of the value line's streams (None: device order; capi.ORDER_REFERENCE)


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
