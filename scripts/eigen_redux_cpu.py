"""Developer script (CPU, where /root/reference exists): how far is a build of the reference whose Eigen VECTORISES its reductions
from the build the oracle and the device's reference order are pinned to?

oracle/_ref/libdftpav_ref.so sums every reduction sequentially (oracle/ref_shim/Eigen/Eigen states the contract);
oracle/_ref/libdftpav_ref_eigen.so is the same sources with the reductions of dynamic vectors -- lbfgs.hpp's .dot() / .norm() /
.squaredNorm(), :300, 339, 551, 685-704, 725, 735 -- added in the order of Eigen 3.3's SSE2 reduction (two 2-lane packet
accumulators, then the lanes, then the odd term).  Both solve the bench's trajectories (BASELINE configs[2]/[3]'s layout); the
one-ulp control (x0 of the sequential build moved by one ulp in one coordinate) says what ANY change of the last bit does to this
solver.    python scripts/eigen_redux_cpu.py [n_trajectories] [processes]  ->  profiles/r05_eigen_redux_cpu.json"""
import json
import os
import sys
from multiprocessing import Pool
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np

N = int(sys.argv[1]) if len(sys.argv) > 1 else 2048
PROCS = int(sys.argv[2]) if len(sys.argv) > 2 else 8


def work(chunk):
    from dftpav_amd import scenarios as sc
    from oracle import pyoracle as po, pyref as pr
    p = po.default_params()
    s = sc.baseline_config(3, B=N, seed=20240)
    s.apply_resolution(p)
    out = []
    for b in chunk:
        a = pr.RefProblem(p, s, b).optimize()
        e = pr.RefProblem(p, s, b, eigen_redux=True).optimize()
        s2 = s.subset(np.array([b]))
        s2.inner_pts = s2.inner_pts.copy()
        s2.inner_pts[0].flat[0] = np.nextafter(s2.inner_pts[0].flat[0], np.inf)  # the control: one waypoint coordinate, one ulp
        c = pr.RefProblem(p, s2, 0).optimize()
        out.append((b, a["final_cost"], e["final_cost"], c["final_cost"], a["iters"], e["iters"], c["iters"], int(a["ok"]), int(e["ok"]),
                    bool(np.array_equal(a["x"], e["x"]) and a["final_cost"] == e["final_cost"])))
    return out


def stats(f0, f1):
    rel = np.abs(f1 - f0) / np.abs(f0)
    lr = np.log(f1 / f0)
    se = lr.std(ddof=1) / np.sqrt(len(lr))
    better = int((f1 < f0).sum())
    return {"frac_within_1e-5": float((rel < 1e-5).mean()), "abs_rel_p50": float(np.percentile(rel, 50)), "abs_rel_p95": float(np.percentile(rel, 95)),
            "log_ratio_mean": float(lr.mean()), "log_ratio_ci95": [float(lr.mean() - 1.96 * se), float(lr.mean() + 1.96 * se)],
            "lower_cost_in": better, "of": int(len(f0))}


if __name__ == "__main__":
    from oracle import pyref as pr
    pr.build()
    chunks = [list(range(i, N, PROCS * 4)) for i in range(PROCS * 4)]
    with Pool(PROCS) as pool:
        rows = sorted(r for part in pool.map(work, chunks) for r in part)
    A = np.array([[r[1], r[2], r[3], r[4], r[5], r[6], r[7], r[8], r[9]] for r in rows], dtype=np.float64)
    rec = {"workload": "BASELINE configs[2]/[3] layout (16 pieces x 33 points), %d trajectories, seed 20240" % N,
           "builds": {"sequential": "oracle/_ref/libdftpav_ref.so", "eigen_redux": "oracle/_ref/libdftpav_ref_eigen.so (DFTPAV_SHIM_EIGEN_REDUX=1)"},
           "bit_equal_solves": int(A[:, 8].sum()), "of": N,
           "eigen_redux_vs_sequential": stats(A[:, 0], A[:, 1]),
           "one_ulp_control_vs_sequential": stats(A[:, 0], A[:, 2]),
           "mean_iterations": {"sequential": float(A[:, 3].mean()), "eigen_redux": float(A[:, 4].mean()), "one_ulp_control": float(A[:, 5].mean())},
           "success": {"sequential": int(A[:, 6].sum()), "eigen_redux": int(A[:, 7].sum())},
           "median_cost": {"sequential": float(np.median(A[:, 0])), "eigen_redux": float(np.median(A[:, 1])), "one_ulp_control": float(np.median(A[:, 2]))}}
    out = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "profiles", "r05_eigen_redux_cpu.json")
    with open(out, "w") as f:
        json.dump(rec, f, indent=1)
    print(json.dumps(rec, indent=1))
