"""Developer script (CPU): is the dense direction (oracle order 3) biased against the reference?  The solver is chaotic: any
change of rounding sends a solve down another path, so orders are compared statistically, against a control -- the literal
reference order (order 0) with one coordinate of x0 moved by one ulp.  Paired over the same trajectories of BASELINE configs[2]'s
problem: log ratio of the final costs (mean + bootstrap CI), sign test, iteration counts.
    python scripts/dense_bias_cpu.py [n_trajectories] > profiles/r04_dense_bias_cpu.json"""
import json
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from dftpav_amd import scenarios as sc  # noqa: E402
from oracle import pyoracle as po  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
po.build()
p = po.default_params()
s = sc.baseline_config(3, B=B, seed=20240)
s.apply_resolution(p)
T = min(8, os.cpu_count() or 1)
ref = po.solve_batch(p, s, nthreads=T, order=0)
s1 = s.subset(np.arange(B))
s1.inner_pts = np.ascontiguousarray(s1.inner_pts).copy()
s1.inner_pts[:, 0] = np.nextafter(s1.inner_pts[:, 0], np.inf)
runs = {"control_reference_with_x0_moved_one_ulp": po.solve_batch(p, s1, nthreads=T, order=0),
        "device_order_two_loop": po.solve_batch(p, s, nthreads=T, order=1),
        "device_order_dense_direction": po.solve_batch(p, s, nthreads=T, order=3)}


def paired(a, b, seed):
    from scipy import stats
    lr = np.log(a / b)
    rng = np.random.default_rng(seed)
    boot = np.array([lr[rng.integers(0, len(lr), len(lr))].mean() for _ in range(2000)])
    npos, nneg = int((lr > 0).sum()), int((lr < 0).sum())
    return {"log_ratio_mean": float(lr.mean()), "log_ratio_mean_ci95": [float(np.percentile(boot, 2.5)), float(np.percentile(boot, 97.5))],
            "ci_covers_0": bool(np.percentile(boot, 2.5) <= 0.0 <= np.percentile(boot, 97.5)),
            "n_higher": npos, "n_lower": nneg, "sign_test_p": float(stats.binomtest(npos, npos + nneg, 0.5).pvalue) if npos + nneg else 1.0,
            "abs_rel_diff_p50": float(np.median(np.abs(a - b) / b)), "frac_within_1e-5": float((np.abs(a - b) / b <= 1e-5).mean())}


out = {"trajectories": B, "problem": "BASELINE configs[2]/[3]: 16 pieces x 33 points, 50 obstacles", "against": "oracle order 0 = the reference's program (bit-equal to oracle/_ref)",
       "reference": {"mean_iters": float(ref["iters"].mean()), "mean_evals": float(ref["evals"].mean()), "success_rate": float(ref["success"].mean()),
                     "median_cost": float(np.median(ref["final_cost"]))}}
for i, (name, r) in enumerate(runs.items()):
    out[name] = dict(paired(r["final_cost"], ref["final_cost"], i + 1), mean_iters=float(r["iters"].mean()), mean_evals=float(r["evals"].mean()),
                     success_rate=float(r["success"].mean()), median_cost=float(np.median(r["final_cost"])),
                     thread_seconds=float(r["seconds"].sum()))
out["reference"]["thread_seconds"] = float(ref["seconds"].sum())
print(json.dumps(out, indent=1))
