"""Developer script: GPU-vs-oracle parity + timing on the BASELINE configs (run through gpurun)."""
import sys, os, time, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from dftpav_amd import capi, scenarios as sc
from oracle import pyoracle as po

def run(cfg, B, nthreads=8, do_solve=True):
    p = capi.default_params()
    s = sc.baseline_config(cfg, B=B)
    s.apply_resolution(p)
    h = capi.Handle(p)
    if s.surround is not None:
        h.set_surround(s.surround)
    bt = capi.Batch(h, s.layout, B)
    bt.upload(s)
    x0 = bt.x0()
    rng = np.random.default_rng(1)
    out = {"cfg": cfg, "B": B}
    for tag, x in (("x0", x0), ("pert", x0 + rng.normal(0, 0.05, x0.shape))):
        f, g = bt.eval(x)
        fe = np.zeros(B); ge = np.zeros_like(g)
        for b in range(B):
            pr = po.OracleProblem(p, s, b, order=1)
            if tag == "x0":
                assert np.array_equal(pr.x0(), x0[b]), "x0 packing differs"
            fe[b], ge[b] = pr.eval(x[b])
        rf = np.abs(f - fe) / np.maximum(1.0, np.abs(fe))
        rg = np.abs(g - ge).max(axis=1) / np.maximum(1.0, np.abs(ge).max(axis=1))
        out["eval_" + tag] = dict(max_rel_f=float(rf.max()), max_rel_g=float(rg.max()),
                                  bit_f=float((f == fe).mean()), bit_g=float((g == ge).all(axis=1).mean()))
    if do_solve:
        t0 = time.time(); ro = po.solve_batch(p, s, nthreads=nthreads, order=1); t_cpu = time.time() - t0
        bt.solve_async(); bt.sync()  # warm
        t0 = time.time(); bt.solve_async(); bt.sync(); t_gpu = time.time() - t0
        ms = bt.last_solve_ms()
        r = bt.results()
        rel = np.abs(r["final_cost"] - ro["final_cost"]) / np.maximum(1.0, np.abs(ro["final_cost"]))
        out["solve"] = dict(
            bit_cost=float((r["final_cost"] == ro["final_cost"]).mean()), bit_x=float((r["x"] == ro["x"]).all(axis=1).mean()),
            same_evals=float((r["evals"] == ro["evals"]).mean()), same_hist=float((r["hist_sum"] == ro["hist_sum"]).mean()),
            max_rel_cost=float(rel.max()), med_rel_cost=float(np.median(rel)), frac_1e5=float((rel <= 1e-5).mean()),
            same_iters=float((r["iters"] == ro["iters"]).mean()), same_status=float((r["status"] == ro["status"]).mean()),
            iters_med=int(np.median(r["iters"])), iters_max=int(r["iters"].max()), evals_med=int(np.median(r["evals"])),
            gpu_ms=ms, gpu_wall_s=t_gpu, cpu_wall_s=t_cpu, cpu_threads=nthreads,
            gpu_solves_per_s=B / (ms * 1e-3), cpu_solves_per_s=B / t_cpu,
            max_x_diff=float(np.abs(r["x"] - ro["x"]).max()))
    bt.close(); h.close()
    return out

if __name__ == "__main__":
    which = sys.argv[1:] or ["1:4", "2:4", "3:64", "5:4"]
    res = []
    for w in which:
        cfg, B = w.split(":")
        o = run(int(cfg), int(B))
        print(json.dumps(o), flush=True)
        res.append(o)
    os.makedirs("gpurun_out", exist_ok=True)
    json.dump(res, open("gpurun_out/dev_parity.json", "w"), indent=1)
