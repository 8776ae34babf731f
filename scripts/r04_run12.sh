#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out
cd $R
timeout 1500 python -m pytest tests/test_gpu_reference_order.py tests/test_gpu_dropin.py -x -q 2>&1 | tail -5 > $O/r04_t11.txt
{ timeout 300 python - <<'PY'
import sys, os
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import numpy as np
from dftpav_amd import capi, scenarios as sc
for cfg, B in ((2, 1), (2, 64), (2, 4096)):
    ms, its = [], []
    for seed in range(3 if B == 1 else 1):
        p = capi.default_params()
        s = sc.baseline_config(cfg, B=B, seed=20240 + seed); s.apply_resolution(p)
        h = capi.Handle(p); bt = capi.Batch(h, s.layout, B); bt.upload(s)
        bt.set_order(capi.ORDER_REFERENCE)
        bt.solve_async(); bt.sync(); bt.solve_async(); bt.sync()
        r = bt.results(); ms.append(bt.last_solve_ms()); its.append(int(r["iters"].max()))
        bt.close(); h.close()
    print("configs[1] layout (8 + 8 pieces, n = 33), reference order, batch", B, ": kernel ms", np.round(ms, 2), "longest solve", its, "iterations ->",
          round(1e3 * sum(ms) / sum(its), 1), "us per iteration of the longest;", round(B * len(ms) / (sum(ms) * 1e-3), 1), "solves/s")
PY
} > $O/r04_cap40.txt 2>&1
