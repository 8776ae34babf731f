#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out
cd $R
timeout 1200 python -m pytest tests/test_gpu_parity.py tests/test_gpu_reference_order.py -x -q -k "moving or cfg5 or 5 or surround or obstacle or live or wave_shape or configs4 or more_than" 2>&1 | tail -6 > $O/r04_t9.txt
{ timeout 600 python scripts/cfg5_time.py 1024; } > $O/r04_cfg5.txt 2>&1
{ timeout 600 python - <<'PY'
import sys, os, time
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import numpy as np
from dftpav_amd import capi, scenarios as sc
p = capi.default_params()
s = sc.baseline_config(5, B=256); s.apply_resolution(p)
h = capi.Handle(p); h.set_surround(s.surround)
bt = capi.Batch(h, s.layout, s.B); bt.upload(s)
bt.set_order(capi.ORDER_REFERENCE)
bt.solve_async(); bt.sync()
bt.solve_async(); bt.sync()
r = bt.results()
print("configs[4] reference order, batch 256: kernel ms", bt.last_solve_ms(), "max iters", r["iters"].max(), "us/iter of the longest", 1e3 * bt.last_solve_ms() / r["iters"].max())
PY
} >> $O/r04_cfg5.txt 2>&1
