#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out
cd $R
for v in "" narrow48; do
if [ -n "$v" ]; then export DFTPAV_LIB=$R/dftpav_amd/variants/libdftpav_hip_$v.so; else unset DFTPAV_LIB; fi
echo "=== variant '$v'"
timeout 300 python - <<'PY'
import sys, os
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import numpy as np
from dftpav_amd import capi, scenarios as sc
from oracle import pyoracle as po
p = capi.default_params()
s = sc.baseline_config(2, B=4096); s.apply_resolution(p)
h = capi.Handle(p); bt = capi.Batch(h, s.layout, s.B); bt.upload(s)
bt.set_order(capi.ORDER_REFERENCE)
bt.solve_async(); bt.sync()
ms = []
for _ in range(2):
    bt.solve_async(); bt.sync(); ms.append(bt.last_solve_ms())
r = bt.results()
pick = np.array([0, 1000, 4095])
o2 = po.solve_batch(p, s.subset(pick), nthreads=2, order=2)
print("gear-shift batch 4096 (n = 33), reference order: kernel ms", np.round(ms, 1), "->", round(4096 / (min(ms) * 1e-3)), "solves/s; bit-equal on 3 sampled:",
      all(np.array_equal(o2[k], r[k][pick]) for k in ("final_cost", "x", "iters", "evals")))
PY
done > $O/r04_narrow.txt 2>&1
