"""Developer script: the per-trajectory cycle counts (Prof) of a STREAM of batches in the reference order, against the same batch alone:
does a pass of the QUAD kernels take longer when four batches share the device?  CFG=2|3 scripts/stream_profile.py [depth]"""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("GPU_MAX_HW_QUEUES", "16")
import numpy as np
import torch
from dftpav_amd import capi
from benchlib.common import Ctx
from benchlib.stream import Stream

cfg = int(os.environ.get("CFG", 3))
depth = int(sys.argv[1]) if len(sys.argv) > 1 else 4
ctx = Ctx("overlap", 0, 1, 0, False, capi.default_params(), n_cu=torch.cuda.get_device_properties(0).multi_processor_count)
st = Stream(ctx, 4096, cfg, 20240, depth=depth, order=capi.ORDER_REFERENCE)
for b in st.bts:
    b.profile(True)
res = st.run(3 * depth, depth)
print("cfg", cfg, "depth", depth, "solves/s", round(res["value"]), "ms per step", round(res["ms_per_step"], 1))
for i, b in enumerate(st.bts):
    pr = b.read_profile().astype(np.float64)
    r = res["rs"][i]
    cyc = pr.copy()
    cyc[:, [9, 11]] = 0.0
    tot = cyc.sum(axis=1)
    per_eval = cyc[:, [0, 1, 2, 3, 4, 5, 10]].sum(axis=1) / r["evals"]
    per_iter = cyc[:, [6, 7, 8]].sum(axis=1) / np.maximum(r["iters"], 1)
    print(" batch", i, "cycles per trajectory mean %.1f M" % (tot.mean() / 1e6), "per eval %.0f k" % (per_eval.mean() / 1e3), "per iter %.0f k" % (per_iter.mean() / 1e3),
          "time in service ms mean/max %.0f / %.0f" % (r["latency_us"].mean() / 1e3, r["latency_us"].max() / 1e3), "evals", int(r["evals"].sum()))
st.close()
