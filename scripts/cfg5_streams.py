"""Developer script (GPU box): BASELINE configs[4] as a stream of batches -- 8 resident batches of B on 8 HIP streams in the
throughput residency (what bench.py's `moving_obstacles_1024.stream_of_batches` times), and one isolated batch.
DFTPAV_LIB selects the library (A/B of two builds on the same box)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from dftpav_amd import capi, scenarios as sc
B = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
STREAMS = len(sys.argv) <= 2 or sys.argv[2] != "isolated"      # "isolated": skip the 8-stream part
p = capi.default_params()
s = sc.baseline_config(5, B=B); s.apply_resolution(p)
hx = [capi.Handle(p) for _ in range(8 if STREAMS else 0)]
bx = []
for hh in hx:
    hh.set_surround(s.surround)
    bb = capi.Batch(hh, s.layout, B, residency=2); bb.upload(s); bx.append(bb)
for bb in bx: bb.solve_async()
for bb in bx: bb.sync()
out = []
for rep in range(2 if STREAMS else 0):
    t1 = time.perf_counter()
    for _ in range(2):
        for bb in bx: bb.solve_async()
    for bb in bx: bb.sync()
    out.append(16 * B / (time.perf_counter() - t1))
h = capi.Handle(p); h.set_surround(s.surround)
b1 = capi.Batch(h, s.layout, B); b1.upload(s); b1.solve_async(); b1.sync()
ms = []
for _ in range(2):
    b1.solve_async(); b1.sync(); ms.append(b1.last_solve_ms())
print("lib %s: 8 streams x %d: %s solves/s; isolated %s ms; same results: %s" % (os.path.basename(capi.LIB_PATH), B, np.round(out, 0), np.round(ms, 1),
      np.array_equal(bx[0].results()["x"], b1.results()["x"]) if bx else None))
