"""Developer script: rate of the device corridor generator beside the CPU oracle (run through gpurun)."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from dftpav_amd import capi, scenarios as sc
from oracle import pyoracle as po

s = sc.baseline_config(3, B=2048)  # 128 hypotheses x 528 constraint points
st = s.meta["states"].reshape(-1, 3)
c = (0.5 * (st[:, 0].min() + st[:, 0].max()), 0.5 * (st[:, 1].min() + st[:, 1].max()))
span = max(st[:, 0].max() - st[:, 0].min(), st[:, 1].max() - st[:, 1].min()) + 40.0
grid, origin = sc.occupancy_grid(s.meta["obstacles"], arena=span, centre=c)
h = capi.Handle(capi.default_params())
h.set_grid_map(grid, sc.MAP_RESL, origin)
H = h.corridor_rectangles(st)
t = []
for _ in range(5):
    t0 = time.perf_counter(); H = h.corridor_rectangles(st); t.append(time.perf_counter() - t0)
t0 = time.perf_counter(); Ho = po.corridor_rectangles(grid, sc.MAP_RESL, origin, st[:20000], order=1); tc = time.perf_counter() - t0
km = h.corridor_last_ms()
print("states", len(st), "map", grid.shape, "kernel %.2f ms -> %.1f M rectangles/s;" % (km, len(st) / km / 1e3),
      "GPU incl. PCIe both ways %.2f ms -> %.2f M rectangles/s" % (1e3 * min(t), len(st) / min(t) / 1e6),
      "| CPU oracle 1 thread %.1f k rectangles/s" % (20000 / tc / 1e3), "| bit-identical:", np.array_equal(H[:20000], Ho))
