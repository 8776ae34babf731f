"""Developer script: Reeds-Shepp shots on the device, timed, against the CPU oracle (run through gpurun).
  python scripts/shot_time.py [n]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from dftpav_amd import capi, scenarios as sc
from oracle import pyoracle as po

n = int(sys.argv[1]) if len(sys.argv) > 1 else 65536
rng = np.random.default_rng(1)
f = np.column_stack([rng.uniform(-40, 40, n), rng.uniform(-40, 40, n), rng.uniform(-np.pi, np.pi, n)])
t = np.column_stack([rng.uniform(-40, 40, n), rng.uniform(-40, 40, n), rng.uniform(-np.pi, np.pi, n)])
obs = np.column_stack([rng.uniform(-40, 40, 80), rng.uniform(-40, 40, 80), rng.uniform(0.5, 2.0, 80)])
grid, origin = sc.occupancy_grid(obs, arena=100.0)
h = capi.Handle(capi.default_params())
h.set_grid_map(grid, sc.MAP_RESL, origin)
h.reeds_shepp_shots(f[:64], t[:64], max_samples=512, check_collision=True)
t0 = time.perf_counter(); r = h.reeds_shepp_shots(f, t, max_cur=1.0, checkl=0.2, max_samples=512, check_collision=True); wall = time.perf_counter() - t0
ms = h.corridor_last_ms()
m = min(n, 512)
t1 = time.perf_counter()
o = po.reeds_shepp_shots(f[:m], t[:m], max_cur=1.0, checkl=0.2, max_samples=512, grid=grid, resolution=sc.MAP_RESL, origin=origin, order=1)
cpu = (time.perf_counter() - t1) / m
same = all(np.array_equal(r[k][:m], o[k]) for k in o)
print("%d shots, %.1f M poses (mean length %.1f m), %.0f %% free: kernel %.2f ms -> %.2f M shots/s, %.0f M poses/s; with PCIe %.1f ms | "
      "CPU oracle %.3f ms per shot (1 thread) | bit-identical on %d: %s" %
      (n, r["n_samples"].sum() / 1e6, r["length"].mean(), 100 * (1 - r["collides"].mean()), ms, n / ms / 1e3, r["n_samples"].sum() / ms / 1e3,
       wall * 1e3, cpu * 1e3, m, same))
