"""Developer script: every stage of a plan cycle on the device, timed (run through gpurun).

searched paths -> front-end resampling -> restarts -> corridor of every trajectory -> solve -> collision re-check,
for the largest group of hypotheses that share a layout (a batch needs one layout)."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from dftpav_amd import capi, scenarios as sc
from dftpav_amd.pods import FrontendParams, LayoutSpec
from dftpav_amd.scenarios import Scenario

n_hyp = int(sys.argv[1]) if len(sys.argv) > 1 else 512
n_restarts = int(sys.argv[2]) if len(sys.argv) > 2 else 32
K = Kd = 32
t0 = time.perf_counter()
P, pl, ss, es, ct = sc.searched_paths(n_hyp, seed=11, gears=(1,), seg_duration=16.0, max_path=512)
t_gen = time.perf_counter() - t0
p = capi.default_params(); p.traj_resolution, p.des_traj_resolution = K, Kd
h = capi.Handle(p)
fp = FrontendParams.default(K=K, Kd=Kd)
h.frontend_resample(P[:2], pl[:2], ss[:2], es[:2], ct[:2], fp)  # warm-up
t0 = time.perf_counter(); fe = h.frontend_resample(P, pl, ss, es, ct, fp); t_fe = time.perf_counter() - t0
pieces = fe["piece_nums"][:, 0]
vals, counts = np.unique(pieces, return_counts=True)
N = int(vals[np.argmax(counts)])
grp = np.nonzero(pieces == N)[0]
lay = LayoutSpec([N], [1], 4)
npts = lay.n_points(K, Kd)
B = len(grp) * n_restarts
inner = fe["inner_pts"][grp, 0, :N - 1].reshape(len(grp), -1)
durs = (fe["piece_dt"][grp, 0] * N)[:, None]
h.sample_restarts(inner[:1], durs[:1], 2)
t0 = time.perf_counter(); ri, rd = h.sample_restarts(inner, durs, n_restarts, seed=5); t_rs = time.perf_counter() - t0
rng = np.random.default_rng(0)
obs = np.column_stack([rng.uniform(-60, 60, 400), rng.uniform(-60, 60, 400), rng.uniform(0.5, 1.5, 400)])
d = np.hypot(obs[:, None, 0] - P[grp][:, ::8, 0].reshape(1, -1), obs[:, None, 1] - P[grp][:, ::8, 1].reshape(1, -1)).min(axis=1)
obs = obs[d > 4.0]
grid, origin = sc.occupancy_grid(obs, arena=160.0)
h.set_grid_map(grid, sc.MAP_RESL, origin)
states = np.ascontiguousarray(fe["states"][grp, 0, :npts])  # [hypotheses][npts][3]: restarts share their hypothesis' corridor
s = Scenario("pipe", lay, K, Kd, B, np.repeat(fe["ini_states"][grp, :1], n_restarts, axis=0).copy(),
             np.repeat(fe["fin_states"][grp, :1], n_restarts, axis=0).copy(), ri, rd, np.zeros((1, 1, 4, 4)))
bt = capi.Batch(h, lay, B)
t0 = time.perf_counter(); bt.upload(s, with_corridor=False); t_up = time.perf_counter() - t0
t0 = time.perf_counter(); bt.corridor_from_states(states, n_restarts); t_cor = time.perf_counter() - t0; k_cor = h.corridor_last_ms()
bt.solve_async(); bt.sync()
t0 = time.perf_counter(); bt.solve_async(); bt.sync(); t_solve = time.perf_counter() - t0
r = bt.results()
t0 = time.perf_counter(); col, first = bt.validate(); t_val = time.perf_counter() - t0; k_val = h.corridor_last_ms()
bt.sample_states(sample_dt=0.1, n_samples=8)
t0 = time.perf_counter(); st, nv = bt.sample_states(sample_dt=0.1, n_samples=int(rd.max() / 0.1) + 2); t_rd = time.perf_counter() - t0
k_rd = h.corridor_last_ms()
tot = t_fe + t_rs + t_up + t_cor + t_solve + t_val + t_rd
print("hypotheses %d (of %d searched, %d pieces), restarts %d -> B = %d trajectories, %d obstacles, map %s" %
      (len(grp), n_hyp, N, n_restarts, B, len(obs), grid.shape))
print("  resampling of %d paths        %8.2f ms" % (n_hyp, 1e3 * t_fe))
print("  restarts                        %8.2f ms" % (1e3 * t_rs))
print("  upload without corridor         %8.2f ms" % (1e3 * t_up))
print("  corridor of every hypothesis    %8.2f ms  (kernel %.2f ms, %.1f M rectangles/s, written to its %d restarts)" %
      (1e3 * t_cor, k_cor, len(grp) * npts / k_cor / 1e3, n_restarts))
print("  solve                           %8.2f ms  (%.0f solves/s, success %.3f, mean iterations %.0f)" %
      (1e3 * t_solve, B / t_solve, r["success"].mean(), r["iters"].mean()))
print("  collision re-check              %8.2f ms  (kernel %.2f ms), colliding %d" % (1e3 * t_val, k_val, int(col.sum())))
print("  read-out every 0.1 s            %8.2f ms  (kernel %.3f ms, %d states, %.0f MB to the host)" %
      (1e3 * t_rd, k_rd, int(nv.sum()), st.nbytes / 1e6))
print("  whole cycle                     %8.2f ms  -> %.0f planned trajectories/s" % (1e3 * tot, B / tot))
