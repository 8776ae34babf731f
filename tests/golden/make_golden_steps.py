"""Freezes small inputs and bit-stable outputs of the oracles of the steps around the solve.  Corridor, validation, state read-out,
obstacle fit and front-end resampling: the restatement of the reference's functions with every libm call CORRECTLY ROUNDED (oracle
order 2: cos / sin / tan / atan / atan2 / x^3 from binary128, an implementation that shares nothing with the kernels' double-double
functions) -- what the HIP kernels must reproduce bit for bit, discrete outputs included.  Restart sampler and Reeds-Shepp shots
(no libm call of the reference to follow: OMPL's arithmetic is not in the reference tree): the portable functions, order 1.
The reference has no expected values for any of them.  Run from the repo root:
    python tests/golden/make_golden_steps.py
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from dftpav_amd import scenarios as sc  # noqa: E402
from dftpav_amd.pods import FrontendParams  # noqa: E402
from oracle import pyoracle as po  # noqa: E402


def main():
    out_dir = os.path.dirname(os.path.abspath(__file__))
    rng = np.random.default_rng(20240)
    rec = {}
    obs = np.column_stack([rng.uniform(-20, 20, 30), rng.uniform(-20, 20, 30), rng.uniform(0.5, 1.5, 30)])
    grid, origin = sc.occupancy_grid(obs, arena=60.0)
    st = np.column_stack([rng.uniform(-22, 22, 60), rng.uniform(-22, 22, 60), rng.uniform(-7, 7, 60)])
    rec.update(grid=grid, origin=np.array(origin), cor_states=st,
               cor_out=po.corridor_rectangles(grid, sc.MAP_RESL, origin, st, order=2))
    # trajectories for validation / read-out: random quintic pieces (two segments, the second reversing)
    B, pn, sg = 3, np.array([3, 2], dtype=np.int32), np.array([1, -1], dtype=np.int32)
    co = rng.normal(0, 1, size=(B, 5, 6, 2)) * np.array([8.0, 2.0, 0.6, 0.2, 0.05, 0.01])[None, None, :, None]
    dts = rng.uniform(0.6, 1.4, size=(B, 2))
    col, first = po.validate_trajectories(grid, sc.MAP_RESL, origin, co, dts, pn, sg, sample_dt=0.05, vertex_res=0.1, order=2)
    sts, nv = po.sample_states(co, dts, pn, sg, t0=-0.1, sample_dt=0.03, n_samples=220, filter_singularity=True, order=2)
    rec.update(traj_coeffs=co, traj_dt=dts, traj_pn=pn, traj_sg=sg, val_col=col, val_first=first, rd_states=sts, rd_valid=nv)
    # obstacle fit
    ps = np.zeros((2, 9, 7))
    ps[..., 0:2] = np.cumsum(rng.normal(0, 1.5, size=(2, 9, 2)), axis=1)
    ps[..., 2] = rng.uniform(-3, 3, size=(2, 9)); ps[..., 3] = rng.uniform(0, 6, size=(2, 9))
    ps[..., 4] = rng.normal(0, 1, size=(2, 9)); ps[..., 5] = rng.normal(0, 0.1, size=(2, 9))
    ps[..., 6] = 0.5 + 0.8 * np.arange(9)[None, :]
    ft = po.fit_surround(ps, order=2)
    rec.update(fit_states=ps, fit_dur=ft["durations"], fit_coef=ft["coeffs"], fit_total=ft["total"], fit_start=ft["start"])
    # front end
    P, pl, ss, es, ct = sc.searched_paths(3, seed=9, gears=(1, -1), seg_duration=5.0)
    fp = FrontendParams.default(K=6, Kd=9)
    fe = po.frontend_resample(P, pl, ss, es, ct, fp, order=2)
    rec.update(fe_paths=P, fe_len=pl, fe_ss=ss, fe_es=es, fe_ct=ct)
    for k, v in fe.items():
        rec["fe_out_" + k] = v
    # restarts
    inner, durs = rng.normal(size=(2, 6)), rng.uniform(2, 8, size=(2, 2))
    ri, rd = po.sample_restarts(inner, durs, 5, sigma=0.3, lo=0.8, hi=1.25, seed=77)
    rec.update(rs_inner=inner, rs_durs=durs, rs_out_inner=ri, rs_out_durs=rd)
    # Reeds-Shepp shots
    f = np.column_stack([rng.uniform(-15, 15, 40), rng.uniform(-15, 15, 40), rng.uniform(-4, 4, 40)])
    t = np.column_stack([rng.uniform(-15, 15, 40), rng.uniform(-15, 15, 40), rng.uniform(-4, 4, 40)])
    sh = po.reeds_shepp_shots(f, t, max_cur=0.8, checkl=0.25, max_samples=96, grid=grid, resolution=sc.MAP_RESL, origin=origin, order=1)
    rec.update(shot_from=f, shot_to=t)
    for k, v in sh.items():
        rec["shot_out_" + k] = v
    np.savez_compressed(os.path.join(out_dir, "steps.npz"), **rec)
    print("steps.npz:", {k: np.asarray(v).shape for k, v in rec.items() if k.startswith(("cor_out", "val_", "rd_valid", "shot_out_type"))})


if __name__ == "__main__":
    main()
