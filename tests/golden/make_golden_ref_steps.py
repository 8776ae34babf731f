"""Golden vectors of the steps either side of the solve path written by the REFERENCE'S OWN CODE --
oracle/_ref/libdftpav_ref_next.so: getRectangleConst, CheckReplan's re-check, GetState + the server's playback,
ConverSurroundTrajFromPoints, getKinoNode / RunMINCOParking's resampling, cut verbatim out of /root/reference by
oracle/ref_slices.py -- not by the restatements.  The inputs are those stored in tests/golden/steps.npz.  Runs only where
/root/reference exists (this container); the vectors travel as a fixture:  python tests/golden/make_golden_ref_steps.py
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from dftpav_amd import scenarios as sc  # noqa: E402
from dftpav_amd.pods import FrontendParams  # noqa: E402
from oracle import pyoracle as po  # noqa: E402


def run(Z, **how):
    """every step on the stored inputs; how = dict(ref=True) for the reference's code, dict(order=0) for the restatement"""
    grid, origin = Z["grid"], tuple(Z["origin"])
    rec = {"cor_out": po.corridor_rectangles(grid, sc.MAP_RESL, origin, Z["cor_states"], **how)}
    rec["val_col"], rec["val_first"] = po.validate_trajectories(grid, sc.MAP_RESL, origin, Z["traj_coeffs"], Z["traj_dt"], Z["traj_pn"],
                                                                Z["traj_sg"], sample_dt=0.05, vertex_res=0.1, **how)
    rec["rd_states"], rec["rd_valid"] = po.sample_states(Z["traj_coeffs"], Z["traj_dt"], Z["traj_pn"], Z["traj_sg"], t0=-0.1, sample_dt=0.03,
                                                         n_samples=220, filter_singularity=True, **how)
    ft = po.fit_surround(Z["fit_states"], **how)
    rec.update(fit_dur=ft["durations"], fit_coef=ft["coeffs"], fit_total=ft["total"], fit_start=ft["start"])
    fe = po.frontend_resample(Z["fe_paths"], Z["fe_len"], Z["fe_ss"], Z["fe_es"], Z["fe_ct"], FrontendParams.default(K=6, Kd=9), **how)
    for k, v in fe.items():
        rec["fe_out_" + k] = v
    return rec


def main():
    out_dir = os.path.dirname(os.path.abspath(__file__))
    Z = np.load(os.path.join(out_dir, "steps.npz"))
    rec = run(Z, ref=True)
    np.savez_compressed(os.path.join(out_dir, "ref_steps.npz"), **rec)
    print("ref_steps.npz:", {k: np.asarray(v).shape for k, v in rec.items()})


if __name__ == "__main__":
    main()
