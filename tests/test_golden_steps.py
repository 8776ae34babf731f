"""The oracles of the steps around the solve against their frozen vectors (tests/golden/steps.npz, made by
tests/golden/make_golden_steps.py).  Corridor / validation / read-out / fit / front end: written at order 2 -- the reference's
functions with every libm call correctly rounded, from binary128; order 1 -- the same functions over the double-double cos / sin / tan
/ atan / atan2 / x^3 the HIP kernels call (cr_trig.h, compiled for the host) -- must reproduce them bit for bit: two implementations
of "correctly rounded" that share nothing.  Restarts and shots: the portable functions (order 1), bit-stable across hosts."""
import os

import numpy as np

from dftpav_amd import scenarios as sc
from dftpav_amd.pods import FrontendParams

Z = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "steps.npz"))


import pytest


@pytest.mark.parametrize("order", [1, 2])
def test_corridor_validation_readout(oracle, order):
    grid, origin = Z["grid"], tuple(Z["origin"])
    assert np.array_equal(oracle.corridor_rectangles(grid, sc.MAP_RESL, origin, Z["cor_states"], order=order), Z["cor_out"])
    col, first = oracle.validate_trajectories(grid, sc.MAP_RESL, origin, Z["traj_coeffs"], Z["traj_dt"], Z["traj_pn"], Z["traj_sg"],
                                              sample_dt=0.05, vertex_res=0.1, order=order)
    assert np.array_equal(col, Z["val_col"]) and np.array_equal(first, Z["val_first"])
    sts, nv = oracle.sample_states(Z["traj_coeffs"], Z["traj_dt"], Z["traj_pn"], Z["traj_sg"], t0=-0.1, sample_dt=0.03,
                                   n_samples=220, filter_singularity=True, order=order)
    assert np.array_equal(sts, Z["rd_states"]) and np.array_equal(nv, Z["rd_valid"])


def test_the_libm_order_is_within_rounding_of_the_correctly_rounded_one(oracle):
    """order 0 (this host's libm, as the reference calls it) against the vectors: the discrete outputs agree on these inputs, the
    continuous ones to the rounding of a libm call"""
    grid, origin = Z["grid"], tuple(Z["origin"])
    assert np.abs(oracle.corridor_rectangles(grid, sc.MAP_RESL, origin, Z["cor_states"], order=0) - Z["cor_out"]).max() < 1e-11
    col, first = oracle.validate_trajectories(grid, sc.MAP_RESL, origin, Z["traj_coeffs"], Z["traj_dt"], Z["traj_pn"], Z["traj_sg"],
                                              sample_dt=0.05, vertex_res=0.1, order=0)
    assert np.array_equal(col, Z["val_col"]) and np.array_equal(first, Z["val_first"])
    sts, nv = oracle.sample_states(Z["traj_coeffs"], Z["traj_dt"], Z["traj_pn"], Z["traj_sg"], t0=-0.1, sample_dt=0.03,
                                   n_samples=220, filter_singularity=True, order=0)
    assert np.array_equal(nv, Z["rd_valid"]) and np.abs(sts - Z["rd_states"]).max() < 1e-9


@pytest.mark.parametrize("order", [1, 2])
def test_fit_frontend_restarts_shots(oracle, order):
    ft = oracle.fit_surround(Z["fit_states"], order=order)
    assert np.array_equal(ft["durations"], Z["fit_dur"]) and np.array_equal(ft["coeffs"], Z["fit_coef"])
    assert np.array_equal(ft["total"], Z["fit_total"]) and np.array_equal(ft["start"], Z["fit_start"])
    fe = oracle.frontend_resample(Z["fe_paths"], Z["fe_len"], Z["fe_ss"], Z["fe_es"], Z["fe_ct"], FrontendParams.default(K=6, Kd=9),
                                  order=order)
    for k, v in fe.items():
        assert np.array_equal(v, Z["fe_out_" + k]), k
    ri, rd = oracle.sample_restarts(Z["rs_inner"], Z["rs_durs"], 5, sigma=0.3, lo=0.8, hi=1.25, seed=77)
    assert np.array_equal(ri, Z["rs_out_inner"]) and np.array_equal(rd, Z["rs_out_durs"])
    sh = oracle.reeds_shepp_shots(Z["shot_from"], Z["shot_to"], max_cur=0.8, checkl=0.25, max_samples=96, grid=Z["grid"],
                                  resolution=sc.MAP_RESL, origin=tuple(Z["origin"]), order=1)
    for k, v in sh.items():
        assert np.array_equal(v, Z["shot_out_" + k]), k
    assert len(set(Z["shot_out_type"].tolist())) >= 6
