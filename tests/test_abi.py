"""C-ABI surface: the library loads without a GPU, exports every symbol of
include/dftpav_hip.h, PODs have the documented layout, and there is no CPU fallback."""
import ctypes as C
import os
import re

import numpy as np
import pytest

from dftpav_amd import pods


def test_exports_every_declared_symbol(hiplib):
    hdr = open(os.path.join(os.path.dirname(__file__), "..", "include", "dftpav_hip.h")).read()
    declared = set(re.findall(r"\b(dftpav_[a-z0-9_]+)\s*\(", hdr))
    assert declared, "no declarations parsed"
    assert declared == set(hiplib.EXPORTS)
    L = hiplib.lib()
    for name in declared:
        assert getattr(L, name) is not None, name


def test_pod_sizes_match_compiled_layout(hiplib):
    L = hiplib.lib()
    assert L.dftpav_abi_sizeof_params() == C.sizeof(pods.Params)
    assert L.dftpav_abi_sizeof_layout() == C.sizeof(pods.Layout)
    assert L.dftpav_abi_sizeof_batch_data() == C.sizeof(pods.BatchData)
    assert L.dftpav_abi_sizeof_surround() == C.sizeof(pods.Surround)


def test_default_params_are_the_reference_values(hiplib, oracle):
    """Product and oracle restate config/minco_config.pb.txt:65-100 independently; they must agree."""
    a = hiplib.default_params().as_dict()
    b = oracle.default_params().as_dict()
    assert a == b
    assert a["traj_resolution"] == 16 and a["des_traj_resolution"] == 32
    assert a["wei_obs"] == 1000.0 and a["wei_surround"] == 5000.0 and a["wei_feas"] == 2500.0
    assert a["wei_time"] == 500.0 and a["lbfgs_mem_size"] == 256 and a["lbfgs_past"] == 3
    assert a["lbfgs_delta"] == 1e-4 and a["non_sinv"] == 0.24 and a["mini_T"] == 0.1


def test_num_vars_and_points(hiplib):
    p = hiplib.default_params()
    for pieces, n, npts in (([8], 15, 168), ([8, 8], 33, None), ([16], 31, None)):
        lay = pods.LayoutSpec(pieces, [1] * len(pieces))
        ls = lay.c_struct()
        assert hiplib.lib().dftpav_num_vars(C.byref(ls)) == n == lay.n_vars
        if npts is not None:  # 6*17 + 2*33
            assert hiplib.lib().dftpav_num_points(C.byref(p), C.byref(ls)) == npts


@pytest.mark.parametrize("N", [2, 3, 8, 16, 32])
def test_minco_operator_bits_equal_oracle(hiplib, oracle, N):
    """The kernels' dense MINCO operator is built with the reference's banded LU
    (poly_traj_utils.hpp:776-826); product and oracle must produce the same bits."""
    out = np.zeros((6 * N, N + 5))
    fn = hiplib.lib().dftpav_debug_minco_operator
    fn.argtypes = [C.c_int, pods.c_double_p]
    assert fn(N, pods.dptr(out)) == 0
    ref = oracle.minco_operator(N)
    assert np.array_equal(out, ref)


def test_host_tables_of_the_moving_obstacles(hiplib):
    """The two tables the host derives from the obstacles' pieces (capi.cpp): theta reproduces the piece index of the
    reference's walk t -= duration (poly_traj_utils.hpp:510-528) for every t, and the box of a piece contains the
    obstacle over the whole piece (so skipping a pair that is farther from the box than the gate never changes the gate)."""
    rng = np.random.default_rng(3)
    S = 5
    nps = rng.integers(1, 40, S)
    off = np.concatenate([[0], np.cumsum(nps)]).astype(np.int32)
    npz = int(off[-1])
    dur = rng.uniform(0.05, 3.0, npz)
    dur[: nps[0]] = 1.0                                    # one obstacle with equal pieces
    coef = rng.normal(0, 1, (npz, 6, 2)) * np.array([0.01, 0.05, 0.2, 1.0, 3.0, 20.0])[None, :, None]   # col 0 multiplies t^5
    theta = np.zeros(npz); box = np.zeros((npz, 4))
    fn = hiplib.lib().dftpav_debug_surround_tables
    fn.argtypes = [C.c_int, C.POINTER(C.c_int), pods.c_double_p, pods.c_double_p, pods.c_double_p, pods.c_double_p]
    assert fn(S, off.ctypes.data_as(C.POINTER(C.c_int)), pods.dptr(dur), pods.dptr(coef), pods.dptr(theta), pods.dptr(box)) == 1
    for u in range(S):
        d = dur[off[u]:off[u + 1]]; th = theta[off[u]:off[u + 1]]
        assert np.all(np.diff(th) >= 0)
        ts = np.concatenate([rng.uniform(0, d.sum() * 1.1, 400), th, np.nextafter(th, np.inf), np.nextafter(th, -np.inf)])
        for t in ts:
            tt, idx = t, 0
            while idx < len(d) and tt > d[idx]:             # the reference's walk
                tt -= d[idx]; idx += 1
            assert idx == int(np.sum(t > th))
    s = np.linspace(0.0, 1.0, 257)
    for k in range(npz):
        t = s * dur[k]
        pw = np.stack([t ** 5, t ** 4, t ** 3, t ** 2, t, np.ones_like(t)], axis=1)      # [257][6]
        pos = pw @ coef[k]                                                                # [257][2]
        assert pos[:, 0].min() >= box[k, 0] and pos[:, 0].max() <= box[k, 1]
        assert pos[:, 1].min() >= box[k, 2] and pos[:, 1].max() <= box[k, 3]
        # and it is a tight box, not a trivial one: within the hull of a quintic's control points
        assert box[k, 1] - box[k, 0] <= 3.0 * (pos[:, 0].max() - pos[:, 0].min()) + 1e-3 + 2.5 * np.abs(coef[k, :5, 0] * dur[k] ** np.arange(5, 0, -1)).sum()


def test_gate_with_tables_equals_the_walk_also_before_the_obstacle_starts(hiplib):
    """The distance gate of traj_optimizer.cpp:1393 as the kernels decide it with the host's tables (threshold search,
    piece boxes) against the reference's walk -- including obstacles whose trajectory starts after t_now: piece 0 is then
    extrapolated backwards (negative local time, traj_optimizer.cpp:1374-1378), outside its hull box.  Round-2 advisor
    repro: one piece x = 10 + 2 t, start 3, t_now 0, t = 1, ego at the origin -> the obstacle is at (6, 0), inside the gate."""
    fn = hiplib.lib().dftpav_debug_gate
    ip = C.POINTER(C.c_int)
    fn.argtypes = [C.c_int, ip, pods.c_double_p, pods.c_double_p, pods.c_double_p, pods.c_double_p, C.c_int, C.c_double, C.c_double,
                   C.c_double, C.c_int, pods.c_double_p, pods.c_double_p, ip, ip]

    def gate(off, dur, coef, total, start, u, t_now, trajtime, infl, sig, t):
        off = np.ascontiguousarray(off, dtype=np.int32); sig = np.ascontiguousarray(sig, dtype=np.float64)
        t = np.ascontiguousarray(t, dtype=np.float64)
        a = np.zeros(len(t), dtype=np.int32); w = np.zeros(len(t), dtype=np.int32)
        rc = fn(len(total), off.ctypes.data_as(ip), pods.dptr(dur), pods.dptr(coef), pods.dptr(total), pods.dptr(start), u, t_now, trajtime,
                infl, len(t), pods.dptr(sig), pods.dptr(t), a.ctypes.data_as(ip), w.ctypes.data_as(ip))
        assert rc == 1
        return a, w

    coef = np.zeros((1, 6, 2)); coef[0, 4, 0] = 2.0; coef[0, 5, 0] = 10.0        # column 0 multiplies t^5: x = 10 + 2 t, y = 0
    a, w = gate([0, 1], np.array([5.0]), coef, np.array([5.0]), np.array([3.0]), 0, 0.0, 0.0, 5.0, np.zeros((1, 2)), np.array([1.0]))
    assert w[0] == 1 and a[0] == 1
    rng = np.random.default_rng(8)
    S = 4
    nps = rng.integers(1, 12, S)
    off = np.concatenate([[0], np.cumsum(nps)])
    npz = int(off[-1])
    dur = rng.uniform(0.3, 2.0, npz)
    coef = rng.normal(0, 1, (npz, 6, 2)) * np.array([0.002, 0.01, 0.05, 0.3, 2.0, 8.0])[None, :, None]
    total = np.array([dur[off[u]:off[u + 1]].sum() for u in range(S)])
    start = np.array([6.0, 0.0, 2.5, -1.0])
    npts = 4000
    sig = rng.normal(0, 12, (npts, 2)); t = rng.uniform(0, 12, npts)
    seen_neg = 0
    for u in range(S):
        a, w = gate(off, dur, coef, total, start, u, 0.5, 1.0, 4.0, sig, t)
        assert np.array_equal(a, w), u
        neg = (0.5 - start[u] + 1.0) + t < 0
        seen_neg += int((w[neg] == 1).sum())
    assert seen_neg > 0      # pairs that pass the gate while the obstacle has not started yet do occur


def test_no_cpu_fallback(hiplib):
    """Without a usable HIP device the product must fail loudly, never compute on the CPU."""
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    with pytest.raises(hiplib.DftpavError) as e:
        hiplib.Handle(hiplib.default_params())
    assert e.value.code == hiplib.E_NO_DEVICE


def test_product_does_not_reference_oracle():
    """Only tests/, smoke() and bench.py's cpu_baseline may touch oracle/."""
    root = os.path.join(os.path.dirname(__file__), "..", "dftpav_amd")
    for dp, _, fns in os.walk(root):
        for fn in fns:
            if fn.endswith((".py", ".cpp", ".hip", ".h")):
                txt = open(os.path.join(dp, fn)).read()
                assert "pyoracle" not in txt and "dftpav_oracle.h" not in txt and "libdftpav_oracle" not in txt, fn


def test_launch_shapes_of_the_reference_order(hiplib):
    """Host logic of dftpav_batch_set_order(DFTPAV_ORDER_REFERENCE), without a device (solver_ref.hip: reference_order_supported /
    reference_order_plan through a test hook): what is supported, and which launch shape a batch takes on a 256-CU device --
    a workgroup per trajectory up to five trajectories per CU, beyond that one wave per trajectory, as many waves per workgroup
    as keep the most trajectories resident, the shared tables plus the waves' own state inside one CU's 160 KB of LDS."""
    import ctypes as C
    from dftpav_amd import scenarios as sc
    fn = hiplib.lib().dftpav_debug_reference_plan
    fn.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.POINTER(C.c_longlong)]

    def plan(pieces, singuls, K, Kd, S, B, H=4):
        from dftpav_amd.pods import LayoutSpec
        p = hiplib.default_params()
        p.traj_resolution, p.des_traj_resolution = K, Kd
        lay = LayoutSpec(pieces, singuls, H=H)
        out = (C.c_longlong * 8)()
        assert fn(C.byref(lay.c_struct()), C.byref(p), S, B, 256, out) == 0
        return dict(zip(("supported", "wave", "threads", "wg_per_cu", "slots", "slice", "lds", "cap"), [int(v) for v in out]))

    # BASELINE configs[2] / [3]: 16 pieces x 33 points, n = 31
    # (beyond five trajectories per CU this layout -- one gear segment, 16 pieces, n = 31, no moving obstacles -- takes the QUAD shape
    # of solver_ref4.hip: "wave" = 3, four trajectories per wave, workgroups of one wave, four per CU at one wave per SIMD; a launch
    # takes half as many rows as the batch has trajectories, at most the device's 1024 waves)
    for B, wave in ((1, 0), (256, 0), (1024, 0), (1280, 0), (1281, 3), (2048, 3), (4096, 3), (16384, 3)):
        q = plan([16], [1], 32, 32, 0, B)
        assert q["supported"] == 1 and q["wave"] == wave and q["cap"] == 32, (B, q)
        assert q["lds"] <= 160 * 1024
        if wave:
            assert q["threads"] == 64 and q["wg_per_cu"] == 4 and q["slots"] == min(1024, (B + 7) // 8) and q["slice"] == 64
            assert q["wg_per_cu"] * q["lds"] <= 160 * 1024
        else:
            assert q["threads"] == (128 if B > 768 else 256)
    # 8 + 9 pieces in ONE segment do not fit a row of 16 lanes: the WAVE shape, as before
    q = plan([17], [1], 16, 16, 0, 4096)
    assert q["supported"] == 1 and q["wave"] == 1 and q["cap"] == 40
    # configs[1]: 8 + 8 pieces with a gear shift, n = 33: sums of 40 terms (the WAVE shape's kernel, built for 256 registers since round 5)
    # -- in the WAVE shape, which only the hand-over of a batch's last trajectories still uses: a full batch takes the QUAD shape
    # for several gear segments (solver_ref4m.hip, "wave" = 5): the two segments' pieces side by side on a row's sixteen lanes, the two
    # segments of eight pieces sharing one copy of the sweep tables, four waves per CU
    q = plan([8, 8], [1, -1], 32, 32, 0, 4096)
    assert q["supported"] == 1 and q["cap"] == 40 and q["wave"] == 5 and q["threads"] == 64 and q["wg_per_cu"] == 4 and q["slots"] == 512
    assert q["wg_per_cu"] * q["lds"] <= 160 * 1024 and q["slice"] == 64
    q = plan([8, 8], [1, -1], 32, 32, 0, 1024)        # up to five per CU: the TEAM shape, as before
    assert q["supported"] == 1 and q["wave"] == 0
    q = plan([5, 4, 6], [1, -1, 1], 16, 16, 0, 4096)  # three segments of unequal length: three tables, workgroups of two waves
    assert q["supported"] == 1 and q["wave"] == 5 and q["threads"] * q["wg_per_cu"] == 256 and q["wg_per_cu"] * q["lds"] <= 160 * 1024
    q = plan([5, 4, 6], [1, -1, 1], 16, 16, 3, 4096)  # with moving obstacles: the WAVE shape
    assert q["supported"] == 1 and q["wave"] == 1
    # 48 terms and more: the wide kernels, four waves per CU
    q = plan([11, 12], [1, -1], 16, 16, 0, 4096)
    assert q["supported"] == 1 and q["cap"] == 48 and q["wave"] == 1 and q["threads"] * q["wg_per_cu"] <= 256
    # configs[4]: 32 pieces x 65 points with four moving obstacles, n = 63
    q = plan([32], [1], 64, 64, 4, 4096)
    assert q["supported"] == 1 and q["cap"] == 64 and q["wave"] == 1 and q["threads"] * q["wg_per_cu"] <= 256 and q["lds"] <= 160 * 1024
    # the reference's live case, 12 obstacles (5 H + S + 4 = 36 terms: the 64-bit mask)
    assert plan([5, 4, 6], [1, -1, 1], 12, 16, 12, 64)["supported"] == 1
    # beyond a wave of variables / five half-planes: the generic TEAM kernel whatever the batch size (round 6)
    for kw in (dict(pieces=[40]), dict(pieces=[8], H=6), dict(pieces=[8], H=12)):
        q = plan(kw["pieces"], [1], 8, 8, 0, 4096, H=kw.get("H", 4))
        assert q["supported"] == 1 and q["wave"] == 0 and q["lds"] <= 160 * 1024, (kw, q)
    # outside the mode: more than 12 half-planes, more than 64 terms per point, more than 256 variables
    assert plan([8], [1], 8, 8, 0, 1, H=13)["supported"] == 0
    assert plan([8], [1], 8, 8, 45, 1)["supported"] == 0
    assert plan([130], [1], 4, 4, 0, 1)["supported"] == 0
