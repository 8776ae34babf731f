"""GPU tests of the EXPERIMENTAL dense direction (csrc/dense_dir.h, dftpav_debug_set_direction) -- written in round 4 together
with the device path, which that round could compile but not run.  They are SKIPPED unless DFTPAV_TEST_DENSE=1: the mode is off
by default, and a test tier must not go red over a path nobody has switched on.  Round 5 runs them first
(scripts/r05_first.sh, scripts/dense_check.py)."""
import os

import numpy as np
import pytest

from dftpav_amd import scenarios as sc
from golden_util import GOLDEN_DIR, load

pytestmark = [pytest.mark.gpu, pytest.mark.skipif(os.environ.get("DFTPAV_TEST_DENSE") != "1", reason="experimental mode: set DFTPAV_TEST_DENSE=1")]
KEYS = ("final_cost", "x", "status", "iters", "evals", "hist_sum", "success")


def _batch(hiplib, s, p, residency=None):
    h = hiplib.Handle(p)
    h.set_surround(s.surround)
    bt = hiplib.Batch(h, s.layout, s.B) if residency is None else hiplib.Batch(h, s.layout, s.B, residency=residency)
    bt.upload(s)
    bt.debug_set_direction(True)
    return h, bt


@pytest.mark.parametrize("cfg,B,mem", [(3, 16, 256), (3, 8, 32), (3, 8, 8), (3, 4, 1), (2, 8, 256), (1, 8, 256), (5, 4, 256), (2, 6, 17), (3, 64, 256)])
def test_whole_solves_equal_oracle_order3(hiplib, oracle, cfg, B, mem):
    p = hiplib.default_params()
    p.lbfgs_mem_size = mem
    s = sc.baseline_config(cfg, B=B)
    s.apply_resolution(p)
    h, bt = _batch(hiplib, s, p)
    want = oracle.solve_batch(p, s, nthreads=4, order=3)
    for rep in range(2):      # the second solve starts from aggregates the first one left behind
        r = bt.solve()
        for k in KEYS:
            assert np.array_equal(r[k], want[k]), (k, rep)
    bt.debug_set_direction(False)      # and back: the two-loop recursion's bits again
    r = bt.solve()
    w1 = oracle.solve_batch(p, s, nthreads=4, order=1)
    for k in KEYS:
        assert np.array_equal(r[k], w1[k]), k
    bt.close()
    h.close()


@pytest.mark.parametrize("name,mem", [("cfg1", 256), ("cfg2", 256), ("cfg3", 256), ("cfg3", 8), ("cfg5", 256)])
def test_stored_vectors(hiplib, name, mem):
    z = np.load(os.path.join(GOLDEN_DIR, "dense.npz"))
    s, _ = load(name)
    p = hiplib.default_params()
    p.lbfgs_mem_size = mem
    s.apply_resolution(p)
    h, bt = _batch(hiplib, s, p)
    r = bt.solve()
    for k in KEYS:
        assert np.array_equal(r[k], z["%s_m%d_%s" % (name, mem, k)]), k
    bt.close()
    h.close()


@pytest.mark.parametrize("cfg,B,slots,slice_,hand_over", [(3, 48, 8, 7, 3), (3, 48, 5, 40, 0), (2, 12, 4, 1, 12)])
def test_time_sliced_schedule_is_bit_identical(hiplib, oracle, monkeypatch, cfg, B, slots, slice_, hand_over):
    """suspended trajectories carry their aggregates (HBM) and the queue's front index (state record) across workgroups"""
    monkeypatch.setenv("DFTPAV_SCHED", "1")
    monkeypatch.setenv("DFTPAV_SLOTS", str(slots))
    monkeypatch.setenv("DFTPAV_SLICE", str(slice_))
    monkeypatch.setenv("DFTPAV_HANDOVER", str(hand_over))
    p = hiplib.default_params()
    p.lbfgs_mem_size = 32          # the window slides inside the test
    s = sc.baseline_config(cfg, B=B)
    s.apply_resolution(p)
    h, bt = _batch(hiplib, s, p)
    want = oracle.solve_batch(p, s, nthreads=8, order=3)
    for rep in range(2):
        r = bt.solve()
        for k in KEYS:
            assert np.array_equal(r[k], want[k]), (k, rep)
    bt.close()
    h.close()


def test_throughput_residency_and_the_reference_order_guard(hiplib, oracle):
    p = hiplib.default_params()
    s = sc.baseline_config(3, B=96)
    s.apply_resolution(p)
    h, bt = _batch(hiplib, s, p, residency=2)       # one wave per trajectory
    want = oracle.solve_batch(p, s, nthreads=8, order=3)
    r = bt.solve()
    for k in KEYS:
        assert np.array_equal(r[k], want[k]), k
    with pytest.raises(hiplib.DftpavError):          # a variant of the device order: not under the reference order
        bt.set_order(hiplib.ORDER_REFERENCE)
    bt.close()
    h.close()


@pytest.mark.parametrize("index", [1615, 2454, 3927, 3542])
def test_windows_that_close_the_gate(hiplib, oracle, index):
    """layouts of scripts/fuzz_dense_cpu.py whose windows hold ill-conditioned pairs: those iterations take the plain recursion
    (dense_dir.h: kNuGate) -- on the device as in oracle order 3, bit for bit"""
    import sys
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "scripts"))
    from dense_cases import make_case
    p, s, _ = make_case(index)
    h, bt = _batch(hiplib, s, p)
    want = oracle.solve_batch(p, s, nthreads=3, order=3)
    r = bt.solve()
    for k in KEYS:
        assert np.array_equal(r[k], want[k]), k
    bt.close()
    h.close()
