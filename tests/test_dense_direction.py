"""The dense form of the L-BFGS direction (dftpav_amd/csrc/dense_dir.h, oracle order 3) -- EXPERIMENTAL device-order mode.

H_k of L-BFGS is the composition of the maps X -> V_j^T X V_j + rho_j s_j s_j^T over the window, applied to gamma I; with
n = 31 .. 63 variables and a window of m = 256 pairs it is cheaper, and free of the two-loop recursion's 2 x bound DEPENDENT
reductions, to keep the composition as two n x n matrices (H = gamma A^T A + C), update them per accepted pair, and keep the
window with a two-stack queue of suffix aggregates.  dense_dir.h holds the arithmetic (shared by the kernel's lanes and the
oracle's loop over lanes); these tests are its CPU validation:
  * every direction against the plain two-loop recursion (lbfgs.hpp:716-739) over the same window in 80-bit arithmetic;
  * the window logic (rebuilds, suffix stepping) at memories 1, 8, 32 and 256;
  * whole solves: same success, statistically the same optima and iteration counts as the two-loop device order;
  * nothing but the direction changes: evaluations are order 1's bit for bit.
The device path is off by default and has not run on a GPU this round."""
import numpy as np
import pytest

from dftpav_amd import scenarios as sc


def _solve_all(oracle, p, s, order):
    return [oracle.OracleProblem(p, s, b, order=order).solve()[1] for b in range(s.B)]


@pytest.mark.parametrize("cfg,B,mem", [(3, 10, 256), (3, 6, 32), (3, 6, 8), (3, 4, 1), (2, 6, 256), (1, 6, 256), (5, 2, 256), (2, 4, 17)])
def test_every_direction_agrees_with_the_two_loop_recursion(oracle, cfg, B, mem):
    """max-norm relative difference of d = -H g to the two-loop recursion in 80-bit arithmetic over whole solves.  The fp64
    two-loop recursion itself sits at 1e-15 .. 3e-13 against that yardstick; the dense form (repeated rank-2 updates of C) at
    1e-14 typical, 1e-9 worst seen; the bar is 1e-8, five orders below what the line search can tell apart."""
    p = oracle.default_params()
    p.lbfgs_mem_size = mem
    s = sc.baseline_config(cfg, B=B)
    s.apply_resolution(p)
    oracle.dense_check(True)
    try:
        res = _solve_all(oracle, p, s, 3)
        st = oracle.dense_stats()
    finally:
        oracle.dense_check(False)
    assert all(r.success for r in res)
    assert st["directions"] >= sum(r.iters for r in res) * 0.9 - B       # (rejected pairs take d = -g and are not compared)
    deepest = min(mem, max(r.iters for r in res))
    assert deepest - 2 <= st["deepest_window"] <= deepest       # (the first iteration has no pair yet)
    assert st["max_rel_d"] <= 1e-8, st
    if 1 < mem <= 32:
        assert st["with_front"] > 0     # the window slid: suffix aggregates were in play


def test_the_window_slides_at_the_reference_memory(oracle):
    """m = 256 (pb.txt:96) on solves longer than 256 iterations: rebuild + suffix stepping at full size"""
    p = oracle.default_params()
    s = sc.baseline_config(3, B=16, seed=20240)
    s.apply_resolution(p)
    oracle.dense_check(True)
    try:
        res = _solve_all(oracle, p, s, 3)
        st = oracle.dense_stats()
    finally:
        oracle.dense_check(False)
    assert max(r.iters for r in res) > 300 and st["with_front"] > 100 and st["deepest_window"] == 256
    assert st["max_rel_d"] <= 1e-8, st


def test_whole_solves_are_statistically_those_of_the_two_loop_device_order(oracle):
    p = oracle.default_params()
    s = sc.baseline_config(3, B=96)
    s.apply_resolution(p)
    a = oracle.solve_batch(p, s, nthreads=8, order=1)
    b = oracle.solve_batch(p, s, nthreads=8, order=3)
    assert a["success"].all() and b["success"].all()
    assert abs(np.median(a["final_cost"]) - np.median(b["final_cost"])) < 0.03 * np.median(a["final_cost"])
    assert abs(a["iters"].mean() - b["iters"].mean()) < 0.15 * a["iters"].mean()
    lr = np.log(b["final_cost"] / a["final_cost"])
    assert abs(lr.mean()) < 3.0 * lr.std() / np.sqrt(len(lr)) + 1e-3       # no bias beyond the spread of a chaotic solver
    assert np.array_equal(a["hist_sum"] > 0, b["hist_sum"] > 0)


def test_only_the_direction_differs(oracle):
    """evaluations in order 3 are order 1's; the first iteration (d = -g, no pair yet) is too: the first trial point agrees"""
    p = oracle.default_params()
    s = sc.baseline_config(2, B=2)
    s.apply_resolution(p)
    rng = np.random.default_rng(4)
    for b in range(2):
        o1, o3 = oracle.OracleProblem(p, s, b, order=1), oracle.OracleProblem(p, s, b, order=3)
        x = o1.x0() + rng.normal(0, 0.2, o1.n)
        f1, g1 = o1.eval(x)
        f3, g3 = o3.eval(x)
        assert f1 == f3 and np.array_equal(g1, g3)
    a = oracle.solve_batch(p, s, nthreads=1, order=3)
    b_ = oracle.solve_batch(p, s, nthreads=2, order=3)
    assert np.array_equal(a["x"], b_["x"]) and np.array_equal(a["final_cost"], b_["final_cost"])     # deterministic


@pytest.mark.parametrize("name,mem", [("cfg1", 256), ("cfg2", 256), ("cfg3", 256), ("cfg3", 8), ("cfg5", 256)])
def test_order3_bits_are_pinned(oracle, name, mem):
    """tests/golden/dense.npz (tests/golden/make_golden_dense.py): the order's arithmetic is portable, so its whole solves are
    the same bits on every host -- and must stay so when dense_dir.h is restructured (batched reads, other loop shapes)"""
    import os
    from golden_util import GOLDEN_DIR, load
    z = np.load(os.path.join(GOLDEN_DIR, "dense.npz"))
    s, _ = load(name)
    p = oracle.default_params()
    p.lbfgs_mem_size = mem
    s.apply_resolution(p)
    r = oracle.solve_batch(p, s, nthreads=2, order=3)
    for k in ("x", "final_cost", "status", "iters", "evals", "hist_sum", "success"):
        assert np.array_equal(r[k], z["%s_m%d_%s" % (name, mem, k)]), k


def test_ill_conditioned_windows_go_to_the_plain_recursion():
    """Forming H explicitly loses accuracy where a pair of the window has a small curvature (|V| = |y||s| / (y.s) large): without
    the gate of dense_dir.h these layouts of scripts/fuzz_dense_cpu.py had directions 0.4 (relative) away from the recursion.
    With it the iterations of such windows take the plain recursion: every direction within 1e-5 (the fp64 recursion itself is
    at 2.5e-6 on the worst of them), and the gate is seen closing."""
    import subprocess
    import sys
    import os
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    for n_cases, first in ((8, 3920), (1, 2454), (1, 1615)):       # the fuzz's cases 2920-2927, 1454, 615 of seed block 1000
        out = subprocess.run([sys.executable, os.path.join(root, "scripts", "fuzz_dense_cpu.py"), str(n_cases), str(first)], capture_output=True, text=True, timeout=600)
        assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-2000:]
        line = out.stdout.strip().splitlines()[-1]
        assert "0 failures" in line, line
    assert "from the plain recursion" in line
    closed = int(line.split(" from the plain recursion")[0].split(", ")[-1])
    assert closed > 0, line          # case 615's third trajectory closes the gate
