"""RETIRED CROSS-CHECK (rounds 3-5; NOT a parity pin): the restatement oracle/dftpav_oracle.c (literal order) against a build of the
reference's sources over STAND-IN headers.

oracle/_ref/libdftpav_ref.so was /root/reference/src/Plan/traj_planner/{src/traj_optimizer.cpp,
include/plan_utils/poly_traj_utils.hpp, include/geo_utils2d/lbfgs.hpp} compiled unmodified (oracle/Makefile.ref) against
interface stand-ins for Eigen / ROS / the protobuf config written in this repository (oracle/ref_shim).  The reference needs those
libraries, so it is unbuildable here and such a build is not a reference build: since round 6 nothing builds it and the `ref` tests
below SKIP (they run only for a developer who sets DFTPAV_STANDIN_BUILD=1, oracle/pyref.py).  Two kinds of test:
  * `ref` tests call that build beside the oracle (skipped);
  * golden tests check the oracle against the vectors that build once wrote (tests/golden/ref_*.npz: regression data, see
    tests/golden/README.md) -- they run anywhere.
The bar is bit equality: same evaluation points, same costs, same gradients, same iterates, same counts.
"""
import hashlib
import os

import numpy as np
import pytest

import ref_cases
from dftpav_amd import scenarios as sc
from golden_util import CASES, GOLDEN_DIR, load

REF_SRC = "/root/reference/src/Plan/traj_planner"
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def ref():
    from oracle import pyref
    if os.path.exists(REF_SRC):
        pyref.build()
    if not pyref.available():
        pytest.skip("oracle/_ref is not built here and /root/reference is absent")
    pyref.lib()
    return pyref


def test_ref_is_built_from_the_reference_tree_not_from_copies(ref):
    """the recipe compiles the files where they lie; nothing of them is in the repository"""
    mk = open(os.path.join(ROOT, "oracle", "Makefile.ref")).read()
    assert "$(REF)/src/traj_optimizer.cpp" in mk and "REF ?= /root/reference/src/Plan/traj_planner" in mk
    for dirpath, _, files in os.walk(ROOT):
        if ".git" in dirpath or "gpurun_out" in dirpath:
            continue
        assert "traj_optimizer.cpp" not in files and "lbfgs.hpp" not in files and "poly_traj_utils.hpp" not in files
    if os.path.exists(REF_SRC):
        for line in open(os.path.join(ROOT, "oracle", "_ref", "SOURCES.sha256")):
            digest, path = line.split()
            assert hashlib.sha256(open(path, "rb").read()).hexdigest() == digest


@pytest.mark.parametrize("case", [c[0] for c in ref_cases.LBFGS_CASES])
def test_lbfgs_is_bit_equal_to_the_reference(ref, oracle, case):
    """lbfgs_optimize + line_search_lewisoverton (lbfgs.hpp:276-390, 440-751) on analytic functions"""
    name, fn, x0, kw = [c for c in ref_cases.LBFGS_CASES if c[0] == case][0]
    p = ref_cases.with_fields(oracle.default_params(), kw)
    a, b = ref.lbfgs(fn, x0, p), oracle.lbfgs(fn, x0, p)
    assert a["ret"] == b["ret"] and a["iters"] == b["iters"] and a["evals"] == b["evals"]
    assert a["f"] == b["f"] and np.array_equal(a["x"], b["x"])
    assert a["iters"] >= 2


def test_banded_system_is_bit_equal_to_the_reference(ref, oracle):
    """BandedSystem::factorizeLU / solve (poly_traj_utils.hpp:776-826): the oracle's dense operator is built column by
    column with its restatement of that LU, so each column must equal the reference's solve of the same unit vector."""
    for N in (2, 5, 16):
        A = sc.minco_matrix(N)
        op = oracle.minco_operator(N)  # [6N][N+5]: columns = non-zero RHS rows 0,1,2, 6i+5, 6N-3..6N-1
        rows = [0, 1, 2] + [6 * i + 5 for i in range(N - 1)] + [6 * N - 3, 6 * N - 2, 6 * N - 1]
        E = np.zeros((6 * N, len(rows)))
        for c, r in enumerate(rows):
            E[r, c] = 1.0
        X = ref.banded_solve(A, 6, 6, E)
        assert np.array_equal(X, op)


@pytest.mark.parametrize("N", [2, 3, 8, 16, 32])
def test_minco_is_bit_equal_to_the_reference(ref, oracle, N):
    """MinJerkOpt::reset / generate / getTrajJerkCost (poly_traj_utils.hpp:880-1009)"""
    inner, dT, head, tail = ref_cases.minco_inputs(N, 100 + N)
    r = ref.minco(inner, dT, head, tail)
    c, J = oracle.minco_generate(inner, dT, head, tail)
    assert np.array_equal(c, r["coeffs"]) and J == r["energy"]


def _compare_problem(ref, oracle, p, s, b):
    o = oracle.OracleProblem(p, s, b, order=0)
    r = ref.RefProblem(p, s, b)
    rr = r.optimize(trace=True)
    x0 = o.x0()
    assert np.array_equal(rr["eval_x"][0], x0)            # packing + RealT2VirtualT (traj_optimizer.cpp:96-115, 360-369)
    fo, go = o.eval(x0)
    assert fo == rr["eval_f"][0] and np.array_equal(go, rr["eval_g"][0])
    co, dto = o.coeffs()
    fr, gr = r.eval(x0)
    cr, dtr = r.coeffs()
    assert fo == fr and np.array_equal(go, gr)
    assert np.array_equal(co, cr) and np.array_equal(dto, dtr)  # generate + VirtualT2RealT
    xo, ro = o.solve()
    assert np.array_equal(xo, rr["x"]) and ro.final_cost == rr["final_cost"]
    assert (ro.status, ro.iters, ro.evals, bool(ro.success)) == (rr["status"], rr["iters"], rr["evals"], rr["ok"])
    # a point in the middle of the solve, where penalties are active differently than at x0
    xm = rr["iter_x"][len(rr["iter_x"]) // 2]
    fo, go = o.eval(xm)
    fr, gr = r.eval(xm)
    assert fo == fr and np.array_equal(go, gr)
    return rr


@pytest.mark.parametrize("cfg,B", [(1, 3), (2, 3), (3, 4), (5, 2)])
def test_cost_gradient_and_whole_solve_are_bit_equal_to_the_reference(ref, oracle, cfg, B):
    """costFunctionCallback and OptimizeTrajectory of the reference on the BASELINE configs against the literal oracle"""
    p = oracle.default_params()
    s = sc.baseline_config(cfg, B=B)
    s.apply_resolution(p)
    for b in range(B):
        rr = _compare_problem(ref, oracle, p, s, b)
        assert rr["ok"] and rr["iters"] > 20


def test_random_layouts_are_bit_equal_to_the_reference(ref, oracle):
    """a seeded slice of scripts/fuzz_ref.py: gear patterns, resolutions, limits, weights, help_eps, obstacle clock, memory"""
    for c in range(16):
        rng = np.random.default_rng(7000 + c)
        M = int(rng.choice([1, 2, 3]))
        pieces = [int(rng.integers(2, 9)) for _ in range(M)]
        sing = [int(rng.choice([1, -1]))]
        for _ in range(M - 1):
            sing.append(-sing[-1])
        moving = c % 4 == 0 and sum(pieces) <= 12
        p = oracle.default_params()
        s = sc.make_scenario(pieces, sing, int(rng.integers(3, 17)), int(rng.integers(3, 17)), 1, seed=8000 + c, with_moving=moving,
                             n_obs=int(rng.integers(0, 60)))
        s.apply_resolution(p)
        if c % 3 == 0:
            p.lbfgs_mem_size = int(rng.choice([4, 8, 17]))
        if c % 2 == 0:
            p.max_forward_vel *= 0.5; p.max_backward_vel *= 0.6; p.max_forward_acc *= 0.4; p.max_forward_cur *= 0.3
            p.wei_obs *= 3.0; p.wei_time *= 0.3
        if c % 5 == 0:
            s.help_eps = 1e-3
        if moving:
            s.t_now = float(rng.uniform(0.0, 5.0))
        _compare_problem(ref, oracle, p, s, 0)


def test_scalar_pieces_match_the_reference(ref, oracle):
    """positiveSmoothedL1 (traj_optimizer.cpp:783-806)"""
    p = oracle.default_params()
    import ctypes as C
    for x in np.concatenate([np.linspace(-1e-4, 3e-4, 41), [1e-4, 0.99999e-4, 1.0, 37.5]]):
        f, df = C.c_double(0), C.c_double(0)
        oracle.lib().oracle_smoothed_l1(float(x), C.byref(f), C.byref(df))
        assert (f.value, df.value) == ref.smoothed_l1(p, x)


# ---------------------------------------------------------------- golden vectors written by the reference build
@pytest.mark.parametrize("name", CASES)
def test_oracle_reproduces_the_retired_standin_builds_vectors(oracle, name):
    s, _ = load(name)
    z = np.load(os.path.join(GOLDEN_DIR, "ref_" + name + ".npz"))
    p = oracle.default_params()
    s.apply_resolution(p)
    for b in range(s.B):
        o = oracle.OracleProblem(p, s, b, order=0)
        x0 = o.x0()
        assert np.array_equal(x0, z["x0"][b])
        f, g = o.eval(x0)
        assert f == z["f0"][b] and np.array_equal(g, z["g0"][b])
        x, r = o.solve()
        assert np.array_equal(x, z["x"][b]) and r.final_cost == z["cost"][b]
        assert (r.status, r.iters, r.evals, r.success) == (z["status"][b], z["iters"][b], z["evals"][b], z["ok"][b])
        # every accepted iterate of the reference's solve is reproduced when the oracle is evaluated there
        xi, fi = z["iter_x_%d" % b], z["iter_fx_%d" % b]
        for k in (0, len(fi) // 3, len(fi) - 1):
            assert o.eval(xi[k])[0] == fi[k]


def test_oracle_lbfgs_and_minco_reproduce_the_retired_standin_builds_vectors(oracle):
    z = np.load(os.path.join(GOLDEN_DIR, "ref_units.npz"))
    p = oracle.default_params()
    got = ref_cases.run_lbfgs(oracle, p)
    for k, v in got.items():
        assert np.array_equal(v, z[k]), k
    for N in (2, 3, 8, 16, 32):
        inner, dT, head, tail = ref_cases.minco_inputs(N, 100 + N)
        c, J = oracle.minco_generate(inner, dT, head, tail)
        assert np.array_equal(c, z["minco_%d_coeffs" % N]) and J == z["minco_%d_energy" % N][0]
    import ctypes as C
    for x, (f0, d0) in zip(z["l1_x"], z["l1_f"]):
        f, df = C.c_double(0), C.c_double(0)
        oracle.lib().oracle_smoothed_l1(float(x), C.byref(f), C.byref(df))
        assert (f.value, df.value) == (f0, d0)


# ---- the reference-order device mode's substitution tables (dftpav_amd/csrc/solver_ref.hip: sweep)
_INTERIOR = [[0x3f, 0x1f, 0x0f, 0x00, 0x00, 0x3e], [0x00, 0x18, 0x30, 0x31, 0x21, 0x06],
             [0x00, 0x00, 0x00, 0x35, 0x3b, 0x30], [0x03, 0x07, 0x0f, 0x1e, 0x3c, 0x38]]


def _row_sweep(tab, q, b):
    """the kernel's row-oriented sweep q over one right-hand side, in Python's IEEE doubles (multiply, then subtract;
    the division is the one the kernel reproduces with its stored reciprocal)"""
    n6 = tab.shape[1]
    desc, div = q in (1, 3), q in (1, 2)
    b = b.copy()
    for idx in range(n6):
        i = n6 - 1 - idx if desc else idx
        generic = idx < 6 or idx >= n6 - 6
        acc = b[i]
        for k in range(6):
            c = tab[q, i, k]
            src = i + 6 - k if desc else i - 6 + k
            if generic:
                if c != 0.0 and 0 <= src < n6:
                    acc = acc - c * b[src]
            elif _INTERIOR[q][i % 6] >> k & 1:
                acc = acc - c * b[src]
        if div:
            acc = acc / tab[q, i, 6]
        b[i] = acc
    return b


@pytest.mark.parametrize("N", [2, 3, 4, 8, 10, 16, 31, 32])
def test_row_sweeps_of_the_reference_order_kernel_equal_banded_solve(ref, hiplib, N):
    """BandedSystem::solve / solveAdj (poly_traj_utils.hpp:805-852) are column loops; the reference-order kernel runs them
    row by row from tables the host derives from the same LU (capi.cpp: reference_order_tables), the middle blocks with a
    fixed non-zero pattern.  Row form == column form, bit for bit, on random right-hand sides; the stored reciprocal of
    the diagonal is the correctly rounded one."""
    import ctypes as C
    from dftpav_amd import pods
    fn = hiplib.lib().dftpav_debug_reference_tables
    fn.argtypes = [C.c_int, pods.c_double_p]
    tab = np.zeros((4, 6 * N, 8))
    assert fn(N, pods.dptr(tab)) == 1          # the middle blocks have the pattern the kernel assumes
    assert np.array_equal(tab[:, :, 7], 1.0 / tab[:, :, 6])
    A = sc.minco_matrix(N)
    rng = np.random.default_rng(N)
    for trial in range(3):
        b = rng.normal(0, 10.0 ** rng.integers(-2, 3), 6 * N)
        if trial == 0:                          # the sparsity of a real right-hand side (poly_traj_utils.hpp:968-977)
            keep = [0, 1, 2] + [6 * i + 5 for i in range(N - 1)] + [6 * N - 3, 6 * N - 2, 6 * N - 1]
            z = np.zeros_like(b)
            z[keep] = b[keep]
            b = z
        want = ref.banded_solve(A, 6, 6, b)[:, 0]
        got = _row_sweep(tab, 1, _row_sweep(tab, 0, b))
        assert np.array_equal(got, want)
        want_adj = ref.banded_solve(A, 6, 6, b, adjoint=True)[:, 0]
        got_adj = _row_sweep(tab, 3, _row_sweep(tab, 2, b))
        assert np.array_equal(got_adj, want_adj)


def _packed_sweep(pk, off, N, q, b):
    """the kernel's sweep q over one right-hand side from the PACKED table (solver_ref.hip: "The table of one sweep"): blocks of six
    rows in traversal order; the two end blocks hold whole rows (six coefficients, diagonal, 1 / diagonal) and test every
    coefficient; an interior block holds only the coefficients of the pattern _INTERIOR, in (row, k) order, then -- for the sweeps
    that divide -- (diagonal, 1 / diagonal) of its six rows from an even slot on."""
    n6 = 6 * N
    desc, div = q in (1, 3), q in (1, 2)
    masks = [_INTERIOR[q][5 - r] if desc else _INTERIOR[q][r] for r in range(6)]
    ncoef = sum(bin(m).count("1") for m in masks)
    diag0 = (ncoef + 1) & ~1
    size = diag0 + 12 if div else (ncoef + 1) & ~1
    b = b.copy()
    w = [0.0] * 6
    o = off
    for blk in range(N):
        end = blk == 0 or blk == N - 1
        for r in range(6):
            i = n6 - 1 - (6 * blk + r) if desc else 6 * blk + r
            acc = b[i]
            if end:
                row = pk[o + 8 * r: o + 8 * r + 8]
                for k in range(6):
                    if row[k] != 0.0:
                        acc = acc - row[k] * w[(r + k) % 6]
                if div:
                    assert row[7] == 1.0 / row[6]
                    acc = acc / row[6]
            else:
                pos = o + sum(bin(masks[rr]).count("1") for rr in range(r))
                for k in range(6):
                    if masks[r] >> k & 1:
                        acc = acc - pk[pos] * w[(r + k) % 6]
                        pos += 1
                if div:
                    assert pk[o + diag0 + 2 * r + 1] == 1.0 / pk[o + diag0 + 2 * r]
                    acc = acc / pk[o + diag0 + 2 * r]
            w[r] = acc
            b[i] = acc
        o += 48 if end else size
    return b, o


@pytest.mark.parametrize("N", [2, 3, 4, 9, 16, 32])
def test_packed_sweep_tables_of_the_reference_order_kernel(ref, hiplib, N):
    """What the kernel reads is the packed form of those tables (12.9 KB instead of 24.6 KB for 16 pieces): sweeping over it as
    the kernel does -- the window of the six previous results indexed (r + k) mod 6 -- gives BandedSystem::solve / solveAdj of the
    reference build bit for bit, and the four sweeps fill exactly the size the host allocates."""
    import ctypes as C
    from dftpav_amd import pods
    fn = hiplib.lib().dftpav_debug_reference_tables_packed
    fn.argtypes = [C.c_int, pods.c_double_p, C.POINTER(C.c_int)]
    nd = C.c_int(0)
    assert fn(N, None, C.byref(nd)) == 0 and nd.value == 384 + 88 * (N - 2)
    pk = np.zeros(nd.value)
    assert fn(N, pods.dptr(pk), C.byref(nd)) == 1
    A = sc.minco_matrix(N)
    rng = np.random.default_rng(100 + N)
    for trial in range(2):
        b = rng.normal(0, 10.0 ** rng.integers(-2, 3), 6 * N)
        y, o = _packed_sweep(pk, 0, N, 0, b)
        x, o = _packed_sweep(pk, o, N, 1, y)
        assert np.array_equal(x, ref.banded_solve(A, 6, 6, b)[:, 0])
        y, o = _packed_sweep(pk, o, N, 2, b)
        x, o = _packed_sweep(pk, o, N, 3, y)
        assert np.array_equal(x, ref.banded_solve(A, 6, 6, b, adjoint=True)[:, 0])
        assert o == nd.value



@pytest.mark.parametrize("layout,seed", [(([7, 6], [1, -1]), 81), (([5, 4, 6], [1, -1, 1]), 82)])
def test_live_case_gear_shifts_with_moving_obstacles(ref, oracle, layout, seed):
    """What the reference's only live caller passes (traj_manager.cpp:604-610): a multi-segment layout AND the moving cars,
    obstacle clocks that differ from the ego's (traj_optimizer.cpp:1367-1369).  The literal oracle (order 0) is bit-equal
    to the reference build through the whole solve; order 2 -- the same program with correctly rounded cos / sin / exp /
    log / pow, the order the GPU's reference-order kernel is held bit-equal to -- stays within 1e-12 of it per evaluation,
    and the moving-obstacle term is active on the points compared."""
    pieces, sing = layout
    B = 4
    p = oracle.default_params()
    s = sc.make_scenario(pieces, sing, 12, 16, B, seed=seed, with_moving=True, n_obs=25, start_centre=(-38.0, 5.0))
    s.surround.start_time[:] = [0.5, 0.0, 1.5, 0.25]
    s.t_now = 0.6
    s.apply_resolution(p)
    rng = np.random.default_rng(seed)
    active = 0
    for b in range(B):
        rr = _compare_problem(ref, oracle, p, s, b)
        o2 = oracle.OracleProblem(p, s, b, order=2)
        r = ref.RefProblem(p, s, b)
        x0 = o2.x0()
        for x in (x0, x0 + rng.normal(0, 0.25, x0.shape), rr["iter_x"][len(rr["iter_x"]) // 2]):
            f2, g2 = o2.eval(x)
            fr, gr = r.eval(x)
            assert abs(f2 - fr) <= 1e-12 * abs(fr), (layout, b, f2, fr)
            assert np.abs(g2 - gr).max() <= 1e-12 * max(1.0, np.abs(gr).max())
            active += o2.cost_terms()[3] > 0.0
    assert active >= B, active


# ---------------------------------------------------------------------------------------------------------------------------
# The pin of ORDER 2 -- the order the device's reference-order kernel implements -- to the reference's own code.
# Where the reference's loop calls libm (cos / sin of the junction angles of a gear shift; exp / log / pow(., 3) of the moving
# obstacles' soft min / max) the bits of oracle/_ref belong to this host's glibc.  oracle/_ref/libdftpav_ref_cr.so is the SAME
# objects (traj_optimizer.cpp compiled unmodified) linked against a correctly rounded exp / log / pow / sin / cos
# (oracle/cr_libm.c, binary128): the reference's compiled program on the one libm every host agrees on.  Order 2 must be
# bit-equal to it: every evaluation, every iterate, every count.

def _compare_order2_with_the_reference_on_a_correctly_rounded_libm(ref, oracle, p, s, b, rng):
    o2 = oracle.OracleProblem(p, s, b, order=2)
    r = ref.RefProblem(p, s, b, cr=True)
    rr = r.optimize(trace=True)
    x0 = o2.x0()
    assert np.array_equal(rr["eval_x"][0], x0)
    for x in (x0, x0 + rng.normal(0, 0.25, x0.shape), rr["iter_x"][len(rr["iter_x"]) // 2]):
        f2, g2 = o2.eval(x)
        fr, gr = r.eval(x)
        assert f2 == fr and np.array_equal(g2, gr), (b, f2, fr)
        c2, dt2 = o2.coeffs()
        cr_, dtr = r.coeffs()
        assert np.array_equal(c2, cr_) and np.array_equal(dt2, dtr)
    x2, r2 = o2.solve()
    assert np.array_equal(x2, rr["x"]) and r2.final_cost == rr["final_cost"]
    assert (r2.status, r2.iters, r2.evals, bool(r2.success)) == (rr["status"], rr["iters"], rr["evals"], rr["ok"])
    return rr


def test_cr_build_is_the_reference_objects_without_libm_imports(ref):
    """the recipe links the SAME two objects as oracle/_ref plus cr_libm.o; the result imports no exp / log / pow / sin / cos"""
    import subprocess
    if not ref.cr_available():
        pytest.skip("oracle/_ref/libdftpav_ref_cr.so is not built here")
    mk = open(os.path.join(ROOT, "oracle", "Makefile.ref")).read()
    assert "$(OUT_CR): _ref/traj_optimizer.o _ref/ref_driver.o _ref/cr_libm.o" in mk
    und = subprocess.run(["nm", "-D", "--undefined-only", os.path.join(ROOT, "oracle", "_ref", "libdftpav_ref_cr.so")], capture_output=True, text=True).stdout
    names = {ln.split()[-1].split("@")[0] for ln in und.splitlines() if ln.strip()}
    assert not names & {"exp", "log", "pow", "sin", "cos", "sincos"} and {"expq", "logq", "sinq", "cosq"} <= names


def test_order2_is_bit_equal_to_the_reference_on_a_correctly_rounded_libm_gear_shifts(ref, oracle):
    """BASELINE configs[1] (one gear shift): the nine instances bench.py's `single` times.  The glibc build agrees with
    order 2 on a part of them only (whenever sincos rounded every junction angle correctly)"""
    p = oracle.default_params()
    rng = np.random.default_rng(1)
    same_as_glibc = 0
    for sd in range(9):
        s = sc.baseline_config(2, B=1, seed=20240 + 17 * sd)
        s.apply_resolution(p)
        rr = _compare_order2_with_the_reference_on_a_correctly_rounded_libm(ref, oracle, p, s, 0, rng)
        rg = ref.RefProblem(p, s, 0).optimize()
        same_as_glibc += bool(np.array_equal(rg["x"], rr["x"]))
    assert same_as_glibc < 9      # or the correctly rounded libm would be an empty distinction on this host


@pytest.mark.parametrize("layout,seed", [(([7, 6], [1, -1]), 81), (([5, 4, 6], [1, -1, 1]), 82)])
def test_order2_is_bit_equal_to_the_reference_on_a_correctly_rounded_libm_live_case(ref, oracle, layout, seed):
    """gear shifts AND moving obstacles (traj_manager.cpp:604-610), the scenario of tests/test_gpu_reference_order.py"""
    pieces, sing = layout
    B = 6
    p = oracle.default_params()
    s = sc.make_scenario(pieces, sing, 12, 16, B, seed=seed, with_moving=True, n_obs=25, start_centre=(-38.0, 5.0))
    s.surround.start_time[:] = [0.5, 0.0, 1.5, 0.25]
    s.t_now = 0.6
    s.apply_resolution(p)
    rng = np.random.default_rng(seed)
    for b in range(B):
        _compare_order2_with_the_reference_on_a_correctly_rounded_libm(ref, oracle, p, s, b, rng)


def test_order2_is_bit_equal_to_the_reference_on_a_correctly_rounded_libm_configs4(ref, oracle):
    """BASELINE configs[4] (32 pieces x 65 points, four moving cars): one whole solve (binary128 exp / log: seconds per solve)"""
    p = oracle.default_params()
    s = sc.baseline_config(5, B=1)
    s.apply_resolution(p)
    _compare_order2_with_the_reference_on_a_correctly_rounded_libm(ref, oracle, p, s, 0, np.random.default_rng(5))


# ---------------------------------------------------------------------------------------------------------------------------
# The steps either side of the solve path: the restatements oracle/{corridor,validate,states,fit,frontend}_oracle.cpp
# against the reference's OWN code -- oracle/_ref/libdftpav_ref_next.so = the functions cut verbatim out of /root/reference by
# oracle/ref_slices.py (traj_manager.cpp, map_adapter.cpp, kino_astar.cpp, traj_server_ros.cpp, the simulator's grid / outline
# code) compiled by oracle/ref_next_driver.cpp.  Same process, same libm: order 0 of every restatement must agree BIT FOR BIT, on
# the scenarios the GPU tests of these steps use (tests/test_gpu_parity.py).  The device's own order (order 1: portable
# transcendentals) is tied to order 0 by the existing oracle tests.
@pytest.fixture(scope="module")
def refnext():
    from oracle import pyoracle as po
    if os.path.exists(REF_SRC):
        from oracle import pyref
        pyref.build()
    if not os.path.exists(os.path.join(ROOT, "oracle", "_ref", "libdftpav_ref_next.so")):
        pytest.skip("oracle/_ref/libdftpav_ref_next.so is not built here and /root/reference is absent")
    po.ref_next_lib()
    return po


def test_ref_next_is_cut_from_the_reference_tree_by_line_range(refnext):
    """every slice is where the manifest says, none of it is in the repository, and the library holds the five entry points"""
    import subprocess
    import sys
    if os.path.exists(REF_SRC):
        out = subprocess.run([sys.executable, os.path.join(ROOT, "oracle", "ref_slices.py"), "/root/reference", os.path.join(ROOT, "oracle", "_ref", "slices")],
                             capture_output=True, text=True)
        assert out.returncode == 0, out.stdout + out.stderr
    tracked = subprocess.run(["git", "-C", ROOT, "ls-files", "oracle/_ref"], capture_output=True, text=True).stdout.strip()
    assert tracked == ""
    L = refnext.ref_next_lib()
    for name in ("corridor_rectangles", "validate_trajectories", "sample_states", "fit_surround", "frontend_resample"):
        assert hasattr(L, "ref_" + name)


def _corridor_scene(seed):  # tests/test_gpu_parity.py::_corridor_scene
    rng = np.random.default_rng(seed)
    obs = np.column_stack([rng.uniform(-25, 25, 60), rng.uniform(-25, 25, 60), rng.uniform(0.5, 1.5, 60)])
    grid, origin = sc.occupancy_grid(obs, arena=80.0)
    states = np.column_stack([rng.uniform(-22, 22, 1500), rng.uniform(-22, 22, 1500), rng.uniform(-7.0, 7.0, 1500)])
    return grid, origin, states


@pytest.mark.parametrize("seed", [1, 2])
def test_corridor_oracle_is_bit_equal_to_getRectangleConst(refnext, seed):
    """§8(f)-1: corridor_oracle.cpp == TrajPlanner::getRectangleConst (traj_manager.cpp:1213-1469) over
    TrajPlannerAdapter::CheckIfCollisionUsingLine (map_adapter.cpp:117-129) and the simulator's GridMapND"""
    po = refnext
    grid, origin, states = _corridor_scene(seed)
    want = po.corridor_rectangles(grid, sc.MAP_RESL, origin, states, ref=True)
    got = po.corridor_rectangles(grid, sc.MAP_RESL, origin, states, order=0)
    assert np.array_equal(got, want)
    # the rectangles are not trivial: sides stop at obstacles at many different lengths
    ext = np.einsum("pki,pki->pk", want[:, :, :2], want[:, :, 2:] - states[:, None, :2])
    assert len(np.unique(np.round(ext, 6))) > 50
    # a pose outside the map (every sample out of range counts as free), an empty map of another resolution
    far = np.array([[500.0, -300.0, 0.7]])
    assert np.array_equal(po.corridor_rectangles(grid, sc.MAP_RESL, origin, far, order=0), po.corridor_rectangles(grid, sc.MAP_RESL, origin, far, ref=True))
    empty = np.full((50, 70), 127, dtype=np.uint8)
    assert np.array_equal(po.corridor_rectangles(empty, 0.25, (-3.0, -4.0), states[:40], order=0),
                          po.corridor_rectangles(empty, 0.25, (-3.0, -4.0), states[:40], ref=True))
    # the path poses of a BASELINE scenario on its own map (the corridor the solver's inputs are made of)
    s = sc.baseline_config(3, B=4)
    st = s.meta["states"]
    c = (0.5 * (st[..., 0].min() + st[..., 0].max()), 0.5 * (st[..., 1].min() + st[..., 1].max()))
    g2, o2 = sc.occupancy_grid(s.meta["obstacles"], arena=120.0, centre=c)
    assert np.array_equal(po.corridor_rectangles(g2, sc.MAP_RESL, o2, st.reshape(-1, 3), order=0),
                          po.corridor_rectangles(g2, sc.MAP_RESL, o2, st.reshape(-1, 3), ref=True))


def _solved(po, cfg, B):
    """optimised trajectories of a BASELINE config (the restatement's own solves): coefficients, piece durations, scenario"""
    p = po.default_params()
    s = sc.baseline_config(cfg, B=B)
    s.apply_resolution(p)
    r = po.solve_batch(p, s, nthreads=4, order=1)
    co, dts = [], []
    for b in range(s.B):
        pr = po.OracleProblem(p, s, b, order=1)
        pr.eval(r["x"][b])
        a, d = pr.coeffs()
        co.append(a)
        dts.append(d)
    return np.array(co), np.array(dts), s, p


@pytest.mark.parametrize("cfg,B", [(3, 24), (2, 6)])
def test_validate_oracle_is_bit_equal_to_CheckReplan(refnext, cfg, B):
    """§8(f)-2: validate_oracle.cpp == the collision loop of TrajPlannerServer::CheckReplan (traj_server_ros.cpp:385-397) over
    Trajectory::getPos / getAngle, SemanticMapManager::CheckCollisionUsingPosAndYaw (semantic_map_manager.cc:639-662) and
    ShapeUtils::GetDenseVerticesOfOrientedBoundingBox (shapes.cc:110-149)"""
    po = refnext
    co, dts, s, _ = _solved(po, cfg, B)
    lay = s.layout
    st = s.meta["states"]
    c = (0.5 * (st[..., 0].min() + st[..., 0].max()), 0.5 * (st[..., 1].min() + st[..., 1].max()))
    obs = s.meta["obstacles"]
    grid, origin = sc.occupancy_grid(obs, arena=140.0, centre=c)
    a = po.validate_trajectories(grid, sc.MAP_RESL, origin, co, dts, lay.piece_nums, lay.singuls, order=0)
    b = po.validate_trajectories(grid, sc.MAP_RESL, origin, co, dts, lay.piece_nums, lay.singuls, ref=True)
    assert np.array_equal(a[0], b[0]) and np.array_equal(a[1], b[1])
    # obstacles dropped onto the nominal paths: most trajectories flagged, at the same first sample
    rng = np.random.default_rng(cfg)
    extra = [[st[h, k, 0], st[h, k, 1], 0.8] for h in range(st.shape[0]) for k in rng.choice(np.arange(st.shape[1] // 3, st.shape[1]), 3, replace=False)]
    grid2, origin2 = sc.occupancy_grid(np.vstack([obs, np.array(extra)]), arena=140.0, centre=c)
    a = po.validate_trajectories(grid2, sc.MAP_RESL, origin2, co, dts, lay.piece_nums, lay.singuls, order=0)
    b = po.validate_trajectories(grid2, sc.MAP_RESL, origin2, co, dts, lay.piece_nums, lay.singuls, ref=True)
    assert np.array_equal(a[0], b[0]) and np.array_equal(a[1], b[1])
    assert b[0].mean() > 0.5 and (b[1][b[0] == 1] > 0).all() and len(np.unique(b[1])) > 1
    # the server's 0.05 s and the outline's 0.1 m are constants of the reference: its code takes no others
    with pytest.raises(ValueError):
        po.validate_trajectories(grid, sc.MAP_RESL, origin, co, dts, lay.piece_nums, lay.singuls, sample_dt=0.21, ref=True)


@pytest.mark.parametrize("cfg,B", [(2, 6), (3, 16)])
def test_states_oracle_is_bit_equal_to_GetState_and_the_servers_playback(refnext, cfg, B):
    """§8(f)-2, the read-out: states_oracle.cpp == Trajectory::GetState (poly_traj_utils.hpp:378-406) driven by the playback tick
    of TrajPlannerServer::PublishData (traj_server_ros.cpp:248-259) with FilterSingularityState (:335-356), the segments chained
    by TrajContainer::addSingulTraj as traj_manager.cpp:618-625 does"""
    po = refnext
    co, dts, s, p = _solved(po, cfg, B)
    lay = s.layout
    total = (dts * lay.piece_nums[None, :]).sum(axis=1)
    for t0, dt, n, filt in [(0.0, 0.01, int(total.max() / 0.01) + 40, True), (0.0, 0.01, 300, False), (-0.3, 0.037, 600, True),
                            (2.5, 0.2, 7, True)]:
        a, na = po.sample_states(co, dts, lay.piece_nums, lay.singuls, t0=t0, sample_dt=dt, n_samples=n, filter_singularity=filt, order=0)
        b, nb = po.sample_states(co, dts, lay.piece_nums, lay.singuls, t0=t0, sample_dt=dt, n_samples=n, filter_singularity=filt, ref=True)
        assert np.array_equal(na, nb)
        assert np.array_equal(a, b)
    assert (nb > 0).all()
    # a standstill where the filter acts: a reversing segment entered at zero speed (the heading jumps by pi between samples)
    co2 = np.zeros((1, 5, 6, 2))
    for q in range(3):
        co2[0, q, 0] = [1.0 * q, 0.0]
        co2[0, q, 1] = [1.0, 0.0]
    for q in range(2):
        co2[0, 3 + q, 0] = [3.0 - 0.05 * 1.5 * q, 0.0]
        co2[0, 3 + q, 1] = [-0.05, 1e-9]
    a, na = po.sample_states(co2, [[1.0, 1.5]], [3, 2], [1, -1], sample_dt=0.01, n_samples=700, order=0)
    b, nb = po.sample_states(co2, [[1.0, 1.5]], [3, 2], [1, -1], sample_dt=0.01, n_samples=700, ref=True)
    assert np.array_equal(na, nb) and np.array_equal(a, b)


def test_fit_oracle_is_bit_equal_to_ConverSurroundTrajFromPoints(refnext):
    """§8(f)-4: fit_oracle.cpp == TrajPlanner::ConverSurroundTrajFromPoints + state_to_flat_output (traj_manager.cpp:743-789,
    139-158) over MinJerkOpt::reset / generate / getTraj"""
    po = refnext
    st = sc.predicted_states()
    a, b = po.fit_surround(st, order=0), po.fit_surround(st, ref=True)
    for k in a:
        assert np.array_equal(a[k], b[k]), k
    rng = np.random.default_rng(5)
    st2 = sc.predicted_states(pre_time=9.0, deltatime=0.75, cars=[(rng.uniform(-30, 30), rng.uniform(-30, 30), rng.uniform(1, 6),
                                                                   rng.uniform(3, 20), rng.uniform(0, 6.28)) for _ in range(7)])
    for sub in (st2, st2[:1], st2[:, :3]):
        a, b = po.fit_surround(sub, order=0), po.fit_surround(sub, ref=True)
        for k in a:
            assert np.array_equal(a[k], b[k]), k
    z = st.copy()
    z[:, 0, 3] = 0.0  # a car at rest: the 1e-5 speed rule of state_to_flat_output
    a, b = po.fit_surround(z, order=0), po.fit_surround(z, ref=True)
    for k in a:
        assert np.array_equal(a[k], b[k]), k


@pytest.mark.parametrize("gears,K,Kd", [((1, -1), 16, 32), ((-1, 1, -1, 1), 32, 32), ((1,), 7, 11)])
def test_frontend_oracle_is_bit_equal_to_getKinoNode_and_RunMINCOParking(refnext, gears, K, Kd):
    """§8(f)-3: frontend_oracle.cpp == KinoAstar::getKinoNode from SampleTraj on (kino_astar.cpp:613-743), evaluatePos (:468-521),
    evaluateDuration / evaluateLength (:744-795), getFlatState (:834-857) and the resampling loop of TrajPlanner::RunMINCOParking
    (traj_manager.cpp:531-568).  (What builds SampleTraj -- the A* nodes and the OMPL Reeds-Shepp shot -- is not in the reference
    tree's reach here: oracle/shot_oracle*.cpp stay property-pinned.)"""
    from dftpav_amd.pods import FrontendParams
    po = refnext
    P, pl, ss, es, ct = sc.searched_paths(40, seed=len(gears) + K, gears=gears, seg_duration=6.0)
    fp = FrontendParams.default(K=K, Kd=Kd)
    a = po.frontend_resample(P, pl, ss, es, ct, fp, order=0)
    b = po.frontend_resample(P, pl, ss, es, ct, fp, ref=True)
    for k in a:
        assert np.array_equal(a[k], b[k]), k
    assert (b["n_seg"] == len(gears)).all() and (b["n_states"][:, :len(gears)] > 0).all()
    # fewer segments than gear changes: the counts are reported, nothing else is produced
    a = po.frontend_resample(P, pl, ss, es, ct, fp, order=0, max_seg=max(1, len(gears) - 1))
    b = po.frontend_resample(P, pl, ss, es, ct, fp, ref=True, max_seg=max(1, len(gears) - 1))
    for k in a:
        assert np.array_equal(a[k], b[k]), k


def test_step_oracles_reproduce_the_retired_standin_builds_step_vectors(oracle):
    """anywhere (no /root/reference needed): order 0 of the five restatements against tests/golden/ref_steps.npz, written by the
    reference's own code on the inputs of tests/golden/steps.npz (tests/golden/make_golden_ref_steps.py).  Same libm as the
    writer's (this image): bit equality."""
    import importlib.util
    spec = importlib.util.spec_from_file_location("make_golden_ref_steps", os.path.join(GOLDEN_DIR, "make_golden_ref_steps.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    Z, R = np.load(os.path.join(GOLDEN_DIR, "steps.npz")), np.load(os.path.join(GOLDEN_DIR, "ref_steps.npz"))
    got = mod.run(Z, order=0)
    assert set(got) == set(R.files)
    for k in got:
        assert np.array_equal(got[k], R[k]), k


def test_eigen_redux_build_differs_only_in_the_dynamic_reductions(ref):
    """oracle/_ref/libdftpav_ref_eigen.so (the stand-in Eigen's reductions of dynamic vectors in Eigen 3.3's SSE2 order,
    scripts/eigen_redux_cpu.py): costFunctionCallback has no such reduction -- same bits as the sequential build on every
    evaluation -- while lbfgs_optimize's dot products do, so whole solves part ways (profiles/r05_eigen_redux_cpu.json: 0 of 2048
    keep their bits, with the spread of a one-ulp perturbation)."""
    if not os.path.exists(REF_SRC):
        pytest.skip("needs /root/reference to build the variant")
    from oracle import pyoracle as po
    p = po.default_params()
    s = sc.baseline_config(3, B=3, seed=20240)
    s.apply_resolution(p)
    parted = 0
    for b in range(3):
        a, e = ref.RefProblem(p, s, b), ref.RefProblem(p, s, b, eigen_redux=True)
        ra, re_ = a.optimize(trace=True), e.optimize(trace=True)
        assert np.array_equal(ra["eval_x"][0], re_["eval_x"][0])
        assert ra["eval_f"][0] == re_["eval_f"][0] and np.array_equal(ra["eval_g"][0], re_["eval_g"][0])
        x = ra["x"]
        fa, ga = a.eval(x)
        fe, ge = e.eval(x)
        assert fa == fe and np.array_equal(ga, ge)
        parted += int(not np.array_equal(ra["x"], re_["x"]))
    assert parted >= 2
