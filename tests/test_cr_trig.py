"""The correctly rounded sin / cos of the reference-order kernel (dftpav_amd/csrc/cr_trig.h, double-double arithmetic) against
binary128 (libquadmath) -- on the host, where the same header compiles (its fused multiply-adds are IEEE operations with the
same result on gfx950, scripts/ieee_probe.hip).  Also measures how often this host's libm is NOT correctly rounded: the reason
a device cannot reproduce "the reference's bits" on layouts with a gear shift."""
import ctypes as C
import math
import os
import subprocess
import tempfile

import numpy as np

from dftpav_amd import pods

SRC = r"""
#include <quadmath.h>
void cr_sincos_q(int n, const double *x, double *s, double *c) {
  for (int i = 0; i < n; i++) { __float128 q = (__float128)x[i]; s[i] = (double)sinq(q); c[i] = (double)cosq(q); }
}
"""


def _quad():
    d = tempfile.mkdtemp()
    open(os.path.join(d, "q.c"), "w").write(SRC)
    so = os.path.join(d, "libq.so")
    subprocess.check_call(["gcc", "-O2", "-fPIC", "-shared", os.path.join(d, "q.c"), "-o", so, "-lquadmath"])
    q = C.CDLL(so)
    q.cr_sincos_q.argtypes = [C.c_int, pods.c_double_p, pods.c_double_p, pods.c_double_p]
    return q


def test_cr_sincos_equals_binary128_rounded(hiplib):
    fn = hiplib.lib().dftpav_debug_cr_sincos
    fn.argtypes = [C.c_int, pods.c_double_p, pods.c_double_p, pods.c_double_p]
    q = _quad()
    rng = np.random.default_rng(0)
    n = 200000
    libm_off = libm_n = 0
    # 1e6 straddles the switch to the Payne-Hanek reduction at 2^20; a line search can try junction angles of 1e10 and beyond
    for scale in (3.2, 10.0, 100.0, 1.0e4, 1.0e-3, 1.0e-8, 2.0e6, 1.0e10, 1.0e22, 1.0e150, 1.7e308):
        x = rng.uniform(-1.0, 1.0, n) * scale
        if scale >= 1.0e22:     # every binade up to the largest
            x = x * 2.0 ** -rng.integers(0, int(np.log2(scale)) - 20, n)
        if scale == 3.2:        # close to the multiples of pi / 2, where the reduction loses the most
            x[: n // 2] = rng.integers(-50, 50, n // 2) * (np.pi / 2) + rng.normal(0, 1e-9, n // 2)
            x[n // 2: n // 2 + 4] = [0.0, -0.0, np.pi / 2, -np.pi]
        if scale == 1.0e10:     # doubles next to large multiples of pi / 2: the reduction cancels up to 60 bits
            kk = rng.integers(1, 2 ** 40, n // 2).astype(np.float64)
            x[: n // 2] = kk * (np.pi / 2)
        s, c, sq, cq = (np.zeros(n) for _ in range(4))
        assert fn(n, pods.dptr(x), pods.dptr(s), pods.dptr(c)) == 0
        q.cr_sincos_q(n, pods.dptr(x), pods.dptr(sq), pods.dptr(cq))
        assert np.array_equal(s, sq) and np.array_equal(c, cq), scale
        assert np.array_equal(np.signbit(s), np.signbit(sq))        # the zeros keep their sign
        if scale == 10.0:
            ls = np.array([math.sin(v) for v in x[:50000]])
            lc = np.array([math.cos(v) for v in x[:50000]])
            libm_off += int((ls != sq[:50000]).sum() + (lc != cq[:50000]).sum())
            libm_n += 100000
    x = np.array([np.inf, -np.inf, np.nan, 6381956970095103.0 * 2.0 ** 797, 1.7976931348623157e308, 5e-324, -2.85456754454990e10])
    s, c, sq, cq = (np.zeros(len(x)) for _ in range(4))
    fn(len(x), pods.dptr(x), pods.dptr(s), pods.dptr(c))
    q.cr_sincos_q(len(x), pods.dptr(x), pods.dptr(sq), pods.dptr(cq))
    assert np.array_equal(s, sq, equal_nan=True) and np.array_equal(c, cq, equal_nan=True)      # 6381956970095103 2^797: the double closest to a multiple of pi / 2
    print("this host's libm differs from the correctly rounded sin / cos for %d of %d arguments" % (libm_off, libm_n))


SRC2 = r"""
#include <quadmath.h>
void q_fn(int which, int n, const double *x, double *y) {
  for (int i = 0; i < n; i++) { __float128 q = (__float128)x[i]; y[i] = which == 0 ? (double)expq(q) : (which == 1 ? (double)logq(q) : (double)(q * q * q)); }
}
"""


def test_cr_exp_log_cube_equal_binary128_rounded(hiplib):
    """exp, log and x^3 of cr_trig.h (the moving-obstacle term of the reference calls libm's exp / log / pow) against binary128"""
    fn = hiplib.lib().dftpav_debug_cr_fn
    fn.argtypes = [C.c_int, C.c_int, pods.c_double_p, pods.c_double_p]
    d = tempfile.mkdtemp()
    open(os.path.join(d, "q.c"), "w").write(SRC2)
    so = os.path.join(d, "libq2.so")
    subprocess.check_call(["gcc", "-O2", "-fPIC", "-shared", os.path.join(d, "q.c"), "-o", so, "-lquadmath"])
    q = C.CDLL(so)
    q.q_fn.argtypes = [C.c_int, C.c_int, pods.c_double_p, pods.c_double_p]
    rng = np.random.default_rng(1)
    n = 200000
    sets = {0: [rng.uniform(-700, 1, n), rng.uniform(-30, 0, n), rng.uniform(-1e-3, 1e-3, n), np.array([0.0, -0.0, -745.0, -800.0] + [-1.0] * (n - 4))],
            1: [rng.uniform(1, 8, n), rng.uniform(1e-3, 1e3, n), 1 + rng.uniform(-1e-6, 1e-6, n), np.exp(rng.uniform(-300, 300, n)),
                np.array([1.0, 2.0, 8.0, 0.5] + [3.0] * (n - 4))],
            2: [rng.uniform(0, 30, n), rng.normal(0, 1e3, n), rng.uniform(0, 1e-3, n)]}
    inf, nan = np.inf, np.nan
    special = {0: [709.0, 709.78, 709.79, 710.0, 1e10, inf, -inf, nan, -708.5, -744.0, -745.1, -745.2, -1e10, 1e-300, 5e-324],
               1: [inf, 0.0, -0.0, -1.0, nan, 5e-324, 1e-310, 2.2250738585072014e-308, 1e300, 1.7976931348623157e308],
               2: [1e100, 5e102, 1e103, -1e103, inf, -inf, nan, 1e-110, 1e-105, -1e-108, 0.0, -0.0, 5e-324]}
    for which, xs in special.items():   # overflowed penalties of a far trial point: what libm returns for them
        x = np.array(xs)
        y, yq = np.zeros(len(x)), np.zeros(len(x))
        fn(which, len(x), pods.dptr(x), pods.dptr(y))
        q.q_fn(which, len(x), pods.dptr(x), pods.dptr(yq))
        assert np.array_equal(y, yq, equal_nan=True) and np.array_equal(np.signbit(y), np.signbit(yq)), (which, y, yq)
    off = {0: 0, 1: 0, 2: 0}
    for which, xs in sets.items():
        for x in xs:
            x = np.ascontiguousarray(x, dtype=np.float64)
            y, yq = np.zeros(n), np.zeros(n)
            assert fn(which, n, pods.dptr(x), pods.dptr(y)) == 0
            q.q_fn(which, n, pods.dptr(x), pods.dptr(yq))
            keep = (yq >= 2.3e-308) | (yq == 0.0) if which == 0 else np.ones(n, bool)   # subnormal exp: rounded twice (absorbed by the sums)
            assert np.array_equal(y[keep], yq[keep]), which
            if which < 2:   # exp / log answer from a quick phase when its rounding test is certain: never different from the accurate phase
                ya = np.zeros(n)
                assert fn(which + 3, n, pods.dptr(x), pods.dptr(ya)) == 0
                assert np.array_equal(y, ya, equal_nan=True), which
            lib = [math.exp, math.log, lambda v: math.pow(v, 3)][which]
            off[which] += int(sum(lib(v) != w for v, w in zip(x[:20000], yq[:20000])))
    print("this host's libm differs from the correctly rounded value: exp %d, log %d, pow(x, 3) %d of 80000 / 100000 / 60000 arguments" %
          (off[0], off[1], off[2]))


def test_quick_phases_never_disagree_with_the_accurate_ones(tmp_path):
    """scripts/cr_quick_check.cpp (the header compiled for the host, OpenMP): exp / log / sincos through their quick phase + rounding
    test against the double-double series alone on 2M random arguments per range -- no mismatch, and the quick phase hands over for
    about one argument in 1 400 (exp, log) / 640 (sincos: two roundings) -- a test that never hands over would not be a test."""
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    exe = str(tmp_path / "cr_quick_check")
    subprocess.check_call(["g++", "-O2", "-fopenmp", "-std=c++17", "-ffp-contract=off", "-I", os.path.join(root, "dftpav_amd", "csrc"),
                           os.path.join(root, "scripts", "cr_quick_check.cpp"), "-o", exe])
    out = subprocess.run([exe, "2000000"], capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stdout
    lines = [ln for ln in out.stdout.splitlines() if "arguments" in ln]
    assert len(lines) == 12
    for ln in lines:
        assert " 0 mismatches" in ln, ln
        handed = int(ln.split("handed over")[1].split()[0])
        if ln.startswith("sin / cos"):
            assert 2000000 / 1300 < handed < 2000000 / 300, ln
        elif "e^[" not in ln:   # (the logarithm of a correctly rounded exponential sits next to a double: its rounding is never in doubt)
            assert 2000000 / 3000 < handed < 2000000 / 700, ln


SRC_ATAN = r"""
#include <quadmath.h>
void q_atan2(int n, const double *y, const double *x, double *o) { for (int i = 0; i < n; i++) o[i] = (double)atan2q((__float128)y[i], (__float128)x[i]); }
void q_atan(int n, const double *x, double *o) { for (int i = 0; i < n; i++) o[i] = (double)atanq((__float128)x[i]); }
void q_tan(int n, const double *x, double *o) { for (int i = 0; i < n; i++) o[i] = (double)tanq((__float128)x[i]); }
"""


def test_cr_atan2_atan_tan_equal_binary128_rounded(hiplib):
    """The correctly rounded atan2 / atan / tan of the step kernels' reference order (cr_trig.h: the heading of a sampled state, the
    steering angle, the front end's curvature) against binary128 rounded to double: every quadrant, the axes and the signed zeros,
    quotients from 2^-1000 to 2^1000, arguments next to the table's nodes k / 64, large arguments of tan."""
    L = hiplib.lib()
    d = tempfile.mkdtemp()
    open(os.path.join(d, "q.c"), "w").write(SRC_ATAN)
    so = os.path.join(d, "libqa.so")
    subprocess.check_call(["gcc", "-O2", "-fPIC", "-shared", os.path.join(d, "q.c"), "-o", so, "-lquadmath"])
    q = C.CDLL(so)
    dp = pods.c_double_p
    q.q_atan2.argtypes = [C.c_int, dp, dp, dp]
    q.q_atan.argtypes = q.q_tan.argtypes = [C.c_int, dp, dp]
    L.dftpav_debug_cr_atan2.argtypes = [C.c_int, dp, dp, dp]
    L.dftpav_debug_cr_fn.argtypes = [C.c_int, C.c_int, dp, dp]
    rng = np.random.default_rng(3)
    n = 200000
    for scale in (1.0, 1.0e-3, 1.0e3, 1.0e-30, 1.0e30, 1.0e-200, 1.0e200, 1.0e-300, 1.0e300):
        y = rng.normal(0, 1, n)
        x = rng.normal(0, 1, n) * scale
        if scale == 1.0:
            k = rng.integers(0, 65, n // 4)
            y[: n // 4] = (k / 64.0) * (1.0 + rng.normal(0, 1e-12, n // 4)) * np.abs(x[: n // 4])   # next to the table's nodes
            y[n // 4: n // 4 + 12] = [0.0, -0.0, 0.0, -0.0, 1.0, -1.0, np.inf, -np.inf, np.inf, 1.0, 5e-324, 1.0]
            x[n // 4: n // 4 + 12] = [1.0, 1.0, -1.0, -1.0, 0.0, -0.0, np.inf, -np.inf, 1.0, np.inf, 1.0, 5e-324]
            y[n // 2: n // 2 + 1000] = x[n // 2: n // 2 + 1000]          # the diagonal
        o, oq = np.zeros(n), np.zeros(n)
        assert L.dftpav_debug_cr_atan2(n, pods.dptr(y), pods.dptr(x), pods.dptr(o)) == 0
        q.q_atan2(n, pods.dptr(y), pods.dptr(x), pods.dptr(oq))
        assert np.array_equal(o, oq), (scale, int((o != oq).sum()), y[o != oq][:3], x[o != oq][:3])
        assert np.array_equal(np.signbit(o), np.signbit(oq))
    for scale in (1.0, 64.0, 1.0e-3, 1.0e-20, 1.0e6, 1.0e40, 1.0e300):
        x = rng.normal(0, 1, n) * scale
        x[:5] = [0.0, -0.0, 1.0, -1.0, 0.5]
        o, oq = np.zeros(n), np.zeros(n)
        assert L.dftpav_debug_cr_fn(5, n, pods.dptr(x), pods.dptr(o)) == 0
        q.q_atan(n, pods.dptr(x), pods.dptr(oq))
        assert np.array_equal(o, oq), ("atan", scale, int((o != oq).sum()))
    for scale in (1.0, 1.5, 1.0e-3, 1.0e-20, 100.0, 1.0e6, 1.0e10, 1.0e100):
        x = rng.uniform(-1, 1, n) * scale
        if scale == 1.5:        # next to the poles
            x[: n // 2] = (2 * rng.integers(-20, 20, n // 2) + 1) * (np.pi / 2) + rng.normal(0, 1e-9, n // 2)
        o, oq = np.zeros(n), np.zeros(n)
        assert L.dftpav_debug_cr_fn(6, n, pods.dptr(x), pods.dptr(o)) == 0
        q.q_tan(n, pods.dptr(x), pods.dptr(oq))
        assert np.array_equal(o, oq), ("tan", scale, int((o != oq).sum()))
