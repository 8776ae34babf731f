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
    for scale in (3.2, 10.0, 100.0, 1.0e4, 1.0e-3, 1.0e-8):
        x = rng.uniform(-scale, scale, n)
        if scale == 3.2:        # close to the multiples of pi / 2, where the reduction loses the most
            x[: n // 2] = rng.integers(-50, 50, n // 2) * (np.pi / 2) + rng.normal(0, 1e-9, n // 2)
            x[n // 2: n // 2 + 4] = [0.0, -0.0, np.pi / 2, -np.pi]
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
    print("this host's libm differs from the correctly rounded sin / cos for %d of %d arguments" % (libm_off, libm_n))
