"""Writes tests/golden/dense.npz: whole solves of oracle order 3 (the dense direction, dftpav_amd/csrc/dense_dir.h) on the inputs
of the existing golden cases -- the pin of that order's BITS (its arithmetic is portable: no libm call), so that a restructuring
of dense_dir.h that changes a rounding shows, and the device path has stored vectors to be compared with.
    python tests/golden/make_golden_dense.py"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from golden_util import GOLDEN_DIR, load  # noqa: E402
from oracle import pyoracle as po  # noqa: E402

po.build()
out = {}
for name, mem in (("cfg1", 256), ("cfg2", 256), ("cfg3", 256), ("cfg3", 8), ("cfg5", 256)):
    s, _ = load(name)
    p = po.default_params()
    p.lbfgs_mem_size = mem
    s.apply_resolution(p)
    r = po.solve_batch(p, s, nthreads=4, order=3)
    for k in ("x", "final_cost", "status", "iters", "evals", "hist_sum", "success"):
        out["%s_m%d_%s" % (name, mem, k)] = r[k]
    print(name, mem, "iters", r["iters"])
np.savez_compressed(os.path.join(GOLDEN_DIR, "dense.npz"), **out)
