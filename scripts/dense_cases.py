"""The random problems of scripts/fuzz_dense_cpu.py by index (shared with tests/test_gpu_dense.py and scripts/dense_check.py)."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from dftpav_amd import scenarios as sc  # noqa: E402
from oracle import pyoracle as po  # noqa: E402


def make_case(index):
    """the index-th random problem of this fuzz: (params, scenario, pieces)"""
    rng = np.random.default_rng(31000 + index)
    M = int(rng.choice([1, 1, 2, 3]))
    pieces = [int(rng.integers(2, 11)) for _ in range(M)]
    while 2 * (sum(pieces) - M) + M + 3 * (M - 1) > 64:
        pieces[int(np.argmax(pieces))] -= 1
    sing = [int(rng.choice([1, -1]))]
    for _ in range(M - 1):
        sing.append(-sing[-1])
    moving = bool(rng.uniform() < 0.25) and sum(pieces) <= 12
    p = po.default_params()
    p.lbfgs_mem_size = int(rng.choice([1, 2, 3, 5, 8, 15, 16, 17, 31, 32, 33, 47, 64, 100, 256, 300]))
    B = int(rng.integers(1, 4))
    s = sc.make_scenario(pieces, sing, int(rng.integers(3, 21)), int(rng.integers(3, 21)), B, seed=32000 + index, with_moving=moving,
                         n_obs=int(rng.integers(0, 60)))
    s.apply_resolution(p)
    if rng.uniform() < 0.4:
        p.max_forward_vel *= float(rng.uniform(0.3, 1.0)); p.max_forward_acc *= float(rng.uniform(0.2, 1.0)); p.max_forward_cur *= float(rng.uniform(0.2, 1.0))
        p.wei_obs *= float(rng.uniform(0.1, 10)); p.wei_feas *= float(rng.uniform(0.1, 10)); p.wei_time *= float(rng.uniform(0.1, 10))
    if moving:
        s.t_now = float(rng.uniform(0.0, 5.0))
    return p, s, pieces


