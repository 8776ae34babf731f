"""dftpav_amd — MI355X-native batched MINCO / L-BFGS trajectory optimiser.

Drop-in for the solve path of Dftpav's traj_planner
(PolyTrajOptimizer::OptimizeTrajectory, traj_optimizer.cpp:7-202).  The compute
lives in libdftpav_hip.so (hand-written HIP for gfx950) behind the C-ABI of
include/dftpav_hip.h; this package is the thin Python host side used by the
tests and bench.py.
"""
from .pods import BatchData, Layout, LayoutSpec, Params, Surround, SurroundSet  # noqa: F401
