"""Times the read-out kernel (dftpav_batch_sample_states) on solved batches and counts how often the server's
singularity filter engages; checks a sample of trajectories against the oracle bit for bit.
  python scripts/states_time.py [B]"""
import sys, time
import numpy as np
sys.path.insert(0, ".")
from dftpav_amd import capi, scenarios as sc
from oracle import pyoracle as po

B = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
for cfg, b in ((2, 64), (3, B)):
    p = capi.default_params()
    s = sc.baseline_config(cfg, B=b)
    s.apply_resolution(p)
    h = capi.Handle(p)
    h.set_surround(s.surround)
    bt = capi.Batch(h, s.layout, s.B)
    bt.upload(s)
    bt.solve()
    co, dts = bt.coeffs()
    total = (dts * s.layout.piece_nums[None, :]).sum(axis=1)
    n = int(total.max() / 0.01) + 2
    t0 = time.perf_counter()
    raw, nv = bt.sample_states(sample_dt=0.01, n_samples=n, filter_singularity=False)
    fil, _ = bt.sample_states(sample_dt=0.01, n_samples=n, filter_singularity=True)
    wall = (time.perf_counter() - t0) / 2
    ms = h.corridor_last_ms()
    changed = int((raw[..., 3] != fil[..., 3]).sum())
    slow = int((np.abs(raw[..., 5]) < 0.1).sum() - (raw[..., 0] == 0).sum() + b)
    idx = np.arange(min(b, 32)) * max(1, b // 32)
    so, no = po.sample_states(co[idx], dts[idx], s.layout.piece_nums, s.layout.singuls, sample_dt=0.01, n_samples=n,
                              filter_singularity=True, wheel_base=p.veh_wheel_base, order=1)
    ok = bool(np.array_equal(so, fil[idx]) and np.array_equal(no, nv[idx]))
    t1 = time.perf_counter()
    po.sample_states(co[idx], dts[idx], s.layout.piece_nums, s.layout.singuls, sample_dt=0.01, n_samples=n, order=0)
    cpu = (time.perf_counter() - t1) / len(idx)
    print("cfg %d: %d trajectories x %d samples (%.0f MB): kernel %.3f ms -> %.1f M states/s, %.0f GB/s written; with download "
          "%.1f ms | slow samples %d, headings held by the filter %d | CPU oracle %.2f ms per trajectory (1 thread) | "
          "bit-identical: %s" % (cfg, b, n, raw.nbytes / 1e6, ms, nv.sum() / ms / 1e3, raw.nbytes / ms / 1e6, wall * 1e3, slow,
                                 changed, cpu * 1e3, ok))
    bt.close(); h.close()
