"""Developer script: throughput of the time-sliced schedule vs the plain launch (run through gpurun)."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from dftpav_amd import capi, scenarios as sc

B = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
cfg = int(sys.argv[2]) if len(sys.argv) > 2 else 3          # 5: BASELINE configs[4] (its default: slice 32, hand-over 256)
p = capi.default_params()
s = sc.baseline_config(cfg, B=B); s.apply_resolution(p)
ref = None
CFG5 = [("default (s32 h256)", {}), ("s16", {"DFTPAV_SLICE": "16"}), ("s64", {"DFTPAV_SLICE": "64"}), ("s128", {"DFTPAV_SLICE": "128"}),
        ("s100000 (run to completion)", {"DFTPAV_SLICE": "100000"}), ("h128", {"DFTPAV_HANDOVER": "128"}), ("h512", {"DFTPAV_HANDOVER": "512"}),
        ("h768", {"DFTPAV_HANDOVER": "768"}), ("h0", {"DFTPAV_HANDOVER": "0"}), ("s64 h512", {"DFTPAV_SLICE": "64", "DFTPAV_HANDOVER": "512"}),
        ("s16 h512", {"DFTPAV_SLICE": "16", "DFTPAV_HANDOVER": "512"}), ("plain", {"DFTPAV_SCHED": "0"})]
for tag, env in CFG5 if cfg == 5 else [("plain", {"DFTPAV_SCHED": "0"}),
                 ("queue s48 h256", {"DFTPAV_SCHED": "1"}),
                 ("queue s24 h256", {"DFTPAV_SCHED": "1", "DFTPAV_SLICE": "24"}),
                 ("queue s96 h256", {"DFTPAV_SCHED": "1", "DFTPAV_SLICE": "96"}),
                 ("queue s48 h512", {"DFTPAV_SCHED": "1", "DFTPAV_HANDOVER": "512"}),
                 ("queue s48 h128", {"DFTPAV_SCHED": "1", "DFTPAV_HANDOVER": "128"}),
                 ("queue s48 h0", {"DFTPAV_SCHED": "1", "DFTPAV_HANDOVER": "0"}),
                 ("queue s32 h384", {"DFTPAV_SCHED": "1", "DFTPAV_SLICE": "32", "DFTPAV_HANDOVER": "384"}),
                 ("queue s64 h192", {"DFTPAV_SCHED": "1", "DFTPAV_SLICE": "64", "DFTPAV_HANDOVER": "192"}),
                 ("queue s48 h256 768 slots", {"DFTPAV_SCHED": "1", "DFTPAV_SLOTS": "768"})]:
    for k in ("DFTPAV_SCHED", "DFTPAV_SLICE", "DFTPAV_HANDOVER", "DFTPAV_SLOTS"):
        os.environ.pop(k, None)
    os.environ.update(env)
    h = capi.Handle(p); h.set_surround(s.surround); bt = capi.Batch(h, s.layout, B); bt.upload(s)
    bt.solve_async(); bt.sync()
    ms = []
    for _ in range(3):
        bt.solve_async(); bt.sync(); ms.append(bt.last_solve_ms())
    r = bt.results()
    if ref is None: ref = r
    same = all(np.array_equal(r[k], ref[k]) for k in ("final_cost", "x", "iters", "evals", "status"))
    print("%-26s kernel ms %s  solves/s %8.0f  identical to plain: %s" % (tag, np.round(ms, 1), B / (np.mean(ms) * 1e-3), same), flush=True)
    bt.close(); h.close()
