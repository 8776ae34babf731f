"""TEST INFRASTRUCTURE: bench.main() with tests/device_standin.py where the device library stands and gloo where RCCL stands.
Run by tests/test_bench_preflight.py (alone, or under torch.distributed.run).  The line it prints is NOT a measurement: it
carries "data": "STAND-IN" so that it cannot be mistaken for one."""
import json
import os
import sys
import types

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

import bench  # noqa: E402
import device_standin as standin  # noqa: E402

# the device: tensors on the host, no streams to wait for
import benchlib.common as _common  # noqa: E402
_common.DEV = bench.DEV = "cpu"
torch.cuda.set_device = lambda d: None
torch.cuda.synchronize = lambda *a: None
torch.cuda.get_device_properties = lambda d: types.SimpleNamespace(multi_processor_count=256)
# the library
for name in ("Handle", "Batch"):
    setattr(bench.capi, name, getattr(standin, name))
if os.environ.get("STANDIN_COMM", "standin") == "standin":
    bench.dd.RcclComm = standin.Comm     # else: the real RcclComm decides, collectively, that it cannot be set up here
# the process group
_init = dist.init_process_group


def init_gloo(backend, rank, world_size, device_id=None, **kw):
    assert backend == "nccl" and device_id is not None      # what bench.py asks for on the GPU box
    return _init("gloo", rank=rank, world_size=world_size, **kw)


dist.init_process_group = init_gloo
# side runs at sizes a CPU finishes in seconds
_side_batch = bench.side_batch
bench.side_batch = lambda ctx, args, po, cores, cfg, B, reps, n_check: _side_batch(ctx, args, po, cores, cfg, min(B, 8), reps, min(n_check, 2))
_refb = bench.side_reference_order_batch
bench.side_reference_order_batch = lambda ctx, args, po, cores, cfg, B, golden: _refb(ctx, args, po, cores, cfg, min(B, 4), golden)
import benchlib.counters as _counters  # noqa: E402  (hbm_traffic looks live_pmc up in its own module)
_counters.live_pmc = bench.live_pmc = lambda args, schedule: (None, "stand-in: no counters")

if os.environ.get("STANDIN_TIMES") == "1":   # where the time of a full run goes
    import time

    def timed(name):
        fn = getattr(bench, name)

        def w(*a, **k):
            t0 = time.perf_counter()
            r = fn(*a, **k)
            print("[standin] %-40s %7.1f s" % (name, time.perf_counter() - t0), file=sys.stderr, flush=True)
            return r
        setattr(bench, name, w)
    for n_ in ("run_strong_shard", "side_isolated", "side_batch", "side_single", "side_reference_order_batch", "side_reference_order_other_configs",
               "side_neighbours", "cpu_baseline", "parity_device_order", "parity_reference_order", "parity_literal", "parity_lockstep", "with_upload"):
        timed(n_)

_dumps = json.dumps


def dumps(o, **kw):
    if isinstance(o, dict) and "metric" in o:
        o = dict(o, data="STAND-IN", calls=[list(c) for c in standin.CALLS if c[0] != "handle_close"][:4096])
    return _dumps(o, **kw)


bench.json = types.SimpleNamespace(dumps=dumps, load=json.load, loads=json.loads)
bench.main()
