"""Developer script: the stream of 4096-batches in the reference order WITH the C-ABI communicator at world size 1 (the delivery of a step is
dftpav_batch_allgather_results: at one rank RCCL's all-gather is a device-to-device copy on the solve's stream) against the same stream
without it -- does a collective's kernel queue behind the persistent waves?   CFG=3 scripts/stream_comm_time.py [depth ...]"""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("GPU_MAX_HW_QUEUES", "16")
os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
os.environ.setdefault("MASTER_PORT", "29533")
import torch
import torch.distributed as dist
from dftpav_amd import capi
from benchlib.common import Ctx
from benchlib.stream import Stream

dist.init_process_group("nccl", rank=0, world_size=1)
torch.cuda.set_device(0)
cfg = int(os.environ.get("CFG", 3))
n_cu = torch.cuda.get_device_properties(0).multi_processor_count
for depth in [int(a) for a in sys.argv[1:]] or [4]:
    for distributed in (False, True):
        ctx = Ctx("overlap", 0, 1, 0, distributed, capi.default_params(), n_cu=n_cu)
        st = Stream(ctx, 4096, cfg, 20240, depth=depth, order=capi.ORDER_REFERENCE)
        res = st.run(3 * depth, depth)
        print("cfg", cfg, "depth", depth, "communicator" if distributed else "no communicator", "| solves/s", round(res["value"]), "ms per step", round(res["ms_per_step"], 1),
              "to result ms", round(res["to_result_ms"]), "| delivery:", st.via, flush=True)
        st.close()
dist.destroy_process_group()
