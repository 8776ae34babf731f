"""Developer script: the bench's overlapped stream of 4096-batches in the reference order, in a given launch shape
(DFTPAV_REF_SHAPE=wave|quad) and depth.  scripts/ref_stream_time.py [depth ...]"""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("GPU_MAX_HW_QUEUES", "16")
import torch
from dftpav_amd import capi
from benchlib.common import Ctx
from benchlib.stream import Stream

ctx = Ctx("overlap", 0, 1, 0, False, capi.default_params(), n_cu=torch.cuda.get_device_properties(0).multi_processor_count)
for depth in [int(a) for a in sys.argv[1:]] or [4]:
    st = Stream(ctx, 4096, int(os.environ.get('CFG', 3)), 20240, depth=depth, order=capi.ORDER_REFERENCE)
    res = st.run(int(os.environ.get('STEPS', 3 * depth)), int(os.environ.get('WARMUP', depth)))
    print("cfg", os.environ.get("CFG", "3"), "shape", os.environ.get("DFTPAV_REF_SHAPE", "default"), "quadm off" if os.environ.get("DFTPAV_REF_QUADM_OFF") else "", "depth", depth, "solves/s", round(res["value"]), "ms per step", round(res["ms_per_step"], 1),
          "to result ms", round(res["to_result_ms"]), flush=True)
    st.close()
