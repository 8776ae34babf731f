"""N>1 path on CPU: world_size-2 gloo, contiguous sharding + the single all-gather of records."""
import os
import socket
import sys

import numpy as np
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from dftpav_amd import distributed as dd


def test_shard_ranges_partition_the_batch():
    for B in (1, 2, 7, 256, 4096):
        for G in (1, 2, 3, 8):
            r = [dd.shard_range(B, k, G) for k in range(G)]
            assert r[0][0] == 0 and r[-1][1] == B
            assert all(r[k][1] == r[k + 1][0] for k in range(G - 1))
            sizes = [hi - lo for lo, hi in r]
            assert max(sizes) - min(sizes) <= 1


def test_record_roundtrip():
    rng = np.random.default_rng(0)
    c = rng.normal(size=37)
    s = rng.integers(-1024, 3, 37).astype(np.int32)
    it = rng.integers(0, 12000, 37).astype(np.int32)
    c2, s2, i2 = dd.unpack_records(dd.pack_records(c, s, it))
    assert np.array_equal(c, c2) and np.array_equal(s, s2) and np.array_equal(it, i2)
    assert dd.best_of(np.array([5.0, 1.0, 3.0]), np.array([1, -1007, 0])) == 2


def _worker(rank, world, port, B, q):
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, root)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from dftpav_amd import scenarios as sc
    from oracle import pyoracle as po
    p = po.default_params()
    s = sc.baseline_config(1, B=B)
    s.apply_resolution(p)
    lo, hi = dd.shard_range(B, rank, world)
    r = po.solve_batch(p, s.subset(np.arange(lo, hi)), nthreads=1, order=1)  # stand-in for the GPU shard solve
    rec = torch.from_numpy(dd.pack_records(r["final_cost"], r["status"], r["iters"]))
    allrec = dd.allgather_records(rec, B)
    q.put((rank, allrec.numpy().copy()))
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_allgather_equals_single_process():
    from dftpav_amd import scenarios as sc
    from oracle import pyoracle as po
    B, world = 5, 2  # uneven shards: 2 + 3
    sock = socket.socket()
    sock.bind(("127.0.0.1", 0))
    port = sock.getsockname()[1]
    sock.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, world, port, B, q)) for r in range(world)]
    for pr in procs:
        pr.start()
    got = dict(q.get(timeout=240) for _ in range(world))
    for pr in procs:
        pr.join(timeout=60)
        assert pr.exitcode == 0
    assert np.array_equal(got[0], got[1])
    p = po.default_params()
    s = sc.baseline_config(1, B=B)
    s.apply_resolution(p)
    ref = po.solve_batch(p, s, nthreads=1, order=1)
    c, st, it = dd.unpack_records(got[0])
    assert np.array_equal(c, ref["final_cost"]) and np.array_equal(st, ref["status"]) and np.array_equal(it, ref["iters"])


def test_cpp_host_shard_layout_and_record_placement_without_a_device():
    """dftpav_comm_layout + the placement of the 16-byte records in the gathered buffer, through the C++ host
    (dftpav_amd/csrc/host/host_example --placement N B: every rank's send block built as dftpav_batch_allgather_results builds
    it, the collective replaced by a plain copy) for even and uneven shards: the pre-flight of the first real N > 1 run."""
    import os
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    host = os.path.join(root, "dftpav_amd", "csrc", "host")
    subprocess.check_call(["make", "-C", host, "-s"])
    for nranks, B in [(2, 7), (2, 8), (8, 4096), (8, 4099), (8, 4103), (4, 3), (3, 1), (16, 512), (7, 1000)]:
        out = subprocess.run([os.path.join(host, "host_example"), "--placement", str(nranks), str(B)], capture_output=True, text=True, timeout=60)
        assert out.returncode == 0 and "every record in its place" in out.stdout, (nranks, B, out.stdout, out.stderr)
