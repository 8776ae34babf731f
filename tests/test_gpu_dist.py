"""The N > 1 path of SURVEY §8(e) with the PRODUCT in it, on the one GPU a test box has:
  * RCCL (backend "nccl") at world size 1: init, the all-gather of device-packed records, barrier — the calls the bench makes;
  * two processes, gloo between them, each solving ITS shard with the HIP solver on cuda:0 and packing its records on the
    device (dftpav_batch_pack_results): the gathered records equal those of one process solving the whole batch;
  * bench.py end to end with DFTPAV_BENCH_FORCE_DIST=1.
(Two RCCL ranks cannot share one device; the 8-GPU run is the driver's.)"""
import json
import os
import socket
import subprocess
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from dftpav_amd import distributed as dd

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    sock = socket.socket()
    sock.bind(("127.0.0.1", 0))
    port = sock.getsockname()[1]
    sock.close()
    return port


def _solve_shard(B, lo, hi):
    from dftpav_amd import capi, scenarios as sc
    p = capi.default_params()
    s = sc.baseline_config(3, B=B)
    s.apply_resolution(p)
    sub = s.subset(np.arange(lo, hi))
    h = capi.Handle(p, device=0)
    bt = capi.Batch(h, sub.layout, sub.B)
    bt.upload(sub)
    bt.solve_async()
    rec = torch.zeros((sub.B, dd.RECORD_BYTES), dtype=torch.uint8, device="cuda:0")
    bt.pack_results(rec.data_ptr())
    bt.sync()
    r = bt.results()
    bt.close()
    h.close()
    return rec, r


def test_rccl_allgather_of_device_packed_records(hiplib):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(_free_port())
    torch.cuda.set_device(0)
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
    try:
        B = 6
        rec, r = _solve_shard(B, 0, B)
        allrec = dd.allgather_records(rec, B)
        dist.barrier()
        torch.cuda.synchronize()
        c, st, it = dd.unpack_records(allrec.cpu().numpy())
        assert np.array_equal(c, r["final_cost"]) and np.array_equal(st, r["status"]) and np.array_equal(it, r["iters"])
        t = torch.tensor([1.5], dtype=torch.float64, device="cuda")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        assert float(t.item()) == 1.5
    finally:
        dist.destroy_process_group()


def _worker(rank, world, port, B, q):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    lo, hi = dd.shard_range(B, rank, world)
    rec, _ = _solve_shard(B, lo, hi)                   # the product: HIP solve + device-side record packing
    allrec = dd.allgather_records(rec.cpu(), B)        # gloo carries the bytes
    q.put((rank, allrec.numpy().copy()))
    dist.barrier()
    dist.destroy_process_group()


def test_two_ranks_with_the_hip_solver_equal_one_process(hiplib):
    B, world = 7, 2  # uneven shards: 3 + 4
    port = _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, world, port, B, q)) for r in range(world)]
    for pr in procs:
        pr.start()
    got = dict(q.get(timeout=300) for _ in range(world))
    for pr in procs:
        pr.join(timeout=60)
        assert pr.exitcode == 0
    assert np.array_equal(got[0], got[1])
    _, ref = _solve_shard(B, 0, B)
    c, st, it = dd.unpack_records(got[0])
    assert np.array_equal(c, ref["final_cost"]) and np.array_equal(st, ref["status"]) and np.array_equal(it, ref["iters"])


def test_bench_takes_the_rccl_path_when_forced(hiplib):
    env = dict(os.environ, DFTPAV_BENCH_FORCE_DIST="1", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(_free_port()), RANK="0",
               WORLD_SIZE="1", LOCAL_RANK="0")
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--steps", "2", "--warmup", "1",
                          "--batch-per-gpu", "1024", "--no-extras", "--cpu-sample", "0"], env=env, capture_output=True, text=True,
                         timeout=600)
    assert out.returncode == 0, out.stderr[-2000:]
    line = [l for l in out.stdout.splitlines() if l.startswith("{")][-1]
    d = json.loads(line)
    assert d["n_gpus"] == 1 and d["value"] > 0 and d["config"]["global_batch"] == 1024


def test_bench_side_run_of_the_other_scaling_mode(hiplib):
    """At N > 1 the bench times the other scaling mode after the value line, several steps deep (one RCCL communicator per
    handle) and under a watchdog.  Forced here on one GPU: four steps in flight in the one-wave shape, and a time limit too
    short for the side run, which must cost its own entry and not the line."""
    base = dict(os.environ, DFTPAV_BENCH_FORCE_DIST="1", DFTPAV_BENCH_FORCE_OTHER="1", MASTER_ADDR="127.0.0.1", RANK="0",
                WORLD_SIZE="1", LOCAL_RANK="0")
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--steps", "4", "--warmup", "1", "--batch-per-gpu", "512",
           "--no-extras", "--cpu-sample", "0"]
    out = subprocess.run(cmd, env=dict(base, MASTER_PORT=str(_free_port()), DFTPAV_BENCH_DEPTH="4,2"), capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stderr[-2000:]
    d = json.loads([l for l in out.stdout.splitlines() if l.startswith("{")][-1])
    o = d["other_scaling"]
    assert "error" not in o and o["value"] > 0 and o["steps_in_flight"] == 4 and o["steps"] >= 8
    assert "ncclAllGather behind the C-ABI" in json.dumps(d)
    out = subprocess.run(cmd, env=dict(base, MASTER_PORT=str(_free_port()), DFTPAV_BENCH_SIDE_LIMIT_S="0.05"), capture_output=True, text=True,
                         timeout=600)
    assert out.returncode == 0, out.stderr[-2000:]
    d = json.loads([l for l in out.stdout.splitlines() if l.startswith("{")][-1])
    assert d["value"] > 0 and "time limit" in d["other_scaling"]["error"]


def test_c_abi_communicator_gathers_the_records(hiplib):
    """dftpav_comm_create / dftpav_batch_allgather_results: RCCL behind the C-ABI, no torch.distributed in the data path
    (world size 1 on this box: init, pack, ncclAllGather on the handle's stream, layout with its zero pad)."""
    from dftpav_amd import capi, scenarios as sc
    p = capi.default_params()
    B = 6
    s = sc.baseline_config(3, B=B)
    s.apply_resolution(p)
    h = capi.Handle(p, device=0)
    comm = dd.RcclComm(h)                       # no process group: world size 1, the id never leaves the process
    assert comm.world == 1 and capi.comm_layout(7, 2, 0) == (0, 3, 4) and capi.comm_layout(7, 2, 1) == (3, 4, 4)
    bt = capi.Batch(h, s.layout, B)
    bt.upload(s)
    bt.solve_async()
    rec = comm.allgather(bt, B)
    r = bt.results()
    c, st, it = dd.unpack_records(rec.cpu().numpy())
    assert np.array_equal(c, r["final_cost"]) and np.array_equal(st, r["status"]) and np.array_equal(it, r["iters"])
    with pytest.raises(capi.DftpavError):       # a batch that is not the rank's shard is refused
        bt.allgather_results(B + 1, comm._recv.data_ptr())
    comm.close()
    bt.close()
    h.close()


def test_cpp_host_runs_the_collective_through_the_c_abi(hiplib):
    """host_example --ranks 1: a C++ host, one thread per rank, sharding + solve + the RCCL all-gather with nothing but
    include/dftpav_hip.h"""
    exe = os.path.join(ROOT, "dftpav_amd", "csrc", "host", "host_example")
    if not os.path.exists(exe):
        subprocess.check_call(["make", "-C", os.path.dirname(exe), "-s"])
    out = subprocess.run([exe, "--ranks", "1"], capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stdout + out.stderr
    assert "identical on every rank" in out.stdout


@pytest.mark.parametrize("mode", ["device", "device_sliced", "reference", "reference_wave"])
def test_records_are_written_by_the_solve_kernels(hiplib, monkeypatch, mode):
    """The 16-byte records of the all-gather are written by the solver's epilogue when a trajectory finishes (no packing kernel):
    dftpav_batch_records / dftpav_batch_pack_results return {final cost, status, iterations} of dftpav_batch_results for every
    launch shape -- plain and time-sliced device order (the straggler launch writes its own), both shapes of the reference order --
    and a second solve of the same batch overwrites them."""
    from dftpav_amd import capi, scenarios as sc
    if mode == "device_sliced":
        monkeypatch.setenv("DFTPAV_SCHED", "1")
        monkeypatch.setenv("DFTPAV_SLOTS", "3")
        monkeypatch.setenv("DFTPAV_SLICE", "7")
        monkeypatch.setenv("DFTPAV_HANDOVER", "2")
    if mode == "reference_wave":
        monkeypatch.setenv("DFTPAV_REF_SHAPE", "wave")
        monkeypatch.setenv("DFTPAV_REF_SLOTS", "1")
        monkeypatch.setenv("DFTPAV_REF_SLICE", "9")
    p = capi.default_params()
    s = sc.baseline_config(1, B=20)
    s.apply_resolution(p)
    h = capi.Handle(p, device=0)
    bt = capi.Batch(h, s.layout, s.B)
    bt.upload(s)
    if mode.startswith("reference"):
        bt.set_order(capi.ORDER_REFERENCE)
    for rep in range(2):
        bt.solve_async()
        rec_dev = torch.zeros((s.B, dd.RECORD_BYTES), dtype=torch.uint8, device="cuda:0")
        bt.pack_results(rec_dev.data_ptr())
        rec_host = bt.records()
        r = bt.results()
        for rec in (rec_host, rec_dev.cpu().numpy()):
            cost, status, iters = dd.unpack_records(rec)
            assert np.array_equal(cost, r["final_cost"]) and np.array_equal(status, r["status"]) and np.array_equal(iters, r["iters"]), (mode, rep)
    bt.close()
    h.close()


def test_one_communicator_shared_by_several_handles(hiplib):
    """dftpav_comm_share at world size 1: three handles (= HIP streams) of one process, ONE communicator -- the owner's -- and an
    all-gather of the epilogue-written records on each of the three streams, round-robin, twice; the handle that created the
    communicator is destroyed FIRST (the communicator is reference-counted).  (What bench.py does at N > 1 instead of a communicator per handle.)"""
    from dftpav_amd import capi, scenarios as sc
    p = capi.default_params()
    B = 9
    s = sc.baseline_config(1, B=B)
    s.apply_resolution(p)
    hs = [capi.Handle(p, device=0) for _ in range(3)]
    owner = dd.RcclComm(hs[0])
    comms = [owner] + [dd.RcclComm(h_, share=owner) for h_ in hs[1:]]
    bts = []
    for k, h_ in enumerate(hs):
        sub = s.subset(np.arange(B))
        sub.inner_pts = np.ascontiguousarray(sub.inner_pts).copy()
        sub.inner_pts[:, 1] += 0.01 * k          # three different batches
        bt = capi.Batch(h_, sub.layout, B)
        bt.upload(sub)
        bts.append(bt)
    for rep in range(2):
        for bt in bts:
            bt.solve_async()
        for c_, bt in zip(comms, bts):
            allrec = c_.allgather(bt, B)
            cost, status, iters = dd.unpack_records(allrec.cpu().numpy())
            r = bt.results()
            assert np.array_equal(cost, r["final_cost"]) and np.array_equal(status, r["status"]) and np.array_equal(iters, r["iters"])
    for bt in bts:
        bt.close()
    # the communicator belongs to its holders together: it lives until the last of them lets go, in whatever order (here the
    # handle that created it goes first, and a handle shares from a handle that itself only shares)
    extra = capi.Handle(p, device=0)
    extra.comm_share(hs[2])
    comms[0].close()
    hs[0].close()
    for c_ in comms[1:]:
        c_.close()
    extra.comm_destroy()
    extra.close()
    for h_ in hs[1:]:
        h_.close()
