"""Pre-flight of bench.py's HOST LOGIC without a GPU: bench.main() runs with tests/device_standin.py where the device library
stands (the oracle computes the solves) and gloo where RCCL stands, launched exactly as the driver launches the N > 1 bench
(python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P ... --gpus N).  What
this pins: rank arithmetic and shard ranges, the order of the collectives on every rank (a mismatch hangs, the timeout fails the
test), the collective decisions (communicator or torch path, the same on every rank), the one JSON line of the contract on rank
0 only, the other scaling mode behind it, every side run's bookkeeping.  What it cannot pin: anything about the device.  The
GPU twin of these tests is tests/test_gpu_dist.py (world size 1, forced)."""
import json
import os
import socket
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
MAIN = os.path.join(ROOT, "tests", "bench_standin_main.py")
CONTRACT = ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype",
            "data", "config", "roofline")


def _port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _bench(flags, nproc=1, env=None, timeout=600):
    e = dict(os.environ, OMP_NUM_THREADS="1")
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_PORT"):
        e.pop(k, None)
    e.update(env or {})
    if nproc == 1:
        cmd = [sys.executable, MAIN, "--gpus", "1"] + flags
    else:
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(nproc), "--master-addr", "127.0.0.1",
               "--master-port", str(_port()), MAIN, "--gpus", str(nproc)] + flags
    p = subprocess.run(cmd, cwd=ROOT, env=e, capture_output=True, text=True, timeout=timeout)
    assert p.returncode == 0, p.stderr[-3000:]
    lines = [ln for ln in p.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, "rank 0 prints ONE line, the other ranks none: %d" % len(lines)
    out = json.loads(lines[0])
    assert out["data"] == "STAND-IN"      # never a measurement
    for k in CONTRACT:
        assert k in out, k
    assert out["roofline"]["bound"] == "hbm" and out["roofline"]["unit"] == "GB/s" and out["unit"] == "solves/s"
    assert abs(out["roofline"]["frac"] - out["roofline"]["achieved"] / out["roofline"]["peak"]) < 1e-12
    # value = the trajectories ALL ranks solved in the timed steps / the (max over ranks) time of those steps
    assert abs(out["value"] - out["config"]["global_batch"] * out["steps"] / (out["ms_per_step"] * out["steps"] * 1e-3)) <= 1e-6 * out["value"]
    return out


def _count(out, what, arg=None):
    return sum(1 for c in out["calls"] if c[0] == what and (arg is None or c[1] == arg))


def test_one_rank_line_and_step_count():
    out = _bench(["--steps", "3", "--warmup", "2", "--batch-per-gpu", "6", "--no-extras", "--cpu-sample", "0", "--order", "device"])
    assert out["order"].startswith("device") and out["roofline"]["kernel"] == "solver_kernel"
    out = _bench(["--steps", "3", "--warmup", "2", "--batch-per-gpu", "6", "--no-extras", "--cpu-sample", "0"])
    assert out["order"].startswith("reference")
    assert out["n_gpus"] == 1 and out["steps"] == 3 and out["warmup"] == 2 and out["scaling"] == "weak"
    assert out["config"]["global_batch"] == 6 and out["config"]["allgather_via"].startswith("none")
    assert out["config"]["steps_in_flight"] == 4                                # --depth's default
    assert _count(out, "solve", 6) == 5 and _count(out, "allgather") == 0     # warm-up + timed steps, one solve each; no collective
    assert "other_scaling" not in out and "cpu_baseline" not in out


@pytest.mark.parametrize("nproc", [2, 4])
def test_ranks_weak_value_line_and_the_strong_side_run(nproc):
    B = 8
    out = _bench(["--steps", "3", "--warmup", "1", "--batch-per-gpu", str(B), "--depth", "2", "--no-extras", "--cpu-sample", "0"], nproc=nproc)
    assert out["n_gpus"] == nproc and out["scaling"] == "weak" and out["config"]["global_batch"] == nproc * B
    assert out["config"]["steps_in_flight"] == 2
    assert "dftpav_batch_allgather_results" in out["config"]["allgather_via"]
    o = out["other_scaling"]
    assert o["scaling"] == "strong" and o["global_batch"] == B and o["per_gpu"] == B // nproc
    depth = min(8, max(2, min(16, 2 * B // (B // nproc))))
    assert o["steps_in_flight"] == depth and o["steps"] == max(3, 2 * depth)
    # rank 0's calls: one communicator per stream of cycles (the other handles borrow it); one all-gather per step delivered
    assert _count(out, "comm_create", nproc) == 2
    assert _count(out, "solve", B) == 4 and _count(out, "allgather", nproc * B) == 4
    assert _count(out, "solve", B // nproc) == depth + o["steps"] and _count(out, "allgather", B) == depth + o["steps"]


def test_two_ranks_strong_with_shards_of_unequal_size():
    out = _bench(["--steps", "2", "--warmup", "1", "--batch-per-gpu", "9", "--scaling", "strong", "--no-extras", "--cpu-sample", "0"], nproc=2)
    assert out["scaling"] == "strong" and out["config"]["global_batch"] == 9      # rank 0 owns 4, rank 1 owns 5
    assert _count(out, "solve", 4) >= 3 and _count(out, "allgather", 9) >= 3
    assert out["other_scaling"]["scaling"] == "weak" and out["other_scaling"]["global_batch"] == 18 and out["other_scaling"]["per_gpu"] == 9


def test_a_rank_without_communicator_moves_every_rank_to_the_torch_path():
    """dftpav_comm_create fails on rank 1 only: dd.RcclComm (the real one) must raise on BOTH ranks, bench.py then takes
    torch.distributed's all_gather_into_tensor on both -- a rank left on the other path would hang the job"""
    out = _bench(["--steps", "2", "--warmup", "1", "--batch-per-gpu", "4", "--no-extras", "--cpu-sample", "0"], nproc=2,
                 env={"STANDIN_COMM": "fail_rank1"})
    via = out["config"]["allgather_via"]
    assert via.startswith("torch.distributed all_gather_into_tensor") and "failed on another rank" in via
    assert _count(out, "allgather") == 0 and "value" in out["other_scaling"]


def test_torch_path_on_request_and_the_other_schedules():
    out = _bench(["--steps", "2", "--warmup", "1", "--batch-per-gpu", "4", "--no-extras", "--cpu-sample", "0"], nproc=2,
                 env={"DFTPAV_BENCH_COMM": "torch"})
    assert out["config"]["allgather_via"] == "torch.distributed all_gather_into_tensor (RCCL)"
    for schedule in ("chain", "plain"):
        out = _bench(["--steps", "3", "--warmup", "1", "--batch-per-gpu", "4", "--no-extras", "--cpu-sample", "0", "--schedule", schedule], nproc=2)
        assert out["schedule"].startswith(schedule) and _count(out, "solve", 4) == 4 and _count(out, "allgather", 8) == 4


def test_world_size_one_through_the_collective_path():
    """what tests/test_gpu_dist.py runs on the GPU box (DFTPAV_BENCH_FORCE_DIST / _FORCE_OTHER)"""
    out = _bench(["--steps", "2", "--warmup", "1", "--batch-per-gpu", "8", "--no-extras", "--cpu-sample", "0"],
                 env={"DFTPAV_BENCH_FORCE_DIST": "1", "DFTPAV_BENCH_FORCE_OTHER": "1", "MASTER_PORT": str(_port())})
    assert out["n_gpus"] == 1 and out["other_scaling"]["scaling"] == "strong" and _count(out, "allgather", 8) >= 3


def test_every_side_run_of_the_one_gpu_line():
    """the default N = 1 line with all its side runs, at sizes the oracle finishes in half a minute: no side run fails, every
    entry the documents quote is there"""
    out = _bench(["--steps", "2", "--warmup", "1", "--batch-per-gpu", "16", "--cpu-sample", "16"], timeout=900)
    assert "side_run_errors" not in out, out.get("side_run_errors")
    for k in ("strong_shard", "isolated", "batch256", "single", "moving_obstacles_1024", "validate", "readout", "shots", "corridor",
              "cpu_baseline", "with_upload", "parity", "device_order", "order"):
        assert k in out, k
    assert out["order"].startswith("reference") and "ref4_kernel" in out["roofline"]["kernel"]       # the parity-grade order is the value line
    do = out["device_order"]
    assert do["solves_per_s"] > 0 and "depth_2" in do and "depth_4" in do and do["steps_in_flight"] == 4 and "_results" not in do
    cb = out["cpu_baseline"]
    assert cb["kind"] == "port" and cb["unit"] == "solves/s" and cb["value"] > 0 and cb["cores"] == 1 and "sample" in cb   # the reference cannot be built here
    par = out["parity"]
    for k in ("reference_order", "reference_order_other_configs", "bias", "literal", "lockstep"):
        assert k in par, k
    assert par["reference_order"]["bit_equal"] == par["reference_order"]["trajectories"] == 16
    assert par["reference_order"]["frac_within_1e-5_of_cpu"] == 1.0 and par["reference_order"]["solves_per_s"] == out["value"]
    assert par["reference_order"]["first_batch_equals_the_isolated_solve"] is True
    assert par["lockstep"] == {"failed": par["lockstep"]["failed"]}      # the stand-in keeps no trace: reported, not fatal
    assert out["moving_obstacles_1024"]["reference_order"]["batch"] == 4
    # (the stand-in build of the reference's sources is retired, oracle/pyref.py: its entries stay empty)
    assert out["single"]["reference_order"]["bit_equal_to_the_retired_standin_build_on_a_correctly_rounded_libm"] is None
    assert out["single"]["reference_order"]["bit_equal_to_the_reference_program_with_correctly_rounded_cos_sin"] == 9
    assert "standin_build_of_the_reference_sources" not in out["cpu_baseline"]
    for k in ("moving_obstacles_1024", ):
        assert out[k]["reference_order"]["bit_equal_to_the_reference_program_with_correctly_rounded_libm_calls_on_4_sampled"] is True
    ro = out["gear_shift_4096_reference_order"]
    assert ro["batch"] == 4 and ro["bit_equal_to_the_reference_program_with_correctly_rounded_libm_calls_on_4_sampled"] is True and ro["roofline"]["frac"] > 0
    assert ro["overlapped"]["first_batch_equals_the_isolated_solve"] is True and ro["overlapped"]["steps_in_flight"] == 4   # the gear shift as a stream, like the value line
    assert out["single"]["reference_order"]["best_of_64_restarts_in_one_launch"]["slot0_bit_equal_to_the_lone_solve_on_all"] is True
    live = par["reference_order_other_configs"]["gear_shifts_with_moving_obstacles"]
    assert live["bit_equal_to_the_reference_program_with_correctly_rounded_libm_calls"] == live["trajectories"] == 8
    assert "against_retired_standin_build" not in live
    assert out["strong_shard"]["per_gpu"] == 2 and out["strong_shard"]["steps_in_flight"] == 16


def test_a_side_run_over_its_time_limit_costs_its_entry_not_the_line():
    """the watchdog of the other scaling mode (what tests/test_gpu_dist.py checks on the device)"""
    out = _bench(["--steps", "4", "--warmup", "1", "--batch-per-gpu", "16", "--no-extras", "--cpu-sample", "0"],
                 env={"DFTPAV_BENCH_FORCE_DIST": "1", "DFTPAV_BENCH_FORCE_OTHER": "1", "MASTER_PORT": str(_port()), "DFTPAV_BENCH_SIDE_LIMIT_S": "0.05"})
    assert out["value"] > 0 and "time limit" in out["other_scaling"]["error"]
    out = _bench(["--steps", "4", "--warmup", "1", "--batch-per-gpu", "16", "--no-extras", "--cpu-sample", "0"],
                 env={"DFTPAV_BENCH_FORCE_DIST": "1", "DFTPAV_BENCH_FORCE_OTHER": "1", "MASTER_PORT": str(_port()), "DFTPAV_BENCH_DEPTH": "4,2"})
    assert out["other_scaling"]["steps_in_flight"] == 4 and out["other_scaling"]["steps"] >= 8
