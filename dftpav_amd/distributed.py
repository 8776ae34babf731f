"""Multi-GPU layer of the solve path.

Trajectories are independent, so the batch is the only shard axis: rank r of G
owns the contiguous block [r*B/G, (r+1)*B/G), solves it with no communication,
and ONE all-gather of 16-byte records {f64 final_cost, i32 status, i32 iters}
makes every rank see every result (tens of KB: latency-bound, so a single small
collective over xGMI, not a ring of large chunks).  Backend "nccl" is RCCL on
ROCm; "gloo" is used by the CPU tests.
"""
import numpy as np
import torch
import torch.distributed as dist

RECORD_BYTES = 16


def shard_range(B, rank, world):
    """Contiguous block of rank `rank`: [lo, hi)."""
    lo = (B * rank) // world
    hi = (B * (rank + 1)) // world
    return lo, hi


def pack_records(final_cost, status, iters):
    """Host-side twin of dftpav_batch_pack_results: uint8 [n][16]."""
    n = len(final_cost)
    rec = np.zeros((n, RECORD_BYTES), dtype=np.uint8)
    rec[:, :8] = np.ascontiguousarray(final_cost, dtype=np.float64).view(np.uint8).reshape(n, 8)
    rec[:, 8:12] = np.ascontiguousarray(status, dtype=np.int32).view(np.uint8).reshape(n, 4)
    rec[:, 12:16] = np.ascontiguousarray(iters, dtype=np.int32).view(np.uint8).reshape(n, 4)
    return rec


def unpack_records(rec):
    rec = np.ascontiguousarray(rec, dtype=np.uint8).reshape(-1, RECORD_BYTES)
    cost = rec[:, :8].copy().view(np.float64).reshape(-1)
    status = rec[:, 8:12].copy().view(np.int32).reshape(-1)
    iters = rec[:, 12:16].copy().view(np.int32).reshape(-1)
    return cost, status, iters


def allgather_records(local_rec, B, group=None):
    """One all-gather of the per-rank record blocks.

    local_rec: uint8 tensor [n_local][16] on the device of the backend.  Shards may
    differ by one trajectory when B % world != 0, so blocks are padded to the
    largest shard.  Returns a uint8 tensor [B][16] identical on every rank."""
    world = dist.get_world_size(group)
    rank = dist.get_rank(group)
    sizes = [shard_range(B, r, world)[1] - shard_range(B, r, world)[0] for r in range(world)]
    nmax = max(sizes)
    assert local_rec.shape[0] == sizes[rank]
    send = torch.zeros((nmax, RECORD_BYTES), dtype=torch.uint8, device=local_rec.device)
    send[:sizes[rank]] = local_rec
    recv = torch.empty((world * nmax, RECORD_BYTES), dtype=torch.uint8, device=local_rec.device)
    dist.all_gather_into_tensor(recv, send, group=group)
    recv = recv.view(world, nmax, RECORD_BYTES)
    return torch.cat([recv[r, :sizes[r]] for r in range(world)], dim=0)


class RcclComm:
    """The communicator of the C-ABI (dftpav_comm_create) for one handle, set up from a torch.distributed job: rank 0 makes the
    128-byte id through the library, torch.distributed only carries those bytes to the other ranks; the all-gather itself
    is the library's (ncclAllGather on the handle's stream), so a C++ host without PyTorch runs the identical path
    (dftpav_amd/csrc/host/host_example.cpp --ranks N)."""

    def __init__(self, handle, group=None, share=None):
        """share: an RcclComm of this process whose communicator this handle borrows (dftpav_comm_share) -- a host with k batches
        in flight on k handles sets up ONE communicator per rank; no collective call is made here then."""
        from . import capi
        self.handle = handle
        if share is not None:
            self.world, self.rank = share.world, share.rank
            handle.comm_share(share.handle)
            self._recv = None
            return
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        self.rank = dist.get_rank(group) if dist.is_initialized() else 0
        # Every decision on the way is taken by ALL ranks together: a rank that raised on its own would leave the others inside a
        # broadcast or inside ncclCommInitRank.  (1) can every rank reach RCCL through the library at all?  (2) the id, with a
        # status byte behind it;  (3) did every rank's dftpav_comm_create succeed?
        dev = "cpu"
        if self.world > 1:
            dev = "cuda" if dist.get_backend(group) == "nccl" else "cpu"
        uid, ok = np.zeros(128, dtype=np.uint8), 1
        try:
            # every rank probes that RCCL loads behind the C-ABI (dlopen only); rank 0 alone makes an id -- ncclGetUniqueId starts
            # a bootstrap root (a listening socket and a thread) that only the id's maker should own
            if not capi.comm_available():
                raise RuntimeError("RCCL (librccl.so.1) is not loadable")
            if self.rank == 0:
                uid = capi.comm_unique_id()
        except Exception as ex:  # noqa: BLE001
            ok, self._err = 0, str(ex)
        if self.world > 1:
            t = torch.from_numpy(np.concatenate([uid, np.array([ok], dtype=np.uint8)])).to(dev)
            dist.broadcast(t, src=0, group=group)
            got = t.cpu().numpy()
            uid, ok0 = got[:128].copy(), int(got[128])
            flag = torch.tensor([min(ok, ok0)], dtype=torch.int32, device=dev)
            dist.all_reduce(flag, op=dist.ReduceOp.MIN, group=group)
            ok = int(flag.item())
        if not ok:
            raise RuntimeError("RCCL behind the C-ABI is not available on every rank: %s" % getattr(self, "_err", "another rank failed"))
        created = 1
        try:
            handle.comm_create(self.world, self.rank, uid)
        except Exception as ex:  # noqa: BLE001
            created, self._err = 0, str(ex)
        if self.world > 1:
            flag = torch.tensor([created], dtype=torch.int32, device=dev)
            dist.all_reduce(flag, op=dist.ReduceOp.MIN, group=group)
            if int(flag.item()) == 0 and created:
                handle.comm_destroy()
                created, self._err = 0, "dftpav_comm_create failed on another rank"
        if not created:
            raise RuntimeError("dftpav_comm_create: %s" % self._err)
        self._recv = None

    def allgather(self, batch, B):
        """-> uint8 tensor [B][16] on the device, identical on every rank, valid once the handle's stream is synchronised"""
        from . import capi
        _, count, block = capi.comm_layout(B, self.world, self.rank)
        assert count == batch.B
        if self._recv is None or self._recv.shape[0] != self.world * block:
            self._recv = torch.zeros((self.world * block, RECORD_BYTES), dtype=torch.uint8, device="cuda")
        batch.allgather_results(B, self._recv.data_ptr())
        batch.sync()
        sizes = [shard_range(B, r, self.world)[1] - shard_range(B, r, self.world)[0] for r in range(self.world)]
        blocks = self._recv.view(self.world, block, RECORD_BYTES)
        return torch.cat([blocks[r, :sizes[r]] for r in range(self.world)], dim=0)

    def close(self):
        self.handle.comm_destroy()


def best_of(cost, status):
    """Host-side argmin over successful restarts (status as lbfgs.hpp:135-184;
    success rule of traj_optimizer.cpp:176-201 without the cost cap)."""
    ok = np.isin(status, (0, 1, 2, -1008, -1009))
    c = np.where(ok, cost, np.inf)
    i = int(np.argmin(c))
    return i if np.isfinite(c[i]) else -1
