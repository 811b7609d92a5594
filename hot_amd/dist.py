"""One connected body over several ranks: the host side (one process per GPU, torch.distributed).

The C library never communicates by itself (include/hot_mi355x.h, `hot_comm`): at the points where its sharded solve
needs data from other ranks it calls three collectives with raw pointers — all-reduce, all-gather and a personalised
all-to-all.  `TorchComm` implements them over a torch.distributed process group: backend "nccl" (= RCCL over xGMI on the
GPU box; device pointers are wrapped as tensors without copies) or "gloo" (CPU tests and the single-GPU test box: the
payload is staged through host tensors).  `shard_by_page_order` cuts a particle cloud into the contiguous ranges of the
global sort order that the library expects as shards.

SURVEY.md §8(e): particles are sharded, node tiles are summed with one all-reduce per scatter, matrix rows are owned by
one rank each and completed by an exchange of partial rows, and every colour of a Gauss-Seidel sweep is handed to the
other ranks before the next colour starts (the reference's update order, MultigridPreconditioner.h:266-318)."""
import ctypes as C

import numpy as np

_TORCH_DT = None


def _dtypes():
    global _TORCH_DT
    if _TORCH_DT is None:
        import torch
        _TORCH_DT = {0: torch.float32, 1: torch.float64, 2: torch.int32, 3: torch.int64}
    return _TORCH_DT


_ALLREDUCE = C.CFUNCTYPE(C.c_int32, C.c_void_p, C.c_void_p, C.c_int64, C.c_int32, C.c_int32, C.c_int32)
_ALLGATHER = C.CFUNCTYPE(C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_int32)
_ALLTOALLV = C.CFUNCTYPE(C.c_int32, C.c_void_p, C.c_void_p, C.POINTER(C.c_int64), C.POINTER(C.c_int64), C.c_void_p, C.POINTER(C.c_int64), C.POINTER(C.c_int64), C.c_int32)


class hot_comm(C.Structure):  # include/hot_mi355x.h
    _fields_ = [("rank", C.c_int32), ("size", C.c_int32), ("user", C.c_void_p), ("allreduce", _ALLREDUCE), ("allgather", _ALLGATHER), ("alltoallv", _ALLTOALLV),
                ("partition_min_rows", C.c_int32), ("stream_ordered", C.c_int32), ("reserved", C.c_int32 * 2)]


class _DevMem:
    """`nbytes` of device memory at `ptr` as an object torch.as_tensor can wrap without copying."""

    def __init__(self, ptr, nbytes):
        self.__cuda_array_interface__ = {"shape": (int(nbytes),), "typestr": "|u1", "data": (int(ptr), False), "version": 2}


class TorchComm:
    """hot_comm over torch.distributed.  Keep the object alive as long as the context uses it (it owns the callbacks)."""

    def __init__(self, group=None, device=None, partition_min_rows=0):
        import torch
        import torch.distributed as dist
        self.torch, self.dist, self.group = torch, dist, group
        self.rank, self.size = dist.get_rank(group), dist.get_world_size(group)
        self.backend = dist.get_backend(group)
        self.device = device  # torch.device of this rank's GPU, or None for a library that works on host memory (the CPU checker of the tests)
        self.calls = dict(allreduce=0, allgather=0, alltoallv=0, bytes=0)
        self._hip = None
        self._cb = (_ALLREDUCE(self._allreduce), _ALLGATHER(self._allgather), _ALLTOALLV(self._alltoallv))
        self.struct = hot_comm(self.rank, self.size, None, self._cb[0], self._cb[1], self._cb[2], int(partition_min_rows), 0, (C.c_int32 * 2)(0, 0))

    # ---- raw memory <-> tensors
    def _host(self, ptr, nbytes):
        return self.torch.frombuffer((C.c_char * nbytes).from_address(ptr), dtype=self.torch.uint8)

    def _dev(self, ptr, nbytes):
        return self.torch.as_tensor(_DevMem(ptr, nbytes), device=self.device)

    def _view(self, ptr, nbytes, on_device):
        """(tensor to communicate on, write-back function).  nccl communicates on the library's own device memory; gloo needs
        host tensors, so device payloads are staged through a host copy."""
        if nbytes == 0:
            return self.torch.empty(0, dtype=self.torch.uint8, device=self.device if self.backend == "nccl" else "cpu"), (lambda: None)
        if not on_device:
            h = self._host(ptr, nbytes)
            if self.backend != "nccl" or self.device is None:
                return h, (lambda: None)
            d = h.to(self.device)  # RCCL communicates device memory only: small host payloads (scalars, counts) take a round trip
            return d, (lambda: h.copy_(d))
        d = self._dev(ptr, nbytes)
        if self.backend == "nccl":
            return d, (lambda: None)
        h = d.cpu()
        return h, (lambda: d.copy_(h))

    def _done(self, on_device):
        if on_device and self.device is not None:
            self.torch.cuda.synchronize(self.device)  # the library continues on its own stream

    # ---- the three collectives of hot_comm
    def _allreduce(self, user, buf, n, dtype, op, on_device):
        try:
            dt = _dtypes()[dtype]
            nbytes = n * dt.itemsize
            t, back = self._view(buf, nbytes, on_device)
            self.dist.all_reduce(t.view(dt), op=self.dist.ReduceOp.MAX if op == 1 else self.dist.ReduceOp.SUM, group=self.group)
            back()
            self._done(on_device)
            self.calls["allreduce"] += 1
            self.calls["bytes"] += nbytes
            return 0
        except Exception as e:  # never let an exception cross the C boundary
            print("hot_comm.allreduce failed:", repr(e), flush=True)
            return 1

    def _allgather(self, user, send, recv, nbytes, on_device):
        try:
            s, _ = self._view(send, nbytes, on_device)
            r, back = self._view(recv, nbytes * self.size, on_device)
            self.dist.all_gather_into_tensor(r, s, group=self.group) if self.backend == "nccl" else self.dist.all_gather(list(r.view(self.size, nbytes).unbind(0)), s, group=self.group)
            back()
            self._done(on_device)
            self.calls["allgather"] += 1
            self.calls["bytes"] += nbytes * self.size
            return 0
        except Exception as e:
            print("hot_comm.allgather failed:", repr(e), flush=True)
            return 1

    def _alltoallv(self, user, send, soff, sbytes, recv, roff, rbytes, on_device):
        try:
            R = self.size
            so, sb, ro, rb = ([int(a[i]) for i in range(R)] for a in (soff, sbytes, roff, rbytes))
            stot, rtot = max((so[i] + sb[i] for i in range(R)), default=0), max((ro[i] + rb[i] for i in range(R)), default=0)
            s, _ = self._view(send, stot, on_device)
            r, back = self._view(recv, rtot, on_device)
            ops = []
            for p in range(R):
                if p == self.rank:
                    continue
                if rb[p]:
                    ops.append(self.dist.P2POp(self.dist.irecv, r[ro[p]:ro[p] + rb[p]], p, self.group))
                if sb[p]:
                    ops.append(self.dist.P2POp(self.dist.isend, s[so[p]:so[p] + sb[p]], p, self.group))
            if ops:
                for w in self.dist.batch_isend_irecv(ops):
                    w.wait()
            back()
            self._done(on_device)
            self.calls["alltoallv"] += 1
            self.calls["bytes"] += sum(sb)
            return 0
        except Exception as e:
            print("hot_comm.alltoallv failed:", repr(e), flush=True)
            return 1


def attach_rccl(ctx, group=None, partition_min_rows=0):
    """Native, stream-ordered RCCL communicator for a HIP-library context (hot_amd/csrc/rccl_comm.hip): rank 0 creates the
    ncclUniqueId, torch.distributed only carries its 128 bytes to the other ranks.  Raises HotError if RCCL is unavailable."""
    import torch.distributed as dist
    rank, size = dist.get_rank(group), dist.get_world_size(group)
    box = [ctx.rccl_unique_id() if rank == 0 else None]
    dist.broadcast_object_list(box, src=0, group=group)
    ctx.rccl_attach(box[0], rank, size, partition_min_rows)


def attach(ctx, comm):
    """Install `comm` (a TorchComm) in a binding.Context; call before set_particles.  Returns comm."""
    ctx.set_comm(comm)
    return comm


# ---------------------------------------------------------------------------------------------------- sharding a cloud
def _spread3(v):
    v = v.astype(np.uint64) & np.uint64(0x1fffff)
    v = (v | (v << np.uint64(32))) & np.uint64(0x1f00000000ffff)
    v = (v | (v << np.uint64(16))) & np.uint64(0x1f0000ff0000ff)
    v = (v | (v << np.uint64(8))) & np.uint64(0x100f00f00f00f00f)
    v = (v | (v << np.uint64(4))) & np.uint64(0x10c30c30c30c30c3)
    v = (v | (v << np.uint64(2))) & np.uint64(0x1249249249249249)
    return v


def page_keys(X, dx, dtype):
    """SPGrid page id (Linear_Offset(baseNode(x / dx)) >> 12, Lib/SPGrid/Core/SPGrid_Mask.h:150-157) of every particle, in the
    arithmetic of the simulation's scalar type like MpmSimulationBase.cpp:1080-1085: the high part of the particle sort key."""
    T = np.dtype(dtype).type
    data_bits = 6 if T == np.float32 else 7
    block_bits = 12 - data_bits
    zb, yb, xb = block_bits // 3 + (block_bits % 3 > 0), block_bits // 3 + (block_bits % 3 > 1), block_bits // 3
    sh = 3 - block_bits % 3

    def lo(start):
        b = start
        while b < 12:
            b += 3
        return b
    zlo, ylo, xlo = lo(sh), lo(sh + 1), lo(sh + 2)
    inv = T(1) / T(dx)
    xi = np.asarray(X, T) * inv - T(0.5)
    b = xi.astype(np.int64)
    b -= (b > xi)  # int_floor
    off = (_spread3(b[:, 0] >> xb) << np.uint64(xlo)) | (_spread3(b[:, 1] >> yb) << np.uint64(ylo)) | (_spread3(b[:, 2] >> zb) << np.uint64(zlo))
    return off >> np.uint64(12)


def shard_by_page_order(cloud, rank, world, keys=("X", "V", "mass", "vol", "mu", "lam")):
    """Rank `rank`'s shard of `cloud` (dict of per-particle arrays + "dx"): the particles are ordered by SPGrid page (the
    library's own sort order), the page list is cut into `world` contiguous runs of nearly equal particle counts, and the
    particles of a run keep their original relative order.  Whole pages only: a particle group is never split."""
    pk = page_keys(cloud["X"], cloud["dx"], cloud["X"].dtype)
    order = np.argsort(pk, kind="stable")
    spk = pk[order]
    n = len(spk)
    cuts = [0]
    for r in range(1, world):
        t = n * r // world
        while 0 < t < n and spk[t] == spk[t - 1]:  # move the cut to the next page boundary
            t += 1
        cuts.append(max(t, cuts[-1]))
    cuts.append(n)
    idx = np.sort(order[cuts[rank]:cuts[rank + 1]])
    out = {k: np.ascontiguousarray(cloud[k][idx]) for k in keys if cloud.get(k) is not None}
    for k, v in cloud.items():
        if k not in out and k not in keys:
            out[k] = v
    out["index"] = idx  # positions of the shard's particles in the whole cloud
    return out
