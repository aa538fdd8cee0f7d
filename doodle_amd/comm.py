"""The sharded render's one collective: an all-gather of image shards.

Two interchangeable transports behind one small class:

* ``rccl`` — libhelio_comm.so (include/helio_comm.h): ``ncclAllGather`` enqueued directly on
  a dedicated HIP side stream, so the gather of step k overlaps the render of step k+1 and
  costs a few microseconds of host time.  The communicator is bootstrapped by broadcasting
  RCCL's unique id over the already initialised ``torch.distributed`` group.
* ``torch`` — ``torch.distributed.all_gather_into_tensor`` on the caller's process group
  (RCCL with the "nccl" backend on GPUs, gloo in the CPU tests).

* ``p2p`` — opt-in, correctness only (no multi-GPU box has run it): every rank maps the other ranks' receive buffers
  once (HIP IPC) and a gather is ONE kernel per rank that stores its shard straight into every rank's buffer — all
  seven xGMI links written at once, no ring — then publishes a per-source flag to a table in host memory shared by
  the node's processes; the receiver's host polls its row (no kernel ever spins).  ``helio_p2p_*`` in
  include/helio_comm.h.  RCCL stays the default.

One process per GPU.
"""
from __future__ import annotations

import ctypes
import os

import torch
import torch.distributed as dist

_HERE = os.path.dirname(os.path.abspath(__file__))
COMM_LIB_PATH = os.path.join(_HERE, "libhelio_comm.so")
COMM_EXPORTS = ("helio_comm_unique_id", "helio_comm_init", "helio_comm_allgather_f32", "helio_comm_count",
                "helio_comm_destroy", "helio_p2p_alloc", "helio_p2p_open", "helio_p2p_close", "helio_p2p_free",
                "helio_p2p_register_host", "helio_p2p_unregister_host", "helio_p2p_scatter_f32")
_IPC_BYTES = 64
_P2P_SLOTS = 8          # distinct shard sizes one gather object serves (images, actual, refl, …)
_ID_BYTES = 128
_lib = None


def load_comm_library(path: str = COMM_LIB_PATH) -> ctypes.CDLL:
    global _lib
    if _lib is None:
        if not os.path.exists(path):
            raise RuntimeError(f"{path} is missing: run `python -m doodle_amd.build`")
        lib = ctypes.CDLL(path)
        vp, i, l = ctypes.c_void_p, ctypes.c_int, ctypes.c_long
        lib.helio_comm_unique_id.argtypes, lib.helio_comm_unique_id.restype = [ctypes.c_char_p, i], i
        lib.helio_comm_init.argtypes, lib.helio_comm_init.restype = [ctypes.POINTER(vp), i, i, ctypes.c_char_p, i], i
        lib.helio_comm_allgather_f32.argtypes, lib.helio_comm_allgather_f32.restype = [vp, vp, vp, l, vp], i
        lib.helio_comm_destroy.argtypes, lib.helio_comm_destroy.restype = [vp], i
        lib.helio_comm_count.argtypes = [vp, ctypes.POINTER(i), ctypes.POINTER(i)]
        lib.helio_comm_count.restype = i
        pvp, u = ctypes.POINTER(vp), ctypes.c_uint
        lib.helio_p2p_alloc.argtypes, lib.helio_p2p_alloc.restype = [l, pvp, ctypes.c_char_p, i], i
        lib.helio_p2p_open.argtypes, lib.helio_p2p_open.restype = [ctypes.c_char_p, i, pvp], i
        lib.helio_p2p_close.argtypes, lib.helio_p2p_close.restype = [vp], i
        lib.helio_p2p_free.argtypes, lib.helio_p2p_free.restype = [vp], i
        lib.helio_p2p_register_host.argtypes, lib.helio_p2p_register_host.restype = [vp, l, pvp], i
        lib.helio_p2p_unregister_host.argtypes, lib.helio_p2p_unregister_host.restype = [vp], i
        lib.helio_p2p_scatter_f32.argtypes = [vp, l, i, i, pvp, vp, i, vp, vp]
        lib.helio_p2p_scatter_f32.restype = i
        _lib = lib
    return _lib


class ImageGather:
    """all-gather of equally sized fp32 blocks: ``out[r*n:(r+1)*n] = block of rank r``."""

    def __init__(self, group=None, transport: str = "auto"):
        self.group = group
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        self.rank = dist.get_rank(group) if dist.is_initialized() else 0
        self.comm = None
        self.stream = None
        self._p2p = None
        if transport == "auto":
            transport = os.environ.get("HELIO_COMM") or (
                "rccl" if (torch.cuda.is_available() and dist.is_initialized()
                           and dist.get_backend(group) == "nccl") else "torch")
        if transport == "rccl":
            try:
                self._init_rccl()
            except (RuntimeError, OSError) as e:
                # no silent change of transport: on an 8-GPU run it would change what is measured, and "auto" picks
                # RCCL only where it is expected to work (GPUs, an nccl process group).  The other transport is one
                # explicit word away.
                raise RuntimeError(
                    f"ImageGather: the RCCL transport (libhelio_comm.so) could not be set up: {e}.  Build it with "
                    "`python -m doodle_amd.build`, or ask for the torch.distributed transport explicitly "
                    "(transport=\"torch\" / HELIO_COMM=torch)") from e
        elif transport == "p2p":
            self._p2p = _PeerStores(self.group, self.world, self.rank)
        elif transport != "torch":
            raise ValueError(f"ImageGather: unknown transport {transport!r} (auto, rccl, torch, p2p)")
        self.transport = "rccl" if self.comm is not None else ("p2p" if self._p2p is not None else "torch")

    def _init_rccl(self):
        lib = load_comm_library()
        buf = ctypes.create_string_buffer(_ID_BYTES)
        if self.rank == 0 and lib.helio_comm_unique_id(buf, _ID_BYTES) <= 0:
            raise RuntimeError("ncclGetUniqueId failed")
        ident = torch.frombuffer(bytearray(buf.raw), dtype=torch.uint8).clone()
        if self.world > 1:
            dev_ident = ident.cuda()
            dist.broadcast(dev_ident, src=dist.get_global_rank(self.group, 0) if self.group else 0, group=self.group)
            ident = dev_ident.cpu()
        comm = ctypes.c_void_p()
        rc = lib.helio_comm_init(ctypes.byref(comm), self.world, self.rank, bytes(ident.numpy().tobytes()), _ID_BYTES)
        if rc != 0:
            raise RuntimeError(f"ncclCommInitRank failed (code {rc})")
        self.comm, self._libc = comm, lib
        self.stream = torch.cuda.Stream()
        self._ready, self._done, self._issued = torch.cuda.Event(), [None, None], 0

    def gather(self, local: torch.Tensor, out: torch.Tensor, overlap: bool = False) -> torch.Tensor:
        """Enqueue the all-gather of ``local`` into ``out``.

        ``overlap=False`` (default): the collective is enqueued on the caller's CURRENT stream,
        after the render that produced ``local`` and before anything that reads ``out`` — plain
        stream order, no events, one RCCL call of host time.
        ``overlap=True`` (rccl transport): the collective runs on a side stream so that it
        overlaps later work of the current stream; at most two gathers are in flight (the
        current stream is made to wait for the one before last); call :meth:`wait` before
        reading ``out``.
        """
        local = local.contiguous()
        assert out.numel() == self.world * local.numel() and out.is_contiguous()
        if self._p2p is not None:
            return self._p2p.gather(local, out)         # (stream-ordered; `overlap` has no peer-store form yet)
        if self.transport == "torch":
            if self.world == 1 and not dist.is_initialized():
                out.view(-1).copy_(local.view(-1))
            else:
                dist.all_gather_into_tensor(out, local, group=self.group)
            return out
        from .native import _stream
        if not overlap:
            rc = self._libc.helio_comm_allgather_f32(self.comm, local.data_ptr(), out.data_ptr(), local.numel(),
                                                     _stream())
            if rc != 0:
                raise RuntimeError(f"ncclAllGather failed (code {rc})")
            return out
        cur = torch.cuda.current_stream()
        k = self._issued & 1
        if self._done[k] is not None:
            cur.wait_event(self._done[k])                  # back-pressure: gather k-2 has finished
        else:
            self._done[k] = torch.cuda.Event()
        self._ready.record(cur)
        self.stream.wait_event(self._ready)                # the shard must be rendered first
        local.record_stream(self.stream)
        out.record_stream(self.stream)
        rc = self._libc.helio_comm_allgather_f32(self.comm, local.data_ptr(), out.data_ptr(), local.numel(),
                                                 self.stream.cuda_stream)
        if rc != 0:
            raise RuntimeError(f"ncclAllGather failed (code {rc})")
        self._done[k].record(self.stream)
        self._issued += 1
        return out

    @property
    def rccl_ranks(self):
        """Ranks in the RCCL communicator as RCCL itself counts them (ncclCommCount); None on the
        torch.distributed transport."""
        if self.comm is None:
            return None
        n, r = ctypes.c_int(), ctypes.c_int()
        if self._libc.helio_comm_count(self.comm, ctypes.byref(n), ctypes.byref(r)) != 0:
            raise RuntimeError("ncclCommCount failed")
        assert r.value == self.rank
        return n.value

    def wait(self):
        """Make the current stream wait for every gather enqueued with ``overlap=True``."""
        if self.stream is not None:
            torch.cuda.current_stream().wait_stream(self.stream)

    def close(self):
        if self._p2p is not None:
            self._p2p.close()
            self._p2p = None
        if self.comm is not None:
            torch.cuda.synchronize()
            self._libc.helio_comm_destroy(self.comm)
            self.comm = None


class _DeviceBlock:
    """A device allocation that is not torch's, as something ``torch.as_tensor`` can view (__cuda_array_interface__)."""

    def __init__(self, ptr: int, floats: int):
        self.__cuda_array_interface__ = {"shape": (floats,), "typestr": "<f4", "data": (ptr, False), "version": 2}


class _PeerStores:
    """The ``p2p`` transport (module docstring; include/helio_comm.h ``helio_p2p_*``).

    Per shard size (``count`` floats) a *slot*: two receive buffers of ``world·count`` floats (steps alternate between
    them), every other rank's two buffers mapped through their IPC handles, an epoch counter, and row ``rank`` of the
    slot's ``[world][world]`` flag table in the node-wide shared host segment.  A gather:

      1. ``helio_p2p_scatter_f32`` on the caller's stream — this rank's shard into slot position ``rank`` of EVERY
         rank's buffer of this step's parity, then ``flags[p][rank] = epoch`` for every p, behind a system-scope fence;
      2. the HOST polls ``flags[rank][:]`` until every source has published this epoch (timeout → RuntimeError: a rank
         that never arrives is an exception here, never a wave that spins);
      3. a device copy of the buffer into the caller's ``out``, on the same stream.

    Why two buffers are enough: rank A enqueues its scatter of step k+2 (into parity k) only after its host has seen
    every rank's flag of step k+1; rank B's scatter of step k+1 is stream-ordered behind B's copy of step k out of its
    parity-k buffer — so nobody still reads what A overwrites.  All on one stream; ``overlap`` is not offered."""

    def __init__(self, group, world, rank, timeout_s: float = 60.0):
        if not torch.cuda.is_available():
            raise RuntimeError("ImageGather(transport='p2p') needs a HIP device")
        if world > 1 and not dist.is_initialized():
            raise RuntimeError("ImageGather(transport='p2p') needs an initialised torch.distributed group")
        if world > 16:
            raise RuntimeError("ImageGather(transport='p2p') serves one node: at most 16 ranks")
        from multiprocessing import resource_tracker, shared_memory
        self.group, self.world, self.rank, self.timeout_s = group, world, rank, timeout_s
        self.lib = load_comm_library()
        self.slots = {}                       # count → slot state
        nbytes = -(-(_P2P_SLOTS * world * world * 4) // 4096) * 4096
        if rank == 0:
            self.shm = shared_memory.SharedMemory(create=True, size=nbytes)
            self.shm.buf[:nbytes] = bytes(nbytes)
        name = [self.shm.name if rank == 0 else None]
        if world > 1:
            dist.broadcast_object_list(name, src=dist.get_global_rank(group, 0) if group else 0, group=group)
        if rank != 0:
            self.shm = shared_memory.SharedMemory(name=name[0])
            try:        # (Python < 3.13 registers an ATTACHED segment with the resource tracker, which would unlink it at exit)
                resource_tracker.unregister(self.shm._name, "shared_memory")
            except Exception:
                pass
        import numpy as np
        self.flags = np.frombuffer(self.shm.buf, dtype=np.int32, count=_P2P_SLOTS * world * world)
        self._host_ptr = ctypes.addressof(ctypes.c_char.from_buffer(self.shm.buf))
        dev = ctypes.c_void_p()
        if self.lib.helio_p2p_register_host(self._host_ptr, nbytes, ctypes.byref(dev)) != 0:
            raise RuntimeError("ImageGather(transport='p2p'): hipHostRegister of the shared flag table failed")
        self._flags_dev = dev.value
        self._arrived = torch.zeros(_P2P_SLOTS, dtype=torch.int32, device="cuda")
        if world > 1:
            dist.barrier(group=group)          # the table is zeroed and registered everywhere before anyone publishes

    def _slot(self, count: int):
        st = self.slots.get(count)
        if st is not None:
            return st
        if len(self.slots) == _P2P_SLOTS:
            raise RuntimeError(f"ImageGather(transport='p2p'): more than {_P2P_SLOTS} distinct shard sizes")
        lib, world = self.lib, self.world
        own, handles = [], []
        for _ in range(2):
            ptr, h = ctypes.c_void_p(), ctypes.create_string_buffer(_IPC_BYTES)
            if lib.helio_p2p_alloc(4 * world * max(count, 1), ctypes.byref(ptr), h, _IPC_BYTES) <= 0:
                raise RuntimeError("ImageGather(transport='p2p'): hipMalloc / hipIpcGetMemHandle failed")
            own.append(ptr.value)
            handles.append(h.raw)
        everyone = [None] * world
        if world > 1:
            dist.all_gather_object(everyone, handles, group=self.group)
        else:
            everyone[0] = handles
        peers, opened = [], []
        for par in range(2):
            arr = (ctypes.c_void_p * world)()
            for r in range(world):
                if r == self.rank:
                    arr[r] = own[par]
                else:
                    p = ctypes.c_void_p()
                    if lib.helio_p2p_open(everyone[r][par], _IPC_BYTES, ctypes.byref(p)) != 0:
                        raise RuntimeError(f"ImageGather(transport='p2p'): hipIpcOpenMemHandle of rank {r}'s buffer failed")
                    arr[r] = p.value
                    opened.append(p.value)
            peers.append(arr)
        index = len(self.slots)
        views = [torch.as_tensor(_DeviceBlock(own[par], world * max(count, 1)), device="cuda") for par in range(2)]
        st = {"index": index, "own": own, "peers": peers, "opened": opened, "views": views, "epoch": 0}
        self.slots[count] = st
        if world > 1:
            dist.barrier(group=self.group)     # every rank has mapped every buffer before the first store
        return st

    def gather(self, local: torch.Tensor, out: torch.Tensor) -> torch.Tensor:
        import time
        from .native import _stream
        if not (local.is_cuda and local.dtype == torch.float32 and out.is_cuda and out.dtype == torch.float32):
            raise RuntimeError("ImageGather(transport='p2p') gathers float32 device tensors")
        count, world = local.numel(), self.world
        st = self._slot(count)
        st["epoch"] += 1
        epoch, par, k = st["epoch"], st["epoch"] & 1, st["index"]
        if count:
            rc = self.lib.helio_p2p_scatter_f32(local.data_ptr(), count, self.rank, world, st["peers"][par],
                                                self._flags_dev + 4 * k * world * world, epoch,
                                                self._arrived.data_ptr() + 4 * k, _stream())
            if rc != 0:
                raise RuntimeError(f"helio_p2p_scatter_f32 failed (code {rc})")
            row = self.flags[(k * world + self.rank) * world:(k * world + self.rank + 1) * world]
            deadline = time.monotonic() + self.timeout_s
            while not bool((row >= epoch).all()):
                if time.monotonic() > deadline:
                    raise RuntimeError(f"ImageGather(transport='p2p'): rank(s) {[r for r in range(world) if row[r] < epoch]} "
                                       f"did not publish step {epoch} within {self.timeout_s} s")
                time.sleep(0)
            out.view(-1).copy_(st["views"][par][:world * count])
        return out

    def close(self):
        torch.cuda.synchronize()
        if self.world > 1 and dist.is_initialized():
            dist.barrier(group=self.group)     # nobody unmaps a buffer another rank may still store into
        for st in self.slots.values():
            st["views"] = None
            for p in st["opened"]:
                self.lib.helio_p2p_close(p)
            for p in st["own"]:
                self.lib.helio_p2p_free(p)
        self.slots = {}
        self.lib.helio_p2p_unregister_host(self._host_ptr)
        self.flags = None
        try:
            self.shm.close()
        except BufferError:
            pass                               # (a view of the segment is still alive somewhere: the OS reclaims it at exit)
        if self.rank == 0:
            try:
                self.shm.unlink()
            except FileNotFoundError:
                pass
