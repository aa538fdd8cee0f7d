"""The sharded render's one collective: an all-gather of image shards.

Two interchangeable transports behind one small class:

* ``rccl`` — libhelio_comm.so (include/helio_comm.h): ``ncclAllGather`` enqueued directly on
  a dedicated HIP side stream, so the gather of step k overlaps the render of step k+1 and
  costs a few microseconds of host time.  The communicator is bootstrapped by broadcasting
  RCCL's unique id over the already initialised ``torch.distributed`` group.
* ``torch`` — ``torch.distributed.all_gather_into_tensor`` on the caller's process group
  (RCCL with the "nccl" backend on GPUs, gloo in the CPU tests).

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
                "helio_comm_destroy")
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
        elif transport != "torch":
            raise ValueError(f"ImageGather: unknown transport {transport!r} (auto, rccl, torch)")
        self.transport = "rccl" if self.comm is not None else "torch"

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
        if self.comm is not None:
            torch.cuda.synchronize()
            self._libc.helio_comm_destroy(self.comm)
            self.comm = None
