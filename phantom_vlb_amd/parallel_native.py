"""``DirectComm``: the collectives of ``parallel.ShardedFlatState`` on libvlb's own RCCL entry points
(``vlb_comm_*`` in include/vlb.h) instead of torch.distributed's - selected with ``VLB_COMM=direct``.

All-pairs ("direct") schedules over the point-to-point xGMI links and a rank-ordered local reduction: the reduced
gradients are bit-reproducible.  torch.distributed is still what bootstraps the job (the 128-byte RCCL unique id
travels through its store / a broadcast) and what the default transport uses; this class needs a GPU per rank.
"""
from __future__ import annotations

import ctypes

import torch
import torch.distributed as dist

from ._lib import check, lib


class DirectComm:
    @classmethod
    def loopback(cls, world: int, device=None):
        """`world` DirectComm objects in THIS process on libvlb's loopback transport (vlb_comm_init_loopback): the same
        schedules and the same Python wrapper code as over RCCL, with ranks 0..world-1 driven by one host thread each
        (tests/test_gpu_comm.py) - the way to exercise offsets and pairing at world 2..8 on a one-GPU box."""
        device = device or torch.device("cuda", torch.cuda.current_device())
        stage = torch.zeros(lib.vlb_comm_loopback_stage_bytes(world) // 4, dtype=torch.float32, device=device)
        handles = (ctypes.c_void_p * world)()
        check(lib.vlb_comm_init_loopback(world, stage.data_ptr(), stage.numel() * 4, handles), "vlb_comm_init_loopback")
        out = []
        for r in range(world):
            c = cls.__new__(cls)
            c.group, c.world, c.rank = None, world, r
            c._h = ctypes.c_void_p(handles[r])
            c._stage, c._loop_stage = None, stage             # the staging buffer lives as long as any communicator
            c.stream = torch.cuda.Stream(device=device)
            out.append(c)
        return out

    def __init__(self, group=None, stream=None):
        self.group = group
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        self.rank = dist.get_rank(group) if dist.is_initialized() else 0
        buf = (ctypes.c_ubyte * 128)()
        if self.rank == 0:
            check(lib.vlb_comm_unique_id(buf), "vlb_comm_unique_id")
        if self.world > 1:
            ids = [bytes(buf)]
            dist.broadcast_object_list(ids, src=0, group=group)
            buf = (ctypes.c_ubyte * 128).from_buffer_copy(ids[0])
        self._h = ctypes.c_void_p()
        check(lib.vlb_comm_init(self.rank, self.world, buf, ctypes.byref(self._h)), "vlb_comm_init")
        self._stage = None
        # collectives run on a side stream (event hand-off with the compute stream), like torch's NCCL stream
        self.stream = stream if stream is not None else torch.cuda.Stream()

    def __del__(self):
        h = getattr(self, "_h", None)
        if h:
            lib.vlb_comm_destroy(h)
            self._h = None

    def _enter(self):
        self.stream.wait_stream(torch.cuda.current_stream())

    def _handle(self):
        ev = torch.cuda.Event()
        ev.record(self.stream)

        class _W:
            def wait(_s):
                torch.cuda.current_stream().wait_event(ev)
        return _W()

    def reduce_scatter(self, out, inp):
        """fp32 (head / LoRA store) or bf16 (the full fine-tune's backbone store): slices travel in their own dtype and are
        summed in rank order with fp32 accumulation by one local kernel."""
        n = out.numel()
        assert inp.numel() == n * self.world and out.dtype == inp.dtype and out.dtype in (torch.float32, torch.bfloat16)
        bf16 = out.dtype == torch.bfloat16
        fn = lib.vlb_reducescatter_direct_bf16 if bf16 else lib.vlb_reducescatter_direct
        self._enter()
        with torch.cuda.stream(self.stream):
            stage = torch.empty(n * self.world, dtype=out.dtype, device=out.device)      # stream-ordered allocation on the side stream
            check(fn(self._h, inp.data_ptr(), out.data_ptr(), n, stage.data_ptr(), self.stream.cuda_stream),
                  "vlb_reducescatter_direct_bf16" if bf16 else "vlb_reducescatter_direct")
        return self._handle()

    def all_gather(self, out, inp):
        nbytes = inp.numel() * inp.element_size()
        assert out.numel() * out.element_size() == nbytes * self.world
        self._enter()
        with torch.cuda.stream(self.stream):
            check(lib.vlb_allgather_direct(self._h, inp.data_ptr(), out.data_ptr(), nbytes, self.stream.cuda_stream),
                  "vlb_allgather_direct")
        return self._handle()

    def all_reduce_scalar(self, t):
        assert t.dtype == torch.float32
        self._enter()                      # same stream as the other collectives: one issue order per communicator
        with torch.cuda.stream(self.stream):
            check(lib.vlb_allreduce_scalar(self._h, t.data_ptr(), t.numel(), self.stream.cuda_stream), "vlb_allreduce_scalar")
        self._handle().wait()
