"""Multi-GPU: one process per GPU, torch.distributed over RCCL/xGMI ("nccl" backend on ROCm).

The path shards over CLIPS: every rank runs the same step on its own clips; the
only exchange per optimiser step is the gradient of the trainables.  The reference has no
multi-GPU mainline - only an unused Accelerate FSDP config (``fsdp.yaml``: FULL_SHARD,
TRANSFORMER_BASED_WRAP on the decoder layer, BACKWARD_PRE, forward prefetch) - so this module is
the MI355X-native equivalent of that config, not a translation of torch FSDP:

* ``ShardedFlatState``  - what trains (head, LoRA adapters; for the full fine-tune also the backbone store) lives in flat
  buffers cut into segments; per step every segment's gradient is REDUCE-SCATTERED (started from the backward pass as
  soon as the segment's layers are differentiated), each rank runs the clip + AdamW on its 1/world slice only (fp32
  master and both moments exist only for the owned slice) and the refreshed bf16 copies are ALL-GATHERED back.  The
  frozen 7B never enters a gradient collective: with ``use_orig_params`` torch FSDP would reduce-scatter 436 MB per
  layer of zeros.  Loss scaling: each rank backpropagates mse/world and lambda||W||^2/world, so the SUM over ranks is
  the gradient of the single-process objective on the concatenated batch (penalty counted once).  The full
  fine-tune's backbone store is FULL_SHARD for its trained bf16 WEIGHTS as well (``full_shard=True``, the default:
  1/world of every decoder layer per rank, a layer gathered for its forward and again for its backward into two
  rotating buffers; ``VLB_FSDP_STRATEGY=SHARD_GRAD_OP`` keeps them replicated), and so are the frozen ones when opted into:
* ``ShardedLayerStore`` - the fsdp.yaml-equivalent parameter sharding for the frozen decoder layers:
  each rank keeps 1/world of every layer's flat bf16 weights (436 MB/layer -> 54.5 MB at 8 ranks);
  the full layer is all-gathered into one of two buffers on a side stream, one layer ahead of
  compute (forward and, in reverse order, backward).  On xGMI (point-to-point links) RCCL's
  all-gather moves each 54.5 MB shard over its own link, ~0.36 ms/layer, hidden behind ~2 ms of
  layer compute.  With 288 GB of HBM the replicated variant (no gathers) is the faster default;
  sharding is opt-in (``shard_frozen=True``) for memory parity with the reference's FSDP intent.  Under
  ``VLB_COMM=direct`` the gathers go through libvlb's own all-pairs schedule (``vlb_allgather_direct``).

Everything here works on CPU tensors with the gloo backend (tests/test_cpu_parallel.py).
"""
from __future__ import annotations

import os

import torch
import torch.distributed as dist


def local_device_index() -> int:
    """GPU of this process: LOCAL_RANK, wrapped onto the visible devices so that several ranks can
    rehearse on a one-GPU box (gloo backend); on a real node every rank gets its own GPU."""
    n = torch.cuda.device_count() if torch.cuda.is_available() else 0
    local = int(os.environ.get("LOCAL_RANK", "0"))
    return local % n if n > 0 else local


def init_distributed(backend: str | None = None):
    """Initialise from torchrun's env (RANK/WORLD_SIZE/LOCAL_RANK/MASTER_*). Returns (rank, world)."""
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    if world > 1 and not dist.is_initialized():
        if backend is None:       # VLB_DIST_BACKEND=gloo lets several ranks rehearse on one GPU
            backend = os.environ.get("VLB_DIST_BACKEND") or ("nccl" if torch.cuda.is_available() else "gloo")
        kw = {}
        if backend == "nccl":
            kw["device_id"] = torch.device("cuda", local_device_index())
        dist.init_process_group(backend, **kw)
    return rank, world


def dp_loss_scales(world: int):
    """(mse gradient scale, ridge-penalty gradient scale) each rank applies before the SUM all-reduce."""
    return 1.0 / world, 1.0 / world


class TorchComm:
    """The three collectives the sharded step needs, over torch.distributed (backend "nccl" = RCCL on ROCm, gloo on
    CPU).  ``parallel_native.DirectComm`` offers the same interface on libvlb's own RCCL entry points
    (vlb_comm_* in include/vlb.h: all-pairs schedules over the xGMI links)."""

    def __init__(self, group=None):
        self.group = group
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        self.rank = dist.get_rank(group) if dist.is_initialized() else 0

    def reduce_scatter(self, out, inp):
        """out[n/world] = sum over ranks of inp[rank*n/world : ...]; asynchronous, returns a handle with wait()."""
        return dist.reduce_scatter_tensor(out, inp, op=dist.ReduceOp.SUM, group=self.group, async_op=True)

    def all_gather(self, out, inp):
        return dist.all_gather_into_tensor(out, inp, group=self.group, async_op=True)

    def all_reduce_scalar(self, t):
        dist.all_reduce(t, op=dist.ReduceOp.SUM, group=self.group)


class ShardedFlatState:
    """fsdp.yaml's FULL_SHARD for everything the optimiser owns (reference fsdp.yaml:5-14, never loaded by its
    mainline): the flat trainables of ``flat.FlatTrainables`` are cut into segments (the head; runs of whole decoder
    layers) and every rank owns the rank-th 1/world slice of each segment.

    Per step and segment: ``reduce_scatter`` of the fp32 gradient (started from the backward pass the moment the
    segment's layers are differentiated, so the xGMI exchange runs under the remaining backward) -> each rank holds
    the summed gradient of its slice only; the clip norm is the all-reduced sum of the slices' squares; AdamW runs
    on the slice (fp32 master and both moments exist ONLY for the owned slice: 16 B/param/world); the updated bf16
    compute copies are ``all_gather``-ed back into the full buffer the kernels read.  Wire bytes per step:
    (world-1)/world x (4 + 2) B per trainable, against 8 B for the all-reduce of replicated optimisers.
    At world 1 the slices ARE the flat buffers (aliases, no copies, no collectives): one code path.

    ``flat.master`` stays allocated at full size as a staging area: ``gather_masters()`` refreshes it (checkpoints,
    tests) and ``load_masters()`` re-reads the owned slices from it (resume, sync_module_states)."""

    NAMES = ("master", "grad", "m", "v")

    def __init__(self, flat, comm=None, chunks: int = 4, force_collectives: bool = False, full_shard: bool = False):
        """``force_collectives``: run the collective path (separate shard buffers, reduce-scatter / all-gather calls)
        even at world size 1 - rehearses the RCCL calls on a one-GPU box.

        ``full_shard`` (the full fine-tune's backbone store only): ``fsdp_sharding_strategy: FULL_SHARD``
        (reference fsdp.yaml:11) for the trained bf16 WEIGHTS as well - see ``_enter_full_shard``."""
        self.flat = flat
        self.comm = comm if comm is not None else TorchComm()
        w, r = self.comm.world, self.comm.rank
        self.world, self.rank = w, r
        segs = [flat.head_range]
        L = len(flat.layer_ranges)
        if full_shard:
            chunks = max(L, 1)                    # one segment per decoder layer: the unit of every gather and reduce
        per = max(1, -(-L // chunks)) if L else 1
        self.layer_seg = {}                       # first layer of a chunk -> segment index
        for lo in range(0, L, per):
            hi = min(lo + per, L) - 1
            self.layer_seg[lo] = len(segs)
            segs.append((flat.layer_ranges[lo][0], flat.layer_ranges[hi][1]))
        for s, e in segs:
            if (e - s) % (8 * w):
                raise ValueError(f"world size {w} does not divide the flat segment alignment ({e - s} elements): "
                                 f"supported world sizes divide 840 (1..8, 10, 12, ...)")
        self.segments = segs
        self.shard_off, off = [], 0
        for s, e in segs:
            self.shard_off.append(off)
            off += (e - s) // w
        self.numel = off
        assert off * w == flat.numel
        self.active = w > 1 or (force_collectives and (dist.is_initialized() or not isinstance(self.comm, TorchComm)))
        if not self.active:
            self.master, self.compute, self.grad, self.m, self.v = flat.master, flat.compute, flat.grad, flat.m, flat.v
        else:
            dev = flat.master.device
            self.master = torch.empty(off, dtype=torch.float32, device=dev)
            self.compute = torch.empty(off, dtype=flat.compute.dtype, device=dev)
            self.grad = torch.zeros(off, dtype=flat.grad.dtype, device=dev)      # fp32, or bf16 for the backbone store
            self.m = torch.zeros(off, dtype=torch.float32, device=dev)
            self.v = torch.zeros(off, dtype=torch.float32, device=dev)
            self.load_masters()
            flat.m = flat.v = None                # the full-size moments are released: sharded state only
        self._pending = {}
        self.full_shard = bool(full_shard and self.active)
        if self.full_shard:
            self._enter_full_shard()

    # ---- FULL_SHARD for trained weights (fsdp.yaml:11; the full fine-tune's backbone store)
    def _enter_full_shard(self):
        """Between uses a rank holds only its 1/world slice of every decoder layer's trained bf16 weights (``self.compute``,
        the buffer AdamW writes) and of their gradients (``self.grad``).  The full-size ``flat.compute`` / ``flat.grad`` /
        ``flat.master`` are cut down to the TAIL segment (connector, embeddings, final norm: the root unit, gathered for the
        whole step like FSDP's root module), and two layer-sized weight buffers + two layer-sized gradient buffers take the
        place of the 32 layers: ``layer_weights(li)`` all-gathers layer li into buffer li % 2 (the next layer's gather is
        started first so it runs under this layer's kernels: fsdp_forward_prefetch / BACKWARD_PRE), ``layer_grads(li)``
        hands out the gradient buffer the backward writes, and ``on_layer_done(li)`` reduce-scatters it into the owned
        slice.  The alignment pads inside a layer stay zero (buffers zeroed once; the kernels write whole tensors only)."""
        flat, dev = self.flat, self.flat.compute.device
        assert flat.head_range[0] == 0 and len(self.segments) == len(flat.layer_ranges) + 1
        self.load_full("compute", flat.compute)               # owned bf16 slices <- the (already synchronised) full weights
        maxlayer = max(e - s for s, e in flat.layer_ranges)
        flat.release_layers()                                  # tail-only compute / grad, master released, layer views dropped
        self.wpool = [torch.zeros(maxlayer, dtype=self.compute.dtype, device=dev) for _ in range(2)]
        self.gpool = [torch.zeros(maxlayer, dtype=self.grad.dtype, device=dev) for _ in range(2)]
        self._in_w = [None, None]                 # layer currently (being) gathered into each weight buffer
        self._w_work = [None, None]
        torch.cuda.empty_cache()

    def _start_gather(self, li: int):
        b = li % 2
        if self._in_w[b] == li:
            return
        s, e = self.segments[li + 1]
        # (the collective is ordered after everything already enqueued on the current stream: the buffer's previous reader)
        self._w_work[b] = self.comm.all_gather(self.wpool[b][:e - s], self.compute[self._own(li + 1)[1]])
        self._in_w[b] = li

    def layer_weights(self, li: int, then: int | None = None):
        """Flat bf16 weights of decoder layer li, gathered (views: ``flat.layer_views``).  ``then``: the layer needed next -
        its gather is started now, into the other buffer."""
        self._start_gather(li)
        b = li % 2
        if self._w_work[b] is not None:
            self._w_work[b].wait()
            self._w_work[b] = None
        if then is not None and 0 <= then < len(self.flat.layer_ranges) and then % 2 != b:
            self._start_gather(then)
        s, e = self.segments[li + 1]
        return self.wpool[b][:e - s]

    def layer_grads(self, li: int):
        """The buffer layer li's backward writes its gradients into (the reduce-scatter of the layer that used it last is
        waited for first)."""
        b = li % 2
        prev = self._pending.get(li + 2 + 1)
        if prev is not None:
            prev.wait()
        s, e = self.segments[li + 1]
        return self.gpool[b][:e - s]

    # the backbone.layer_weights() store interface (validation / inference forward through the frozen-path code)
    def get(self, li: int) -> dict:
        return self.flat.layer_views(self.layer_weights(li), li)

    def prefetch(self, li: int):
        if 0 <= li < len(self.flat.layer_ranges):
            self._start_gather(li)

    def _own(self, si):
        """(slice of the flat buffers this rank owns in segment si, its slice of the shard buffers)."""
        s, e = self.segments[si]
        n = (e - s) // self.world
        return slice(s + self.rank * n, s + (self.rank + 1) * n), slice(self.shard_off[si], self.shard_off[si] + n)

    # ---- gradients
    def reduce_segment(self, si: int):
        """Start the reduce-scatter of segment si NOW (ordered after everything enqueued on the current stream)."""
        if not self.active or si in self._pending:
            return
        s, e = self.segments[si]
        _, mine = self._own(si)
        src = self.gpool[(si - 1) % 2][:e - s] if (self.full_shard and si > 0) else self.flat.grad[s:e]
        self._pending[si] = self.comm.reduce_scatter(self.grad[mine], src)

    def on_layer_done(self, li: int):
        """LoRA / full backward hook: layer li (walking L-1 .. 0) has its final gradients."""
        si = self.layer_seg.get(li)
        if si is not None:
            self.reduce_segment(si)

    def finish_reduce(self):
        for si in range(len(self.segments)):
            self.reduce_segment(si)
        for work in self._pending.values():
            work.wait()
        self._pending = {}

    def all_reduce_scalar(self, t):
        if self.active:
            self.comm.all_reduce_scalar(t)

    # ---- parameters
    def gather_compute(self):
        """After the update: every rank's refreshed bf16 slices -> the full compute buffer the kernels read."""
        if not self.active:
            return
        if self.full_shard:          # only the tail is held gathered; the layers' buffers are stale now, layer 0 is fetched ahead
            s, e = self.segments[0]
            self.comm.all_gather(self.flat.compute[s:e], self.compute[self._own(0)[1]]).wait()
            self._in_w, self._w_work = [None, None], [None, None]
            if len(self.segments) > 1:
                self._start_gather(0)
            return
        works = []
        for si, (s, e) in enumerate(self.segments):
            works.append(self.comm.all_gather(self.flat.compute[s:e], self.compute[self._own(si)[1]]))
        for wk in works:
            wk.wait()

    def gather_full(self, name: str):
        """Full-size fp32 copy of a sharded buffer ('master' | 'grad' | 'm' | 'v') on every rank (checkpoints, tests)."""
        src = getattr(self, name)
        if not self.active:
            return src
        out = torch.empty(self.flat.numel, dtype=src.dtype, device=src.device)
        works = [self.comm.all_gather(out[s:e], src[self._own(si)[1]]) for si, (s, e) in enumerate(self.segments)]
        for wk in works:
            wk.wait()
        return out

    def gather_masters(self):
        """Refresh the full-size fp32 staging copy ``flat.master`` (checkpoints, tests).  Under FULL_SHARD there is no
        standing one: it is created here and dropped again by ``release_staging()``."""
        if not self.active:
            return
        if self.full_shard:
            self.flat.master = self.gather_full("master")
        else:
            self.flat.master.copy_(self.gather_full("master"))

    def release_staging(self):
        if self.full_shard:
            self.flat.master = None

    def compute_from_master(self):
        """bf16 weights <- fp32 masters after a restore (checkpoint resume)."""
        if self.full_shard:
            self.compute.copy_(self.master)
            self.gather_compute()
        else:
            f = self.flat
            step = 1 << 28
            for a in range(0, f.numel, step):
                f.compute[a:a + step].copy_(f.master[a:a + step])

    def load_full(self, name: str, full):
        if not self.active:
            getattr(self, name).copy_(full)
            return
        dst = getattr(self, name)
        for si in range(len(self.segments)):
            whole, mine = self._own(si)
            dst[mine].copy_(full[whole])

    def load_masters(self):
        """Owned master slices <- flat.master (after a checkpoint load / broadcast wrote the full buffer)."""
        if self.active and self.flat.master is not None:
            self.load_full("master", self.flat.master)


def full_shard_default() -> bool:
    """``fsdp_sharding_strategy`` for the TRAINED backbone weights of the full fine-tune: FULL_SHARD, what the reference's
    fsdp.yaml:11 says, unless ``VLB_FSDP_STRATEGY=SHARD_GRAD_OP`` keeps them replicated (ZeRO-2: no per-layer weight
    gathers; 28 GB more per rank at 7B, which a 288 GB card has)."""
    v = os.environ.get("VLB_FSDP_STRATEGY", "FULL_SHARD").upper()
    if v not in ("FULL_SHARD", "SHARD_GRAD_OP"):
        raise ValueError(f"VLB_FSDP_STRATEGY={v!r}: FULL_SHARD or SHARD_GRAD_OP")
    return v == "FULL_SHARD"


def attach_data_parallel(module, optimizer, group=None, comm=None, force_collectives: bool = False, full_shard: bool | None = None,
                         src: int = 0):
    """Wire a VLBLitModule + VlbAdamW for clip-sharded data parallelism with sharded optimiser state.

    Full fine-tune: ``full_shard`` (default: ``full_shard_default()``) shards the trained decoder weights 1/world per layer
    as well.  Because a rank then no longer holds whole weights, the fsdp_sync_module_states broadcast from rank ``src``
    (fsdp.yaml:13) happens HERE, before the layers are cut up; ``sync_module_states`` afterwards covers the rest."""
    if comm is None:
        comm = make_comm(group)
    state = ShardedFlatState(module.flat, comm, force_collectives=force_collectives)
    optimizer.attach_sharded(state)
    lora, full = getattr(module, "lora", None), getattr(module, "full", None)
    if lora is not None:
        lora.grad_hook = state.on_layer_done
    module.sharded_backbone = None
    if full is not None:            # full fine-tune: the backbone store is sharded the same way (bf16 gradients)
        fs = full_shard_default() if full_shard is None else bool(full_shard)
        if fs:
            _sync_backbone_store(full, src, group)
        sb = module.sharded_backbone = ShardedFlatState(full.flat, comm, force_collectives=force_collectives, full_shard=fs)
        optimizer.attach_sharded(sb)
        full.grad_hook = sb.on_layer_done
        if sb.full_shard:
            full.enter_full_shard(sb)
            empty = torch.empty(0, dtype=torch.float32, device=sb.master.device)
            for n, p_ in zip(optimizer.names, optimizer.param_groups[0]["params"]):
                if n.startswith("backbone."):
                    p_.data = empty            # the optimiser's handles were views of the released full-size master
            torch.cuda.empty_cache()
    module.world_size, module.rank = state.world, state.rank
    module.sharded = state
    return state


_direct_comms = []          # [(live ProcessGroup object, DirectComm)]: identity-keyed, so a communicator never outlives its job


def make_comm(group=None):
    """VLB_COMM=direct selects libvlb's own RCCL schedules (vlb_comm_*); default: torch.distributed.  One DirectComm (one
    RCCL communicator, one issue order) per process group: the gradient exchange and the frozen-layer gathers share it.
    The cache is keyed by the live ProcessGroup OBJECT (``None`` = the current default group): after
    ``destroy_process_group()`` + a new ``init_process_group()`` the default group is a new object and gets a new
    communicator instead of one bootstrapped by the dead job."""
    if not dist.is_initialized():
        _direct_comms.clear()
        return TorchComm(group)
    if os.environ.get("VLB_COMM", "torch") == "direct" and dist.get_backend(group) == "nccl":
        from .parallel_native import DirectComm
        pg = dist.group.WORLD if group is None else group
        for live, comm in _direct_comms:
            if live is pg:
                return comm
        comm = DirectComm(group)
        _direct_comms.append((pg, comm))
        return comm
    return TorchComm(group)


def broadcast_parameters(tensors, src: int = 0, group=None):
    """fsdp_sync_module_states equivalent: make every rank start from rank `src`'s trainables."""
    if dist.is_initialized():
        for t in tensors:
            dist.broadcast(t, src=src, group=group)


def _sync_backbone_store(full, src: int = 0, group=None):
    """Rank `src`'s trained backbone weights everywhere (bf16; the fp32 masters are their exact widening)."""
    broadcast_parameters([full.flat.compute], src, group)
    step = 1 << 28
    for a in range(0, full.flat.numel, step):
        full.flat.master[a:a + step].copy_(full.flat.compute[a:a + step])


def sync_module_states(module, src: int = 0, group=None):
    """Broadcast the trainable masters from rank `src` and rebuild everything derived from them on this rank:
    the bf16 compute copies and (LoRA) the transposed / padded adapter layouts the kernels read."""
    flat = getattr(module, "flat", None)
    if flat is not None:
        broadcast_parameters([flat.master], src, group)
        flat.compute.copy_(flat.master)
        sharded = getattr(module, "sharded", None)
        if sharded is not None:
            sharded.load_masters()
        full = getattr(module, "full", None)
        sb = getattr(module, "sharded_backbone", None)
        if full is not None and not (sb is not None and sb.full_shard):       # (FULL_SHARD: done by attach_data_parallel)
            _sync_backbone_store(full, src, group)
            if sb is not None:
                sb.load_masters()
            full.refresh_transposed()
    else:
        broadcast_parameters([p.data for p in module.parameters()], src, group)
        for n, t in module.head.master.items():
            module.head.compute[n].copy_(t)
    lora = getattr(module, "lora", None)
    if lora is not None:
        lora.refresh(from_master=flat is None)


class ShardedLayerStore:
    """1/world shard of every layer's flat weights + double-buffered, prefetched all-gather."""

    def __init__(self, layers: list[dict], keys: tuple, group=None, stream=None, comm=None):
        """``comm``: a TorchComm / DirectComm whose ``all_gather`` does the per-layer gathers (``make_comm(group)`` by
        default: ``VLB_COMM=direct`` puts the 436 MB-per-layer gathers on vlb_allgather_direct's all-pairs schedule)."""
        self.group = group
        self.comm = comm
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        self.rank = dist.get_rank(group) if dist.is_initialized() else 0
        self.keys = keys
        self.meta = []          # per layer: [(key, shape, offset, numel)]
        self.shards = []
        maxpad = 0
        for lw in layers:
            off, meta = 0, []
            for k in keys:
                n = lw[k].numel()
                meta.append((k, tuple(lw[k].shape), off, n))
                off += (n + 7) // 8 * 8            # keep every tensor 16-byte aligned inside the flat layer
            pad = (off + self.world * 8 - 1) // (self.world * 8) * (self.world * 8)
            flat = torch.zeros(pad, dtype=lw[keys[0]].dtype, device=lw[keys[0]].device)
            for k, shp, o, n in meta:
                flat[o:o + n] = lw[k].reshape(-1)
            per = pad // self.world
            self.shards.append(flat[self.rank * per:(self.rank + 1) * per].clone())
            self.meta.append((meta, pad))
            maxpad = max(maxpad, pad)
        dev = self.shards[0].device
        self.buf = [torch.empty(maxpad, dtype=self.shards[0].dtype, device=dev) for _ in range(2)]
        self.stream = stream
        self.events = [None, None]
        self.in_buf = [None, None]

    def shard_bytes(self):
        return sum(s.numel() * s.element_size() for s in self.shards)

    def prefetch(self, i: int):
        """Start gathering layer i into buffer i%2 (on the side stream when one was given)."""
        if i < 0 or i >= len(self.shards) or self.in_buf[i % 2] == i:
            return
        b = i % 2
        pad = self.meta[i][1]
        out = self.buf[b][:pad]
        if self.stream is not None:
            self.stream.wait_stream(torch.cuda.current_stream())   # the buffer's previous consumer must be done
            with torch.cuda.stream(self.stream):
                self._gather(out, self.shards[i])
                ev = torch.cuda.Event()
                ev.record(self.stream)
                self.events[b] = ev
        else:
            self._gather(out, self.shards[i])
        self.in_buf[b] = i

    def _gather(self, out, shard):
        if self.world == 1 and self.comm is None:
            out.copy_(shard)
            return
        if self.comm is None:
            self.comm = make_comm(self.group)
        self.comm.all_gather(out[:shard.numel() * self.world], shard).wait()      # ordered on the calling (side) stream

    def get(self, i: int) -> dict:
        """Full weights of layer i as views into the gather buffer (valid until layer i+2 is fetched)."""
        self.prefetch(i)
        b = i % 2
        if self.stream is not None and self.events[b] is not None:
            torch.cuda.current_stream().wait_event(self.events[b])
        meta, _ = self.meta[i]
        return {k: self.buf[b][o:o + n].view(shp) for k, shp, o, n in meta}
