"""Data-parallel training plumbing: one process per GPU, clips sharded over ranks, ONE all-reduce per step.

Reference behaviour (pyskl/apis/train.py:94-102): mmcv ``MMDistributedDataParallel`` = torch DDP with
``broadcast_buffers=False`` and ``find_unused_parameters=True`` (20 ``conv2_se`` tensors never get a
gradient), BatchNorm statistics stay rank-local.  Here: parameters and gradients live in two flat fp32
buffers (5.5 MB each for DS-STGCN), every ``p.grad`` is a view into the gradient buffer (autograd
accumulates in place; unused parameters simply stay zero), and the exchange step is one RCCL
all-reduce (mean) of that buffer over xGMI — latency-bound (~50 us), no bucketing needed.
"""
import torch
import torch.distributed as dist


class FlatParams:
    """Re-homes a module's parameters and grads into contiguous flat buffers (views keep the module API)."""

    def __init__(self, module, gather=False):
        """gather=False: every ``p.grad`` is a view of the flat gradient buffer and autograd accumulates into it in
        place (one small add kernel per parameter tensor per step).  gather=True: ``zero_grad`` drops the grads,
        autograd hands each parameter its gradient tensor as is, and ``collect_grads`` packs them into the flat buffer
        with a few multi-tensor copies (~360 fewer launches per DS-STGCN step)."""
        self.module = module
        self.gather = gather
        self.params = [p for p in module.parameters() if p.requires_grad]
        if not self.params:
            raise ValueError('module has no trainable parameters')
        dev, dt = self.params[0].device, self.params[0].dtype
        total = sum(p.numel() for p in self.params)
        self.flat_p = torch.empty(total, device=dev, dtype=dt)
        self.flat_g = torch.zeros(total, device=dev, dtype=dt)
        # layout order: parameters() order, except that the groups a module declares in flat_groups() (tensors its
        # forward concatenates) are placed back to back at the position of their first member
        index = {id(p): i for i, p in enumerate(self.params)}
        group_of = {}
        for m in module.modules():
            for grp in (m.flat_groups() if hasattr(m, 'flat_groups') else []):
                ids = [index[id(p)] for p in grp if id(p) in index]
                if len(ids) == len(grp) and len(ids) > 1 and not any(i in group_of for i in ids):
                    for i in ids:
                        group_of[i] = ids
        order, placed = [], set()
        for i in range(len(self.params)):
            for j in (group_of.get(i, [i]) if i not in placed else []):
                if j not in placed:
                    placed.add(j)
                    order.append(j)
        self.slices = [None] * len(self.params)
        off = 0
        with torch.no_grad():
            for i in order:
                p = self.params[i]
                n = p.numel()
                self.flat_p[off:off + n].copy_(p.detach().reshape(-1))
                p.data = self.flat_p[off:off + n].view(p.shape)
                p.grad = self.flat_g[off:off + n].view(p.shape)
                self.slices[i] = (off, n)
                off += n
        self.numel = total
        self.views = [self.flat_g[off:off + n].view(p.shape) for p, (off, n) in zip(self.params, self.slices)]
        self.flat_views = [self.flat_g[off:off + n] for (off, n) in self.slices]

    def zero_grad(self):
        if self.gather:
            for p in self.params:
                p.grad = None
        else:
            self.flat_g.zero_()

    @torch.no_grad()
    def collect_grads(self):
        """gather mode: pack the gradients autograd produced into the flat buffer and re-point every ``p.grad`` at its
        view (parameters that received none — the dead conv2_se tensors — read as zero).  No-op otherwise."""
        if not self.gather:
            return
        base, esz = self.flat_g.data_ptr(), self.flat_g.element_size()
        first = next(((p, off) for p, (off, _) in zip(self.params, self.slices) if p.grad is not None), None)
        if first is not None and first[0].grad.data_ptr() == base + first[1] * esz:
            raise RuntimeError('FlatParams(gather=True): a gradient still aliases the flat buffer — call zero_grad() '
                               'before every backward (a second backward would accumulate into the views and this '
                               'call would wipe it)')
        if self.flat_g.is_cuda:
            # one launch writes the whole buffer: a copy per gradient, zeros where none arrived
            self._pack_cuda([(p.grad, off, n) for p, (off, n) in zip(self.params, self.slices)])
        else:
            self.flat_g.zero_()
            live = [(p.grad, off, n) for p, (off, n) in zip(self.params, self.slices) if p.grad is not None]
            if live:
                torch._foreach_copy_([self.flat_g[off:off + n] for _, off, n in live], [g.reshape(-1) for g, _, _ in live])
        for p, v in zip(self.params, self.views):
            p.grad = v

    def check_views(self):
        """True while every p.grad still aliases the flat buffer (autograd accumulates in place)."""
        base = self.flat_g.data_ptr()
        return all(p.grad is not None and p.grad.data_ptr() == base + off * self.flat_g.element_size()
                   for p, (off, _) in zip(self.params, self.slices))


    _PACK_SLOTS = 4

    def _pack_tables(self, k):
        """A (pinned host, device) table pair for one dsgcn_pack call.  The H2D copy of the table is asynchronous: the
        DMA reads the pinned memory when it EXECUTES, so a table must not be rewritten before its copy has run (the
        host may be a whole step ahead of the GPU: nothing in the step syncs).  Eager calls rotate over a few slots and
        wait on the slot's copy event before reusing it; a call under hipGraph capture gets tables of its own that are
        never rewritten (every replay re-reads them; the gradient addresses they hold live in the graph's pool)."""
        cap = max(k, len(self.params))
        dev = self.flat_g.device

        def new_slot():
            host = torch.empty((3, cap), dtype=torch.int64).pin_memory()
            return dict(host=host, dev=torch.empty_like(host, device=dev), event=None)

        if torch.cuda.is_current_stream_capturing():
            # the slot reserved by an earlier eager call (pinned allocation is not a capturable operation in the
            # default capture mode); further captures allocate (works under capture_error_mode='thread_local')
            slot = self.__dict__.pop('_pack_reserved', None) or new_slot()
            self.__dict__.setdefault('_pack_graph_slots', []).append(slot)      # alive as long as the graph may replay
            return slot
        if self.__dict__.get('_pack_reserved') is None:
            self._pack_reserved = new_slot()
        slots = self.__dict__.setdefault('_pack_slots', [])
        self._pack_next = (getattr(self, '_pack_next', -1) + 1) % self._PACK_SLOTS
        if len(slots) <= self._pack_next:
            slots.append(new_slot())
        slot = slots[self._pack_next]
        if slot['host'].shape[1] < cap:
            slots[self._pack_next] = slot = new_slot()
        if slot['event'] is not None:
            slot['event'].synchronize()
        return slot

    def _pack_cuda(self, live):
        """One HIP launch (dsgcn_pack_fill) instead of one blit per tensor and a fill: a (pointer, offset, length) table of
        EVERY parameter goes to the device through pinned staging memory (see _pack_tables for its lifetime rules)."""
        from . import native
        k = len(live)
        grads = [g if (g is None or g.is_contiguous()) else g.contiguous() for g, _, _ in live]
        slot = self._pack_tables(k)
        slot['keep'] = grads                                      # alive until the slot is reused (>= one full step)
        tab = slot['host'].numpy()
        tab[0, :k] = [0 if g is None else g.data_ptr() for g in grads]         # 0: no gradient arrived -> zeros
        tab[1, :k] = [off for _, off, _ in live]
        tab[2, :k] = [n for _, _, n in live]
        slot['dev'].copy_(slot['host'], non_blocking=True)
        if not torch.cuda.is_current_stream_capturing():
            slot['event'] = torch.cuda.Event()
            slot['event'].record()
        st = torch.cuda.current_stream().cuda_stream
        rc = native.lib().dsgcn_pack_fill(slot['dev'][0].data_ptr(), slot['dev'][1].data_ptr(), slot['dev'][2].data_ptr(), k,
                                          self.flat_g.data_ptr(), st)
        native.check(rc, 'dsgcn_pack_fill')


class FlatDataParallel:
    """Gradient averaging across ranks for a ``FlatParams`` (no-op at world size 1).  At wrap time rank 0's parameters
    AND buffers (BatchNorm running statistics, ``num_batches_tracked``, ``backbone.A``) are broadcast once — what torch
    DDP's constructor does with the module state (``_sync_module_states``: parameters and buffers) even under the
    reference's ``broadcast_buffers=False`` (pyskl/apis/train.py:98-102), which only switches the PER-STEP buffer sync
    off: BatchNorm statistics then evolve rank-locally, as here."""

    def __init__(self, flat, process_group=None, broadcast_params=True):
        self.flat = flat
        self.group = process_group
        self.world = dist.get_world_size(process_group) if dist.is_available() and dist.is_initialized() else 1
        if self.world > 1 and broadcast_params:
            dist.broadcast(self.flat.flat_p, src=0, group=process_group)     # C1b: params rank0 -> all
            self._broadcast_buffers()

    @torch.no_grad()
    def _broadcast_buffers(self):
        """All module buffers in two packed collectives (floating / integer) instead of one per tensor."""
        bufs = [b for b in self.flat.module.buffers() if b.numel() > 0]
        for pick, dt in ((lambda b: b.dtype.is_floating_point, self.flat.flat_p.dtype), (lambda b: not b.dtype.is_floating_point, torch.int64)):
            grp = [b for b in bufs if pick(b)]
            if not grp:
                continue
            packed = torch.cat([b.detach().reshape(-1).to(dt) for b in grp])
            dist.broadcast(packed, src=0, group=self.group)
            off = 0
            for b in grp:
                b.copy_(packed[off:off + b.numel()].view_as(b).to(b.dtype))
                off += b.numel()

    def allreduce_grads(self):
        """The step's ONE collective.  On RCCL the mean is the collective's own reduction (``ReduceOp.AVG``): no scale
        launch in the only un-graphed stretch of the step.  gloo (the CPU tests) has no AVG: sum, then scale."""
        if self.world > 1:
            if dist.get_backend(self.group) == 'nccl':
                dist.all_reduce(self.flat.flat_g, op=dist.ReduceOp.AVG, group=self.group)
            else:
                dist.all_reduce(self.flat.flat_g, op=dist.ReduceOp.SUM, group=self.group)
                self.flat.flat_g.mul_(1.0 / self.world)


def shard_batch(batch_size, rank, world):
    """Contiguous clip range of this rank: [lo, hi) — weak scaling keeps hi-lo fixed per GPU."""
    per = batch_size // world
    return rank * per, (rank + 1) * per
