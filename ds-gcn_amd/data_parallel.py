"""Data-parallel training plumbing: one process per GPU, clips sharded over ranks, ONE all-reduce per step.

Reference behaviour (pyskl/apis/train.py:94-102): mmcv ``MMDistributedDataParallel`` = torch DDP with
``broadcast_buffers=False`` and ``find_unused_parameters=True`` (20 ``conv2_se`` tensors never get a
gradient), BatchNorm statistics stay rank-local.  Here: parameters and gradients live in two flat fp32
buffers (5.5 MB each for DS-STGCN), every ``p.grad`` is a view into the gradient buffer (autograd
accumulates in place; unused parameters simply stay zero), and the exchange step is one RCCL
all-reduce of that buffer over xGMI followed by a scale — latency-bound (~50 us), no bucketing needed.
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
        off = 0
        self.slices = []
        with torch.no_grad():
            for p in self.params:
                n = p.numel()
                self.flat_p[off:off + n].copy_(p.detach().reshape(-1))
                p.data = self.flat_p[off:off + n].view(p.shape)
                p.grad = self.flat_g[off:off + n].view(p.shape)
                self.slices.append((off, n))
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
        self.flat_g.zero_()
        live = [(p.grad, off, n) for p, (off, n) in zip(self.params, self.slices) if p.grad is not None]
        if live and self.flat_g.is_cuda:
            self._pack_cuda(live)
        elif live:
            torch._foreach_copy_([self.flat_g[off:off + n] for _, off, n in live], [g.reshape(-1) for g, _, _ in live])
        for p, v in zip(self.params, self.views):
            p.grad = v

    def check_views(self):
        """True while every p.grad still aliases the flat buffer (autograd accumulates in place)."""
        base = self.flat_g.data_ptr()
        return all(p.grad is not None and p.grad.data_ptr() == base + off * self.flat_g.element_size()
                   for p, (off, _) in zip(self.params, self.slices))


    def _pack_cuda(self, live):
        """One HIP launch (dsgcn_pack) instead of one blit per tensor.  The (pointer, offset, length) table goes to the
        device through a pinned staging buffer that stays alive, so the copy is capturable in a hipGraph (replays
        see the same gradient addresses: they live in the graph's private pool)."""
        import numpy as np
        from . import native
        k = len(live)
        if getattr(self, '_pack_host', None) is None or self._pack_host.shape[1] < k:
            self._pack_host = torch.empty((3, max(k, len(self.params))), dtype=torch.int64).pin_memory()
            self._pack_dev = torch.empty_like(self._pack_host, device=self.flat_g.device)
            self._pack_len = torch.empty(self._pack_host.shape[1], dtype=torch.int32, device=self.flat_g.device)
        grads = [g if g.is_contiguous() else g.contiguous() for g, _, _ in live]
        self._pack_keep = grads                                   # alive until the kernel has run
        tab = self._pack_host.numpy()
        tab[0, :k] = [g.data_ptr() for g in grads]
        tab[1, :k] = [off for _, off, _ in live]
        tab[2, :k] = [n for _, _, n in live]
        self._pack_dev.copy_(self._pack_host, non_blocking=True)
        self._pack_len.copy_(self._pack_dev[2])
        st = torch.cuda.current_stream().cuda_stream
        rc = native.lib().dsgcn_pack(self._pack_dev[0].data_ptr(), self._pack_dev[1].data_ptr(),
                                     self._pack_len.data_ptr(), k, self.flat_g.data_ptr(), st)
        native.check(rc, 'dsgcn_pack')


class FlatDataParallel:
    """Gradient averaging across ranks for a ``FlatParams`` (no-op at world size 1)."""

    def __init__(self, flat, process_group=None, broadcast_params=True):
        self.flat = flat
        self.group = process_group
        self.world = dist.get_world_size(process_group) if dist.is_available() and dist.is_initialized() else 1
        if self.world > 1 and broadcast_params:
            dist.broadcast(self.flat.flat_p, src=0, group=process_group)     # C1b: params rank0 -> all

    def allreduce_grads(self):
        if self.world > 1:
            dist.all_reduce(self.flat.flat_g, op=dist.ReduceOp.SUM, group=self.group)
            self.flat.flat_g.mul_(1.0 / self.world)


def shard_batch(batch_size, rank, world):
    """Contiguous clip range of this rank: [lo, hi) — weak scaling keeps hi-lo fixed per GPU."""
    per = batch_size // world
    return rank * per, (rank + 1) * per
