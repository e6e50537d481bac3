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
        dst = [v for v, p in zip(self.views, self.params) if p.grad is not None]
        src = [p.grad for p in self.params if p.grad is not None]
        if dst:
            torch._foreach_copy_(dst, src)
        for p, v in zip(self.params, self.views):
            p.grad = v

    def check_views(self):
        """True while every p.grad still aliases the flat buffer (autograd accumulates in place)."""
        base = self.flat_g.data_ptr()
        return all(p.grad is not None and p.grad.data_ptr() == base + off * self.flat_g.element_size()
                   for p, (off, _) in zip(self.params, self.slices))


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
