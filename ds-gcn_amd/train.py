"""Optimizer / schedule used by the reference's shipped configs (configs/_init_/lr_schedual.py:11-27):
SGD(lr=0.1, momentum=0.9, weight_decay=5e-4, nesterov=True) on ALL parameters (mmcv's default
constructor applies weight decay to BN/alpha/beta/A as well, quirk Q10), cosine annealing per iteration."""
import math

import torch


class FlatSGD:
    """Nesterov SGD over ``FlatParams`` buffers: a handful of elementwise launches per step regardless of
    the 604 parameter tensors.  Matches torch.optim.SGD(nesterov=True, dampening=0) update-for-update."""

    def __init__(self, flat, lr=0.1, momentum=0.9, weight_decay=5e-4, nesterov=True):
        self.flat = flat
        self.lr = lr
        self.base_lr = lr
        self.momentum = momentum
        self.weight_decay = weight_decay
        self.nesterov = nesterov
        self.buf = None

    @torch.no_grad()
    def step(self):
        p, g = self.flat.flat_p, self.flat.flat_g
        if self.weight_decay:
            g = g.add(p, alpha=self.weight_decay)
        if self.momentum:
            if self.buf is None:
                self.buf = g.clone()
            else:
                self.buf.mul_(self.momentum).add_(g)
            g = g.add(self.buf, alpha=self.momentum) if self.nesterov else self.buf
        p.add_(g, alpha=-self.lr)

    def zero_grad(self):
        self.flat.zero_grad()

    def state_dict(self):
        return dict(lr=self.lr, momentum_buffer=None if self.buf is None else self.buf.clone())

    def load_state_dict(self, sd):
        self.lr = sd['lr']
        self.buf = sd['momentum_buffer']


def cosine_lr(base_lr, it, total_iters, min_lr=0.0):
    """mmcv CosineAnnealingLrUpdaterHook(by_epoch=False)."""
    return min_lr + 0.5 * (base_lr - min_lr) * (1 + math.cos(math.pi * it / max(total_iters, 1)))
