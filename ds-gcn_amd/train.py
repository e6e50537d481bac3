"""Optimizer / schedule used by the reference's shipped configs (configs/_init_/lr_schedual.py:11-27):
SGD(lr=0.1, momentum=0.9, weight_decay=5e-4, nesterov=True) on ALL parameters (mmcv's default
constructor applies weight decay to BN/alpha/beta/A as well, quirk Q10), cosine annealing per iteration."""
import math

import torch

from . import kernels, native


class FlatSGD:
    """Nesterov SGD over ``FlatParams`` buffers: a handful of elementwise launches per step regardless of
    the 604 parameter tensors.  Matches torch.optim.SGD(nesterov=True, dampening=0) update-for-update.

    ``capturable=True`` keeps the learning rate in a one-element device tensor (``set_lr`` fills it): a ``step()`` captured
    in a hipGraph then follows the per-iteration schedule on replay instead of freezing the rate of the capture.  The
    momentum buffer is allocated once and only ever written in place (``load_state_dict`` included), so a captured
    ``step()`` keeps updating the live buffer after a resume."""

    def __init__(self, flat, lr=0.1, momentum=0.9, weight_decay=5e-4, nesterov=True, capturable=False):
        self.flat = flat
        self.lr = lr
        self.base_lr = lr
        self.momentum = momentum
        self.weight_decay = weight_decay
        self.nesterov = nesterov
        self.buf = None
        self.capturable = capturable
        self.lr_t = torch.full((1,), float(lr), device=flat.flat_p.device, dtype=flat.flat_p.dtype) if capturable else None
        if capturable and momentum:
            self.buf = torch.zeros_like(flat.flat_p)       # torch's first step sets buf = g; momentum * 0 + g is the same value

    def set_lr(self, lr):
        self.lr = float(lr)
        if self.lr_t is not None:
            self.lr_t.fill_(self.lr)

    @torch.no_grad()
    def step(self):
        p, g = self.flat.flat_p, self.flat.flat_g
        if self.capturable and p.is_cuda and p.dtype == torch.float32 and kernels.FUSED_ENDS:
            # one launch (csrc/head.hip k_sgd) instead of five elementwise passes over the flat buffers
            rc = native.lib().dsgcn_sgd_step(p.data_ptr(), g.data_ptr(), self.buf.data_ptr() if self.momentum else None,
                                             self.lr_t.data_ptr(), float(self.momentum), float(self.weight_decay),
                                             int(bool(self.nesterov)), p.numel(),
                                             torch.cuda.current_stream().cuda_stream)
            native.check(rc, 'dsgcn_sgd_step')
            return
        if self.weight_decay:
            g = g.add(p, alpha=self.weight_decay)
        if self.momentum:
            if self.buf is None:
                self.buf = g.clone()
            else:
                self.buf.mul_(self.momentum).add_(g)
            g = g.add(self.buf, alpha=self.momentum) if self.nesterov else self.buf
        if self.capturable:
            p.addcmul_(g, self.lr_t, value=-1.0)
        else:
            p.add_(g, alpha=-self.lr)

    def zero_grad(self):
        self.flat.zero_grad()

    def state_dict(self):
        """torch.optim.SGD's layout — ``{'state': {i: {'momentum_buffer': tensor}}, 'param_groups': [{...}]}`` with one
        entry per parameter tensor in ``module.parameters()`` order — so the ``optimizer`` entry of a checkpoint is
        interchangeable with the reference's (mmcv saves ``optimizer.state_dict()`` of a torch SGD).  CPU tensors."""
        state = {}
        if self.buf is not None and (not self.capturable or bool(self.buf.any())):     # all-zero = no step taken yet
            for i, (p, (off, n)) in enumerate(zip(self.flat.params, self.flat.slices)):
                state[i] = {'momentum_buffer': self.buf[off:off + n].view(p.shape).detach().cpu().clone()}
        group = dict(lr=self.lr, momentum=self.momentum, dampening=0, weight_decay=self.weight_decay,
                     nesterov=self.nesterov, maximize=False, foreach=None, differentiable=False, fused=None,
                     initial_lr=self.base_lr, params=list(range(len(self.flat.params))))
        return {'state': state, 'param_groups': [group]}

    def load_state_dict(self, sd):
        """Accepts the torch SGD layout (this class's own checkpoints and the reference's)."""
        groups = sd['param_groups']
        if len(groups) != 1 or len(groups[0]['params']) != len(self.flat.params):
            raise ValueError('FlatSGD.load_state_dict: expected one param group over '
                             f'{len(self.flat.params)} tensors, got {[len(g["params"]) for g in groups]}')
        g = groups[0]
        self.set_lr(g['lr'])
        self.base_lr = g.get('initial_lr', self.base_lr)
        self.momentum = g.get('momentum', self.momentum)
        self.weight_decay = g.get('weight_decay', self.weight_decay)
        self.nesterov = g.get('nesterov', self.nesterov)
        state = sd.get('state', {})
        if not state:
            if self.capturable and self.buf is not None:
                self.buf.zero_()
            else:
                self.buf = None
            return
        # in place when the buffer exists: a hipGraph captured around step() holds its address
        buf = self.buf if self.buf is not None else torch.zeros_like(self.flat.flat_p)
        buf.zero_()
        for i, pid in enumerate(g['params']):
            mb = state.get(pid, state.get(str(pid), {})).get('momentum_buffer')
            if mb is not None:
                off, n = self.flat.slices[i]
                buf[off:off + n].copy_(mb.reshape(-1))
        self.buf = buf


def cosine_lr(base_lr, it, total_iters, min_lr=0.0):
    """mmcv CosineAnnealingLrUpdaterHook(by_epoch=False)."""
    return min_lr + 0.5 * (base_lr - min_lr) * (1 + math.cos(math.pi * it / max(total_iters, 1)))


def step_lr(base_lr, progress, step, gamma=0.1, min_lr=None):
    """mmcv StepLrUpdaterHook.get_lr: ``progress`` = the 0-based epoch about to run (by_epoch=True); ``step`` an int
    (decay every ``step`` epochs) or a list of milestones (configs/stgcn/stgcn_vanilla_ntu60_xsub_3dkp/j.py:35)."""
    if isinstance(step, int):
        exp = progress // step
    else:
        exp = len(step)
        for i, s in enumerate(step):
            if progress < s:
                exp = i
                break
    lr = base_lr * gamma ** exp
    if min_lr is not None:
        lr = max(lr, min_lr)
    return lr
