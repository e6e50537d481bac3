"""Checkpoint files in the reference's format (mmcv ``save_checkpoint`` as driven by
pyskl/core/local_runner/epoch_based_sparse_runner.py:145-190): a ``torch.save``d dict
``{'meta': {...}, 'state_dict': OrderedDict, 'optimizer': {...}}``; DDP-wrapped models carry a ``module.`` key prefix
(mmcv strips it on load).  Upstream PYSKL / DS-GCN ``.pth`` files load into the classes of this package unchanged
because the state_dict keys are the reference's (tests/test_host_api.py::test_state_dict_contract)."""
import time
from collections import OrderedDict

import torch


def _strip_prefix(state_dict, prefix='module.'):
    if state_dict and all(k.startswith(prefix) for k in state_dict):
        return OrderedDict((k[len(prefix):], v) for k, v in state_dict.items())
    return state_dict


def load_checkpoint(model, filename, map_location='cpu', strict=False, revise_keys=((r'^module\.', ''),)):
    """mmcv.runner.load_checkpoint semantics: accepts a bare state_dict or a dict with 'state_dict'; returns the
    checkpoint dict.  With ``strict=False`` missing / unexpected keys are reported in the returned dict."""
    import re
    ckpt = torch.load(filename, map_location=map_location, weights_only=False)
    if not isinstance(ckpt, dict):
        raise RuntimeError(f'No state_dict found in checkpoint file {filename}')
    sd = ckpt['state_dict'] if 'state_dict' in ckpt else ckpt
    for pat, rep in revise_keys:
        sd = OrderedDict((re.sub(pat, rep, k), v) for k, v in sd.items())
    res = model.load_state_dict(sd, strict=strict)
    if isinstance(ckpt, dict) and 'state_dict' in ckpt:
        ckpt['missing_keys'], ckpt['unexpected_keys'] = list(res.missing_keys), list(res.unexpected_keys)
    return ckpt


def save_checkpoint(model, filename, optimizer=None, meta=None, create_symlink=False):
    """Writes {'meta', 'state_dict', 'optimizer'} with CPU tensors, like mmcv.runner.save_checkpoint.  ``meta`` carries
    the runner's ``epoch`` / ``iter`` (epoch_based_sparse_runner.py:175).  ``create_symlink`` also points
    ``latest.pth`` in the same directory at the file (epoch_based_sparse_runner.py:185-188) — what auto-resume finds."""
    import os
    meta = dict(meta or {})
    meta.setdefault('time', time.asctime())
    sd = OrderedDict((k, v.detach().cpu()) for k, v in _strip_prefix(model.state_dict()).items())
    ckpt = {'meta': meta, 'state_dict': sd}
    if optimizer is not None:
        ckpt['optimizer'] = optimizer.state_dict()
    torch.save(ckpt, filename)
    if create_symlink:
        dst = os.path.join(os.path.dirname(os.path.abspath(filename)), 'latest.pth')
        if os.path.lexists(dst):
            os.remove(dst)
        os.symlink(os.path.basename(filename), dst)
    return filename


def resume(model, optimizer, filename, map_location='cpu'):
    """mmcv ``BaseRunner.resume``: model weights, optimizer state and the (epoch, iter) counters of a checkpoint.
    -> meta dict (``meta['epoch']`` completed epochs, ``meta['iter']`` completed iterations)."""
    ckpt = load_checkpoint(model, filename, map_location=map_location, strict=True)
    if optimizer is not None and 'optimizer' in ckpt:
        optimizer.load_state_dict(ckpt['optimizer'])
    return dict(ckpt.get('meta', {}))


def find_resume(work_dir, resume_from=None, auto_resume=True):
    """The checkpoint a run in ``work_dir`` continues from (reference tools/train.py:82-86): an explicit ``resume_from``
    wins; otherwise ``work_dir/latest.pth`` when ``auto_resume`` and it exists; else None."""
    import os
    if resume_from is not None:
        return resume_from
    path = os.path.join(work_dir, 'latest.pth')
    return path if auto_resume and os.path.exists(path) else None


def _fold(conv, bn):
    import torch.nn as nn
    scale = bn.weight / torch.sqrt(bn.running_var + bn.eps)
    conv.weight.mul_(scale.view(-1, *([1] * (conv.weight.dim() - 1))))
    bias = conv.bias if conv.bias is not None else torch.zeros_like(bn.running_mean)
    new_bias = (bias - bn.running_mean) * scale + bn.bias
    if conv.bias is None:
        conv.bias = nn.Parameter(new_bias)
    else:
        conv.bias.copy_(new_bias)
    bn.weight.fill_(1.0)
    bn.bias.zero_()
    bn.running_mean.zero_()
    bn.running_var.fill_(1.0 - bn.eps)          # so that 1 / sqrt(var + eps) is exactly 1


@torch.no_grad()
def fuse_conv_bn(module):
    """Inference-time folding of ``Conv2d -> BatchNorm2d`` pairs (what the reference's ``tools/test.py --fuse-conv-bn`` asks
    of mmcv.cnn.fuse_conv_bn, tools/test.py:98-99): the conv takes ``w * gamma / sqrt(var + eps)`` and ``(b - mean) * gamma /
    sqrt(var + eps) + beta``, the BatchNorm becomes the identity (kept as a BatchNorm2d so the module tree and state_dict
    keys do not change).  Only pairs where the BatchNorm really consumes the conv's output are folded: neighbours inside an
    ``nn.Sequential`` (execution order = registration order), and the pairs a unit declares in ``fusable_pairs()`` —
    registration order alone is not dataflow (in ``dggcn`` / ``dgphgcn1`` the projection convs are registered right before
    ``self.bn``, which normalises ``post(...)``; with equal widths a by-order fold would silently change the outputs).
    Eval mode only: with training statistics the fold is meaningless.  Returns ``module``."""
    import torch.nn as nn
    for m in module.modules():
        pairs = []
        if isinstance(m, nn.Sequential):
            kids = list(m.children())
            pairs = [(a, b) for a, b in zip(kids[:-1], kids[1:])]
        elif hasattr(m, 'fusable_pairs'):
            pairs = list(m.fusable_pairs())
        for conv, bn in pairs:
            if (isinstance(conv, nn.Conv2d) and isinstance(bn, nn.BatchNorm2d) and bn.track_running_stats
                    and conv.out_channels == bn.num_features):
                _fold(conv, bn)
    return module
