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


@torch.no_grad()
def fuse_conv_bn(module):
    """Inference-time folding of every ``Conv2d -> BatchNorm2d`` pair that sits back to back in a container (what the
    reference's ``tools/test.py --fuse-conv-bn`` does through mmcv.cnn.fuse_conv_bn, tools/test.py:98-99): the conv takes
    ``w * gamma / sqrt(var + eps)`` and ``(b - mean) * gamma / sqrt(var + eps) + beta``, the BatchNorm becomes the identity
    (gamma 1, beta 0, mean 0, var 1 - eps... kept as a BatchNorm2d so the module tree and state_dict keys do not change).
    Eval mode only: with training statistics the fold is meaningless.  Returns ``module``."""
    import torch.nn as nn
    last_conv = None
    for name, child in module.named_children():
        if isinstance(child, nn.BatchNorm2d):
            if last_conv is not None and child.track_running_stats and last_conv.out_channels == child.num_features:
                scale = child.weight / torch.sqrt(child.running_var + child.eps)
                last_conv.weight.mul_(scale.view(-1, 1, 1, 1))
                bias = last_conv.bias if last_conv.bias is not None else torch.zeros_like(child.running_mean)
                new_bias = (bias - child.running_mean) * scale + child.bias
                if last_conv.bias is None:
                    last_conv.bias = nn.Parameter(new_bias)
                else:
                    last_conv.bias.copy_(new_bias)
                child.weight.fill_(1.0)
                child.bias.zero_()
                child.running_mean.zero_()
                child.running_var.fill_(1.0 - child.eps)          # so that 1 / sqrt(var + eps) is exactly 1
            last_conv = None
        elif isinstance(child, nn.Conv2d):
            last_conv = child
        else:
            last_conv = None
            fuse_conv_bn(child)
    return module
