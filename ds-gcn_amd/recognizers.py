"""``RecognizerGCN`` behind the reference's registry name, constructor and call conventions
(pyskl/models/recognizers/recognizergcn.py:16-148 over recognizers/base.py:21-205):

* ``model(keypoint=(N,1,M,T,V,C), label=(N,1), return_loss=True)`` -> ``dict(top1_acc, top5_acc, loss_cls)``;
* ``model(keypoint=(N,clips,M,T,V,C), return_loss=False)`` -> ``np.ndarray (N, classes)``: clip scores averaged as
  probabilities (``test_cfg['average_clips']`` = 'prob' default, 'score', or None for per-clip scores);
* ``model.train_step(data_batch, optimizer)`` -> ``dict(loss, losses, log_vars, num_samples)`` (what mmcv's runner
  calls, core/local_runner/epoch_based_sparse_runner.py:33-34).

Host-side differences that matter at MI355X step times (a step is ~15 ms; the reference spends four collectives and
four ``.item()`` syncs per step on its log scalars, base.py:150-156):

* ``train_step(..., sync_log_vars=True)`` (default, reference semantics): ONE packed all-reduce of all log scalars and
  ONE device->host read;
* ``train_step(..., sync_log_vars=False)``: no collective and no host read at all — ``log_vars`` are rank-local device
  tensors, safe inside a hipGraph capture; ``reduce_log_vars`` averages them over ranks whenever the caller logs.

Necks, feature/score extraction and multi-view test batching are not reached by the skeleton configs and raise."""
from collections import OrderedDict
from itertools import zip_longest

import torch
import torch.distributed as dist
import torch.nn as nn
import torch.nn.functional as F

from . import builder, kernels
from .builder import RECOGNIZERS


def _dist_world():
    return dist.get_world_size() if dist.is_available() and dist.is_initialized() else 1


def reduce_log_vars(log_vars, sync=True):
    """Average a dict of scalar device tensors over the ranks with one collective.  ``sync=True`` also reads them back
    (one device->host copy) and returns floats."""
    names = list(log_vars)
    packed = torch.stack([torch.as_tensor(log_vars[k]).detach().double() for k in names])
    world = _dist_world()
    if world > 1:
        if torch.cuda.is_available() and torch.cuda.is_current_stream_capturing():
            raise RuntimeError('reduce_log_vars issues a collective: call it outside hipGraph capture '
                               '(use train_step(..., sync_log_vars=False) inside the captured step)')
        packed = packed / world
        dist.all_reduce(packed)
    vals = packed.tolist() if sync else list(packed.unbind(0))
    return OrderedDict(zip(names, vals))


@RECOGNIZERS.register_module()
class RecognizerGCN(nn.Module):

    def __init__(self, backbone, neck=None, cls_head=None, train_cfg=dict(), test_cfg=dict()):
        super().__init__()
        if neck:
            raise NotImplementedError('necks are outside the DS-GCN hot path (no skeleton config uses one)')
        self.backbone = builder.build_backbone(backbone)
        self.cls_head = builder.build_head(cls_head) if cls_head else None
        self.train_cfg = dict(train_cfg or {})
        self.test_cfg = dict(test_cfg or {})
        for key in ('feat_ext', 'score_ext', 'max_testing_views'):
            if self.test_cfg.get(key):
                raise NotImplementedError(f'test_cfg[{key!r}] is outside the hot path')
        mode = self.test_cfg.setdefault('average_clips', 'prob')
        if mode not in ('score', 'prob', None):
            raise ValueError(f'{mode} is not supported. Supported: ["score", "prob", None]')
        self.init_weights()

    with_cls_head = property(lambda self: self.cls_head is not None)

    def init_weights(self):
        self.backbone.init_weights()
        if self.cls_head is not None:
            self.cls_head.init_weights()

    def extract_feat(self, keypoint):
        return self.backbone(keypoint)

    # ---- training ------------------------------------------------------------------------------------------
    def forward_train(self, keypoint, label):
        assert self.cls_head is not None
        assert keypoint.shape[1] == 1, 'training batches carry one clip per sample'
        fused = hasattr(self.cls_head, 'forward_loss')
        # a pooling head only reads the plane means of the features: ask the backbone for those
        pool = (fused and kernels.FUSED_ENDS and getattr(self.backbone, 'supports_pool', False)
                and getattr(self.cls_head, 'mode', None) == 'GCN')
        x = keypoint[:, 0].float()
        feat = self.backbone(x, pool=True) if pool else self.extract_feat(x)
        if fused:
            return self.cls_head.forward_loss(feat, label.squeeze(-1))
        return self.cls_head.loss(self.cls_head(feat), label.squeeze(-1))

    def train_step(self, data_batch, optimizer=None, sync_log_vars=True, **_runner_kwargs):
        losses = self(**data_batch, return_loss=True)
        # (recognizers/base.py:128-143 `_parse_losses`; the mean of a scalar and the `0 +` of sum() are launches here)
        log_vars = OrderedDict((k, v if v.dim() == 0 else v.mean()) for k, v in losses.items())
        loss = None
        for k, v in log_vars.items():
            if 'loss' in k:
                loss = v if loss is None else loss + v
        log_vars['loss'] = loss
        if sync_log_vars:
            log_vars = reduce_log_vars(log_vars)
        else:
            log_vars = OrderedDict((k, v.detach()) for k, v in log_vars.items())
        return dict(loss=loss, losses=losses, log_vars=log_vars, num_samples=len(next(iter(data_batch.values()))))

    val_step = train_step

    # ---- inference -----------------------------------------------------------------------------------------
    @torch.no_grad()
    def forward_test(self, keypoint):
        assert self.cls_head is not None
        N, clips = keypoint.shape[:2]
        feats = self.extract_feat(keypoint.float().flatten(0, 1))
        scores = self.cls_head(feats).view(N, clips, -1)
        mode = self.test_cfg['average_clips']
        if mode == 'prob':
            scores = F.softmax(scores, dim=2).mean(1)
        elif mode == 'score':
            scores = scores.mean(1)
        return scores.cpu().numpy()

    def forward(self, keypoint, label=None, return_loss=True):
        if not return_loss:
            return self.forward_test(keypoint)
        if label is None:
            raise ValueError('Label should not be None.')
        return self.forward_train(keypoint, label)


def gather_results(part, size):
    """Inference results of all ranks in dataset order (what mmcv's ``multi_gpu_test`` hands tools/test.py:107):
    rank r holds samples r, r+world, ... (``DistributedSampler``, padded), so the parts are interleaved and cut to
    ``size``.  Every rank gets the full list."""
    world = _dist_world()
    if world == 1:
        return list(part)[:size]
    parts = [None] * world
    dist.all_gather_object(parts, list(part))
    hole = object()
    out = []
    for group in zip_longest(*parts, fillvalue=hole):
        out.extend(r for r in group if r is not hole)
    return out[:size]
