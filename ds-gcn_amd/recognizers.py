"""Recognizer API (reference: pyskl/models/recognizers/base.py:21-205, recognizergcn.py:16-148).

Same entry points and return conventions — ``forward(keypoint, label, return_loss)``,
``forward_train`` -> dict(top1_acc, top5_acc, loss_cls), ``forward_test`` -> numpy class
probabilities averaged over clips, ``train_step`` -> dict(loss, losses, log_vars, num_samples) —
with one deliberate difference: ``_parse_losses`` reduces all log scalars in ONE all-reduce and
ONE device->host read instead of one collective + ``.item()`` per scalar (base.py:150-156: four
syncs per iteration), because at MI355X step times those syncs bound weak scaling.
"""
from abc import ABCMeta, abstractmethod
from collections import OrderedDict

import numpy as np
import torch
import torch.distributed as dist
import torch.nn as nn
import torch.nn.functional as F

from . import builder
from .builder import RECOGNIZERS


class BaseRecognizer(nn.Module, metaclass=ABCMeta):

    def __init__(self, backbone, neck=None, cls_head=None, train_cfg=dict(), test_cfg=dict()):
        super().__init__()
        self.backbone = builder.build_backbone(backbone)
        if neck:
            raise NotImplementedError('necks are outside the DS-GCN hot path (no BASELINE config uses one)')
        self.neck = None
        self.cls_head = builder.build_head(cls_head) if cls_head else None
        train_cfg = dict() if train_cfg is None else train_cfg
        test_cfg = dict() if test_cfg is None else test_cfg
        assert isinstance(train_cfg, dict)
        assert isinstance(test_cfg, dict)
        self.train_cfg = train_cfg
        self.test_cfg = test_cfg
        self.max_testing_views = test_cfg.get('max_testing_views', None)
        self.init_weights()

    @property
    def with_cls_head(self):
        return getattr(self, 'cls_head', None) is not None

    @property
    def with_neck(self):
        return getattr(self, 'neck', None) is not None

    def init_weights(self):
        self.backbone.init_weights()
        if self.with_cls_head:
            self.cls_head.init_weights()

    def extract_feat(self, imgs):
        return self.backbone(imgs)

    def average_clip(self, cls_score):
        assert len(cls_score.shape) == 3  # (batch, clips, classes)
        average_clips = self.test_cfg.get('average_clips', 'prob')
        if average_clips not in ['score', 'prob', None]:
            raise ValueError(f'{average_clips} is not supported. Supported: ["score", "prob", None]')
        if average_clips is None:
            return cls_score
        if average_clips == 'prob':
            return F.softmax(cls_score, dim=2).mean(dim=1)
        return cls_score.mean(dim=1)

    @abstractmethod
    def forward_train(self, imgs, label, **kwargs):
        pass

    @abstractmethod
    def forward_test(self, imgs, **kwargs):
        pass

    def _parse_losses(self, losses, sync=True):
        """(loss, log_vars, losses).  ``sync=False`` keeps log_vars as device tensors (no host read)."""
        log_vars = OrderedDict()
        for name, value in losses.items():
            if isinstance(value, torch.Tensor):
                log_vars[name] = value.mean()
            elif isinstance(value, list):
                log_vars[name] = sum(v.mean() for v in value)
            else:
                raise TypeError(f'{name} is not a tensor or list of tensors')
        loss = sum(v for k, v in log_vars.items() if 'loss' in k)
        log_vars['loss'] = loss
        names = list(log_vars)
        packed = torch.stack([log_vars[k].detach().double() for k in names])
        if dist.is_available() and dist.is_initialized():
            packed = packed / dist.get_world_size()
            dist.all_reduce(packed)
        if sync:
            vals = packed.tolist()
            log_vars = OrderedDict((k, v) for k, v in zip(names, vals))
        else:
            log_vars = OrderedDict((k, packed[i]) for i, k in enumerate(names))
        return loss, log_vars, losses

    def forward(self, imgs, label=None, return_loss=True, **kwargs):
        if return_loss:
            if label is None:
                raise ValueError('Label should not be None.')
            return self.forward_train(imgs, label, **kwargs)
        return self.forward_test(imgs, **kwargs)

    def train_step(self, data_batch, optimizer=None, **kwargs):
        sync = kwargs.pop('sync_log_vars', True)
        kwargs.pop('current_epoch', None)
        kwargs.pop('total_epoch', None)
        losses = self(**data_batch, return_loss=True, **kwargs)
        loss, log_vars, losses = self._parse_losses(losses, sync=sync)
        return dict(loss=loss, losses=losses, log_vars=log_vars,
                    num_samples=len(next(iter(data_batch.values()))))

    def val_step(self, data_batch, optimizer=None, **kwargs):
        return self.train_step(data_batch, optimizer, **kwargs)


@RECOGNIZERS.register_module()
class RecognizerGCN(BaseRecognizer):

    def forward_train(self, keypoint, label, **kwargs):
        assert self.with_cls_head
        assert keypoint.shape[1] == 1
        if keypoint.dtype != torch.float:
            keypoint = keypoint.float()
        x = self.extract_feat(keypoint[:, 0])
        cls_score = self.cls_head(x)
        gt_label = label.squeeze(-1)
        losses = dict()
        losses.update(self.cls_head.loss(cls_score, gt_label))
        return losses

    def forward_test(self, keypoint, **kwargs):
        assert self.with_cls_head
        if keypoint.dtype != torch.float:
            keypoint = keypoint.float()
        bs, nc = keypoint.shape[:2]
        keypoint = keypoint.reshape((bs * nc, ) + keypoint.shape[2:])
        x = self.extract_feat(keypoint)
        if self.test_cfg.get('feat_ext', False) or self.test_cfg.get('score_ext', False):
            raise NotImplementedError('feature/score extraction modes are outside the hot path')
        cls_score = self.cls_head(x)
        cls_score = cls_score.reshape(bs, nc, cls_score.shape[-1])
        if 'average_clips' not in self.test_cfg:
            self.test_cfg['average_clips'] = 'prob'
        cls_score = self.average_clip(cls_score)
        return cls_score.data.cpu().numpy()

    def forward(self, keypoint, label=None, return_loss=True, **kwargs):
        if return_loss:
            if label is None:
                raise ValueError('Label should not be None.')
            return self.forward_train(keypoint, label, **kwargs)
        return self.forward_test(keypoint, **kwargs)
