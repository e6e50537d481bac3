"""Classification heads (reference: pyskl/models/heads/base.py:11-84, simple_head.py:12-140).

GCNHead: mean over (T,V), mean over persons M, dropout(0), Linear.  ``loss`` adds top-1/top-5
accuracy like the reference, but ranks on the device (one tiny top-k) instead of a D2H copy +
numpy argsort every step (heads/base.py:67-72 forces a sync per iteration); the values are the
same numbers (ties aside) and stay tensors, as in the reference's returned dict."""
from abc import ABCMeta, abstractmethod

import torch
import torch.nn as nn

from .builder import HEADS, build_loss


class BaseHead(nn.Module, metaclass=ABCMeta):

    def __init__(self, num_classes, in_channels, loss_cls=dict(type='CrossEntropyLoss', loss_weight=1.0),
                 multi_class=False, label_smooth_eps=0.0):
        super().__init__()
        self.num_classes = num_classes
        self.in_channels = in_channels
        self.loss_cls = build_loss(loss_cls)
        self.multi_class = multi_class
        self.label_smooth_eps = label_smooth_eps

    @abstractmethod
    def init_weights(self):
        pass

    @abstractmethod
    def forward(self, x):
        pass

    def loss(self, cls_score, label, **kwargs):
        losses = dict()
        if label.shape == torch.Size([]):
            label = label.unsqueeze(0)
        elif label.dim() == 1 and label.size()[0] == self.num_classes and cls_score.size()[0] == 1:
            label = label.unsqueeze(0)
        if not self.multi_class and cls_score.size() != label.size():
            with torch.no_grad():
                kmax = min(5, cls_score.shape[1])
                top = cls_score.detach().topk(kmax, dim=1).indices
                hit = top == label.view(-1, 1)
                losses['top1_acc'] = hit[:, :1].any(1).double().mean()
                losses['top5_acc'] = hit.any(1).double().mean()
        elif self.multi_class and self.label_smooth_eps != 0:
            label = ((1 - self.label_smooth_eps) * label + self.label_smooth_eps / self.num_classes)
        loss_cls = self.loss_cls(cls_score, label, **kwargs)
        if isinstance(loss_cls, dict):
            losses.update(loss_cls)
        else:
            losses['loss_cls'] = loss_cls
        return losses


@HEADS.register_module()
class SimpleHead(BaseHead):

    def __init__(self, num_classes, in_channels, loss_cls=dict(type='CrossEntropyLoss'), dropout=0.5, init_std=0.01,
                 mode='3D', **kwargs):
        super().__init__(num_classes, in_channels, loss_cls, **kwargs)
        self.dropout_ratio = dropout
        self.init_std = init_std
        self.dropout = nn.Dropout(p=self.dropout_ratio) if self.dropout_ratio != 0 else None
        assert mode in ['3D', 'GCN', '2D']
        if mode != 'GCN':
            raise NotImplementedError('only the skeleton (GCN) pooling mode is on this path')
        self.mode = mode
        self.in_c = in_channels
        self.fc_cls = nn.Linear(self.in_c, num_classes)

    def init_weights(self):
        nn.init.normal_(self.fc_cls.weight, 0, self.init_std)
        nn.init.constant_(self.fc_cls.bias, 0)

    def forward(self, x):
        if isinstance(x, list):
            for item in x:
                assert len(item.shape) == 2
            x = torch.stack([item.mean(dim=0) for item in x])
        if len(x.shape) != 2:
            N, M, C, T, V = x.shape
            x = x.reshape(N * M, C, T * V).mean(-1).reshape(N, M, C).mean(dim=1)
        assert x.shape[1] == self.in_c
        if self.dropout is not None:
            x = self.dropout(x)
        return self.fc_cls(x)


@HEADS.register_module()
class GCNHead(SimpleHead):

    def __init__(self, num_classes, in_channels, loss_cls=dict(type='CrossEntropyLoss'), dropout=0., init_std=0.01,
                 **kwargs):
        super().__init__(num_classes, in_channels, loss_cls=loss_cls, dropout=dropout, init_std=init_std, mode='GCN',
                         **kwargs)
