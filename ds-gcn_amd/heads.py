"""Classification head of the skeleton recognizers, behind the reference's registry names ``GCNHead`` /
``SimpleHead`` and constructor kwargs (pyskl/models/heads/simple_head.py:12-140, heads/base.py:11-84).

What the path computes: global average over (T, V), mean over the M persons, ``Linear(in_channels, classes)``;
``loss()`` = the configured loss + top-1 / top-5 accuracy as log scalars.  Two deliberate differences from the
reference's host code: (i) the ranking runs on the device (one small ``topk``) instead of a device->host copy and a
numpy argsort per step (heads/base.py:67-72 syncs every iteration); the numbers are the same, ties aside; (ii) the
2D/3D pooling modes, list inputs, multi-label and label-smoothing branches are not reached by any skeleton config
and are rejected rather than carried along; (iii) ``forward_loss`` is the fused form of ``loss(forward(x), label)`` that
``RecognizerGCN.forward_train`` calls (the two methods stay for everything else)."""
import torch
import torch.nn as nn

from . import kernels
from .builder import HEADS, build_loss


@HEADS.register_module()
class SimpleHead(nn.Module):

    def __init__(self, num_classes, in_channels, loss_cls=dict(type='CrossEntropyLoss'), dropout=0.5, init_std=0.01,
                 mode='3D', multi_class=False, label_smooth_eps=0.0):
        super().__init__()
        assert mode in ['3D', 'GCN', '2D']
        if mode != 'GCN':
            raise NotImplementedError('only the skeleton (GCN) pooling mode is on this path')
        if multi_class or label_smooth_eps:
            raise NotImplementedError('multi-label heads / label smoothing are outside the skeleton configs')
        self.num_classes = num_classes
        self.in_channels = self.in_c = in_channels
        self.mode = mode
        self.multi_class = False
        self.label_smooth_eps = 0.0
        self.loss_cls = build_loss(loss_cls)
        self.dropout_ratio = dropout
        self.init_std = init_std
        self.dropout = nn.Dropout(p=dropout) if dropout != 0 else None
        self.fc_cls = nn.Linear(in_channels, num_classes)

    def init_weights(self):
        nn.init.normal_(self.fc_cls.weight, 0, self.init_std)
        nn.init.constant_(self.fc_cls.bias, 0)

    def forward(self, x):
        """x (N, M, C, T, V) backbone features (or their plane means (N, M, C), or pooled (N, C)) -> scores (N, classes)."""
        if x.dim() == 5:
            N, M, C = x.shape[:3]
            x = x.reshape(N * M, C, -1).mean(-1).reshape(N, M, C).mean(1)
        elif x.dim() == 3:                          # per-person plane means from ``backbone(x, pool=True)``
            x = x.mean(1)
        elif x.dim() != 2:
            raise NotImplementedError(f'GCN head expects (N, M, C, T, V), (N, M, C) or (N, C) features, got {tuple(x.shape)}')
        assert x.shape[1] == self.in_c
        if self.dropout is not None:
            x = self.dropout(x)
        return self.fc_cls(x)

    def forward_loss(self, x, label):
        """``loss(forward(x), label)`` as the training step runs it: person mean, ``fc_cls``, the cross entropy and both
        accuracies in three launches (``kernels.head_loss``) instead of ~25.  Heads with dropout or another loss take the
        two calls."""
        from .losses import CrossEntropyLoss
        if (not kernels.FUSED_ENDS or x.dim() not in (3, 5) or self.dropout is not None or type(self.loss_cls) is not CrossEntropyLoss
                or label.is_floating_point()):
            return self.loss(self(x), label)
        if label.dim() == 0:
            label = label[None]
        N, M, C = x.shape[:3]
        assert C == self.in_c
        if label.shape != (N,):
            raise NotImplementedError(f'CrossEntropyLoss: expects (N,) integer labels for (N, classes) scores, got '
                                      f'{tuple(label.shape)} for {N} clips')
        feat = x.reshape(N * M, C, -1).mean(-1) if x.dim() == 5 else x.reshape(N * M, C)
        loss, acc, _ = kernels.ops().head_loss(feat, self.fc_cls.weight, self.fc_cls.bias, label, M,
                                               self.loss_cls.loss_weight)
        return dict(top1_acc=acc[0], top5_acc=acc[1], loss_cls=loss)

    def loss(self, cls_score, label):
        """-> dict(top1_acc, top5_acc, loss_cls), all device tensors (no host sync)."""
        if label.dim() == 0:
            label = label[None]
        with torch.no_grad():
            top = cls_score.topk(min(5, cls_score.shape[1]), dim=1).indices
            hit = top == label.view(-1, 1)
            out = dict(top1_acc=hit[:, 0].double().mean(), top5_acc=hit.any(1).double().mean())
        out['loss_cls'] = self.loss_cls(cls_score, label)
        return out


@HEADS.register_module()
class GCNHead(SimpleHead):

    def __init__(self, num_classes, in_channels, loss_cls=dict(type='CrossEntropyLoss'), dropout=0., init_std=0.01,
                 **kwargs):
        super().__init__(num_classes, in_channels, loss_cls=loss_cls, dropout=dropout, init_std=init_std, mode='GCN',
                         **kwargs)
