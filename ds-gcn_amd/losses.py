"""The one loss the DS-GCN configs use: hard-label cross entropy, mean over the batch, times ``loss_weight``
(behaviour of the reference's ``CrossEntropyLoss`` on ``(N, classes)`` scores and ``(N,)`` int64 labels:
pyskl/models/losses/cross_entropy_loss.py:75-82 scaled by base.py:38-44).

Registered under the reference's name with the reference's constructor signature so that config dicts
(``loss_cls=dict(type='CrossEntropyLoss')``) build unchanged.  Soft labels, per-class weights and extra
``F.cross_entropy`` kwargs are not reached by any skeleton config and are rejected instead of carried along.
Host-side PyTorch: one (N, classes) tensor per step, not on the HBM-bound path."""
import torch.nn as nn
import torch.nn.functional as F

from .builder import LOSSES


@LOSSES.register_module()
class CrossEntropyLoss(nn.Module):

    def __init__(self, loss_weight=1.0, class_weight=None):
        super().__init__()
        if class_weight is not None:
            raise NotImplementedError('CrossEntropyLoss: class_weight is not used by any skeleton config')
        self.loss_weight = float(loss_weight)

    def forward(self, cls_score, label):
        if cls_score.dim() != 2 or label.shape != cls_score.shape[:1] or label.is_floating_point():
            raise NotImplementedError(
                f'CrossEntropyLoss: expects (N, classes) scores with (N,) integer labels, got '
                f'{tuple(cls_score.shape)} / {tuple(label.shape)} {label.dtype} (soft labels are outside this path)')
        loss = F.cross_entropy(cls_score, label)
        return loss if self.loss_weight == 1.0 else loss * self.loss_weight
