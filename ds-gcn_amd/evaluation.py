"""Accuracy metrics used on the training path (reference: pyskl/core/evaluation.py:85-126)."""
import numpy as np


def top_k_accuracy(scores, labels, topk=(1, )):
    """Fraction of samples whose label is among the k highest scores, per k."""
    scores = np.asarray(scores)
    labels = np.array(labels)[:, np.newaxis]
    res = []
    for k in topk:
        max_k_preds = np.argsort(scores, axis=1)[:, -k:][:, ::-1]
        match = np.logical_or.reduce(max_k_preds == labels, axis=1)
        res.append(match.sum() / match.shape[0])
    return res


def confusion_matrix(y_pred, y_real, normalize=None):
    if normalize not in ['true', 'pred', 'all', None]:
        raise ValueError("normalize must be one of {'true', 'pred', 'all', None}")
    y_pred = np.asarray(y_pred)
    y_real = np.asarray(y_real)
    if y_pred.dtype == np.int32:
        y_pred = y_pred.astype(np.int64)
    if y_real.dtype == np.int32:
        y_real = y_real.astype(np.int64)
    if not np.issubdtype(y_pred.dtype, np.integer) or not np.issubdtype(y_real.dtype, np.integer):
        raise TypeError('y_pred and y_real must be integer arrays')
    label_set = np.unique(np.concatenate((y_pred, y_real)))
    num_labels = len(label_set)
    max_label = label_set[-1]
    label_map = np.zeros(max_label + 1, dtype=np.int64)
    for i, label in enumerate(label_set):
        label_map[label] = i
    cm = np.bincount(num_labels * label_map[y_real] + label_map[y_pred],
                     minlength=num_labels**2).reshape(num_labels, num_labels)
    with np.errstate(all='ignore'):
        if normalize == 'true':
            cm = cm / cm.sum(axis=1, keepdims=True)
        elif normalize == 'pred':
            cm = cm / cm.sum(axis=0, keepdims=True)
        elif normalize == 'all':
            cm = cm / cm.sum()
        cm = np.nan_to_num(cm)
    return cm


def mean_class_accuracy(scores, labels):
    """-> (mean class accuracy, confusion matrix): this fork returns the pair (pyskl/core/evaluation.py:85-104; the
    dataset's ``evaluate`` unpacks it, datasets/base.py:194,201)."""
    pred = np.argmax(scores, axis=1)
    cm = confusion_matrix(pred, labels).astype(float)
    cls_cnt = cm.sum(axis=1)
    cls_hit = np.diag(cm)
    return np.mean([hit / cnt if cnt else 0.0 for cnt, hit in zip(cls_cnt, cls_hit)]), cm
