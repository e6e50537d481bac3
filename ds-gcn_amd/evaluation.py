"""Accuracy metrics used on the training / validation path.  Contract: pyskl/core/evaluation.py:21-126 — ``top_k_accuracy``
(a sample counts when its label is among the k best scores), ``confusion_matrix`` (rows = true class, columns = predicted,
over the classes that OCCUR in either list, in sorted order), ``mean_class_accuracy`` (this fork returns the pair
(mean of the per-class recalls over those classes, confusion matrix); datasets/base.py:194,201 unpacks it)."""
import numpy as np


def top_k_accuracy(scores, labels, topk=(1, )):
    """-> list, per k: the fraction of samples whose label is one of their k highest-scoring classes (ties resolved the
    way a stable ascending argsort resolves them: the later class wins)."""
    scores = np.asarray(scores)
    want = np.asarray(labels).reshape(-1, 1)
    ranked = np.argsort(scores, axis=1)                     # ascending: the best k are the last k columns
    return [float((ranked[:, max(scores.shape[1] - k, 0):] == want).any(axis=1).mean()) for k in topk]


def confusion_matrix(y_pred, y_real, normalize=None):
    """Counts[i, j] = samples of the i-th occurring class predicted as the j-th occurring class (classes = sorted union of
    both inputs).  ``normalize``: None | 'true' (rows sum to 1) | 'pred' (columns) | 'all'; empty rows / columns give 0."""
    if normalize not in ('true', 'pred', 'all', None):
        raise ValueError("normalize must be one of {'true', 'pred', 'all', None}")
    pred, real = np.asarray(y_pred), np.asarray(y_real)
    for name, a in (('y_pred', pred), ('y_real', real)):
        if not np.issubdtype(a.dtype, np.integer):
            raise TypeError(f'{name} must hold integer class ids, got {a.dtype}')
    classes, codes = np.unique(np.concatenate([real.ravel(), pred.ravel()]), return_inverse=True)
    k = len(classes)
    row, col = codes[:real.size], codes[real.size:]
    counts = np.zeros((k, k), dtype=np.int64)
    np.add.at(counts, (row, col), 1)
    if normalize is None:
        return counts
    total = {'true': counts.sum(1, keepdims=True), 'pred': counts.sum(0, keepdims=True), 'all': counts.sum()}[normalize]
    with np.errstate(all='ignore'):
        return np.nan_to_num(counts / total)


def mean_class_accuracy(scores, labels):
    """-> (mean over the occurring classes of hits / samples of that class — a class that only occurs as a prediction
    counts as 0 —, confusion matrix as float)."""
    counts = confusion_matrix(np.argmax(scores, axis=1), labels).astype(float)
    seen = counts.sum(axis=1)
    recall = np.divide(np.diag(counts), seen, out=np.zeros_like(seen), where=seen > 0)
    return float(recall.mean()), counts
