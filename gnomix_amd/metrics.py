"""The two scores the reference's evaluate() methods print (src/Base/base.py:214-228, src/Smooth/smooth.py:67-79: sklearn's
accuracy_score and balanced_accuracy_score, in percent, rounded to two decimals) and its confusion matrix
(src/model.py:93-98), in numpy."""
from __future__ import annotations

import numpy as np


def confusion(y, y_pred):
    """(matrix over the sorted union of labels, those labels) — sklearn.metrics.confusion_matrix's default label set"""
    y = np.asarray(y).reshape(-1)
    y_pred = np.asarray(y_pred).reshape(-1)
    labels = np.unique(np.concatenate([y, y_pred]))
    idx = {int(l): i for i, l in enumerate(labels)}
    cm = np.zeros((len(labels), len(labels)), dtype=np.int64)
    np.add.at(cm, (np.searchsorted(labels, y), np.searchsorted(labels, y_pred)), 1)
    return cm, [int(l) for l in labels]


def accuracy_pair(y, y_pred):
    """(accuracy, balanced accuracy) in percent, two decimals.  Balanced = mean recall over the classes present in y."""
    y = np.asarray(y).reshape(-1)
    y_pred = np.asarray(y_pred).reshape(-1)
    acc = float(np.mean(y == y_pred))
    cm, labels = confusion(y, y_pred)
    support = cm.sum(axis=1)
    recall = np.diag(cm)[support > 0] / support[support > 0]
    return round(acc * 100, 2), round(float(np.mean(recall)) * 100, 2)
