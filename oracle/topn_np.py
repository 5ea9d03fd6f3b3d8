"""Exact L2 top-N retrieval oracle.  TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).

The reference does ``KDTree(ref_f).query(query_f, k=N, return_distance=True,
sort_results=True)`` (evaluation/top-n.py:103-106; also train/train.py:1181-1182 with
k=5): exact Euclidean nearest neighbours, ascending, float64 inside scikit-learn.
``topn_bruteforce`` restates that as a float64 brute force (chunked so the full Q×R
matrix is never held); ``topn_kdtree`` is the reference's own call, usable wherever
scikit-learn is installed (it is in this image) and used to pin the brute force.
"""
import numpy as np


def topn_bruteforce(ref, query, n, chunk=256):
    """Returns (dists [Q,n] float64 ascending, idx [Q,n] int64)."""
    ref = np.asarray(ref, dtype=np.float64)
    query = np.asarray(query, dtype=np.float64)
    q_n = query.shape[0]
    out_d = np.empty((q_n, n), dtype=np.float64)
    out_i = np.empty((q_n, n), dtype=np.int64)
    for s in range(0, q_n, chunk):
        q = query[s:s + chunk]
        # direct (q - r)^2 form, like the tree's rdist, not the Gram expansion
        d2 = ((q[:, None, :] - ref[None, :, :]) ** 2).sum(axis=2)
        part = np.argpartition(d2, n - 1, axis=1)[:, :n]
        pd = np.take_along_axis(d2, part, axis=1)
        order = np.argsort(pd, axis=1, kind='stable')
        out_i[s:s + chunk] = np.take_along_axis(part, order, axis=1)
        out_d[s:s + chunk] = np.sqrt(np.take_along_axis(pd, order, axis=1))
    return out_d, out_i


def topn_kdtree(ref, query, n):
    """The reference's own call (evaluation/top-n.py:103-106)."""
    from sklearn.neighbors import KDTree
    tree = KDTree(np.asarray(ref))
    d, i = tree.query(np.asarray(query), k=n, return_distance=True, sort_results=True)
    return np.asarray(d, dtype=np.float64), np.asarray(i, dtype=np.int64)


def recall_at_threshold(top_g_dists, thresholds, n=None):
    """Localisation recall as the trainer defines it (train/train.py:363-376;
    top-1 form in evaluation/roc.py:213-216): a query is correct at threshold x for
    Top-n if the minimum geographic distance over its first n retrieved references
    is < x.  Returns fractions in [0,1], one per threshold."""
    g = np.asarray(top_g_dists, dtype=np.float64)
    if n is not None:
        g = g[:, :n]
    best = np.min(g, axis=1)
    return np.array([np.mean(best < x) for x in thresholds])
