"""Retrieval counterpart of the reference's ``evaluation/top-n.py``.

``get_top_n`` reproduces the body of the reference function (evaluation/top-n.py:65-119):
PCA(whiten=True, n_components=d) fitted on a PCA set and applied to reference and query
features (:74-77; ``pca_backend='device'`` = the exact float64 Gram eigen-solve of
evaluation/pca.py on the GPU, ``'sklearn'`` = scikit-learn on the host like the reference),
greedy thinning of the reference list by distance ``l`` (:91-94), exact L2 top-N of
every query (:103-108) — here the fused HIP kernel instead of ``KDTree.query`` —
geographic distances of the hits (:110), ground truth (:112-113), translation back to the
original indices (:116-117) and the output list
``[top_i, top_g_dists, top_f_dists, gt_i, gt_g_dist, ref_idx]`` (:119).
"""
import numpy as np
import torch

from . import retrieval
from .pca import PCAWhitening


def thin_reference(ref_xy, l):
    """evaluation/top-n.py:91-94."""
    ref_idx = [0]
    for i in range(len(ref_xy)):
        if sum((ref_xy[i, :] - ref_xy[ref_idx[-1], :]) ** 2) >= l ** 2:
            ref_idx.append(i)
    return ref_idx


def whiten(pca_f, feature_sets, d, device='cuda', backend='device'):
    """Fit on pca_f, transform every array of feature_sets -> float32 device tensors."""
    if backend == 'device':
        pca = PCAWhitening(d, device=device).fit(np.asarray(pca_f, dtype=np.float32))
        return [pca.transform(np.asarray(f, dtype=np.float32)) for f in feature_sets]
    if backend != 'sklearn':
        raise ValueError("pca_backend must be 'device' or 'sklearn', got %r" % (backend,))
    from sklearn.decomposition import PCA
    pca = PCA(whiten=True, n_components=d).fit(np.asarray(pca_f))
    return [torch.from_numpy(pca.transform(np.asarray(f)).astype(np.float32)).to(device)
            for f in feature_sets]


def get_top_n(pca_f, full_ref_f, full_query_f, full_ref_xy, full_query_xy, n=25, d=256, l=0.0,
              device='cuda', pca_backend='device'):
    """Arrays in, the reference's pickle payload out (None when fewer than n references
    survive the thinning, like the reference's ``continue`` at :96-97)."""
    from sklearn.metrics import pairwise_distances
    full_xy_dists = pairwise_distances(full_query_xy, full_ref_xy, metric='euclidean')
    pca_ref_f, pca_query_f = whiten(pca_f, [full_ref_f, full_query_f], d, device, pca_backend)

    ref_idx = thin_reference(np.asarray(full_ref_xy), l)
    if len(ref_idx) < n:
        return None
    ref_f = pca_ref_f[torch.as_tensor(ref_idx, device=pca_ref_f.device)].contiguous()
    xy_dists = full_xy_dists[:, ref_idx]
    num_q = pca_query_f.shape[0]

    # any d: the retrieval layer pads to the kernel's widths or takes its wide-descriptor path
    dist, idx = retrieval.topn_l2(ref_f, pca_query_f, n)
    top_f_dists = dist.cpu().numpy()
    top_i = idx.cpu().numpy().astype(int)
    top_g_dists = [[xy_dists[q, r] for r in top_i[q, :]] for q in range(num_q)]
    gt_i = np.argmin(xy_dists, axis=1)
    gt_g_dist = np.min(xy_dists, axis=1)
    top_i = [[ref_idx[r] for r in top_i[q, :]] for q in range(num_q)]
    gt_i = [ref_idx[r] for r in gt_i]
    return [top_i, top_g_dists, top_f_dists, gt_i, gt_g_dist, ref_idx]


def recall_at(top_g_dists, thresholds, n=1):
    """% of queries whose best geographic distance over the first n hits is below each
    threshold (train/train.py:363-376; evaluation/roc.py:213-216 uses n=1)."""
    g = np.asarray(top_g_dists, dtype=np.float64)[:, :n].min(axis=1)
    return np.array([np.mean(g < x) for x in thresholds])
