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
              device='cuda', pca_backend='device', score='bf16x3'):
    """Arrays in, the reference's pickle payload out (None when fewer than n references
    survive the thinning, like the reference's ``continue`` at :96-97)."""
    from sklearn.metrics import pairwise_distances
    full_xy_dists = pairwise_distances(full_query_xy, full_ref_xy, metric='euclidean')
    pca_ref_f, pca_query_f = whiten(pca_f, [full_ref_f, full_query_f], d, device, pca_backend)
    return retrieve(pca_ref_f, pca_query_f, full_xy_dists, full_ref_xy, n, l, score)


def retrieve(pca_ref_f, pca_query_f, full_xy_dists, full_ref_xy, n, l, score='bf16x3'):
    """evaluation/top-n.py:91-119 for one thinning distance l on whitened features.  ``score``: how the
    candidates are nominated before the float64 re-rank (retrieval.topn_l2); the lists are certified exact
    either way, 'bf16x3' is 2.7 x faster at 100 k x 10 k x 256 and what whitened descriptors (zero mean, unit
    variance per component) suit best."""
    ref_idx = thin_reference(np.asarray(full_ref_xy), l)
    if len(ref_idx) < n:
        return None
    ref_f = pca_ref_f[torch.as_tensor(ref_idx, device=pca_ref_f.device)].contiguous()
    xy_dists = full_xy_dists[:, ref_idx]
    num_q = pca_query_f.shape[0]

    # any d: the retrieval layer pads to the kernel's widths or takes its wide-descriptor path
    dist, idx = retrieval.topn_l2(ref_f, pca_query_f, n, score=score)
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


def out_pickle_path(out_root, query_lv_pickle, l, d):
    """evaluation/top-n.py:43-45, 83-86: <out_root>/l<l>_dim<d>/<query pickle name without dots>.pickle."""
    import os
    name = ''.join(os.path.basename(query_lv_pickle).split('.')[:-1])
    return os.path.join(out_root, 'l{}_dim{}'.format(l, d), '{}.pickle'.format(name))


def main(argv=None):
    """The script (evaluation/top-n.py:23-119, flags :126-139): descriptor pickles of
    evaluation/inference.py + the sets' CSV lists (easting / northing) in, one pickle per
    (thinning distance l, PCA dimension d) out; finished outputs are skipped.  The reference picks
    its sweeps from checkpoint names (:25-39: L = 0, 0.3, 1, 5 and D = 64 .. 4096 for its 'obm'
    models, else L = 0, D = 256); here they are the flags --L / --D with the latter as defaults."""
    import argparse
    import os
    from sklearn.metrics import pairwise_distances
    from ..util import io
    from ..util.meta import get_xy
    p = argparse.ArgumentParser()
    p.add_argument('--pca_lv_pickle', required=True)
    p.add_argument('--query_lv_pickle', required=True)
    p.add_argument('--ref_lv_pickle', required=True)
    p.add_argument('--query_csv', required=True)
    p.add_argument('--ref_csv', required=True)
    p.add_argument('--N', default=25, type=int)
    p.add_argument('--out_root', default='./scl_top_n')
    p.add_argument('--L', default='0.0', help='comma-separated thinning distances in metres')
    p.add_argument('--D', default='256', help='comma-separated PCA dimensions')
    p.add_argument('--pca_backend', default='device', choices=['device', 'sklearn'])
    p.add_argument('--score', default='bf16x3', choices=['f32', 'bf16x3'],
                   help='nomination arithmetic of the retrieval kernel (the emitted lists are exact either way)')
    flags = p.parse_args(argv)
    ls = [float(v) for v in flags.L.split(',')]
    ds = [int(v) for v in flags.D.split(',')]
    todo = [(l, d) for l in ls for d in ds
            if not os.path.exists(out_pickle_path(flags.out_root, flags.query_lv_pickle, l, d))]
    if not todo:
        print('Skipping complete {}'.format(flags.query_lv_pickle))
        return []
    full_ref_xy = get_xy(io.load_csv(flags.ref_csv))
    full_query_xy = get_xy(io.load_csv(flags.query_csv))
    pca_f = np.array(io.load_pickle(flags.pca_lv_pickle))
    full_ref_f = np.array(io.load_pickle(flags.ref_lv_pickle))
    full_query_f = np.array(io.load_pickle(flags.query_lv_pickle))
    full_xy_dists = pairwise_distances(full_query_xy, full_ref_xy, metric='euclidean')
    written = []
    for d in ds:
        if not any(dd == d for _, dd in todo):
            continue
        pca_ref_f, pca_query_f = whiten(pca_f, [full_ref_f, full_query_f], d, 'cuda', flags.pca_backend)
        for l in ls:
            out = out_pickle_path(flags.out_root, flags.query_lv_pickle, l, d)
            if os.path.exists(out):
                print('{} already exists. Skipping.'.format(out))
                continue
            payload = retrieve(pca_ref_f, pca_query_f, full_xy_dists, full_ref_xy, flags.N, l, flags.score)
            if payload is None:                               # fewer than N references left (:96-97)
                continue
            os.makedirs(os.path.dirname(out), exist_ok=True)
            io.save_pickle(payload, out)
            written.append(out)
    return written


if __name__ == '__main__':
    main()
