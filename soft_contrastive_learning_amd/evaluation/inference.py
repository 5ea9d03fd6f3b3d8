"""Feature-extraction counterpart of the reference's ``evaluation/inference.py``.

Reproduces ``infer()`` (evaluation/inference.py:147-192): the image list is padded with
copies of image 0 up to a multiple of ``images_per_pass`` (:172-175), batches go through
``vgg16Netvlad`` forward-only (:89-90, :46), results are re-ordered by original index,
padding rows are dropped (:186-191) and the product is a pickle holding a ``list`` of
``np.ndarray(32768,) float32`` in list order (:192).

Images: ``--csv_root/<set>.csv`` with a ``path`` column relative to ``--img_root``
(evaluation/inference.py:168-170), decoded and sized by ``load_images``' rules (:52-72; util/cv.py,
util/io.py restated in ``soft_contrastive_learning_amd/util``): 'oxs' sets read ``.jpg`` instead of
``.png``, 'achen' sets are portrait (cover ``large_side x small_side``), with the NetVLAD head the
longer side becomes ``large_side`` unless ``--rescale`` is off, without it the frame is brought to
``small_side x large_side``.  Six loader threads feed the device (:155-160).  ``--set synthetic``:
random images, no files.
"""
import argparse
import os
import pickle

import numpy as np
import torch

from .. import checkpoint
from ..model import nets


def pad_indices(num, images_per_pass):
    """evaluation/inference.py:172-175 — note the reference pads a FULL extra pass when
    ``num`` is already a multiple (``images_per_pass - num % images_per_pass``)."""
    padding = [0] * (images_per_pass - (num % images_per_pass))
    return np.concatenate((np.arange(num), np.array(padding, dtype=int))).astype(int)


def extract_features(model, loader, num, images_per_pass=4, device=None, loader_threads=6):
    """Returns ``list`` of ``num`` float32 vectors (length 32768 with the NetVLAD head, H' W' 512
    without: ``ops['full_out']``, evaluation/inference.py:89-92), in index order.  ``loader`` is
    called from ``loader_threads`` threads (cpu_thread, :28-38), one batch ahead of the device."""
    from concurrent.futures import ThreadPoolExecutor
    device = device or next(model.parameters()).device
    order = pad_indices(num, images_per_pass)
    feats = [None] * len(order)
    starts = list(range(0, len(order), images_per_pass))

    def load_batch(s):
        idx = order[s:s + images_per_pass]
        return np.stack([np.asarray(loader(int(i)), dtype=np.float32) for i in idx])
    with torch.no_grad(), ThreadPoolExecutor(max_workers=max(int(loader_threads), 1)) as pool:
        ahead = max(int(loader_threads), 1)
        pending = [pool.submit(load_batch, s) for s in starts[:ahead]]
        for k, s in enumerate(starts):
            batch = pending.pop(0).result()
            if k + ahead < len(starts):
                pending.append(pool.submit(load_batch, starts[k + ahead]))
            out = nets.full_out(torch.from_numpy(batch).to(device), model=model)
            out = out.float().cpu().numpy()
            for slot, f in zip(range(s, s + len(out)), out):
                feats[slot] = f
    return feats[:num]


def csv_loader(set_name, csv_root, img_root, vlad_cores=64, rescale=True, small_side=180, large_side=240):
    """(loader, num) over ``<csv_root>/<set>.csv`` (column ``path``): evaluation/inference.py:52-72,
    168-170."""
    from ..util import cv, io
    meta = io.load_csv(os.path.join(csv_root, '{}.csv'.format(set_name)))
    if not isinstance(meta, dict) or 'path' not in meta:
        raise ValueError('%s.csv needs a "path" column and at least one row' % set_name)
    paths = list(meta['path'])

    def load(i):
        rel = paths[i]
        if 'oxs' in set_name:
            rel = rel.replace('.png', '.jpg')
        img = io.load_img(os.path.join(img_root, rel))
        if 'achen' in set_name:
            return cv.standard_size(img, h=large_side, w=small_side)           # portrait images
        if vlad_cores > 0:
            return cv.resize_img(img, large_side) if rescale else img
        return cv.standard_size(img, h=small_side, w=large_side)
    return load, len(paths)


def save_pickle(data, out_file):
    with open(out_file, 'wb') as f:
        pickle.dump(data, f)


def synthetic_loader(height, width, seed=42):
    def load(i):
        rng = np.random.default_rng(seed + i)
        return rng.integers(0, 256, size=(height, width, 3)).astype(np.float32)
    return load


def main(argv=None):
    p = argparse.ArgumentParser()
    # flag names of evaluation/inference.py:204-230
    p.add_argument('--rescale', default=True,
                   type=lambda v: str(v).lower() not in ('0', 'false', 'no', ''))
    p.add_argument('--small_side', default=180, type=int)
    p.add_argument('--large_side', default=240, type=int)
    p.add_argument('--img_root', default='')
    p.add_argument('--csv_root', default='')
    p.add_argument('--set', default='synthetic')
    p.add_argument('--checkpoint', default='')
    p.add_argument('--out_name', default='scl_amd')
    p.add_argument('--reduction', default='none')
    p.add_argument('--vlad_cores', default=64, type=int)
    p.add_argument('--out_root', default='./scl_lv')
    p.add_argument('--images_per_pass', type=int, default=4)
    p.add_argument('--num_images', type=int, default=10, help='synthetic set size')
    flags = p.parse_args(argv)
    # --reduction pca writes the RAW descriptors, like the reference: "Don't actually do PCA
    # here - doing it after" (evaluation/inference.py:94-95; its projection branch at :111-116
    # is unreachable).  The whitening runs in evaluation/top_n.py.
    if flags.vlad_cores not in (0, 64) or flags.reduction not in ('none', 'pca'):
        raise SystemExit('only --vlad_cores 64 | 0 with --reduction none|pca is on the hot path')
    np.random.seed(42)                                       # inference.py:270-271
    model = nets.VGG16NetVLAD(vlad_cores=flags.vlad_cores).cuda()
    if flags.checkpoint:
        checkpoint.load(model, flags.checkpoint)
    if flags.set == 'synthetic' and not flags.csv_root:
        loader, num = synthetic_loader(flags.small_side, flags.large_side), flags.num_images
    else:
        loader, num = csv_loader(flags.set, flags.csv_root, flags.img_root, flags.vlad_cores,
                                 flags.rescale, flags.small_side, flags.large_side)
    feats = extract_features(model, loader, num, flags.images_per_pass)
    os.makedirs(flags.out_root, exist_ok=True)
    out = os.path.join(flags.out_root, '{}_{}.pickle'.format(flags.set, flags.out_name))
    save_pickle(feats, out)
    print('wrote', out, len(feats), feats[0].shape, feats[0].dtype)


if __name__ == '__main__':
    main()
