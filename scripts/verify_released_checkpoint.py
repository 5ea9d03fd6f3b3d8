"""Settle hypothesis H7 with a released checkpoint: which sign of
``cluster_centers`` and which flatten order of the [512, 64] VLAD matrix reproduce the
descriptors the reference's ``evaluation/inference.py`` pickled?

    python scripts/verify_released_checkpoint.py --checkpoint /path/checkpoint-NNN \
        --reference_pickle /path/<set>_<out_name>.pickle --images frames.npy [--indices 0,5,9]

* ``--checkpoint``: a TensorFlow bundle prefix (``.index`` + ``.data-00000-of-00001``) or ``.npz``.
* ``--reference_pickle``: the reference's inference product — a ``list`` of ``np.ndarray(32768,)``
  float32 in image-list order (evaluation/inference.py:192).
* ``--images``: ``.npy`` [M,H,W,3] (uint8 / float, raw 0..255 RGB, already resized the way
  util/cv.py does), or a directory of ``.npy`` frames; ``--indices``: which rows of the pickle the
  M frames correspond to (default 0..M-1).

Four hypotheses = {centroids added, subtracted} x {D-major ``d*64+k``, K-major ``k*512+d``}.  The
sign is applied to the model (``cluster_centers.neg_()``), the order to the output.  Prints the
mean / min cosine and the max relative error per hypothesis and names the one that holds
(cosine > 0.999); exit code 0 when exactly one does, 1 otherwise.

``--self_test`` needs no files: it writes a bundle with this package's writer, produces the
"reference" descriptors under a chosen convention on the same device, and checks that the script
picks that convention out (what tests/test_gpu_callers.py runs).
"""
import argparse
import os
import pickle
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HYPOTHESES = [('plus', 'd_major'), ('plus', 'k_major'), ('minus', 'd_major'), ('minus', 'k_major')]


def describe(sign, order):
    return 'V = sum a (x %s C), flatten %s' % ('+' if sign == 'plus' else '-',
                                              'index = d*64 + k' if order == 'd_major' else 'index = k*512 + d')


def embed(model, images, sign, order, batch=4):
    """Descriptors of ``images`` under one hypothesis (the model is left as it was found)."""
    from soft_contrastive_learning_amd.model import nets
    dev = next(model.parameters()).device
    outs = []
    with torch.no_grad():
        if sign == 'minus':
            model.cluster_centers.neg_()
        try:
            for s in range(0, len(images), batch):
                img = torch.as_tensor(np.asarray(images[s:s + batch], dtype=np.float32)).to(dev)
                outs.append(nets.vgg16Netvlad(img, model=model).float().cpu())
        finally:
            if sign == 'minus':
                model.cluster_centers.neg_()
    out = torch.cat(outs, 0)
    if order == 'k_major':
        out = out.view(len(out), 512, 64).transpose(1, 2).reshape(len(out), -1)
    return out.numpy()


def compare(model, images, want):
    """-> [(sign, order, mean cosine, min cosine, max relative error)] for the four hypotheses."""
    want = np.asarray(want, dtype=np.float64)
    rows = []
    for sign, order in HYPOTHESES:
        got = embed(model, images, sign, order).astype(np.float64)
        cos = (got * want).sum(1) / np.maximum(np.linalg.norm(got, axis=1) * np.linalg.norm(want, axis=1), 1e-30)
        rel = np.abs(got - want).max() / max(np.abs(want).max(), 1e-30)
        rows.append((sign, order, float(cos.mean()), float(cos.min()), float(rel)))
    return rows


def verdict(rows, threshold=0.999):
    ok = [r for r in rows if r[3] > threshold]
    for sign, order, mean, mn, rel in rows:
        print('%-44s cosine mean %.6f min %.6f   max rel err %.3e%s'
              % (describe(sign, order), mean, mn, rel, '   <== holds' if mn > threshold else ''))
    if len(ok) == 1:
        sign, order = ok[0][:2]
        build = ('plus', 'd_major')
        print('H7: %s.' % describe(sign, order))
        print('This build assumes %s: %s' % (describe(*build),
              'CONFIRMED.' if (sign, order) == build else 'REFUTED — see INTEGRATION.md 3c for the one-line fix.'))
        return ok[0][:2]
    print('H7 undecided: %d hypotheses above cosine %.3f (checkpoint / frames / pickle rows do not belong '
          'together, or the images were preprocessed differently).' % (len(ok), threshold))
    return None


def load_images(path):
    if os.path.isdir(path):
        files = sorted(f for f in os.listdir(path) if f.endswith('.npy'))
        return np.stack([np.load(os.path.join(path, f)) for f in files]).astype(np.float32)
    return np.load(path).astype(np.float32)


def self_test(sign='plus', order='d_major', seed=0, tmp=None):
    """Round trip without any released file: returns the (sign, order) the script decides on."""
    import tempfile
    from soft_contrastive_learning_amd import checkpoint
    from soft_contrastive_learning_amd.model import nets
    dev = torch.device('cuda:0')
    src = nets.VGG16NetVLAD(seed=99).to(dev)
    with torch.no_grad():
        src.cluster_centers.mul_(20.0)       # the released models' centroids are not small
    rng = np.random.default_rng(seed)
    images = rng.integers(0, 256, size=(3, 64, 80, 3)).astype(np.float32)
    want = embed(src, images, sign, order)
    with tempfile.TemporaryDirectory(dir=tmp) as d:
        stem = os.path.join(d, 'checkpoint-7')
        checkpoint.save(src, stem, global_step=7)
        pk = os.path.join(d, 'ref.pickle')
        with open(pk, 'wb') as f:
            pickle.dump([w for w in want], f)
        return main(['--checkpoint', stem, '--reference_pickle', pk, '--images', _save(d, images)],
                    as_function=True)


def _save(d, images):
    p = os.path.join(d, 'frames.npy')
    np.save(p, images)
    return p


def main(argv=None, as_function=False):
    ap = argparse.ArgumentParser()
    ap.add_argument('--checkpoint', default='')
    ap.add_argument('--reference_pickle', default='')
    ap.add_argument('--images', default='')
    ap.add_argument('--indices', default='')
    ap.add_argument('--self_test', action='store_true')
    args = ap.parse_args(argv)
    if args.self_test:
        results = [self_test(s, o) == (s, o) for s, o in HYPOTHESES]
        print('self test:', 'ok' if all(results) else 'FAILED', results)
        return 0 if all(results) else 1
    if not (args.checkpoint and args.reference_pickle and args.images):
        ap.error('--checkpoint, --reference_pickle and --images are required (or --self_test)')
    from soft_contrastive_learning_amd import checkpoint
    from soft_contrastive_learning_amd.model import nets
    model = nets.VGG16NetVLAD().cuda()
    step = checkpoint.load(model, args.checkpoint)
    images = load_images(args.images)
    with open(args.reference_pickle, 'rb') as f:
        ref = pickle.load(f, encoding='latin1')           # the reference pickles under Python 3.5
    idx = [int(i) for i in args.indices.split(',')] if args.indices else list(range(len(images)))
    if len(idx) != len(images):
        raise SystemExit('%d frames but %d pickle rows named' % (len(images), len(idx)))
    want = np.stack([np.asarray(ref[i], dtype=np.float32).reshape(-1) for i in idx])
    if want.shape[1] != 32768:
        raise SystemExit('pickle rows have %d columns, expected 32768 (a --vlad_cores 64 run)' % want.shape[1])
    print('checkpoint %s (global step %d), %d frames %s' % (args.checkpoint, step, len(images), images.shape[1:]))
    got = verdict(compare(model, images, want))
    if as_function:
        return got
    return 0 if got is not None else 1


if __name__ == '__main__':
    sys.exit(main())
