#!/usr/bin/env python
"""Where the one-launch pairwise-loss forward (csrc/gram_loss.hip: gram16x6_persist_kernel /
gram16_persist_kernel) spends its time: scl_debug_set_variant(40) makes thread 0 of every workgroup
write shader-clock stamps into the tail of the workspace; prints, per phase, the median and the
maximum over workgroups in microseconds (s_memtime ticks at 100 MHz).  DIAGNOSTIC ONLY.

    python scripts/loss_stamps.py [--batch 192]
"""
import argparse
import os
import sys

os.environ.setdefault('SCL_DIAG', '1')
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402
import torch  # noqa: E402

from soft_contrastive_learning_amd import _lib  # noqa: E402
from tests import util_data as U  # noqa: E402

# (from stamp, to stamp, what)
SPANS = [(0, 1, 'loads + split of pass 0 -> LDS'), (1, 9, 'pass 0 MFMAs + split of pass 1'),
         (9, 2, 'pass 1 MFMAs + slab stores'), (2, 3, 'barrier 1'), (3, 4, 'phase 2: slab sums'),
         (4, 5, 'barrier 2'), (5, 6, 'phase 3: rows'), (6, 7, 'barrier 3'), (7, 8, 'phase 4: M and the loss')]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--batch', type=int, default=192)
    ap.add_argument('--e', type=int, default=32768)
    ap.add_argument('--ghz', type=float, default=2.1, help='shader clock the s_memtime ticks are priced at')
    args = ap.parse_args()
    dev = torch.device('cuda:0')
    _lib.use_diag()
    lib = _lib.load()
    b, e = args.batch, args.e
    emb = torch.tensor(U.embeddings(b, e), device=dev)
    dist = torch.tensor(U.positions_distances(b)[None], device=dev)
    loss = torch.empty((), device=dev)
    coef = torch.empty((b, b), device=dev)
    ws = _lib.workspace(lib.scl_gram_loss_workspace_bytes(b, e), dev)
    sync = _lib.sync_words(dev)
    lib.scl_debug_set_variant(40)
    rows = []
    for _ in range(8):
        ws.zero_()
        _lib.check(lib.scl_gram_loss_fwd_s(
            _lib.ptr(emb), emb.stride(0), b, e, _lib.MASK_WMS_EXP, _lib.ptr(dist), 1, 0.8, 15.0, None, 2.0,
            50.0, 1.0, 0.1, 1, _lib.SUM_MS, _lib.ptr(loss), _lib.ptr(coef), _lib.ptr(ws), ws.numel(),
            _lib.ptr(sync), _lib.stream_of(emb)))
        torch.cuda.synchronize()
        tail = ws[-256 * 16 * 8:].view(torch.int64).reshape(256, 16).cpu().numpy()
        rows.append(tail)
    lib.scl_debug_set_variant(0)
    st = rows[-1]
    grid = int((st[:, 0] != 0).sum())
    st = st[:grid].astype(np.float64)
    us = 1e-3 / args.ghz
    print('B = %d, E = %d, %d workgroups; loss %.6f; one s_memtime tick = one shader cycle, priced at %.2f GHz'
          % (b, e, grid, float(loss), args.ghz))
    for k0, k1, name in SPANS:
        if (st[:, k0] == 0).all() or (st[:, k1] == 0).all():
            continue                                            # (a stamp this kernel does not take)
        d = (st[:, k1] - st[:, k0]) * us
        print('%-36s median %6.2f us   max %6.2f us' % (name, np.median(d), d.max()))
    d = (st[:, 8] - st[:, 0]) * us
    print('%-36s median %6.2f us   max %6.2f us' % ('first stamp -> last stamp', np.median(d), d.max()))


if __name__ == '__main__':
    main()
