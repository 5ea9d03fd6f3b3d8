#!/usr/bin/env python
"""Pairwise-loss FORWARD (with the matrix M for the backward) at 32 < B <= 256: the product path
(round 6: strip-scheduled Gram kernel + three finishing launches) against the four launches of round
5 (diagnostic variant 37) and against the one persistent launch (41).  Two figures per form:

  stream_us   N calls back to back on one stream between two events / N  (what a training step pays)
  kernels     per-kernel event durations of the scl_prof sink, summed per call

    python scripts/loss_fwd_ab.py [--batches 48,64,96,192,208] [--iters 200]
"""
import argparse
import json
import os
import sys

os.environ.setdefault('SCL_DIAG', '1')
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import torch  # noqa: E402

from soft_contrastive_learning_amd import _lib  # noqa: E402
from tests import util_data as U  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--batches', default='48,64,96,192,208')
    ap.add_argument('--iters', type=int, default=200)
    ap.add_argument('--e', type=int, default=32768)
    args = ap.parse_args()
    dev = torch.device('cuda:0')
    _lib.use_diag()
    lib = _lib.load()
    out = []
    for b in [int(x) for x in args.batches.split(',')]:
        emb = torch.tensor(U.embeddings(b, args.e), device=dev)
        dist = torch.tensor(U.positions_distances(b)[None], device=dev)
        loss = torch.empty((), device=dev)
        coef = torch.empty((b, b), device=dev)
        ws = _lib.workspace(lib.scl_gram_loss_workspace_bytes(b, args.e), dev)
        sync = _lib.sync_words(dev)

        def call():
            _lib.check(lib.scl_gram_loss_fwd_s(
                _lib.ptr(emb), emb.stride(0), b, args.e, _lib.MASK_WMS_EXP, _lib.ptr(dist), 1, 0.8, 15.0,
                None, 2.0, 50.0, 1.0, 0.1, 1, _lib.SUM_MS, _lib.ptr(loss), _lib.ptr(coef), _lib.ptr(ws),
                ws.numel(), _lib.ptr(sync), _lib.stream_of(emb)))
        rec = {'B': b, 'E': args.e}
        for v, name in ((0, 'product_strip_gram_plus_three'), (37, 'four_launches_r05'), (41, 'one_persistent_launch')):
            lib.scl_debug_set_variant(v)
            for _ in range(20):
                call()
            torch.cuda.synchronize()
            best = 1e9
            for _ in range(5):
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for _ in range(args.iters):
                    call()
                e1.record()
                torch.cuda.synchronize()
                best = min(best, e0.elapsed_time(e1) * 1e3 / args.iters)
            with _lib.KernelTimer(capacity=8 * 50) as kt:
                for _ in range(50):
                    call()
                torch.cuda.synchronize()
            ks = {k: round(ms * 1e3, 2) for k, (c, ms) in sorted(kt.summary().items())}
            rec[name] = {'stream_us': round(best, 2), 'loss': float(loss), 'kernel_event_us': ks,
                         'kernel_event_sum_us': round(sum(ks.values()), 2)}
        lib.scl_debug_set_variant(0)
        out.append(rec)
        print(json.dumps(rec))
    return out


if __name__ == '__main__':
    main()
