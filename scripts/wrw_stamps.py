#!/usr/bin/env python
"""Where the cycles of a wrw64_kernel tile go: scl_debug_set_variant(2004) makes wave 0 of every
workgroup write s_memtime stamps (100 MHz-independent shader clock counts) of its first four tiles.

    python scripts/wrw_stamps.py [--layer 4_3] [--pooled]
"""
import argparse
import os

os.environ.setdefault('SCL_DIAG', '1')   # the diagnostic build carries the variants (csrc/Makefile)
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402
import torch  # noqa: E402

from soft_contrastive_learning_amd import _lib as L  # noqa: E402
from soft_contrastive_learning_amd.model import nets  # noqa: E402

LAYERS = {'2_2': (128, 128, 240, 320), '3_2': (256, 256, 120, 160), '4_2': (512, 512, 60, 80),
          '5_2': (512, 512, 30, 40)}
NAMES = ['staging issued', 'first step done (9 products)', 'fourth step done (36)',
         'last product issued (72)', 'own DMA landed', 'barrier passed']


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--layer', default='4_2')
    ap.add_argument('--pooled', action='store_true')
    ap.add_argument('--no-staging', action='store_true', help='timing only: tiles after the first are not staged')
    ap.add_argument('--batch', type=int, default=24)
    args = ap.parse_args()
    dev = torch.device('cuda:0')
    lib = L.load()
    cin, cout, h, w = LAYERS[args.layer]
    b = args.batch
    g = torch.Generator().manual_seed(5)
    cl = torch.channels_last
    x = torch.relu(torch.randn(b, cin, h, w, generator=g)).to(dev).bfloat16().contiguous(memory_format=cl)
    if args.pooled:
        gz = torch.randn(b, cout, h // 2, w // 2, generator=g).to(dev).bfloat16().contiguous(memory_format=cl)
        idx = torch.randint(0, 4, (b, cout, h // 2, w // 2), generator=g, dtype=torch.uint8).to(dev).contiguous(
            memory_format=cl)
    else:
        gz = torch.randn(b, cout, h, w, generator=g).to(dev).bfloat16().contiguous(memory_format=cl)
        idx = None
    wt = torch.empty(cout, cin, 3, 3, device=dev)
    for _ in range(2):
        nets.wrw64(x, gz, wt, None, pool_idx=idx)
    torch.cuda.synchronize()
    ws = L.workspace(lib.scl_wrw3x3_workspace_bytes(cin, cout), dev)
    ws.zero_()
    sk, sc, sh, sw = wt.stride()
    lib.scl_debug_set_variant(2006 if args.no_staging else 2004)
    try:
        if idx is None:
            L.check(lib.scl_wrw3x3_bias(L.ptr(x), L.ptr(gz), b, h, w, cin, cout, L.ptr(wt), sk, sc, sh, sw,
                                        1, None, L.ptr(ws), ws.numel(), L.stream_of(x)))
        else:
            L.check(lib.scl_wrw3x3_pooled(L.ptr(x), L.ptr(gz), L.ptr(idx), b, h, w, cin, cout, L.ptr(wt),
                                          sk, sc, sh, sw, 1, None, L.ptr(ws), ws.numel(), L.stream_of(x)))
        torch.cuda.synchronize()
    finally:
        lib.scl_debug_set_variant(0)
    blocks = (cin // 64) * (cout // 64)
    p = (1024 + blocks - 1) // blocks
    off = (2 * p * blocks * 9 * 64 * 64 * 4 + 255) // 256 * 256
    nwg = 256
    raw = ws.view(torch.uint8)[off:off + nwg * 4 * 8 * 8].cpu().numpy().view(np.uint64).reshape(nwg, 4, 8)
    t = raw.astype(np.int64)
    print('layer %s (%d -> %d, %d x %d), %s gradient; cycles after the tile start, median over %d workgroups'
          % (args.layer, cin, cout, h, w, 'pooled' if args.pooled else 'full-size', nwg))
    for tile in range(4):
        d = t[:, tile, 1:7] - t[:, tile, 0:1]
        ok = (t[:, tile, 0] > 0)
        if not ok.any():
            continue
        print(' tile %d (%d workgroups)' % (tile, int(ok.sum())))
        prev = 0
        for k, nm in enumerate(NAMES):
            med = int(np.median(d[ok, k]))
            print('   %-34s %7d  (+%6d)   p90 %7d' % (nm, med, med - prev, int(np.percentile(d[ok, k], 90))))
            prev = med
    if t.shape[1] > 1:
        per = t[:, 1:4, 0] - t[:, 0:3, 0]
        if (per > 0).any():
            print(' tile start to next tile start: median %d cycles' % int(np.median(per[per > 0])))


if __name__ == '__main__':
    main()
