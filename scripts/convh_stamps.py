#!/usr/bin/env python
"""Where the cycles of a convh_kernel tile go: scl_debug_set_variant(53024) makes wave 0 of every
workgroup write s_memtime stamps of its second and third tile to the tail of the output.

    python scripts/convh_stamps.py [--layer 4_2]
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

LAYERS = {'3_2': (256, 256, 120, 160), '4_2': (512, 512, 60, 80)}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--layer', default='4_2')
    ap.add_argument('--batch', type=int, default=24)
    ap.add_argument('--wave', type=int, default=0, help='which of the eight waves writes the stamps')
    ap.add_argument('--cut', default='2x4', choices=['2x4', '4x2'],
                    help='wave grid of the 12-row block: 2x4 (15 m-tiles x 32 channels per wave, the default since round 4) or 4x2 (8 x 64)')
    args = ap.parse_args()
    dev = torch.device('cuda:0')
    lib = L.load()
    cin, cout, h, w = LAYERS[args.layer]
    b = args.batch
    g = torch.Generator().manual_seed(5)
    cl = torch.channels_last
    x = torch.relu(torch.randn(b, cin, h, w, generator=g)).to(dev).bfloat16().contiguous(memory_format=cl)
    wt = (torch.randn(cout, cin, 3, 3, generator=g) * 0.05).to(dev)
    bias = torch.zeros(cout, device=dev)
    for _ in range(2):
        nets.conv64(x, wt, False, bias=bias, relu=True)
    torch.cuda.synchronize()
    nwg = 256
    n = b * cout * h * w
    store = torch.zeros(n + nwg * 2 * 24 * 4, dtype=torch.bfloat16, device=dev)    # + room for the stamps
    out = store.as_strided((b, cout, h, w), (h * w * cout, 1, w * cout, cout))
    assert out.is_contiguous(memory_format=cl)
    lib.scl_debug_set_variant((53024 if args.cut == '2x4' else 53032) + args.wave)
    try:
        nets.conv64(x, wt, False, bias=bias, relu=True, out=out)
        torch.cuda.synchronize()
    finally:
        lib.scl_debug_set_variant(0)
    raw = store[n:].view(torch.uint8).cpu().numpy().view(np.uint64)
    t = raw.astype(np.int64).reshape(nwg, 2, 24)[:, :, :23]
    names = ['first stage landed + barrier']
    for par in (0, 1):
        for gi in range(3):
            for ph in ('own share waited for', 'barrier passed', 'next requests issued'):
                names.append('chunk %s group %d: %s' % ('even' if par == 0 else 'odd', gi, ph))
    names += ['K loop left', 'barrier', 'epilogue done']
    print('layer %s (%d -> %d, %d x %d) forward; cycles after the tile start, median over workgroups'
          % (args.layer, cin, cout, h, w))
    for tile in range(2):
        ok = t[:, tile, 0] > 0
        if not ok.any():
            continue
        d = t[ok, tile, :] - t[ok, tile, 0:1]
        order = [0] + list(range(1, 23))
        print(' tile %d (%d workgroups)' % (tile + 1, int(ok.sum())))
        # the chunk-pair stamps are those of the LAST pair of the tile: sort by time for reading
        med = np.median(d, axis=0)
        for k in np.argsort(med):
            if k == 0:
                continue
            print('   %-50s %8d   p90 %8d' % (names[k - 1], int(med[k]), int(np.percentile(d[:, k], 90))))
    per = t[:, 1, 0] - t[:, 0, 0]
    per = per[(t[:, 0, 0] > 0) & (t[:, 1, 0] > 0)]
    if per.size:
        print(' tile start to next tile start: median %d cycles' % int(np.median(per)))


if __name__ == '__main__':
    main()
