#!/usr/bin/env python
"""Same-box, same-process A/B of wrw64_kernel variants on the weight-gradient launches of the bench
step (24 x 640x480: conv1_2 .. conv5_3 at their resolutions, post-ReLU-like operands — about half
of x exactly zero, the incoming gradient masked the same way), alternating between the product
kernels (variant 0) and diagnostic variants, e.g. 2200 = round 4's staging without the buffer path.

    python scripts/wrw_ab.py [--variants 0,2200] [--rounds 3] [--layers 4_2,5_1]

Prints microseconds per launch (kernel + reduce, HIP events around `iters` back-to-back calls) and
checks that every variant's gradient is bit-identical to variant 0's.
"""
import argparse
import json
import os
import sys

os.environ.setdefault('SCL_DIAG', '1')   # the diagnostic build carries the variants (csrc/Makefile)
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import torch  # noqa: E402

from soft_contrastive_learning_amd import _lib as L  # noqa: E402
from soft_contrastive_learning_amd.model import nets  # noqa: E402

LAYERS = [('1_2', 64, 64, 1, True), ('2_1', 64, 128, 2, False), ('2_2', 128, 128, 2, True),
          ('3_1', 128, 256, 4, False), ('3_2', 256, 256, 4, False), ('3_3', 256, 256, 4, True),
          ('4_1', 256, 512, 8, False), ('4_2', 512, 512, 8, False), ('4_3', 512, 512, 8, True),
          ('5_1', 512, 512, 16, False), ('5_2', 512, 512, 16, False), ('5_3', 512, 512, 16, False)]


def timed(fn, iters):
    for _ in range(2):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--variants', default='0,2200')
    ap.add_argument('--rounds', type=int, default=3)
    ap.add_argument('--iters', type=int, default=10)
    ap.add_argument('--batch', type=int, default=24)
    ap.add_argument('--height', type=int, default=480)
    ap.add_argument('--width', type=int, default=640)
    ap.add_argument('--layers', default='')
    ap.add_argument('--json', default='')
    args = ap.parse_args()
    variants = [int(v) for v in args.variants.split(',')]
    dev = torch.device('cuda:0')
    lib = L.load()
    assert lib.scl_build_is_diag() == 1
    g = torch.Generator(device=dev).manual_seed(3)
    cl = torch.channels_last
    rows = []
    print('%-5s %5s %5s %9s | %s' % ('layer', 'cin', 'cout', 'GFLOP', '  '.join('v%-6d us (TF/s)' % v for v in variants)))
    for name, cin, cout, div, pooled in LAYERS:
        if args.layers and name not in args.layers.split(','):
            continue
        h, w = args.height // div, args.width // div
        x = torch.relu(torch.randn(args.batch, cin, h, w, device=dev, generator=g)).bfloat16().contiguous(memory_format=cl)
        # pooling layers: the gradient arrives at the pooled map + one-byte window index (PL kernels)
        gh, gw_ = (h // 2, w // 2) if pooled else (h, w)
        gy = torch.randn(args.batch, cout, gh, gw_, device=dev, generator=g)
        gy = (gy * (torch.rand(gy.shape, device=dev, generator=g) > 0.5)).bfloat16().contiguous(memory_format=cl)
        pidx = (torch.randint(0, 4, gy.shape, device=dev, generator=g, dtype=torch.uint8).contiguous(memory_format=cl)
                if pooled else None)
        wt = torch.empty(cout, cin, 3, 3, device=dev)
        gb = torch.empty(cout, device=dev)
        gf = 2.0 * args.batch * h * w * cin * cout * 9 / 1e9
        best = {v: float('inf') for v in variants}
        outs = {}
        for _ in range(args.rounds):
            for v in variants:
                lib.scl_debug_set_variant(v)
                try:
                    best[v] = min(best[v], timed(lambda: nets.wrw64(x, gy, wt, bias_grad=gb, pool_idx=pidx), args.iters))
                    outs[v] = (nets.wrw64(x, gy, wt, bias_grad=gb, pool_idx=pidx).clone(), gb.clone())
                finally:
                    lib.scl_debug_set_variant(0)
        same = all(torch.equal(outs[v][0], outs[variants[0]][0]) and torch.equal(outs[v][1], outs[variants[0]][1])
                   for v in variants)
        print('%-5s %5d %5d %9.1f | %s  %s' % (name, cin, cout, gf, '  '.join(
            '%8.1f (%5.0f)' % (best[v], gf / best[v] * 1e3) for v in variants),
            'bit-identical' if same else 'RESULTS DIFFER'))
        rows.append(dict(layer=name, cin=cin, cout=cout, h=h, w=w, gflop=gf, us=best, identical=same))
        del x, gy
    tot = {v: sum(r['us'][v] for r in rows) for v in variants}
    print('sum of the %d launches: %s' % (len(rows), '  '.join('v%d %.1f us' % (v, tot[v]) for v in variants)))
    if args.json:
        with open(args.json, 'w') as f:
            json.dump(rows, f)


if __name__ == '__main__':
    main()
