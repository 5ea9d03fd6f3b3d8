#!/usr/bin/env python
"""A/B of the two LDS-weights 3x3 convolution kernels (csrc/convg.hip on v_mfma 32x32x16,
csrc/convh.hip on 16x16x32) at the VGG16 layer shapes of the bench (24 x 640 x 480):

    python scripts/conv_ab.py [--iters 10] [--batch 24]

Per-kernel durations from the library's scl_prof_* sink.
"""
import argparse
import json
import os

os.environ.setdefault('SCL_DIAG', '1')   # the diagnostic build carries the variants (csrc/Makefile)
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import torch  # noqa: E402

from soft_contrastive_learning_amd import _lib  # noqa: E402
from soft_contrastive_learning_amd.model import nets  # noqa: E402

LAYERS = [('conv2_x', 128, 128, 240, 320), ('conv3_1', 128, 256, 120, 160),
          ('conv3_x', 256, 256, 120, 160), ('conv4_1', 256, 512, 60, 80),
          ('conv4_x', 512, 512, 60, 80), ('conv5_x', 512, 512, 30, 40)]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--iters', type=int, default=10)
    ap.add_argument('--batch', type=int, default=24)
    ap.add_argument('--rounds', type=int, default=2)
    ap.add_argument('--variants', default='40000,50000')
    args = ap.parse_args()
    dev = torch.device('cuda:0')
    lib = _lib.load()
    g = torch.Generator().manual_seed(3)
    for name, cin, cout, h, w in LAYERS + [('conv2_x on the LDS-weights kernels', 128, 128, 240, 320)]:
        b = args.batch
        if name.endswith('kernels'):
            nets._OWN_CONV_SHAPES.discard((128, 128))
        x = torch.randn(b, cin, h, w, generator=g).to(dev).bfloat16().contiguous(memory_format=torch.channels_last)
        gy = torch.randn(b, cout, h, w, generator=g).to(dev).bfloat16().contiguous(memory_format=torch.channels_last)
        wt = (torch.randn(cout, cin, 3, 3, generator=g) * 0.03).to(dev).bfloat16().contiguous(
            memory_format=torch.channels_last)
        bias = torch.randn(cout, generator=g).to(dev)
        row = dict(layer=name, cin=cin, cout=cout, h=h, w=w)
        outs = {}
        variants = [int(v) for v in args.variants.split(',')]
        for mode in ('fwd', 'bwd_masked'):
            def run():
                if mode == 'fwd':
                    return nets.conv64(x, wt, False, bias=bias, relu=True)
                return nets.conv64(gy, wt, True, mask=x)
            best = {}
            # the first timing after a pause reads 10 % high (clock ramp): one discarded round,
            # then the variants alternate and the best of the rounds counts
            for rnd in range(1 + args.rounds):
                for var in variants:
                    lib.scl_debug_set_variant(var)
                    try:
                        outs[(var, mode)] = run()
                        torch.cuda.synchronize()
                        with _lib.KernelTimer(capacity=8 * args.iters) as kt:
                            for _ in range(args.iters):
                                run()
                            torch.cuda.synchronize()
                    finally:
                        lib.scl_debug_set_variant(0)
                    summ = kt.summary()
                    kname = next(k for k in ('convh_kernel', 'convg_kernel', 'conv3x3_kernel') if k in summ)
                    us = summ[kname][1] * 1e3
                    row['%d_%s_kernel' % (var, mode)] = kname
                    if rnd > 0:
                        best[var] = min(best.get(var, 1e30), us)
            for var in variants:
                row['%d_%s_us' % (var, mode)] = round(best[var], 1)
                row['%d_%s_tflops' % (var, mode)] = round(
                    2.0 * b * h * w * 9 * cin * cout / (best[var] * 1e-6) / 1e12, 1)
        vs = [int(v) for v in args.variants.split(',')]
        if len(vs) > 1:
            for mode in ('fwd', 'bwd_masked'):
                a, c = outs[(vs[0], mode)].float(), outs[(vs[1], mode)].float()
                row['maxdiff_' + mode] = float((a - c).abs().max() / a.abs().max())
                row['differing_' + mode] = float((a != c).float().mean())
        print(json.dumps(row), flush=True)


if __name__ == '__main__':
    main()
