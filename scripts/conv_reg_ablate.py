#!/usr/bin/env python
"""Timing ablation of the register-weights convolution (csrc/conv64.hip) at the bench shapes:
scl_debug_set_variant(60000 + bits), bit 0 = no window staging after the first tile,
bit 1 = no output stores.  Results under a non-zero variant are meaningless.

    python scripts/conv_reg_ablate.py [--iters 10]
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

SHAPES = [('conv1_2', 64, 64, 480, 640), ('conv2_1', 64, 128, 240, 320), ('conv2_2', 128, 128, 240, 320)]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--iters', type=int, default=10)
    ap.add_argument('--batch', type=int, default=24)
    ap.add_argument('--variants', default='0,60001,60002,60003')
    args = ap.parse_args()
    dev = torch.device('cuda:0')
    lib = _lib.load()
    g = torch.Generator().manual_seed(3)
    b = args.batch
    for name, cin, cout, h, w in SHAPES:
        x = torch.randn(b, cin, h, w, generator=g).to(dev).bfloat16().contiguous(memory_format=torch.channels_last)
        gy = torch.randn(b, cout, h, w, generator=g).to(dev).bfloat16().contiguous(memory_format=torch.channels_last)
        wt = (torch.randn(cout, cin, 3, 3, generator=g) * 0.03).to(dev).bfloat16().contiguous(
            memory_format=torch.channels_last)
        bias = torch.randn(cout, generator=g).to(dev)
        modes = {'fwd_bias_relu': lambda: nets.conv64(x, wt, False, bias=bias, relu=True),
                 'bwd_masked': lambda: nets.conv64(gy, wt, True, mask=x)}
        if cin == cout:
            modes['fwd_pool_idx'] = lambda: nets.conv_pool_idx(x, wt, bias)
        for mode, fn in modes.items():
            row = dict(layer=name, mode=mode)
            for var in [int(v) for v in args.variants.split(',')]:
                lib.scl_debug_set_variant(var)
                try:
                    fn()
                    torch.cuda.synchronize()
                    with _lib.KernelTimer(capacity=8 * args.iters) as kt:
                        for _ in range(args.iters):
                            fn()
                        torch.cuda.synchronize()
                    cnt, ms = kt.summary()['conv3x3_kernel']
                    row['v%d_us' % var] = round(ms * 1e3, 1)
                finally:
                    lib.scl_debug_set_variant(0)
            row['tflops'] = round(2.0 * b * h * w * 9 * cin * cout / (row['v0_us'] * 1e-6) / 1e12, 1)
            print(json.dumps(row), flush=True)


if __name__ == '__main__':
    main()
