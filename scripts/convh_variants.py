#!/usr/bin/env python
"""Same-box timing of convh_kernel on one layer shape under a list of scl_debug_set_variant values
(50000 = the kernel as it ships, 53008 / 53012 / 53006 pin the block height, ...), several rounds so
that clock drift shows.  Forward with bias + ReLU or the masked backward-data pass.

    python scripts/convh_variants.py --layer 4_2 --pass fwd --variants 50000,53008 [--rounds 4]
"""
import argparse
import json
import os

os.environ.setdefault('SCL_DIAG', '1')   # the diagnostic build carries the variants (csrc/Makefile)
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import torch  # noqa: E402

from soft_contrastive_learning_amd import _lib as L  # noqa: E402
from soft_contrastive_learning_amd.model import nets  # noqa: E402

LAYERS = {'3_2': (256, 256, 120, 160), '4_2': (512, 512, 60, 80), '5_2': (512, 512, 30, 40)}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--layer', default='4_2')
    ap.add_argument('--pass', dest='which', default='fwd', choices=['fwd', 'bwd', 'pool'])
    ap.add_argument('--iters', type=int, default=20)
    ap.add_argument('--batch', type=int, default=24)
    ap.add_argument('--rounds', type=int, default=4)
    ap.add_argument('--variants', default='50000')
    ap.add_argument('--tag', default='')
    args = ap.parse_args()
    dev = torch.device('cuda:0')
    lib = L.load()
    cin, cout, h, w = LAYERS[args.layer]
    g = torch.Generator().manual_seed(5)
    cl = torch.channels_last
    x = torch.relu(torch.randn(args.batch, cin, h, w, generator=g)).to(dev).bfloat16().contiguous(memory_format=cl)
    wt = (torch.randn(cout, cin, 3, 3, generator=g) * 0.05).to(dev)
    bias = torch.zeros(cout, device=dev)
    out = torch.empty((args.batch, cout, h, w), dtype=torch.bfloat16, device=dev, memory_format=cl)
    gz = torch.randn(args.batch, cout, h, w, generator=g).to(dev).bfloat16().contiguous(memory_format=cl)
    gx = torch.empty_like(x)

    def run():
        if args.which == 'fwd':
            nets.conv64(x, wt, False, bias=bias, relu=True, out=out)
        elif args.which == 'pool':
            nets.conv_pool_idx(x, wt, bias)          # forward with the pooled + window index epilogue
        else:
            nets.conv64(gz, wt, True, mask=x, out=gx)

    for _ in range(30):                      # clocks settle
        run()
    torch.cuda.synchronize()
    for rnd in range(args.rounds):
        res = {}
        for v in [int(t) for t in args.variants.split(',')]:
            lib.scl_debug_set_variant(v)
            try:
                for _ in range(3):
                    run()
                torch.cuda.synchronize()
                with L.KernelTimer(capacity=8 * args.iters) as kt:
                    for _ in range(args.iters):
                        run()
                    torch.cuda.synchronize()
            finally:
                lib.scl_debug_set_variant(0)
            res[v] = [round(ms * 1e3, 1) for k, (cnt, ms) in kt.summary().items() if 'pack' not in k][0]
        print(json.dumps({'tag': args.tag, 'layer': args.layer, 'pass': args.which, 'round': rnd, 'us': res}))


if __name__ == '__main__':
    main()
