#!/usr/bin/env python
"""Timing ablations of conv3x3_kernel (results of the ablated runs are meaningless): production,
no staging after the first tile (60001), no output stores (60002), both (60003).

    python scripts/conv_reg_epi_probe.py [--iters 20]
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

CASES = {'2_1 fwd (64->128, bias+relu)': (64, 128, 240, 320, False),
         '2_1 bwd (128->64, masked)': (128, 64, 240, 320, True),
         '1_2 bwd (64->64, masked)': (64, 64, 480, 640, True)}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--iters', type=int, default=20)
    ap.add_argument('--batch', type=int, default=24)
    args = ap.parse_args()
    dev = torch.device('cuda:0')
    lib = L.load()
    cl = torch.channels_last
    for name, (cin, cout, h, w, masked) in CASES.items():
        g = torch.Generator().manual_seed(5)
        x = torch.relu(torch.randn(args.batch, cin, h, w, generator=g)).to(dev).bfloat16().contiguous(memory_format=cl)
        if masked:
            wt = (torch.randn(cin, cout, 3, 3, generator=g) * 0.05).to(dev)     # conv(., w): cout -> cin
            mask = torch.relu(torch.randn(args.batch, cout, h, w, generator=g)).to(dev).bfloat16().contiguous(
                memory_format=cl)
            run = lambda: nets.conv64(x, wt, True, mask=mask)                   # noqa: E731
        else:
            wt = (torch.randn(cout, cin, 3, 3, generator=g) * 0.05).to(dev)
            bias = torch.zeros(cout, device=dev)
            run = lambda: nets.conv64(x, wt, False, bias=bias, relu=True)       # noqa: E731
        res = {}
        for label, var in (('warm-up (ignore)', 0), ('production', 0), ('no staging', 60001), ('no stores', 60002),
                           ('neither', 60003), ('production again', 0)):
            lib.scl_debug_set_variant(var)
            try:
                for _ in range(2):
                    run()
                torch.cuda.synchronize()
                with L.KernelTimer(capacity=8 * args.iters) as kt:
                    for _ in range(args.iters):
                        run()
                    torch.cuda.synchronize()
            finally:
                lib.scl_debug_set_variant(0)
            res[label] = [round(ms * 1e3, 1) for k, (cnt, ms) in kt.summary().items() if k.startswith('conv3x3_kernel')][0]
        print(json.dumps({'case': name, **res}))


if __name__ == '__main__':
    main()
