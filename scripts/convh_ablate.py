#!/usr/bin/env python
"""Timing ablations of convh_kernel on one layer shape (results of the ablated runs are meaningless):
production, no output stores (53022), no wait / barrier at the top of a tile (53021), both (53023).

    python scripts/convh_ablate.py [--layer 4_2] [--iters 10]
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
    ap.add_argument('--iters', type=int, default=10)
    ap.add_argument('--batch', type=int, default=24)
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
    res = {}
    for name, var in (('warm-up (ignore)', 0), ('production', 0), ('no stores', 53022), ('no top-of-tile wait', 53021),
                      ('neither', 53023), ('production again', 0)):
        lib.scl_debug_set_variant(var)
        try:
            for _ in range(2):
                nets.conv64(x, wt, False, bias=bias, relu=True, out=out)
            torch.cuda.synchronize()
            with L.KernelTimer(capacity=8 * args.iters) as kt:
                for _ in range(args.iters):
                    nets.conv64(x, wt, False, bias=bias, relu=True, out=out)
                torch.cuda.synchronize()
        finally:
            lib.scl_debug_set_variant(0)
        res[name] = {k: round(ms * 1e3, 1) for k, (cnt, ms) in kt.summary().items() if 'pack' not in k}
    print(json.dumps({'layer': args.layer, **res}))


if __name__ == '__main__':
    main()
