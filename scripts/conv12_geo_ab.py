#!/usr/bin/env python
"""conv1_2 at the bench shape (24 x 640x480, 64 -> 64), the two launches of the train step —
forward with the pooled + window-index epilogue, backward-data un-pooling its window and applying
ReLU' — timed in THIS process' geometry: SCL_CONV64_TWO_WG=1 (diagnostic build: two 4-wave workgroups
per CU, 4-row tiles) or unset (the product: one 8-wave workgroup, 8-row tiles).  Run it once per setting.

    SCL_CONV64_TWO_WG=0 python scripts/conv12_geo_ab.py; SCL_CONV64_TWO_WG=1 python scripts/conv12_geo_ab.py
"""
import argparse
import json
import os
import sys

os.environ.setdefault('SCL_DIAG', '1')   # the diagnostic build carries the second geometry

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import torch  # noqa: E402

from soft_contrastive_learning_amd import _lib as L  # noqa: E402
from soft_contrastive_learning_amd.model import nets  # noqa: E402

CL = torch.channels_last


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--iters', type=int, default=20)
    ap.add_argument('--batch', type=int, default=24)
    ap.add_argument('--height', type=int, default=480)
    ap.add_argument('--width', type=int, default=640)
    args = ap.parse_args()
    dev = torch.device('cuda:0')
    b, h, w = args.batch, args.height, args.width
    g = torch.Generator().manual_seed(5)
    x = torch.relu(torch.randn(b, 64, h, w, generator=g)).to(dev).bfloat16().contiguous(memory_format=CL)
    ga = torch.randn(b, 64, h // 2, w // 2, generator=g).to(dev).bfloat16().contiguous(memory_format=CL)
    wt = (torch.randn(64, 64, 3, 3, generator=g) * 0.05).to(dev)
    bias = torch.randn(64, generator=g).to(dev)
    pooled, idx = nets.conv_pool_idx(x, wt, bias)

    def fwd():
        nets.conv_pool_idx(x, wt, bias)

    def bwd():
        nets.conv64(ga, wt, True, mask=x, pool_idx=idx)

    def plain():
        nets.conv64(x, wt, False, bias=bias, relu=True)
    rec = {'two_wg': os.environ.get('SCL_CONV64_TWO_WG', '0'), 'shape': [b, h, w]}
    gf = 2.0 * b * h * w * 64 * 64 * 9 / 1e9
    for name, fn in (('forward_pool_idx', fwd), ('backward_data_unpool_masked', bwd), ('forward_bias_relu', plain)):
        for _ in range(3):
            fn()
        torch.cuda.synchronize()
        best = 1e9
        for _ in range(3):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(args.iters):
                fn()
            e1.record()
            torch.cuda.synchronize()
            best = min(best, e0.elapsed_time(e1) / args.iters * 1e3)
        rec[name] = {'us': round(best, 1), 'tflops': round(gf / best * 1e3, 0)}
    print(json.dumps(rec))


if __name__ == '__main__':
    main()
