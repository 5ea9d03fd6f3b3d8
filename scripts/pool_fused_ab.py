#!/usr/bin/env python
"""A/B of the pooling layers' backward at the bench shapes (24 images, 640x480): the weight-gradient
and backward-data kernels fed the full-size gradient that scl_vgg_pool_bwd_idx writes, against the
same kernels un-pooling the pooled gradient while they stage it.

    python scripts/pool_fused_ab.py [--iters 10] [--layers 1_2,2_2,3_3,4_3]
"""
import argparse
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import torch  # noqa: E402

from soft_contrastive_learning_amd import _lib as L  # noqa: E402
from soft_contrastive_learning_amd.model import nets  # noqa: E402

LAYERS = {'1_2': (64, 64, 480, 640), '2_2': (128, 128, 240, 320), '3_3': (256, 256, 120, 160),
          '4_3': (512, 512, 60, 80)}
CL = torch.channels_last


def unpool(ga, idx, h, w):
    lib = L.load()
    b, c = ga.shape[0], ga.shape[1]
    gz = torch.empty((b, c, h, w), dtype=torch.bfloat16, device=ga.device, memory_format=CL)
    gb = torch.empty(c, device=ga.device)
    ws = L.workspace(lib.scl_vgg_workspace_bytes(c), ga.device)
    L.check(lib.scl_vgg_pool_bwd_idx(L.ptr(ga), None, L.ptr(idx), L.DT_BF16, b, h, w, c,
                                     L.ptr(gz), L.ptr(gb), L.ptr(ws), ws.numel(), L.stream_of(ga)))
    return gz


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--iters', type=int, default=10)
    ap.add_argument('--batch', type=int, default=24)
    ap.add_argument('--layers', default='1_2,2_2,3_3,4_3')
    args = ap.parse_args()
    dev = torch.device('cuda:0')
    b = args.batch
    for name in args.layers.split(','):
        cin, cout, h, w = LAYERS[name]
        g = torch.Generator().manual_seed(5)
        x = torch.relu(torch.randn(b, cin, h, w, generator=g)).to(dev).bfloat16().contiguous(memory_format=CL)
        ga = torch.randn(b, cout, h // 2, w // 2, generator=g).to(dev).bfloat16().contiguous(memory_format=CL)
        idx = torch.randint(0, 4, (b, cout, h // 2, w // 2), generator=g, dtype=torch.uint8).to(dev).contiguous(
            memory_format=CL)
        wt = (torch.randn(cout, cin, 3, 3, generator=g) * 0.05).to(dev)
        gb = torch.empty(cout, device=dev)

        def old():
            gz = unpool(ga, idx, h, w)
            nets.wrw64(x, gz, wt, gb)
            nets.conv64(gz, wt, True, mask=x)

        def new():
            nets.wrw64(x, ga, wt, gb, pool_idx=idx)
            if cin <= 128:
                nets.conv64(ga, wt, True, mask=x, pool_idx=idx)
            else:                                  # LDS-DMA windows: the un-pooling pass stays
                nets.conv64(unpool(ga, idx, h, w), wt, True, mask=x)

        rows = {}
        for label, fn in (('full-size', old), ('pooled', new)):
            for _ in range(2):
                fn()
            torch.cuda.synchronize()
            with L.KernelTimer(capacity=16 * args.iters) as kt:
                for _ in range(args.iters):
                    fn()
                torch.cuda.synchronize()
            rows[label] = {k: round(ms * 1e3, 1) for k, (cnt, ms) in sorted(kt.summary().items())}
            rows[label]['sum'] = round(sum(cnt * ms for cnt, ms in kt.summary().values()) * 1e3 / args.iters, 1)
        print(json.dumps({'layer': name, **rows}))


if __name__ == '__main__':
    main()
