#!/usr/bin/env python
"""conv1_1 + conv1_2 forward: the one-kernel form (scl_conv_first_pool_idx) against the two kernels
it replaces, same box, event-bracketed kernel times (microseconds), 24 x 480 x 640.

    python scripts/first_block_ab.py [--iters 20] [--rounds 3]
"""
import argparse
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import torch  # noqa: E402

from soft_contrastive_learning_amd import _lib as L  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--iters', type=int, default=20)
    ap.add_argument('--rounds', type=int, default=3)
    ap.add_argument('--batch', type=int, default=24)
    args = ap.parse_args()
    dev = torch.device('cuda:0')
    lib = L.load()
    b, h, w = args.batch, 480, 640
    g = torch.Generator().manual_seed(3)
    img = torch.randint(0, 256, (b, h, w, 3), generator=g).float().to(dev)
    avg = torch.tensor([123.68, 116.78, 103.94], device=dev)
    w1 = (torch.randn(64, 3, 3, 3, generator=g) * 0.1).to(dev)
    w2 = (torch.randn(64, 64, 3, 3, generator=g) * 0.05).to(dev)
    b1 = torch.randn(64, generator=g).to(dev)
    b2 = torch.randn(64, generator=g).to(dev)
    ws = L.workspace(lib.scl_conv3x3_workspace_bytes(), dev)
    x0 = torch.empty((b, h, w, 3), dtype=torch.bfloat16, device=dev)
    y1 = torch.empty((b, h, w, 64), dtype=torch.bfloat16, device=dev)
    a = torch.empty((b, h // 2, w // 2, 64), dtype=torch.bfloat16, device=dev)
    idx = torch.empty((b, h // 2, w // 2, 64), dtype=torch.uint8, device=dev)
    s1, s2 = w1.stride(), w2.stride()
    st = L.stream_of(img)

    def two():
        L.check(lib.scl_conv_first(L.ptr(img), L.ptr(avg), L.ptr(w1), *s1, 1, L.ptr(b1), b, h, w,
                                   L.ptr(x0), L.ptr(y1), st))
        L.check(lib.scl_conv3x3_pool_idx(L.ptr(y1), L.ptr(w2), *s2, L.W_F32, b, h, w, 64, 64, L.ptr(b2),
                                         L.ptr(a), L.ptr(idx), L.ptr(ws), ws.numel(), st))

    def one():
        L.check(lib.scl_conv_first_pool_idx(L.ptr(img), L.ptr(avg), L.ptr(w1), *s1, 1, L.ptr(b1),
                                            L.ptr(w2), *s2, L.W_F32, L.ptr(b2), b, h, w, L.ptr(x0),
                                            L.ptr(y1), L.ptr(a), L.ptr(idx), L.ptr(ws), ws.numel(), st))

    for _ in range(10):
        two()
        one()
    torch.cuda.synchronize()
    for rnd in range(args.rounds):
        res = {}
        for name, fn in (('two kernels', two), ('one kernel', one)):
            with L.KernelTimer(capacity=8 * args.iters) as kt:
                for _ in range(args.iters):
                    fn()
                torch.cuda.synchronize()
            res[name] = {k: round(ms * 1e3, 1) for k, (cnt, ms) in kt.summary().items() if 'pack' not in k}
        print(json.dumps({'round': rnd, **res}))


if __name__ == '__main__':
    main()
