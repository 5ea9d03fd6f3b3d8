#!/usr/bin/env python
"""conv2_2 (128 -> 128 channels at 240 x 320): the register-weights kernel (csrc/conv64.hip, four waves,
288 weight registers) against the LDS-weights kernel (csrc/convh.hip, eight waves) — forward with the
pooled epilogue and masked backward-data, same box, microseconds per launch."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import torch  # noqa: E402

from soft_contrastive_learning_amd import _lib as L  # noqa: E402
from soft_contrastive_learning_amd.model import nets  # noqa: E402


def main():
    dev = torch.device('cuda:0')
    L.load()
    b, c, h, w = 24, 128, 240, 320
    g = torch.Generator().manual_seed(5)
    cl = torch.channels_last
    x = torch.relu(torch.randn(b, c, h, w, generator=g)).to(dev).bfloat16().contiguous(memory_format=cl)
    wt = (torch.randn(c, c, 3, 3, generator=g) * 0.03).to(dev)
    bias = torch.zeros(c, device=dev)
    gz = torch.randn(b, c, h, w, generator=g).to(dev).bfloat16().contiguous(memory_format=cl)
    reg_shapes = set(nets._OWN_CONV_SHAPES)
    outs = {}
    for rnd in range(3):
        res = {}
        for kind in ('reg', 'lds'):
            nets._OWN_CONV_SHAPES = reg_shapes if kind == 'reg' else reg_shapes - {(128, 128)}
            try:
                for name, fn in (('fwd+pool', lambda: nets.conv_pool_idx(x, wt, bias)),
                                 ('fwd+bias+relu', lambda: nets.conv64(x, wt, False, bias=bias, relu=True)),
                                 ('masked bwd', lambda: nets.conv64(gz, wt, True, mask=x))):
                    for _ in range(3):
                        o = fn()
                    torch.cuda.synchronize()
                    with L.KernelTimer(capacity=64) as kt:
                        for _ in range(10):
                            fn()
                        torch.cuda.synchronize()
                    res[kind + ' ' + name] = {k: round(ms * 1e3, 1) for k, (cnt, ms) in kt.summary().items()
                                              if 'pack' not in k}
                    outs[(kind, name)] = o
            finally:
                nets._OWN_CONV_SHAPES = reg_shapes
        print(json.dumps({'round': rnd, **res}))
    a0, a1 = outs[('reg', 'fwd+pool')], outs[('lds', 'fwd+pool')]
    print('pooled maps equal:', bool(torch.equal(a0[0], a1[0])), ' indices equal:', bool(torch.equal(a0[1], a1[1])),
          ' max |diff|:', float((a0[0].float() - a1[0].float()).abs().max()))


if __name__ == '__main__':
    main()
