#!/usr/bin/env python
"""Per-layer timing of the VGG16 3x3 convolutions as the library runs them (MIOpen / CK through
PyTorch, bf16 channels-last, find mode on) next to the hand-written kernels where they exist:
forward, backward-data and weight-gradient, microseconds and TFLOP/s.

    python scripts/conv_layers.py [--batch 24] [--height 480] [--width 640]
"""
import argparse
import os
import sys

os.environ.setdefault('MIOPEN_DEBUG_CONV_DIRECT_NAIVE_CONV_FWD', '0')
os.environ.setdefault('MIOPEN_DEBUG_CONV_DIRECT_NAIVE_CONV_BWD', '0')
os.environ.setdefault('MIOPEN_DEBUG_CONV_DIRECT_NAIVE_CONV_WRW', '0')
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import torch  # noqa: E402

from soft_contrastive_learning_amd.model import nets  # noqa: E402

LAYERS = [('1_1', 3, 64, 1), ('1_2', 64, 64, 1), ('2_1', 64, 128, 2), ('2_2', 128, 128, 2),
          ('3_1', 128, 256, 4), ('3_2', 256, 256, 4), ('4_1', 256, 512, 8), ('4_2', 512, 512, 8),
          ('5_1', 512, 512, 16)]
# (the reference's own training resolution, train/train.py:423-428: --height 180 --width 240
# --batch 25 -> conv4 maps 22 x 30, conv5 11 x 15; configs[0]: --height 224 --width 224 --batch 4)
if os.environ.get('SCL_LAYERS'):
    LAYERS = [l for l in LAYERS if l[0] in os.environ['SCL_LAYERS'].split(',')]
ONES = [1, 1]


def timed(fn, iters=5):
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
    ap.add_argument('--batch', type=int, default=24)
    ap.add_argument('--height', type=int, default=480)
    ap.add_argument('--width', type=int, default=640)
    args = ap.parse_args()
    torch.backends.cudnn.benchmark = True
    dev = torch.device('cuda:0')
    print('%-5s %4s %4s %9s | %22s | %22s | %22s' % ('layer', 'cin', 'cout', 'GFLOP', 'forward us (TF/s)',
                                                      'bwd-data us (TF/s)', 'wrw us (TF/s)'))
    for name, cin, cout, div in LAYERS:
        h, w = args.height // div, args.width // div
        x = torch.randn(args.batch, cin, h, w, device=dev).bfloat16().contiguous(
            memory_format=torch.channels_last)
        wt = (torch.randn(cout, cin, 3, 3, device=dev) * 0.05).bfloat16().contiguous(
            memory_format=torch.channels_last)
        gy = torch.randn(args.batch, cout, h, w, device=dev).bfloat16().contiguous(
            memory_format=torch.channels_last)
        gf = 2.0 * args.batch * h * w * cin * cout * 9 / 1e9

        def fwd():
            return torch.ops.aten.convolution(x, wt, None, ONES, ONES, ONES, False, [0, 0], 1)

        def bwd():
            return torch.ops.aten.convolution_backward(gy, x, wt, None, ONES, ONES, ONES, False,
                                                       [0, 0], 1, [True, False, False])

        def wrw():
            return torch.ops.aten.convolution_backward(gy, x, wt, None, ONES, ONES, ONES, False,
                                                       [0, 0], 1, [False, True, False])
        t = [timed(fwd), timed(bwd) if cin > 3 else float('nan'), timed(wrw)]
        cells = ['%8.1f (%6.0f)' % (v, gf / v * 1e3) for v in t]
        print('%-5s %4d %4d %9.1f | %22s | %22s | %22s' % (name, cin, cout, gf, *cells))
        if nets._conv64_ok(x, wt):
            t = [timed(lambda: nets.conv64(x, wt, False)),
                 timed(lambda: nets.conv64(gy, wt, True)) if nets._conv64_ok(gy, wt, True)
                 else float('nan'),
                 timed(lambda: nets.wrw64(x, gy, wt)) if nets._own_wrw_ok(x, gy, wt) else float('nan')]
            cells = ['%8.1f (%6.0f)' % (v, gf / v * 1e3) for v in t]
            print('%-5s %4s %4s %9s | %22s | %22s | %22s' % ('  own', '', '', '', *cells))
            bias = torch.randn(cout, device=dev)
            t = [timed(lambda: nets.conv64(x, wt, False, bias=bias, relu=True)),
                 timed(lambda: nets.conv64(x, wt, False, bias=bias, pool=True))
                 if nets._own_conv_kind(x, wt) == 'reg' else float('nan'), float('nan')]
            cells = ['%8.1f (%6.0f)' % (v, gf / v * 1e3) for v in t]
            print('%-5s %4s %4s %9s | %22s | %22s | %22s' % (' +b/p', '', '', '', *cells))


if __name__ == '__main__':
    main()
