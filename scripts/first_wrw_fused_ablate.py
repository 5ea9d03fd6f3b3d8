#!/usr/bin/env python
"""Timing ablations of conv3x3_kernel<64, 64, 3, 1, FW = 1> (conv1_2's backward-data pass with
conv1_1's parameter gradients computed on its LDS tile) at the bench shape, next to the two kernels
it replaces.  scl_debug_set_variant(61000 + bits): 8 no im2col pass, 16 no products, 32 the tile is
not written to LDS (results meaningless for every bit).

    python scripts/first_wrw_fused_ablate.py [--batch 24 --height 480 --width 640]
"""
import argparse
import os
import sys

os.environ.setdefault('SCL_DIAG', '1')   # the diagnostic build carries the variants (csrc/Makefile)
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import torch  # noqa: E402

from soft_contrastive_learning_amd import _lib as L  # noqa: E402
from soft_contrastive_learning_amd.model import nets  # noqa: E402


def timed(fn, iters=10):
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
    ap.add_argument('--stamps', action='store_true', help='in-kernel clock stamps of waves 0 and 7 (variant 61064 + 128 w)')
    args = ap.parse_args()
    dev = torch.device('cuda:0')
    lib = L.load()
    b, h, w = args.batch, args.height, args.width
    g = torch.Generator(device=dev).manual_seed(1)
    cl = torch.channels_last
    y1 = torch.relu(torch.randn(b, 64, h, w, device=dev, generator=g)).bfloat16().contiguous(memory_format=cl)
    ga = torch.randn(b, 64, h // 2, w // 2, device=dev, generator=g)
    ga = (ga * (torch.rand(ga.shape, device=dev, generator=g) > 0.5)).bfloat16().contiguous(memory_format=cl)
    idx = torch.randint(0, 4, ga.shape, device=dev, generator=g, dtype=torch.uint8).contiguous(memory_format=cl)
    x0 = torch.randn(b, h, w, 3, device=dev, generator=g).bfloat16()
    w1 = (torch.randn(64, 3, 3, 3, device=dev, generator=g) * 0.2)
    b1 = torch.zeros(64, device=dev)
    w2 = (torch.randn(64, 64, 3, 3, device=dev, generator=g) * 0.05)
    link = nets._GradLink()

    def two():
        gx = nets.conv64(ga, w2, True, mask=y1, pool_idx=idx)
        return nets.first_wrw(x0.permute(0, 3, 1, 2), gx, w1, None, w1)
    print('two kernels (conv3x3<pooled> + conv_first_wrw + reduce + davg): %.1f us' % timed(two))
    print('  conv3x3<pooled> alone: %.1f us' % timed(lambda: nets.conv64(ga, w2, True, mask=y1, pool_idx=idx)))
    for var, what in ((0, 'fused, product kernel'), (61008, 'no im2col pass'), (61016, 'no products'),
                      (61024, 'neither'), (61032, 'tile not written to LDS'), (61056, 'none of the three')):
        lib.scl_debug_set_variant(var)
        try:
            t = timed(lambda: nets._masked_pooled_first_wrw(ga, idx, w2, y1, (x0, w1, b1)))
        finally:
            lib.scl_debug_set_variant(0)
        print('variant %5d  %-26s %.1f us' % (var, what, t))
    if args.stamps:
        for wave in (0, 7):
            print_stamps(stamps(lib, ga, idx, w2, y1, x0, w1, wave), wave)


def stamps(lib, ga, idx, w2, y1, x0, w1, wave):
    """One fused launch under variant 61064 + 128 * wave; returns the [workgroups][4 tiles][10] stamps."""
    dev = y1.device
    b, _, h, w = y1.shape
    gw1 = torch.empty(64, 3, 3, 3, device=dev)
    gb1 = torch.empty(64, device=dev)
    davg = torch.empty(3, device=dev)
    ws = L.workspace(lib.scl_conv3x3_workspace_bytes(), dev)
    fws = torch.zeros(lib.scl_conv_first_wrw_workspace_bytes(), dtype=torch.uint8, device=dev)
    s2, s1 = w2.stride(), gw1.stride()
    lib.scl_debug_set_variant(61064 + 128 * wave)
    try:
        L.check(lib.scl_conv3x3_masked_pooled_first_wrw(
            L.ptr(ga), L.ptr(idx), L.ptr(w2), s2[0], s2[1], s2[2], s2[3], L.W_F32, b, h, w, L.ptr(y1), L.ptr(x0),
            L.ptr(gw1), s1[0], s1[1], s1[2], s1[3], 1, L.ptr(gb1), L.ptr(w1), L.ptr(davg), L.ptr(ws),
            ws.numel(), L.ptr(fws), fws.numel(), L.stream_of(y1)))
        torch.cuda.synchronize()
    finally:
        lib.scl_debug_set_variant(0)
    st = fws.view(torch.int64)[1024 * 2048 // 2:1024 * 2048 // 2 + 256 * 4 * 12].view(256, 4, 12)[:, :, :10]
    return st.cpu().numpy()


def print_stamps(st, wave):
    import numpy as np
    names = ['K loop (im2col inside)', 'own loads landed', 'barrier 1', 'masked rows -> LDS planes', 'barrier 2',
             '(empty)', '(empty)', 'products + partial + x0 store', 'barrier 3']
    ok = st[:, :, 0] > 0
    d = np.diff(st.astype(np.int64), axis=2)                    # [wg][tile][9]
    tile = (st[:, :, 9] - st[:, :, 0])
    print('wave %d: cycles per interval, median over %d (workgroup, tile) samples (tiles 2..5 of each workgroup)'
          % (wave, int(ok.sum())))
    for k, n in enumerate(names):
        print('  %-32s %8.0f' % (n, float(np.median(d[:, :, k][ok]))))
    print('  %-32s %8.0f' % ('whole tile', float(np.median(tile[ok]))))


if __name__ == '__main__':
    main()
