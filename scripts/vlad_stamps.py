#!/usr/bin/env python
"""Where the fused NetVLAD kernels (csrc/netvlad.hip: vlad_fwd_kernel, vlad_dx_kernel) spend their
cycles: runs them with scl_debug_set_variant(916 / 917) — wave 0 of every workgroup writes
shader-clock stamps into the tail of the workspace — and prints the median over workgroups of
every phase, in cycles.  DIAGNOSTIC ONLY (the stamps perturb the kernel)."""
import argparse
import os

os.environ.setdefault('SCL_DIAG', '1')   # the diagnostic build carries the variants (csrc/Makefile)
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402
import torch  # noqa: E402

from soft_contrastive_learning_amd import _lib as L  # noqa: E402
from tests import util_data as U  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--kernel', default='fwd', choices=['fwd', 'fwd8', 'dx'])
    ap.add_argument('--batch', type=int, default=24)
    ap.add_argument('--locations', type=int, default=1200)
    args = ap.parse_args()
    lib = L.load()
    dev = torch.device('cuda:0')
    b, n = args.batch, args.locations
    x = torch.tensor(U.feature_map(b, n, seed=5), device=dev).bfloat16()
    w, c = U.vlad_params()
    wt, ct = torch.tensor(w, device=dev), torch.tensor(c, device=dev)
    out = torch.empty(b, 32768, device=dev)
    sa, sl = torch.empty(b, n, 64, device=dev), torch.empty(b, n, 64, device=dev)
    sr, sv = torch.empty(b, n, device=dev), torch.empty(b, L.VLAD_SAVE_ROWS, 64, device=dev)
    ws = L.workspace(lib.scl_netvlad_fwd_workspace_bytes(b, n), dev)
    steps = (n + 31) // 32
    per = -(-steps * b // 256)
    s_cnt = -(-steps // per)
    nwg = b * s_cnt
    lib.scl_debug_set_variant({'fwd': 916, 'fwd8': 918}.get(args.kernel, 0))
    try:
        for _ in range(3):
            ws.zero_()
            L.check(lib.scl_netvlad_fwd(L.ptr(x), L.DT_BF16, L.ptr(wt), L.ptr(ct), b, n, 1, L.ptr(out),
                                        L.ptr(sa), L.ptr(sl), L.ptr(sr), L.ptr(sv), L.ptr(ws), ws.numel(),
                                        L.stream_of(x)))
        torch.cuda.synchronize()
        if args.kernel == 'dx':
            go = torch.randn(b, 32768, device=dev)
            gx, gw, gc = torch.empty_like(x), torch.empty_like(wt), torch.empty_like(ct)
            ws = L.workspace(lib.scl_netvlad_bwd_workspace_bytes(b, n), dev)
            lib.scl_debug_set_variant(917)
            for _ in range(3):
                ws.zero_()
                L.check(lib.scl_netvlad_bwd(L.ptr(x), L.DT_BF16, L.ptr(wt), L.ptr(ct), L.ptr(go), L.ptr(sa),
                                            L.ptr(sl), L.ptr(sr), L.ptr(sv), b, n, 1, L.ptr(gx), L.ptr(gw),
                                            L.ptr(gc), L.ptr(ws), ws.numel(), L.stream_of(x)))
            torch.cuda.synchronize()
    finally:
        lib.scl_debug_set_variant(0)
    tail = ws[-nwg * 32 * 8:].cpu().numpy().view(np.uint64).reshape(nwg, 32).astype(np.int64)
    if args.kernel == 'dx':
        names = {0: 'entry', 1: 'operand registers loaded', 2: 'tile 0 fragments staged', 13: 'last epilogue issued',
                 14: 'stores complete'}
        names.update({3 + t: 'after tile %d (+ epilogue of tile %d)' % (t, t - 1) for t in range(1, 9)})
        names.update({16: 'tile 2: inputs requested', 17: 'tile 2: MFMAs issued', 18: 'tile 2: epilogue of tile 1 issued',
                      19: 'tile 2: next fragments staged', 20: 'tile 2: barrier passed'})
    elif args.kernel == 'fwd8':
        names = {0: 'entry', 1: 'stage DMA issued, W loads issued', 28: 'loop done', 29: 'slab stores issued',
                 30: 'slab stores complete'}
        names.update({2: 'stage 0 + W landed, barrier', 3: 'partial logits of step 0 + barrier', 28: 'last aggregation + slab stores issued'})
        for st in range(4):
            names.update({4 + 6 * st: 'step %d: block P starts' % st,
                          5 + 6 * st: 'step %d: P done (aggregation of step %d || softmax part 1)' % (st, st - 1),
                          6 + 6 * st: 'step %d: stage wait + barrier' % st,
                          7 + 6 * st: 'step %d: Q done (DMA issue, logits of step %d || softmax part 2)' % (st, st + 1),
                          8 + 6 * st: 'step %d: barrier' % st})
    else:
        names = {0: 'entry', 1: 'stage DMA issued, W loads issued', 2: 'W + first stages landed', 28: 'loop done',
                 29: 'slab stores issued', 30: 'slab stores complete'}
        for st in range(4):
            names.update({4 + 6 * st: 'step %d: stage landed + barrier' % st, 5 + 6 * st: 'step %d: logits done' % st,
                          6 + 6 * st: 'step %d: softmax exchange barrier' % st,
                          7 + 6 * st: 'step %d: coefficients written, outputs stored' % st,
                          8 + 6 * st: 'step %d: aggregation done' % st})
    t0 = tail[:, 0:1]
    rel = tail - t0
    prev = 0
    print('workgroups %d, start spread (cycles): %d' % (nwg, int(tail[:, 0].max() - tail[:, 0].min())))
    for k in sorted(names):
        col = rel[:, k]
        col = col[tail[:, k] != 0]
        if col.size == 0:
            continue
        med = int(np.median(col))
        print('%-48s median %7d  (+%6d)   p90 %7d' % (names[k], med, med - prev, int(np.percentile(col, 90))))
        prev = med


if __name__ == '__main__':
    main()
