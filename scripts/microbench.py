#!/usr/bin/env python
"""Kernel-only microbenchmarks of the hand-written hot path (no VGG backbone, so no MIOpen
warm-up): NetVLAD forward+backward at the bench shape, the Gram-loss B-sweep of SURVEY H3,
and the retrieval configuration (BASELINE.json configs[4]).

    python scripts/microbench.py [--what netvlad,loss,topn] [--iters 20] [--json out.json]

Per-kernel durations come from the library's scl_prof_* sink (HIP events on the launch
stream).  Under rocprofv3 run this file directly:  rocprofv3 ... -- python3 scripts/microbench.py
"""
import argparse
import json
import os

os.environ.setdefault('SCL_DIAG', '1')   # the diagnostic build carries the variants (csrc/Makefile)
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402
import torch  # noqa: E402

import bench  # noqa: E402  (kernel_models / price)
from soft_contrastive_learning_amd import _lib  # noqa: E402
from soft_contrastive_learning_amd.evaluation import retrieval  # noqa: E402
from soft_contrastive_learning_amd.model import losses, nets  # noqa: E402
from tests import util_data as U  # noqa: E402


def run_netvlad(dev, b, n, dtype, iters):
    x = torch.tensor(U.feature_map(b, n, seed=5), device=dev).to(dtype).reshape(b, 1, n, 512)
    x.requires_grad_(True)
    w, c = U.vlad_params()
    wt = torch.tensor(w, device=dev, requires_grad=True)
    ct = torch.tensor(c, device=dev, requires_grad=True)
    g = torch.randn(b, 32768, device=dev)
    for _ in range(3):
        nets.netvlad(x, wt, ct, True).backward(g)
    torch.cuda.synchronize()
    with _lib.KernelTimer(capacity=32 * iters) as kt:
        for _ in range(iters):
            nets.netvlad(x, wt, ct, True).backward(g)
        torch.cuda.synchronize()
    models = bench.kernel_models(b, n, b, 2 if dtype == torch.bfloat16 else 4)
    return [bench.price(k, cnt, ms, models.get(k, dict(flops=0.0, bytes=0.0)))
            for k, (cnt, ms) in sorted(kt.summary().items())]


def run_loss(dev, b, iters):
    emb = torch.tensor(U.embeddings(b, 32768), device=dev, requires_grad=True)
    dist = torch.tensor(U.positions_distances(b)[None], device=dev)
    for _ in range(3):
        losses.wms_loss(dist, emb, 0.8, 15.0).backward()
    torch.cuda.synchronize()
    with _lib.KernelTimer(capacity=16 * iters) as kt:
        for _ in range(iters):
            losses.wms_loss(dist, emb, 0.8, 15.0).backward()
        torch.cuda.synchronize()
    models = bench.kernel_models(b, 1200, b, 4)
    rows = [bench.price(k, cnt, ms, models.get(k, dict(flops=0.0, bytes=0.0)))
            for k, (cnt, ms) in sorted(kt.summary().items())]
    for r in rows:
        r['B'] = b
    return rows


def run_topn(dev, r, q, d, n, iters, score='f32'):
    ref, qry = U.retrieval_sets(r, q, d)
    rt, qt = torch.tensor(ref, device=dev), torch.tensor(qry, device=dev)
    retrieval.topn_l2(rt, qt, n, score=score)
    torch.cuda.synchronize()
    with _lib.KernelTimer(capacity=8 * iters) as kt:
        t0 = time.perf_counter()
        for _ in range(iters):
            retrieval.topn_l2(rt, qt, n, score=score)
        torch.cuda.synchronize()
        wall = (time.perf_counter() - t0) / iters
    out = []
    for k, (cnt, ms) in sorted(kt.summary().items()):
        row = dict(kernel=k, launches=cnt, us=round(ms * 1e3, 1))
        if k.startswith('topn_scan'):
            row.update(bench.price_topn_scan(ms, q, r, d, score))
        out.append(row)
    out.append(dict(kernel='topn_l2 (whole call)', us=round(wall * 1e6, 1),
                    queries_per_sec=round(q / wall, 1), R=r, Q=q, d=d, n=n, score=score))
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--what', default='netvlad,loss,topn')
    ap.add_argument('--iters', type=int, default=20)
    ap.add_argument('--json', default='')
    ap.add_argument('--netvlad-batches', default='24')
    ap.add_argument('--topn-refs', type=int, default=100000)
    ap.add_argument('--topn-queries', type=int, default=10000)
    ap.add_argument('--topn-score', default='f32', help="f32, bf16x3 or both (comma list)")
    ap.add_argument('--netvlad-variants', default='',
                    help='comma list of scl_debug_set_variant values to time besides production')
    ap.add_argument('--loss-batches', default='24,48,96,192')
    ap.add_argument('--loss-variants', default='',
                    help='comma list of scl_debug_set_variant values for the loss sweep (35 = multi-launch forward)')
    ap.add_argument('--loss-splits', default='',
                    help='comma list of forced Gram K-split counts for B <= 256 (tuning)')
    ap.add_argument('--topn-splits', default='',
                    help='comma list of forced reference-split counts (tuning; default: planner)')
    args = ap.parse_args()
    dev = torch.device('cuda:0')
    what = args.what.split(',')
    res = {}
    if 'netvlad' in what:
        for b in [int(v) for v in args.netvlad_batches.split(',')]:
            res['netvlad_bf16_b%d_n1200' % b] = run_netvlad(dev, b, 1200, torch.bfloat16, args.iters)
            res['netvlad_f32_b%d_n1200' % b] = run_netvlad(dev, b, 1200, torch.float32, args.iters)
            for var in [int(v) for v in args.netvlad_variants.split(',') if v]:
                _lib.load().scl_debug_set_variant(var)
                try:
                    res['netvlad_bf16_b%d_variant_%d' % (b, var)] = run_netvlad(
                        dev, b, 1200, torch.bfloat16, args.iters)
                finally:
                    _lib.load().scl_debug_set_variant(0)
    if 'loss' in what:
        lb = [int(v) for v in args.loss_batches.split(',')]
        res['wms_loss_sweep'] = sum((run_loss(dev, b, args.iters) for b in lb), [])
        for var in [int(v) for v in args.loss_variants.split(',') if v]:
            _lib.load().scl_debug_set_variant(var)
            try:
                res['wms_loss_variant_%d' % var] = sum((run_loss(dev, b, args.iters) for b in lb), [])
            finally:
                _lib.load().scl_debug_set_variant(0)
        for sp in [int(v) for v in args.loss_splits.split(',') if v]:
            _lib.load().scl_debug_set_variant(100000 * sp)
            try:
                res['wms_loss_splits_%d' % sp] = sum((run_loss(dev, b, args.iters) for b in lb), [])
            finally:
                _lib.load().scl_debug_set_variant(0)
    if 'topn' in what:
        scores = args.topn_score.split(',')
        for sc in scores:
            res['topn' if sc == 'f32' else 'topn_' + sc] = run_topn(
                dev, args.topn_refs, args.topn_queries, 256, 25, max(2, args.iters // 10), sc)
        for sp in [int(v) for v in args.topn_splits.split(',') if v]:
            # < 100: forced split count; otherwise a raw scl_debug_set_variant value
            # (1000 * ablation bits [1 no selection, 2 no staging, 4 no MFMAs] + 100 + splits)
            _lib.load().scl_debug_set_variant(100 + sp if sp < 100 else sp)
            try:
                res['topn_%s_splits_%d' % (scores[-1], sp)] = run_topn(
                    dev, args.topn_refs, args.topn_queries, 256, 25, max(2, args.iters // 10),
                    scores[-1])
            finally:
                _lib.load().scl_debug_set_variant(0)
    for name, rows in res.items():
        print('==', name)
        for r in rows:
            print('  ', json.dumps(r))
    if args.json:
        with open(args.json, 'w') as f:
            json.dump(res, f, indent=1)


if __name__ == '__main__':
    main()
