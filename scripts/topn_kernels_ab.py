#!/usr/bin/env python
"""Retrieval at configs[4] (100k references x 10k queries x 256, top 25): per-kernel event durations of
the threshold scheme (product) against the sorted-list scan (diagnostic variant 8100), both score modes.

    python scripts/topn_kernels_ab.py [--refs 100000] [--queries 10000]
"""
import argparse
import json
import os
import sys

os.environ.setdefault('SCL_DIAG', '1')
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import torch  # noqa: E402

from soft_contrastive_learning_amd import _lib as L  # noqa: E402
from soft_contrastive_learning_amd.evaluation import retrieval  # noqa: E402
from tests import util_data as U  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--refs', type=int, default=100000)
    ap.add_argument('--queries', type=int, default=10000)
    ap.add_argument('--d', type=int, default=256)
    ap.add_argument('--iters', type=int, default=3)
    ap.add_argument('--variants', default='', help='more diagnostic variants, e.g. 1000,9001,9002,9004 (timing ablations)')
    args = ap.parse_args()
    dev = torch.device('cuda:0')
    L.use_diag()
    lib = L.load()
    ref, qry = U.retrieval_sets(args.refs, args.queries, args.d)
    rt, qt = torch.tensor(ref, device=dev), torch.tensor(qry, device=dev)
    for score in ('f32', 'bf16x3'):
        for v, name in ((0, 'threshold'), (8100, 'sorted_lists')) + (tuple((int(x), 'variant %s' % x) for x in args.variants.split(',')) if args.variants else ()):
            lib.scl_debug_set_variant(v)
            st = {}
            retrieval.topn_l2(rt, qt, 25, score=score, stats=st)
            torch.cuda.synchronize()
            with L.KernelTimer(capacity=64 * args.iters) as kt:
                for _ in range(args.iters):
                    retrieval.topn_l2(rt, qt, 25, score=score)
                torch.cuda.synchronize()
            ks = {k: round(ms * 1e3, 1) for k, (c, ms) in sorted(kt.summary().items())}
            print(json.dumps({'score': score, 'form': name, 'uncertified': st.get('uncertified'),
                              'kernel_us': ks, 'sum_ms': round(sum(ks.values()) / 1e3, 3)}))
    lib.scl_debug_set_variant(0)


if __name__ == '__main__':
    main()
