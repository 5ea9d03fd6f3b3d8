#!/usr/bin/env python
"""The pre-pass density of the threshold retrieval scan (diagnostic build: SCL_TAU_STRIDE): a denser
sample costs more pre-pass and gives a tighter threshold = fewer appended candidates.  One child
process per stride (the library reads the variable once); kernels of a certified call at configs[4].

    python scripts/topn_tau_stride_ab.py [--strides 8,16,32]
"""
import argparse
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def child(score):
    sys.path.insert(0, ROOT)
    import torch
    from soft_contrastive_learning_amd import _lib as L
    from soft_contrastive_learning_amd.evaluation import retrieval
    from tests import util_data as U
    L.use_diag()
    dev = torch.device('cuda:0')
    ref, qry = U.retrieval_sets(100000, 10000, 256)
    rt, qt = torch.tensor(ref, device=dev), torch.tensor(qry, device=dev)
    st = {}
    retrieval.topn_l2(rt, qt, 25, score=score, stats=st)
    torch.cuda.synchronize()
    with L.KernelTimer(capacity=256) as kt:
        for _ in range(3):
            retrieval.topn_l2(rt, qt, 25, score=score)
        torch.cuda.synchronize()
    ks = {k: round(ms * 1e3, 1) for k, (c, ms) in sorted(kt.summary().items())}
    print(json.dumps({'stride': os.environ.get('SCL_TAU_STRIDE'), 'score': score, 'uncertified': st.get('uncertified'),
                      'kernel_us': ks, 'sum_ms': round(sum(ks.values()) / 1e3, 3)}))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--strides', default='8,16,32')
    ap.add_argument('--child', default='')
    args = ap.parse_args()
    if args.child:
        return child(args.child)
    for rounds in range(2):
        for s in args.strides.split(','):
            for score in ('bf16x3', 'f32'):
                subprocess.run([sys.executable, os.path.abspath(__file__), '--child', score],
                               env=dict(os.environ, SCL_TAU_STRIDE=s), check=True, stderr=subprocess.DEVNULL)


if __name__ == '__main__':
    main()
