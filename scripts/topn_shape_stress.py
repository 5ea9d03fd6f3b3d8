#!/usr/bin/env python
"""Retrieval on odd shapes around the threshold scheme's switch-over (32768 references), both score
forms, against scikit-learn's KDTree (the reference's own call, evaluation/top-n.py:103-106).
One line per shape; exits non-zero on the first mismatch.  GPU box: python scripts/topn_shape_stress.py"""
import itertools
import sys
import os

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from soft_contrastive_learning_amd.evaluation import retrieval  # noqa: E402


def main():
    from sklearn.neighbors import KDTree
    dev = torch.device('cuda:0')
    rng = np.random.default_rng(2)
    shapes = [(32768, 1, 8, 1), (32769, 31, 24, 25), (33333, 129, 64, 32), (47111, 1000, 200, 25),
              (65537, 257, 256, 25), (131077, 130, 128, 7), (32767, 64, 256, 25), (40000, 5, 512, 25),
              (70001, 77, 16, 30)]
    bad = 0
    for (r, q, d, n), score, clustered in itertools.product(shapes, ('f32', 'bf16x3'), (False, True)):
        ref = rng.standard_normal((r, d)).astype(np.float32)
        if clustered:                     # a few tight clumps + an offset: near-ties and large norms
            centres = rng.standard_normal((37, d)).astype(np.float32) * 3
            ref = (centres[rng.integers(0, 37, r)] + 0.05 * ref + 10.0).astype(np.float32)
        qry = ref[rng.integers(0, r, q)] + 0.02 * rng.standard_normal((q, d)).astype(np.float32)
        st = {}
        got_d, got_i = retrieval.topn_l2(torch.tensor(ref, device=dev), torch.tensor(qry, device=dev), n,
                                         idx_offset=5, score=score, stats=st)
        want_d, want_i = KDTree(ref).query(qry, k=n, return_distance=True, sort_results=True)
        gi, gd = got_i.cpu().numpy() - 5, got_d.cpu().numpy()
        same = np.array_equal(gi, want_i)
        if not same:                      # exact ties (duplicated rows) may swap: compare distances then
            same = np.allclose(gd, want_d, rtol=1e-12, atol=0) and all(
                set(a) == set(b) or np.allclose(np.sort(x), np.sort(y), rtol=1e-12)
                for a, b, x, y in zip(gi, want_i, gd, want_d))
        derr = float(np.abs(gd - want_d).max() / max(want_d.max(), 1e-30))
        print('%-28s %-7s %-9s lists %s  dist %.1e  uncertified %s' % ((r, q, d, n), score,
              'clumped' if clustered else 'gaussian', 'equal' if same else 'DIFFER', derr, st.get('uncertified')))
        bad += not same or derr > 1e-10
    if bad:
        raise SystemExit('%d shapes differ' % bad)
    print('all equal')


if __name__ == '__main__':
    main()
