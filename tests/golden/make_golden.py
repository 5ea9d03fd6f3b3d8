#!/usr/bin/env python
"""Generates tests/golden/golden_v1.json from the CPU oracle.

The reference cannot be imported in the build container (no TensorFlow, third-party
modules absent — SURVEY.md F3-F6), so these vectors come from the oracle restatement and
are therefore "parity unpinned" beyond the hand-derived KATs of tests/test_oracle_kats.py.
They freeze the oracle's outputs: tests/test_golden.py checks (i) that the oracle still
reproduces them (no silent drift of the checker) and (ii) on the GPU, that the HIP path
matches them.

Small cases carry their inputs inline; larger ones are regenerated from the seeded
builders in tests/util_data.py (NumPy Generators are platform-stable).

    python tests/golden/make_golden.py          # rewrites golden_v1.json
"""
import json
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)

from oracle import losses_np as O  # noqa: E402
from oracle import netvlad_np as NV  # noqa: E402
from oracle import topn_np as TN  # noqa: E402
from oracle import twin_torch as TT  # noqa: E402
from tests import util_data as U  # noqa: E402


def grad_summary(g):
    g = np.asarray(g, dtype=np.float64)
    flat = g.reshape(-1)
    pick = np.linspace(0, flat.size - 1, 8).astype(int)
    return {'norm': float(np.linalg.norm(flat)), 'sum': float(flat.sum()),
            'idx': pick.tolist(), 'val': flat[pick].tolist()}


def wms_case(name, b, e, side, seed=99, rank3=True, asym=0.0, mix=0.9, **kw):
    emb = U.embeddings(b, e, seed=seed, mix=mix)
    dist = U.positions_distances(b, side=side)
    if asym:
        dist = dist + np.triu(np.ones_like(dist), 1) * np.float32(asym)
    d = dist[None] if rank3 else dist
    f32 = float(O.wms_loss(d, emb, 0.8, 15.0, **kw))
    t = torch.tensor(emb, dtype=torch.float64, requires_grad=True)
    l64 = TT.wms_loss(d, t, 0.8, 15.0, **kw)
    l64.backward()
    return {'name': name, 'kind': 'wms', 'b': b, 'e': e, 'side': side, 'seed': seed, 'mix': mix,
            'rank3': rank3, 'asym': asym, 'kw': kw, 'loss_f32': f32, 'loss_f64': float(l64),
            'grad': grad_summary(t.grad.numpy())}


def ms_case(name, t_, p, n, e, mining):
    b = t_ * (1 + p + n)
    emb = U.embeddings(b, e, seed=17)
    labels = O.trainer_ms_labels(t_, p, n)
    f32 = float(O.ms_loss(labels, emb, ms_mining=mining))
    x = torch.tensor(emb, dtype=torch.float64, requires_grad=True)
    l64 = TT.ms_loss(labels, x, ms_mining=mining)
    l64.backward()
    return {'name': name, 'kind': 'ms', 't': t_, 'p': p, 'n': n, 'e': e, 'mining': mining,
            'loss_f32': f32, 'loss_f64': float(l64), 'grad': grad_summary(x.grad.numpy())}


def tuple_case(fn, quad, t_, p, n, e):
    shape = [1, p, n] + ([1] if quad else [])
    out = U.tuple_batch(t_, p, n, e, quad=quad)
    flat = out.reshape(t_ * sum(shape), e)
    margins = (0.5, 0.2) if quad else (0.5,)
    f32 = float(getattr(O, fn)(*O.split_tuples(flat, t_, shape), *margins))
    x = torch.tensor(flat, dtype=torch.float64, requires_grad=True)
    l64 = getattr(TT, fn)(*torch.split(x.reshape(t_, sum(shape), e), shape, dim=1), *margins)
    l64.backward()
    return {'name': fn, 'kind': 'tuple', 'fn': fn, 'quad': quad, 't': t_, 'p': p, 'n': n, 'e': e,
            'margins': list(margins), 'loss_f32': f32, 'loss_f64': float(l64),
            'grad': grad_summary(x.grad.numpy())}


def tiny_inline_wms():
    rng = np.random.default_rng(1)
    emb = rng.standard_normal((5, 6)).astype(np.float32)
    dist = U.positions_distances(5, side=30.0, seed=2)
    return {'name': 'wms_tiny_inline', 'kind': 'wms_inline', 'emb': emb.tolist(),
            'dist': dist.tolist(), 'loss_f32': float(O.wms_loss(dist[None], emb, 0.8, 15.0)),
            'loss_f32_plain': float(O.wms_loss(dist[None], emb, 0.8, 15.0, sumfunction='plain')),
            'loss_f32_nomining': float(O.wms_loss(dist[None], emb, 0.8, 15.0, ms_mining=False))}


def netvlad_case(b, n, seed):
    x = U.feature_map(b, n, seed=seed)
    w, c = U.vlad_params()
    out = NV.netvlad_fused(x, w, c)
    g = np.random.default_rng(2).standard_normal((b, 32768)).astype(np.float32)
    x64 = torch.tensor(x, dtype=torch.float64, requires_grad=True)
    w64 = torch.tensor(w, dtype=torch.float64, requires_grad=True)
    c64 = torch.tensor(c, dtype=torch.float64, requires_grad=True)
    TT.netvlad(x64, w64, c64).backward(torch.tensor(g, dtype=torch.float64))
    return {'name': 'netvlad_b%d_n%d' % (b, n), 'kind': 'netvlad', 'b': b, 'n': n, 'seed': seed,
            'out': grad_summary(out), 'grad_x': grad_summary(x64.grad.numpy()),
            'grad_w': grad_summary(w64.grad.numpy()), 'grad_c': grad_summary(c64.grad.numpy())}


def topn_case(r, q, d, n):
    ref, qry = U.retrieval_sets(r, q, d)
    dist, idx = TN.topn_kdtree(ref, qry, n)
    return {'name': 'topn_r%d_q%d_d%d' % (r, q, d), 'kind': 'topn', 'r': r, 'q': q, 'd': d, 'n': n,
            'idx': idx.tolist(), 'dist_first_row': dist[0].tolist()}


def logratio_case():
    p = n = 4
    e = 512
    out = U.tuple_batch(1, p, n, e, seed=33)
    rng = np.random.default_rng(34)
    spd = rng.uniform(1, 200, (1, p, 1)).astype(np.float32)
    snd = rng.uniform(300, 4000, (1, n, 1)).astype(np.float32)
    a, pos, neg = O.split_tuples(out.reshape(-1, e), 1, [1, p, n])
    x = torch.tensor(out, dtype=torch.float64, requires_grad=True)
    l64 = TT.logratio_loss(*torch.split(x, [1, p, n], dim=1), spd, snd)
    l64.backward()
    return {'name': 'logratio', 'kind': 'logratio', 'p': p, 'n': n, 'e': e,
            'loss_f32': float(O.logratio_loss(a, pos, neg, spd, snd)), 'loss_f64': float(l64),
            'grad': grad_summary(x.grad.numpy())}


def main():
    cases = [
        tiny_inline_wms(),
        wms_case('wms_cfg2_b24', 24, 32768, 60.0),
        wms_case('wms_b25_reference_tuple', 25, 32768, 60.0),
        wms_case('wms_far_pairs_b64', 64, 4096, 200.0),
        wms_case('wms_lin', 16, 512, 40.0, wfunction='lin'),
        wms_case('wms_tanh_plain', 16, 512, 40.0, wfunction='tanh', sumfunction='plain'),
        wms_case('wms_nomining', 16, 512, 40.0, ms_mining=False),
        wms_case('wms_rank3_asym', 12, 256, 20.0, seed=3, asym=9.0, mix=3.0),
        wms_case('wms_rank2_asym', 12, 256, 20.0, seed=3, rank3=False, asym=9.0, mix=3.0),
        ms_case('ms_t2', 2, 2, 3, 1024, True),
        ms_case('ms_t1_nomining', 1, 12, 12, 4096, False),
        logratio_case(),
        netvlad_case(2, 50, 7),
        netvlad_case(3, 196, 8),
        topn_case(500, 20, 64, 10),
    ]
    for fn, quad in (('triplet_loss', False), ('lazy_triplet_loss', False),
                     ('evil_triplet_loss', False), ('quadruplet_loss', True),
                     ('lazy_quadruplet_loss', True), ('evil_quadruplet_loss', True)):
        cases.append(tuple_case(fn, quad, 2, 3, 4, 512))
    out = {'version': 1, 'generator': 'tests/golden/make_golden.py',
           'note': 'oracle outputs; parity unpinned by the reference (no tests upstream)',
           'cases': cases}
    with open(os.path.join(HERE, 'golden_v1.json'), 'w') as f:
        json.dump(out, f, indent=1)
    print('wrote', len(cases), 'cases')


if __name__ == '__main__':
    main()
