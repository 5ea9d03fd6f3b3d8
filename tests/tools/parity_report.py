#!/usr/bin/env python
"""Measured parity of the HIP path against the CPU oracle, as numbers (the tests assert the
same comparisons against fixed tolerances; this prints / stores what was actually reached).

    python tests/tools/parity_report.py [--json profiles/rNN/parity_report.json]

Rows: relative loss error vs the float32 NumPy oracle (north_star gate: 1e-4), norm-relative
gradient error vs the float64 autograd twin, descriptor error of the NetVLAD head for both
input dtypes, index-list equality of the retrieval against scikit-learn's KDTree, and the worst
errors against the fixtures made by executing / running the reference's own files (golden_ref_*).
Needs an MI355X; uses oracle/ as the checker only.
"""
import argparse
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402
import torch  # noqa: E402

from oracle import losses_np as O  # noqa: E402
from oracle import netvlad_np as NV  # noqa: E402
from oracle import topn_np as TN  # noqa: E402
from oracle import twin_torch as TT  # noqa: E402
from soft_contrastive_learning_amd import pointnetvlad_cls as P  # noqa: E402
from soft_contrastive_learning_amd.evaluation import retrieval  # noqa: E402
from soft_contrastive_learning_amd.model import losses as M  # noqa: E402
from soft_contrastive_learning_amd.model import nets  # noqa: E402
from tests import util_data as U  # noqa: E402


def rel(a, b):
    a, b = float(a), float(b)
    return abs(a - b) / max(abs(b), 1e-30)


def nrel(got, want):
    got, want = np.asarray(got, np.float64), np.asarray(want, np.float64)
    return float(np.linalg.norm(got - want) / max(np.linalg.norm(want), 1e-30))


def maxrel(got, want):
    got, want = np.asarray(got, np.float64), np.asarray(want, np.float64)
    return float(np.abs(got - want).max() / np.abs(want).max())


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--json', default='')
    args = ap.parse_args()
    dev = torch.device('cuda:0')
    rows = []

    def add(name, **kw):
        rows.append(dict(case=name, **kw))
        print('%-58s %s' % (name, '  '.join('%s=%.3g' % (k, v) if isinstance(v, float)
                                              else '%s=%s' % (k, v) for k, v in kw.items())))

    # ---- pairwise losses -----------------------------------------------------------------
    for b in (24, 192):
        emb = U.embeddings(b, 32768)
        dist = U.positions_distances(b, side=60.0 if b < 64 else 200.0)
        for wf, sf in (('exp', 'ms'), ('lin', 'plain'), ('tanh', 'ms')):
            want = O.wms_loss(dist[None], emb, 0.8, 15.0, wfunction=wf, sumfunction=sf)
            e64 = torch.tensor(emb, dtype=torch.float64, requires_grad=True)
            l64 = TT.wms_loss(dist[None], e64, 0.8, 15.0, wfunction=wf, sumfunction=sf)
            l64.backward()
            et = torch.tensor(emb, device=dev, requires_grad=True)
            loss = M.wms_loss(torch.tensor(dist[None], device=dev), et, 0.8, 15.0, wfunction=wf,
                              sumfunction=sf)
            loss.backward()
            add('wms_loss B=%d %s/%s' % (b, wf, sf), loss_rel_vs_f32_oracle=rel(loss, want),
                loss_rel_vs_f64_twin=rel(loss, l64), grad_nrel_vs_f64_twin=nrel(
                    et.grad.cpu().numpy(), e64.grad.numpy()))
    labels = O.trainer_ms_labels(1, 12, 12)
    emb = U.embeddings(25, 32768, seed=5)
    for mining in (True, False):
        want = O.ms_loss(labels, emb, ms_mining=mining)
        et = torch.tensor(emb, device=dev)
        add('ms_loss T=1 P=N=12 mining=%s' % mining,
            loss_rel_vs_f32_oracle=rel(M.ms_loss(torch.tensor(labels, device=dev), et,
                                                 ms_mining=mining), want))

    # ---- tuple losses --------------------------------------------------------------------
    for name, quad in (('triplet_loss', False), ('lazy_triplet_loss', False),
                       ('evil_triplet_loss', False), ('quadruplet_loss', True),
                       ('lazy_quadruplet_loss', True), ('evil_quadruplet_loss', True)):
        fn = getattr(P, name, None) or getattr(M, name)
        p_, n_ = 12, (11 if quad else 12)
        shape = [1, p_, n_] + ([1] if quad else [])
        out = U.tuple_batch(1, p_, n_, 32768, quad=quad)
        flat = out.reshape(sum(shape), 32768)
        margins = (0.5, 0.2) if quad else (0.5,)
        want = getattr(O, name)(*O.split_tuples(flat, 1, shape), *margins)
        x64 = torch.tensor(flat, dtype=torch.float64, requires_grad=True)
        getattr(TT, name)(*torch.split(x64.reshape(1, sum(shape), -1), shape, dim=1),
                          *margins).backward()
        xt = torch.tensor(flat, device=dev, requires_grad=True)
        loss = fn(*torch.split(xt.reshape(1, sum(shape), -1), shape, dim=1), *margins)
        loss.backward()
        add(name, loss_rel_vs_f32_oracle=rel(loss, want),
            grad_nrel_vs_f64_twin=nrel(xt.grad.cpu().numpy(), x64.grad.numpy()))
    out = U.tuple_batch(1, 12, 12, 32768, seed=33)
    rng = np.random.default_rng(34)
    spd = rng.uniform(1, 200, (1, 12, 1)).astype(np.float32)
    snd = rng.uniform(300, 4000, (1, 12, 1)).astype(np.float32)
    want = O.logratio_loss(*O.split_tuples(out.reshape(-1, 32768), 1, [1, 12, 12]), spd, snd)
    xt = torch.tensor(out, device=dev)
    at, pt, nt = torch.split(xt, [1, 12, 12], dim=1)
    add('logratio_loss', loss_rel_vs_f32_oracle=rel(
        M.logratio_loss(at, pt, nt, torch.tensor(spd, device=dev), torch.tensor(snd, device=dev)),
        want))
    dd = rng.uniform(0.0, 225.0, (1, 12)).astype(np.float32)
    parts = O.split_tuples(out.reshape(-1, 32768), 1, [1, 12, 12])
    want = O.distance_triplet_loss(parts[0], parts[1], parts[2], 0.5, 0.5, dd, 225.0, 2.0)
    add('huber_distance_triplet', loss_rel_vs_f32_oracle=rel(
        M.distance_triplet_loss(at, pt, nt, 0.5, 0.5, torch.tensor(dd, device=dev), 225.0, 2.0),
        want))

    # ---- NetVLAD head --------------------------------------------------------------------
    w, c = U.vlad_params()
    x = U.feature_map(4, 1200, seed=5)
    g = np.random.default_rng(3).standard_normal((4, 32768)).astype(np.float32)
    for dt, label in ((torch.float32, 'f32 x (float32 MFMA)'), (torch.bfloat16, 'bf16 x (fused kernels, two bf16 planes)')):
        xin = torch.tensor(x).to(dt).float().numpy()        # what the kernel really sees
        want = NV.netvlad_fused(xin, w, c)
        xt = torch.tensor(xin, device=dev).to(dt).reshape(4, 1, 1200, 512).requires_grad_(True)
        wt = torch.tensor(w, device=dev).reshape(1, 1, 512, 64).requires_grad_(True)
        ct = torch.tensor(c, device=dev).reshape(1, 1, 1, 512, 64).requires_grad_(True)
        got = nets.netvlad(xt, wt, ct, True)
        got.backward(torch.tensor(g, device=dev))
        x64 = torch.tensor(xin, dtype=torch.float64, requires_grad=True)
        w64 = torch.tensor(w, dtype=torch.float64, requires_grad=True)
        c64 = torch.tensor(c, dtype=torch.float64, requires_grad=True)
        TT.netvlad(x64, w64, c64).backward(torch.tensor(g, dtype=torch.float64))
        add('netvlad 4x1200x512 %s' % label,
            descriptor_maxrel_vs_f32_oracle=maxrel(got.detach().cpu().numpy(), want),
            grad_x_nrel=nrel(xt.grad.float().cpu().numpy().reshape(4, 1200, 512), x64.grad.numpy()),
            grad_w_nrel=nrel(wt.grad.cpu().numpy().reshape(512, 64), w64.grad.numpy()),
            grad_c_nrel=nrel(ct.grad.cpu().numpy().reshape(512, 64), c64.grad.numpy()))

    # ---- retrieval -----------------------------------------------------------------------
    ref, qry = U.retrieval_sets(40000, 500, 256)
    want_d, want_i = TN.topn_kdtree(ref, qry, 25)
    rt, qt = torch.tensor(ref, device=dev), torch.tensor(qry, device=dev)
    for score in ('f32', 'bf16x3'):
        d, i = retrieval.topn_l2(rt, qt, 25, score=score)
        add('top-25 of 40000 x 500 x 256, score=%s' % score,
            index_lists_equal=bool(np.array_equal(i.cpu().numpy(), want_i)),
            dist_maxrel_vs_kdtree=float(np.abs(d.cpu().numpy() - want_d).max() / want_d.max()))

    # ---- against the fixtures made by executing / running the reference's own files ---------
    # (tests/golden/golden_ref_*.json; tests/tools/ref_exec/ made them in the build container)
    import tests.test_golden_ref as GR
    import tests.test_golden_ref_nets as GN
    import tests.test_golden_ref_topn as GT
    import tests.test_golden_ref_trainer as GTR
    from soft_contrastive_learning_amd.train import train as T
    worst, worst_name = 0.0, ''
    for c in GR.BY_KIND['wms']:
        emb, dist = GR._wms_inputs(c)
        got = M.wms_loss(torch.tensor(dist, device=dev), torch.tensor(emb, device=dev), 0.8, 15.0, **c['kw'])
        if rel(got, c['loss']) > worst:
            worst, worst_name = rel(got, c['loss']), c['name']
    add('EXECUTED reference losses.py: wms_loss, %d cases' % len(GR.BY_KIND['wms']),
        worst_loss_rel=worst, worst_case=worst_name)
    worst, worst_name = 0.0, ''
    for c in GTR.CASES:
        f, emb, dist = GTR._inputs(c)
        shape = T.tuple_shape_for(c['loss'], f.positives_per_tuple, f.negatives_per_tuple)
        got = T.compute_loss(f, shape, torch.from_numpy(emb).to(dev), T.batch_distances(f, dist, dev))
        if rel(got, c['loss_value']) > worst:
            worst, worst_name = rel(got, c['loss_value']), c['name']
    add("EXECUTED reference train.py build_model(): trainer's loss, %d configurations" % len(GTR.CASES),
        worst_loss_rel=worst, worst_case=worst_name)
    for dt, label in ((torch.float32, 'f32'), (torch.bfloat16, 'bf16')):
        worst = 0.0
        for c in GN.CASES:
            want = GN._expected(c)
            with torch.no_grad():
                got = GN._model(c, dt).to(dev).forward_vgg16(torch.from_numpy(GN._images(c)).to(dev))
            worst = max(worst, float(np.abs(got.float().cpu().numpy() - want).max() / np.abs(want).max()))
        add('EXECUTED reference nets.py: vgg16 maps, %d cases, %s maps' % (len(GN.CASES), label),
            worst_maxrel=worst)
    import tempfile
    ds, want = GT._dataset(), GT._want()
    with tempfile.TemporaryDirectory() as tmp:
        from soft_contrastive_learning_amd.evaluation import top_n
        from soft_contrastive_learning_amd.util import io as sio
        argv = ['--N', str(GT.C['N']), '--out_root', os.path.join(tmp, 'top_n')]
        for name, xy in (('ref', ds['ref_xy']), ('query', ds['query_xy'])):
            path = os.path.join(tmp, 'set_%s.csv' % name)
            sio.save_csv({'easting': [repr(float(v)) for v in xy[:, 0]],
                          'northing': [repr(float(v)) for v in xy[:, 1]]}, path)
            argv += ['--%s_csv' % name, path]
        for name in ('pca', 'ref', 'query'):
            path = os.path.join(tmp, 'set_%s.v1.pickle' % name)
            sio.save_pickle([row for row in ds[name + '_f']], path)
            argv += ['--%s_lv_pickle' % name, path]
        top_i, _, top_f, gt_i, _, _ = sio.load_pickle(top_n.main(argv)[0])
    add("RUN of the reference's evaluation/top-n.py (real scikit-learn): the script's pickle",
        index_lists_equal=bool(np.array_equal(np.asarray(top_i), want['top_i'])),
        dist_maxrel=float(np.abs(np.asarray(top_f) - want['top_f']).max() / want['top_f'].max()),
        ground_truth_equal=bool(np.array_equal(np.asarray(gt_i), want['gt_i'])))

    if args.json:
        with open(args.json, 'w') as f:
            json.dump(rows, f, indent=1)
        print('wrote', args.json)


if __name__ == '__main__':
    main()
