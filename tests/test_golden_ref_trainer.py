"""tests/golden/golden_ref_trainer_v1.json: what ``build_model()`` of the reference's own
train/train.py (:585-879) put in its ``ops`` when its text was EXECUTED (eager placeholders on
tests/tools/ref_exec/tf_shim.py; build container only — the JSON is what travels): the loss its
own glue computes from a fed descriptor batch — reshape to [T,S,E], ``tf.split`` by tuple_shape,
the ms label constant, the per-loss distance placeholders and their splits, which flags reach
which loss call — the tuple_shape it returns and the learning rate it hands the optimiser
(SURVEY.md section 8 rows A12 and A13).  Cases marked ``uses_recalled_pointnetvlad`` reach the
absent third-party losses, which were the recalled ones of oracle/losses_np.py in that run: they
pin the glue around those losses, not the losses.

CPU part: the oracle's split + losses and the trainer's host logic (distance type, tuple shape,
learning-rate schedule) against those numbers.
GPU part: the package's trainer dispatch ``train.compute_loss`` (HIP losses) against them.
Tolerance 1e-4 relative (BASELINE.json north_star); 1e-5 for the float32 oracle.
"""
import argparse
import json
import os

import numpy as np
import pytest
import torch

from oracle import losses_np as O
from soft_contrastive_learning_amd.train import train as T
from tests import util_data as U

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden', 'golden_ref_trainer_v1.json')
DOC = json.load(open(GOLDEN))
CASES = DOC['cases']
F32 = np.float32


def _flags(c):
    g = {k.lower(): v for k, v in DOC['meta']['defaults'].items()}
    g.update(c['flags'])
    g['loss'] = c['loss']
    return argparse.Namespace(**g)


def _inputs(c):
    """The descriptor batch and distance payload the generator fed (same seeded builders)."""
    f = _flags(c)
    t, p, n = f.tuples_per_batch, f.positives_per_tuple, f.negatives_per_tuple
    s = 1 + p + n
    if 'triplet' in c['loss'] or 'quadruplet' in c['loss']:
        emb = U.tuple_batch(t, p, n, c['e'], seed=c['seed'], scale=0.05).reshape(t * s, c['e'])
    else:
        emb = U.embeddings(t * s, c['e'], seed=c['seed'], mix=0.9)
    rng = np.random.default_rng(c['seed'] + 1000)
    nn = n - 1 if 'quadruplet' in c['loss'] else n
    kind, dist = c['distance_type'], None
    if kind == 'wms':
        dist = np.stack([U.positions_distances(s, side=c.get('side', 60.0), seed=c['seed'] + i) for i in range(t)])
    elif kind == 'logratio':
        dist = np.concatenate([rng.uniform(1.0, 15.0, (t, p)) ** 2, rng.uniform(15.0, 80.0, (t, nn)) ** 2], 1).astype(F32)
    elif kind == 'anchor':
        dist = (rng.uniform(0.0, 15.0, (t, p)) ** 2).astype(F32)
    return f, emb, dist


def _ids(cases):
    return [c['name'] for c in cases]


def test_fixture_file_is_the_generators():
    assert DOC['meta']['made_by'] == 'tests/tools/ref_exec/make_golden_ref_trainer.py'
    assert len(CASES) == 20
    for op in ('placeholder', 'split', 'reshape', 'train.minimize'):
        assert DOC['meta']['shim_ops_called'].get(op, 0) > 0, op


@pytest.mark.parametrize('c', CASES, ids=_ids(CASES))
def test_host_logic_of_the_trainer(c):
    f = _flags(c)
    assert T.distance_type(c['loss']) == c['distance_type']                       # train/train.py:1378-1391
    assert T.tuple_shape_for(c['loss'], f.positives_per_tuple, f.negatives_per_tuple) == c['tuple_shape']
    assert c['negatives_per_tuple_after'] == c['tuple_shape'][2]                  # :589-592 moves the global
    assert c['pn_loss'] is False
    lr = T.get_learning_rate(c['epoch'], f)
    assert abs(lr - c['learning_rate']) <= 1e-6 * c['learning_rate']              # float32 graph vs Python floats
    assert c['optimizer'] == {'adam': 'Adam', 'momentum': 'Momentum'}[f.optimizer]
    s = sum(c['tuple_shape'])
    assert c['outputs_shapes'] == [[f.tuples_per_batch, k, c['e']] for k in c['tuple_shape']]
    if c['distance_type'] == 'wms':
        assert c['distances_shape'] == [f.tuples_per_batch, s, s]                  # rank 3 (:684-686)


def _oracle_loss(c):
    f, emb, dist = _inputs(c)
    t, shape, loss = f.tuples_per_batch, c['tuple_shape'], c['loss']
    outs = O.split_tuples(emb, t, shape)
    if loss == 'wms':
        return O.wms_loss(dist, emb, f.alpha, f.beta, wfunction=f.wfunction, sumfunction=f.sumfunction)
    if loss == 'ms_loss':
        labels = O.trainer_ms_labels(t, f.positives_per_tuple, f.negatives_per_tuple)
        return O.ms_loss(labels, emb, ms_mining=f.msmining)
    if loss == 'logratio':
        p = f.positives_per_tuple
        d = dist.reshape(t, -1, 1)
        return O.logratio_loss(outs[0], outs[1], outs[2], d[:, :p], d[:, p:])
    if 'distance' in loss:
        trip = 'lazy_triplet_loss' if 'lazy' in loss else 'triplet_loss'
        dl = 'huber_distance_loss' if 'huber' in loss else 'distance_loss'
        d_max = float(f.max_pos_radius) ** 2
        if 'quadruplet' in loss:
            return O.distance_quadruplet_loss(outs[0], outs[1], outs[2], outs[3], f.margin_1, f.margin_2, f.lam,
                                              dist, d_max, 2.0, trip, dl)
        return O.distance_triplet_loss(outs[0], outs[1], outs[2], f.margin_1, f.lam, dist, d_max, 2.0, trip, dl)
    fn = getattr(O, loss + '_loss')
    if 'quadruplet' in loss:
        return fn(outs[0], outs[1], outs[2], outs[3], f.margin_1, f.margin_2)
    return fn(outs[0], outs[1], outs[2], f.margin_1)


@pytest.mark.parametrize('c', CASES, ids=_ids(CASES))
def test_oracle_with_its_own_split_gives_the_trainers_loss(c):
    got = float(_oracle_loss(c))
    assert abs(got - c['loss_value']) <= 1e-5 * max(abs(c['loss_value']), 1e-3)


# ------------------------------------------------------------------ GPU: the package's dispatch
@pytest.mark.gpu
@pytest.mark.parametrize('c', CASES, ids=_ids(CASES))
def test_compute_loss_of_the_package_gives_the_reference_trainers_loss(c):
    dev = torch.device('cuda:0')
    f, emb, dist = _inputs(c)
    shape = T.tuple_shape_for(c['loss'], f.positives_per_tuple, f.negatives_per_tuple)
    payload = T.batch_distances(f, dist, dev)            # labels for ms_loss, the fed tensor otherwise
    out = torch.from_numpy(emb).to(dev).requires_grad_(True)
    loss = T.compute_loss(f, shape, out, payload)
    assert abs(loss.item() - c['loss_value']) <= 1e-4 * max(abs(c['loss_value']), 1e-3)
    loss.backward()                                       # the dispatch is differentiable end to end
    assert out.grad is not None and torch.isfinite(out.grad).all()
