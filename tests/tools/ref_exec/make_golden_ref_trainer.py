#!/usr/bin/env python
"""Executes ``build_model()`` of the reference's own ``train/train.py`` (:585-879, where it lies
under /root/reference — nothing of it is copied) on tests/tools/ref_exec/tf_shim.py and freezes
what its ``ops`` hold in tests/golden/golden_ref_trainer_v1.json: the loss the trainer's OWN glue
computes from a batch of descriptors (reshape to [T,S,E], ``tf.split`` by tuple_shape, the ms label
constant, the distance placeholders and their splits, which arguments each loss call gets), the
tuple_shape it returns, and the learning rate it hands the optimiser.
BUILD CONTAINER ONLY: needs /root/reference.

    python tests/tools/ref_exec/make_golden_ref_trainer.py

How the function is made to run eagerly: every ``tf.placeholder`` takes the next fed value of its
dtype and static shape (tf_shim.FEEDS), so each statement of build_model works on real arrays.
The embedder is NOT run: the name ``vgg16Netvlad`` in the trainer's namespace is pointed at a
function returning the fed descriptor batch (the network has its own fixture,
golden_ref_nets_v1.json).  ``DISTANCE_TYPE`` and ``PN_LOSS`` are set by executing the reference's
own two ``if`` chains for them (cut out of its ``__main__`` block with ``ast`` at run time).

Names stood in for so that the file imports (none on the executed path except the last):
``cv2`` and ``netvlad_tf`` (empty), ``learnlarge`` -> /root/reference, helper.{debugging, location, srv_root},
``learnlarge.model.incremental_skl`` / ``.mac`` (missing from the reference),
and ``pointnetvlad.pointnetvlad_cls`` (third party, absent): ITS four losses are the RECALLED ones of
oracle/losses_np.py — cases that reach them say ``uses_recalled_pointnetvlad`` and pin the glue
around them only.
"""
import ast
import json
import os
import sys

import numpy as np

sys.dont_write_bytecode = True          # nothing may be written under /root/reference (no __pycache__)
HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(HERE)))
sys.path.insert(0, ROOT)
sys.path.insert(0, HERE)

import tf_shim  # noqa: E402
from ref_trainer import TRAIN, install_names, load_trainer  # noqa: E402
from tests import util_data as U  # noqa: E402

OUT = os.path.join(ROOT, 'tests', 'golden', 'golden_ref_trainer_v1.json')
F32 = np.float32


def main_block_chain(first_string):
    """The ``if ... elif ... else`` statement of the reference's ``__main__`` block whose first test
    is ``'<first_string>' in LOSS``, compiled for exec in the trainer's namespace."""
    tree = ast.parse(open(TRAIN).read())
    main = [n for n in tree.body if isinstance(n, ast.If) and 'name__' in ast.dump(n.test)][-1]
    for node in main.body:
        if (isinstance(node, ast.If) and isinstance(node.test, ast.Compare)
                and isinstance(node.test.left, ast.Constant) and node.test.left.value == first_string):
            return compile(ast.Module([node], []), TRAIN, 'exec')
    raise LookupError(first_string)


DEFAULTS = dict(POSITIVES_PER_TUPLE=12, NEGATIVES_PER_TUPLE=12, TUPLES_PER_BATCH=1, MARGIN_1=0.1, MARGIN_2=0.2,
                LAM=0.5, ALPHA=0.8, BETA=15, WFUNCTION='exp', SUMFUNCTION='ms', MSMINING=False,
                MAX_POS_RADIUS=15.0, BASE_LR=5e-6, MINIMAL_LR=5e-12, LR_DOWN_FACTOR=0.5, LR_DOWN_FREQUENCY=1.0,
                MOMENTUM=0.9, OPTIMIZER='adam', OUT_DIM=512, LOSS_DIM=512, REDUCTION='none', VLAD_CORES=64, L=3)


def distances_for(c, t, tuple_shape, p, n):
    """What the trainer feeds per distance type (train/train.py:665-691): seeded, metres / squared metres."""
    rng = np.random.default_rng(c['seed'] + 1000)
    kind = c['distance_type']
    s = sum(tuple_shape)
    if kind == 'wms':
        return np.stack([U.positions_distances(s, side=c.get('side', 60.0), seed=c['seed'] + i) for i in range(t)])
    if kind == 'logratio':
        return np.concatenate([rng.uniform(1.0, 15.0, (t, p)) ** 2, rng.uniform(15.0, 80.0, (t, n)) ** 2], 1).astype(F32)
    if kind == 'anchor':
        return (rng.uniform(0.0, 15.0, (t, p)) ** 2).astype(F32)
    return None


def run_case(T, chains, c):
    g = dict(DEFAULTS)
    g.update({k.upper(): v for k, v in c['flags'].items()})
    g['LOSS'] = c['loss']
    for k, v in g.items():
        setattr(T, k, v)
    for code in chains:
        exec(code, T.__dict__)
    c['distance_type'], c['pn_loss'] = T.DISTANCE_TYPE, builtins_bool(T.PN_LOSS)
    t, p, n = g['TUPLES_PER_BATCH'], g['POSITIVES_PER_TUPLE'], g['NEGATIVES_PER_TUPLE']
    s = 1 + p + n                                        # images per tuple as the sampler delivers them
    emb = U.embeddings(t * s, c['e'], seed=c['seed'], mix=0.9)
    if 'triplet' in c['loss'] or 'quadruplet' in c['loss']:
        emb = U.tuple_batch(t, p, n, c['e'], seed=c['seed'], scale=0.05).reshape(t * s, c['e'])
    quad = 'quadruplet' in c['loss']
    tuple_shape = [1, p, n - 1, 1] if quad else [1, p, n]
    dist = distances_for(c, t, tuple_shape, p, n - 1 if quad else n)
    tf_shim.FEEDS[:] = [np.zeros(1, np.bool_), np.zeros((t * s, 8, 8, 3), F32), np.asarray(c['epoch'], F32)]
    if dist is not None:
        tf_shim.FEEDS.append(dist)
    del tf_shim.train.made[:]
    T.vgg16Netvlad = lambda images: tf_shim._t(emb)
    ops, got_shape = T.build_model()
    assert not tf_shim.FEEDS, 'fed values left over: %r' % [np.asarray(v).shape for v in tf_shim.FEEDS]
    c.update({'tuple_shape': [int(v) for v in got_shape], 'negatives_per_tuple_after': int(T.NEGATIVES_PER_TUPLE),
              'loss_value': float(np.asarray(ops['loss'])),
              'learning_rate': float(np.asarray(tf_shim.train.made[0].learning_rate)),
              'optimizer': type(tf_shim.train.made[0]).__name__.strip('_'),
              'outputs_shapes': [list(o.shape) for o in ops['outputs']],
              'distances_shape': None if dist is None else list(dist.shape)})
    return c


def builtins_bool(v):
    return True if v else False


def main():
    install_names()
    T = load_trainer()
    chains = [main_block_chain('eigenvalue'), main_block_chain('pairwise')]
    cases = []

    def case(name, loss, e=512, seed=99, epoch=0.0, recalled=False, **flags):
        c = {'name': name, 'loss': loss, 'e': e, 'seed': seed, 'epoch': epoch, 'flags': flags,
             'uses_recalled_pointnetvlad': recalled}
        cases.append(run_case(T, chains, c))

    case('wms_tu1_p12_n12_reference_command', 'wms', e=32768)
    case('wms_tu1_p5_n6_lin_plain', 'wms', positives_per_tuple=5, negatives_per_tuple=6, wfunction='lin',
         sumfunction='plain', alpha=1.0, beta=20)
    case('wms_tu1_p12_n12_tanh_epoch3', 'wms', wfunction='tanh', epoch=3.0)
    case('ms_tu1_p2_n4', 'ms_loss', positives_per_tuple=2, negatives_per_tuple=4)
    case('ms_tu2_p2_n4_mining', 'ms_loss', positives_per_tuple=2, negatives_per_tuple=4, tuples_per_batch=2,
         msmining=True)
    case('ms_tu3_p3_n3_momentum_epoch40', 'ms_loss', positives_per_tuple=3, negatives_per_tuple=3, tuples_per_batch=3,
         optimizer='momentum', epoch=40.0)
    case('logratio_tu1_p12_n12', 'logratio', seed=21)
    case('logratio_tu1_p4_n4_slow_decay', 'logratio', seed=8, positives_per_tuple=4, negatives_per_tuple=4, epoch=7.0,
         lr_down_factor=0.9, lr_down_frequency=3.0, base_lr=1e-4)
    case('evil_triplet_tu2_p2_n3', 'evil_triplet', positives_per_tuple=2, negatives_per_tuple=3, tuples_per_batch=2,
         margin_1=0.5)
    case('evil_quadruplet_tu2_p2_n4', 'evil_quadruplet', positives_per_tuple=2, negatives_per_tuple=4,
         tuples_per_batch=2, margin_1=0.5, margin_2=0.2)
    for loss in ('triplet', 'lazy_triplet'):
        case(loss + '_tu2_p2_n3', loss, recalled=True, positives_per_tuple=2, negatives_per_tuple=3,
             tuples_per_batch=2, margin_1=0.5)
    for loss in ('quadruplet', 'lazy_quadruplet'):
        case(loss + '_tu2_p2_n4', loss, recalled=True, positives_per_tuple=2, negatives_per_tuple=4,
             tuples_per_batch=2, margin_1=0.5, margin_2=0.2)
    for loss in ('distance_triplet', 'huber_distance_triplet', 'distance_lazy_triplet', 'huber_distance_lazy_triplet'):
        case(loss + '_tu2_p3_n3', loss, recalled=True, seed=41, positives_per_tuple=3, negatives_per_tuple=3,
             tuples_per_batch=2, margin_1=0.5, lam=0.5)
    for loss in ('distance_quadruplet', 'huber_distance_lazy_quadruplet'):
        case(loss + '_tu2_p3_n4', loss, recalled=True, seed=43, positives_per_tuple=3, negatives_per_tuple=4,
             tuples_per_batch=2, margin_1=0.5, margin_2=0.2, lam=0.25, max_pos_radius=10.0)

    meta = {'made_by': 'tests/tools/ref_exec/make_golden_ref_trainer.py',
            'what': "ops of /root/reference/train/train.py's build_model() executed on tests/tools/ref_exec/"
                    'tf_shim.py (eager placeholders; the embedder replaced by the fed descriptors; '
                    'pointnetvlad_cls = the recalled losses of oracle/losses_np.py where a case says so)',
            'numpy': np.__version__, 'defaults': DEFAULTS,
            'shim_ops_called': dict(sorted(tf_shim.CALLS.items()))}
    with open(OUT, 'w') as f:
        json.dump({'meta': meta, 'cases': cases}, f, indent=1)
    for c in cases:
        print('%-44s %-9s %-16s loss %.6f lr %.3e' % (c['name'], c['distance_type'], c['tuple_shape'],
                                                      c['loss_value'], c['learning_rate']))


if __name__ == '__main__':
    main()
