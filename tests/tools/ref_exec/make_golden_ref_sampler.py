#!/usr/bin/env python
"""Runs ``get_tuple`` of the reference's own ``train/train.py`` (:433-582, where it lies under
/root/reference — nothing of it is copied) and freezes what it returns in
tests/golden/golden_ref_sampler_v1.json.  BUILD CONTAINER ONLY: needs /root/reference.

    python tests/tools/ref_exec/make_golden_ref_sampler.py

The function is plain Python on real NumPy and scikit-learn (KDTree.query_radius,
pairwise_distances, np.random.choice): TensorFlow's stand-in is involved only in IMPORTING the file
(tests/tools/ref_exec/ref_trainer.py).  Its flag globals are set per case; ``np.random.seed`` is
called before each call, so a sampler that consumes the same RandomState stream the same way must
return the same tuples.  The mining-cache cases set CACHED_FEATURES / _INDICES / _TREE the way
train_one_epoch does (:1032-1066: KDTree over the cached descriptors).
"""
import json
import os
import sys
import threading

import numpy as np

sys.dont_write_bytecode = True
HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
from ref_trainer import ROOT, install_names, load_trainer  # noqa: E402
from tests import util_data as U  # noqa: E402

OUT = os.path.join(ROOT, 'tests', 'golden', 'golden_ref_sampler_v1.json')


def cache_for(c, n):
    """Cached descriptors: rows of a seeded matrix for `cache_size` dataset indices + the anchors."""
    rng = np.random.default_rng(c['seed'] + 500)
    idx = np.concatenate([np.arange(c['cache_start'], c['cache_start'] + c['cache_size']) % n,
                          np.asarray(c['anchors'])])
    feats = rng.standard_normal((n, 16)).astype(np.float32)
    return feats[idx], idx


def main():
    from sklearn.neighbors import KDTree
    install_names()
    T = load_trainer()
    T.LOG = open(os.devnull, 'w')
    xy, yaw = U.sampler_dataset()
    n = len(yaw)
    meta = {'date': ['d'] * n, 'folder': ['1'] * n, 't': [str(i) for i in range(n)]}
    tree = KDTree(xy)
    cases = []

    def case(name, distance_type, tuple_shape, anchors, seed=42, hard=False, **flags):
        c = {'name': name, 'distance_type': distance_type, 'tuple_shape': tuple_shape, 'anchors': anchors,
             'seed': seed, 'hard': hard, 'flags': flags}
        g = dict(POSITIVES_PER_TUPLE=tuple_shape[1], NEGATIVES_PER_TUPLE=tuple_shape[2], MAX_POS_RADIUS=15.0,
                 MIN_NEG_RADIUS=15.0, HARD_POSITIVES_PER_TUPLE=6, HARD_NEGATIVES_PER_TUPLE=6,
                 MUTUALLY_EXCLUSIVE_NEGS=True, MINING_CACHE_SIZE=1000, ALPHA=0.8, BETA=15)
        g.update({k.upper(): v for k, v in flags.items()})
        for k, v in g.items():
            setattr(T, k, v)
        T.DISTANCE_TYPE = distance_type
        T.CACHED_FEATURE_LOCK = threading.Lock()
        if hard:
            c['cache_start'], c['cache_size'] = flags.get('cache_start', 40), g['MINING_CACHE_SIZE']
            feats, idx = cache_for(c, n)
            T.CACHED_FEATURES, T.CACHED_FEATURE_INDICES, T.CACHED_FEATURE_TREE = feats, idx, KDTree(feats)
        np.random.seed(seed)
        distances, image_info, last = T.get_tuple(anchors, tuple_shape, hard, meta, xy, yaw, tree)
        c['indices'] = [int(info[2]) for info in image_info]
        c['last_tuple_indices'] = [int(v) for v in last]
        c['distances'] = [np.asarray(d, dtype=np.float64).tolist() for d in distances]
        c['next_random'] = float(np.random.random_sample())       # where the stream stands afterwards
        cases.append(c)

    case('none_tu2_p12_n12', 'none', [1, 12, 12], [5, 200])
    case('wms_tu1_p12_n12', 'wms', [1, 12, 12], [77])
    case('anchor_tu3_p4_n6', 'anchor', [1, 4, 6], [0, 150, 301], seed=7)
    case('pairwise_tu2_p3_n2', 'pairwise', [1, 3, 2], [33, 250], seed=8)
    case('logratio_tu1_p12_n12', 'logratio', [1, 12, 12], [120], seed=9)
    case('quadruplet_none_tu2_p2_n3', 'none', [1, 2, 3, 1], [10, 90], seed=10)
    case('quadruplet_anchor_tu1_p12_n11', 'anchor', [1, 12, 11, 1], [300], seed=11)
    case('wide_radius_runs_out_of_negatives', 'none', [1, 2, 12], [15], seed=12, min_neg_radius=60.0)
    case('hard_wms_tu1_p12_n12', 'wms', [1, 12, 12], [64], seed=13, hard=True, mining_cache_size=200)
    case('hard_none_tu2_p4_n8_quadruplet', 'none', [1, 4, 8, 1], [70, 201], seed=14, hard=True,
         mining_cache_size=250, hard_positives_per_tuple=2, hard_negatives_per_tuple=3)
    meta_out = {'made_by': 'tests/tools/ref_exec/make_golden_ref_sampler.py',
                'what': "returns of get_tuple() of /root/reference/train/train.py (real NumPy / scikit-learn; "
                        "np.random.seed(seed) before each call) on tests/util_data.sampler_dataset",
                'numpy': np.__version__}
    import sklearn
    meta_out['sklearn'] = sklearn.__version__
    with open(OUT, 'w') as f:
        json.dump({'meta': meta_out, 'cases': cases}, f, indent=1)
    for c in cases:
        print('%-40s %3d images, %d payloads, first tuple %s' % (c['name'], len(c['indices']), len(c['distances']),
                                                               c['indices'][:sum(c['tuple_shape'])][:8]))


if __name__ == '__main__':
    main()
