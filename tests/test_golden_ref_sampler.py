"""tests/golden/golden_ref_sampler_v1.json: what ``get_tuple`` of the reference's own
train/train.py (:433-582) RETURNED when it was run in the build container — plain Python on real
NumPy and scikit-learn, ``np.random.seed`` before each call (tests/tools/ref_exec/
make_golden_ref_sampler.py; the JSON is what travels).  SURVEY.md section 8(f) ranks 3-4: the tuple
sampler, its per-loss distance payloads, and the walk over the mining cache's sorted neighbours.

The package's TupleSampler given a RandomState of the same seed must return the SAME images in the
same order, bit-identical float64 payloads, and leave the stream at the same position.  All host
logic: CPU tests.  (The cache here is scikit-learn's KDTree like the reference's; the device
MiningCache is compared with that ordering in tests/test_gpu_callers.py.)
"""
import json
import os

import numpy as np
import pytest

from soft_contrastive_learning_amd.train.sampler import TupleSampler
from tests import util_data as U

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden', 'golden_ref_sampler_v1.json')
DOC = json.load(open(GOLDEN))
CASES = DOC['cases']


class _TreeCache:
    """CACHED_FEATURES / _INDICES / _TREE as train_one_epoch builds them (train/train.py:1032-1066)."""

    def __init__(self, c, n):
        from sklearn.neighbors import KDTree
        rng = np.random.default_rng(c['seed'] + 500)
        self.indices = np.concatenate([np.arange(c['cache_start'], c['cache_start'] + c['cache_size']) % n,
                                       np.asarray(c['anchors'])])
        self.features = rng.standard_normal((n, 16)).astype(np.float32)[self.indices]
        self.tree = KDTree(self.features)

    def sorted_neighbours(self, dataset_index, k=None):
        hits = np.where(self.indices == dataset_index)[0]
        if len(hits) == 0:
            return None
        order = self.tree.query(self.features[hits[0]].reshape(1, -1), k=k, return_distance=False,
                                sort_results=True)[0]
        return [int(self.indices[i]) for i in order]


def _sampler(c):
    xy, yaw = U.sampler_dataset()
    f = c['flags']
    shape = c['tuple_shape']
    cache = _TreeCache(c, len(yaw)) if c['hard'] else None
    return TupleSampler(xy, yaw, positives_per_tuple=shape[1], negatives_per_tuple=shape[2],
                        max_pos_radius=f.get('max_pos_radius', 15.0), min_neg_radius=f.get('min_neg_radius', 15.0),
                        hard_positives_per_tuple=f.get('hard_positives_per_tuple', 6),
                        hard_negatives_per_tuple=f.get('hard_negatives_per_tuple', 6),
                        mutually_exclusive_negs=True, distance_type=c['distance_type'], cache=cache,
                        mining_cache_size=f.get('mining_cache_size', 1000), rng=np.random.RandomState(c['seed']))


def test_fixture_file_is_the_generators():
    assert DOC['meta']['made_by'] == 'tests/tools/ref_exec/make_golden_ref_sampler.py'
    assert len(CASES) == 10
    assert {c['distance_type'] for c in CASES} == {'none', 'wms', 'anchor', 'pairwise', 'logratio'}


@pytest.mark.parametrize('c', CASES, ids=[c['name'] for c in CASES])
def test_sampler_returns_the_reference_runs_tuples(c):
    s = _sampler(c)
    distances, indices = s.get_tuple(c['anchors'], c['tuple_shape'], use_hard_negatives=c['hard'])
    assert [int(i) for i in indices] == c['indices']
    if c['indices']:
        per = sum(c['tuple_shape'])
        assert [int(i) for i in indices[-per:]] == c['last_tuple_indices']
        assert [int(indices[k * per]) for k in range(len(c['anchors']))] == c['anchors']     # anchor first (:502)
    assert len(distances) == len(c['distances'])
    for got, want in zip(distances, c['distances']):
        got = np.asarray(got, dtype=np.float64)
        assert got.shape == np.asarray(want).shape
        assert np.array_equal(got, np.asarray(want, dtype=np.float64))
    if c['indices']:
        # the same number of draws were consumed (a dropped batch stops at a different statement)
        assert float(s.rng.random_sample()) == c['next_random']
