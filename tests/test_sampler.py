"""Tuple sampler + input pipeline (train/sampler.py) against the rules of the reference's
get_tuple (train/train.py:433-582), checked on a synthetic route."""
import math

import numpy as np
import pytest

from soft_contrastive_learning_amd.train import sampler as S


def route(n=400, seed=0):
    rng = np.random.default_rng(seed)
    # out and back along a street: the return leg is close in space but opposite in heading
    half = n // 2
    x = np.concatenate([np.arange(half) * 2.0, np.arange(half)[::-1] * 2.0])
    y = np.concatenate([np.zeros(half), np.full(half, 3.0)])
    xy = np.stack([x, y], 1) + rng.normal(0, 0.05, (n, 2))
    yaw = np.concatenate([np.zeros(half), np.full(half, math.pi)]) + rng.normal(0, 0.02, n)
    return xy, yaw


def dist(xy, i, j):
    return float(np.linalg.norm(xy[i] - xy[j]))


@pytest.mark.parametrize('exclusive', [True, False])
def test_tuples_obey_the_sampling_rules(exclusive):
    xy, yaw = route()
    s = S.TupleSampler(xy, yaw, 12, 12, max_pos_radius=15, min_neg_radius=15,
                       mutually_exclusive_negs=exclusive, rng=np.random.RandomState(1))
    d, idx = s.get_tuple([50, 120], [1, 12, 12])
    assert d == [[], []] and idx.shape == (50,)
    for t, anchor in enumerate([50, 120]):
        tup = idx[t * 25:(t + 1) * 25]
        assert tup[0] == anchor
        pos, neg = tup[1:13], tup[13:]
        for p in pos:
            assert p != anchor and dist(xy, anchor, p) <= 15
            assert abs(yaw[anchor] - yaw[p]) % (2 * math.pi) < math.pi / 6      # same heading
        for k, a in enumerate(neg):
            assert dist(xy, anchor, a) > 15
            if exclusive:
                for b in neg[:k]:
                    assert dist(xy, a, b) > 15
        assert len(set(neg.tolist())) == 12


def test_quadruplet_other_negative_is_far_from_everything():
    xy, yaw = route()
    for exclusive in (True, False):
        s = S.TupleSampler(xy, yaw, 4, 5, mutually_exclusive_negs=exclusive,
                           rng=np.random.RandomState(2))
        _, idx = s.get_tuple([30], [1, 4, 5, 1])
        assert len(idx) == 11
        other = idx[-1]
        for a in [idx[0]] + list(idx[5:10]):
            assert dist(xy, other, a) > 15


def test_draws_match_a_literal_restatement_with_the_same_random_state():
    xy, yaw = route(120, seed=3)
    from sklearn.neighbors import KDTree
    tree = KDTree(xy)
    rs = np.random.RandomState(7)
    index, p_want, n_want, r = 40, 3, 4, 15.0
    # the reference's statements, literally (train/train.py:455-498)
    dirty = np.setdiff1d(tree.query_radius(xy[index, :].reshape(1, -1), r=r)[0], [index])
    potential = [p for p in dirty if abs(yaw[index] - yaw[p]) % (2 * math.pi) < (math.pi / 6.0)]
    positives = rs.choice(potential, p_want).tolist()
    excluded = set(tree.query_radius(xy[index, :].reshape(1, -1), r=r)[0])
    negatives = []
    while len(excluded) < len(yaw):
        remaining = [i for i in np.arange(len(yaw)) if i not in excluded]
        nxt = rs.choice(remaining)
        negatives.append(nxt)
        excluded.update(tree.query_radius(xy[nxt, :].reshape(1, -1), r=r)[0])
        if len(negatives) >= n_want:
            break
    want = np.concatenate(([index], positives, negatives)).astype(int)
    s = S.TupleSampler(xy, yaw, p_want, n_want, rng=np.random.RandomState(7))
    _, got = s.get_tuple([index], [1, p_want, n_want])
    assert got.tolist() == want.tolist()


class FakeCache:
    def __init__(self, order):
        self.indices = np.asarray(order)
        self.order = order

    def sorted_neighbours(self, index, k=None):
        return list(self.order) if index == 50 else None


def test_hard_negatives_from_the_front_hard_positives_from_the_back():
    xy, yaw = route()
    # cached neighbours of anchor 50 by descriptor distance: near ones first
    order = [50, 300, 51, 305, 200, 52, 53, 120, 49, 48]
    s = S.TupleSampler(xy, yaw, 4, 6, hard_positives_per_tuple=2, hard_negatives_per_tuple=3,
                       cache=FakeCache(order), rng=np.random.RandomState(4))
    _, idx = s.get_tuple([50], [1, 4, 6], use_hard_negatives=True)
    pos, neg = idx[1:5].tolist(), idx[5:].tolist()
    assert pos[-2:] == [48, 49]               # walked from the back, potential positives only
    # front of the list outside the exclusion zones: 300, then 305 is within 15 m of 300
    # (mutually exclusive) -> skipped, then 200, then 120
    assert neg[-3:] == [300, 200, 120]
    # an anchor that is not cached gets no hard candidates
    _, idx2 = s.get_tuple([60], [1, 4, 6], use_hard_negatives=True)
    assert len(idx2) == 11


def test_distance_payloads():
    xy, yaw = route()
    shape = [1, 3, 4]
    for dt in ('anchor', 'pairwise', 'wms', 'logratio'):
        s = S.TupleSampler(xy, yaw, 3, 4, distance_type=dt, rng=np.random.RandomState(5))
        d, idx = s.get_tuple([70], shape)
        a, pos, neg = idx[0], idx[1:4], idx[4:]
        if dt == 'anchor':
            np.testing.assert_allclose(d[0], [dist(xy, a, p) ** 2 for p in pos])
        elif dt == 'pairwise':
            assert d[0].shape == (4, 4)
            np.testing.assert_allclose(d[0][0, 1:], [dist(xy, a, p) ** 2 for p in pos])
        elif dt == 'wms':
            assert d[0].shape == (8, 8)
            np.testing.assert_allclose(d[0][0], [dist(xy, a, j) for j in idx])     # metres
            np.testing.assert_allclose(d[0], d[0].T)
        else:
            np.testing.assert_allclose(d[0], [dist(xy, a, j) ** 2 for j in idx[1:]])
    with pytest.raises(ValueError):
        S.TupleSampler(xy, yaw, distance_type='swrd')


def test_batches_that_cannot_be_completed_are_dropped():
    xy, yaw = route(40)
    s = S.TupleSampler(xy, yaw, 2, 12, rng=np.random.RandomState(6))      # 80 m street: too short
    assert s.get_tuple([5], [1, 2, 12]) == ([], [])
    lonely_xy = np.array([[0.0, 0.0], [100.0, 0.0], [200.0, 0.0]])
    s = S.TupleSampler(lonely_xy, np.zeros(3), 1, 1)
    assert s.get_tuple([0], [1, 1, 1]) == ([], [])                        # no positives at all
    assert s.get_tuple([0], [1, 1]) == ([], [])                           # invalid tuple shape


def test_input_pipeline_feeds_batches_and_tracks_used_images():
    xy, yaw = route()
    s = S.TupleSampler(xy, yaw, 3, 4, distance_type='wms', rng=np.random.RandomState(8))
    loaded = []

    def load(indices):
        loaded.append(list(indices))
        return np.zeros((len(indices), 6, 8, 3), np.float32) + np.asarray(indices)[:, None, None, None]

    pipe = S.InputPipeline(s, load, [1, 3, 4], use_hard_negatives=False, depth=2)
    try:
        for a in (10, 20, 30):
            pipe.put([a])
        got = [pipe.get(timeout=20) for _ in range(3)]
        pipe.join()
    finally:
        pipe.close()
    for (d, img, idx), a in zip(got, (10, 20, 30)):
        assert idx[0] == a and img.shape == (8, 6, 8, 3) and d[0].shape == (8, 8)
        assert img[:, 0, 0, 0].tolist() == idx.tolist()
    assert pipe.used_images == set(int(i) for g in got for i in g[2])
    assert pipe.dropped == 0


def test_input_pipeline_hands_worker_failures_to_the_consumer():
    """A failing image loader (missing / corrupt file on the dataset route) must surface in the
    training thread's get() — with emit_dropped the loop counts on one get() per put(), so a
    worker that died silently left it blocked for ever."""
    import pytest
    xy, yaw = route()
    s = S.TupleSampler(xy, yaw, 3, 4, distance_type='wms', rng=np.random.RandomState(8))
    calls = []

    def load(indices):
        calls.append(len(indices))
        if len(calls) == 2:
            raise FileNotFoundError('frame 000123.png')
        return np.zeros((len(indices), 4, 4, 3), np.float32)

    pipe = S.InputPipeline(s, load, [1, 3, 4], use_hard_negatives=False, depth=2, emit_dropped=True)
    try:
        for a in (10, 20, 30):
            pipe.put([a])
        assert pipe.get(timeout=20) is not None
        with pytest.raises(RuntimeError, match='000123') as info:
            pipe.get(timeout=20)
        assert isinstance(info.value.__cause__, FileNotFoundError)
        assert pipe.get(timeout=20) is not None            # the worker lives on: one get per put
        pipe.join()
    finally:
        pipe.close()
    # with every worker gone an un-timed get() fails instead of blocking for ever
    with pytest.raises(RuntimeError, match='worker threads have exited'):
        pipe.get()
