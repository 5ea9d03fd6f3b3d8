"""Counterpart of the third-party ``pointnetvlad_cls`` losses the reference trainer
imports (train/train.py:25, call sites :700-712; model/losses.py:256,261 via getattr).

Upstream: github.com/mikacuy/pointnetvlad ``pointnetvlad_cls.py`` (not vendored or
version-pinned by the reference, README.md:11); restated in oracle/losses_np.py.
Same names and argument order; the kernels are csrc/tuple_loss.hip.
"""
from . import _lib as L
from .model.losses import _anchor_sqdists, _tuple

__all__ = ['best_pos_distance', 'triplet_loss', 'lazy_triplet_loss', 'quadruplet_loss',
           'lazy_quadruplet_loss']


def best_pos_distance(query, pos_vecs):
    """min over positives of the squared distance to the query, [T] (forward only)."""
    return _anchor_sqdists(query, pos_vecs).min(dim=1).values


def triplet_loss(q_vec, pos_vecs, neg_vecs, margin):
    return _tuple(L.TUPLE_TRIPLET, q_vec, pos_vecs, neg_vecs, None, margin, 0.0)


def lazy_triplet_loss(q_vec, pos_vecs, neg_vecs, margin):
    return _tuple(L.TUPLE_LAZY_TRIPLET, q_vec, pos_vecs, neg_vecs, None, margin, 0.0)


def quadruplet_loss(q_vec, pos_vecs, neg_vecs, other_neg, m1, m2):
    return _tuple(L.TUPLE_QUADRUPLET, q_vec, pos_vecs, neg_vecs, other_neg, m1, m2)


def lazy_quadruplet_loss(q_vec, pos_vecs, neg_vecs, other_neg, m1, m2):
    return _tuple(L.TUPLE_LAZY_QUADRUPLET, q_vec, pos_vecs, neg_vecs, other_neg, m1, m2)
