"""float32 NumPy restatement of the hot-path losses.  TEST INFRASTRUCTURE ONLY
(see ``oracle/__init__.py``; parity unpinned by the reference).

Every function keeps the reference's tensor ranks, axes and broadcasting, because
NumPy's broadcasting rules and ``np.transpose`` default (full axis reversal) are the
same as TF1's — the shape quirks listed in SURVEY.md §8(a) A6/A8 therefore fall out
for free instead of being "fixed".

All arithmetic is float32 (TF placeholders on this path are tf.float32).
"""
import numpy as np

F32 = np.float32


def _f32(x):
    return np.asarray(x, dtype=F32)


def l2_normalize(x, axis, epsilon=1e-12):
    """tf.nn.l2_normalize: x * rsqrt(max(sum(x^2, axis), epsilon)).

    Used by model/losses.py:7,82 (axis=1) and model/nets.py:66 (axis=-1)."""
    x = _f32(x)
    ss = np.sum(x * x, axis=axis, keepdims=True, dtype=F32)
    return (x * (F32(1.0) / np.sqrt(np.maximum(ss, F32(epsilon))))).astype(F32)


def _ms_core(sim_mat, mask_pos, mask_neg, alpha, beta, lamb, eps, ms_mining,
             sumfunction='ms'):
    """Shared tail of wms_loss (model/losses.py:25-58) and ms_loss (:95-120).

    ``sim_mat`` is [B,B]; the masks may be [B,B] (ms_loss, rank-2 distances) or
    [1,B,B] (wms_loss fed by the trainer's rank-3 placeholder).  axis=1 is applied
    literally, so for rank-3 masks it reduces over the ROW index."""
    alpha, beta, lamb, eps = F32(alpha), F32(beta), F32(lamb), F32(eps)
    sim_mat = np.maximum(sim_mat, F32(0.0))
    pos_mat = (sim_mat * mask_pos).astype(F32)
    neg_mat = (sim_mat * mask_neg).astype(F32)
    if ms_mining:
        max_val = np.max(neg_mat, axis=1, keepdims=True)
        tmp_max_val = np.max(pos_mat, axis=1, keepdims=True)
        min_val = np.min(((sim_mat - tmp_max_val) * mask_pos).astype(F32),
                         axis=1, keepdims=True) + tmp_max_val
        mask_pos = np.where(pos_mat < max_val + eps, mask_pos, F32(0.0)).astype(F32)
        mask_neg = np.where(neg_mat > min_val - eps, mask_neg, F32(0.0)).astype(F32)
    if sumfunction == 'plain':
        pos_sel = np.where(mask_pos > 0.0, pos_mat, F32(0.0)).astype(F32)
        neg_sel = np.where(mask_neg > 0.0, neg_mat, F32(0.0)).astype(F32)
        pos_term = np.sum(pos_sel, axis=1, dtype=F32)
        neg_term = np.sum(neg_sel, axis=1, dtype=F32)
        return F32(np.mean(neg_term - pos_term, dtype=F32))
    with np.errstate(over='ignore'):
        pos_exp = np.exp(-alpha * (pos_mat - lamb)).astype(F32)
        neg_exp = np.exp(beta * (neg_mat - lamb)).astype(F32)
    pos_exp = np.where(mask_pos > 0.0, pos_exp, F32(0.0)).astype(F32)
    neg_exp = np.where(mask_neg > 0.0, neg_exp, F32(0.0)).astype(F32)
    pos_term = np.log(F32(1.0) + np.sum(pos_exp, axis=1, dtype=F32)) / alpha
    neg_term = np.log(F32(1.0) + np.sum(neg_exp, axis=1, dtype=F32)) / beta
    return F32(np.mean((pos_term + neg_term).astype(F32), dtype=F32))


def eigen_fast_tanh_f32(x):
    """tf.tanh on a float32 CPU tensor = Eigen's generic_fast_tanh_float (TF 1.10 bundles Eigen
    3.3.90; unsupported/../MathFunctionsImpl.h, recalled — not part of /root/reference): clamp to
    [-9, 9], odd 13th-degree / even 6th-degree rational approximation, every operation rounded
    to float32 (the stock x86 wheels are built without FMA).  Restated op for op because
    ``1 - tanh(d / d_beta)`` decides pair membership through ``mask_pos > 0``: near saturation
    (d / d_beta in [8.2, 10]) libm, ocml and this approximation disagree in the last bit, this
    one is not even monotone there (8.5 -> 1.0, 8.7 -> 0.99999994, >= 9 -> 1.0), and the wms
    loss at B=192 / 200 m moves by 1.8 % with the choice."""
    x = np.asarray(x, dtype=F32)
    x = np.maximum(F32(-9.0), np.minimum(F32(9.0), x))
    a1, a3, a5, a7, a9, a11, a13 = (F32(v) for v in (
        4.89352455891786e-03, 6.37261928875436e-04, 1.48572235717979e-05, 5.12229709037114e-08,
        -8.60467152213735e-11, 2.00018790482477e-13, -2.76076847742355e-16))
    b0, b2, b4, b6 = (F32(v) for v in (
        4.89352518554385e-03, 2.26843463243900e-03, 1.18534705686654e-04, 1.19825839466702e-06))
    x2 = x * x
    p = x2 * a13 + a11
    p = x2 * p + a9
    p = x2 * p + a7
    p = x2 * p + a5
    p = x2 * p + a3
    p = x2 * p + a1
    p = x * p
    q = x2 * b6 + b4
    q = x2 * q + b2
    q = x2 * q + b0
    return (p / q).astype(F32)


def wms_masks(distances, d_alpha, d_beta, wfunction='exp'):
    """Geographic soft masks of wms_loss before the ``- eye`` (model/losses.py:11-19)."""
    d = _f32(distances)
    d_alpha, d_beta = F32(d_alpha), F32(d_beta)
    if wfunction == 'lin':
        mask_pos = np.where(d < d_beta, F32(1.0) - d / d_beta, F32(0.0))
        mask_neg = np.where(d < d_beta, d / d_beta, F32(1.0))
    elif wfunction == 'tanh':
        t = eigen_fast_tanh_f32(d / d_beta)
        mask_pos = F32(1.0) - t
        mask_neg = t
    else:  # 'exp' and anything else, like the reference's bare else
        with np.errstate(over='ignore'):
            mask_pos = F32(1.0) / (F32(1.0) + np.exp(d_alpha * (d - d_beta)))
            mask_neg = F32(1.0) / (F32(1.0) + np.exp(d_alpha * (d_beta - d)))
    return mask_pos.astype(F32), mask_neg.astype(F32)


def wms_loss(distances, embeddings, d_alpha, d_beta, alpha=2.0, beta=50.0, lamb=1.0,
             eps=0.1, ms_mining=True, wfunction='exp', sumfunction='ms'):
    """model/losses.py:5-60.  ``distances`` [B,B] or [1,B,B]; ``embeddings`` [B,E]."""
    emb = l2_normalize(embeddings, axis=1)
    batch = emb.shape[0]
    mask_pos, mask_neg = wms_masks(distances, d_alpha, d_beta, wfunction)
    mask_pos = mask_pos - np.eye(batch, dtype=F32)
    sim_mat = (emb @ emb.T).astype(F32)
    return _ms_core(sim_mat, mask_pos, mask_neg, alpha, beta, lamb, eps, ms_mining,
                    sumfunction)


def ms_loss(labels, embeddings, alpha=2.0, beta=50.0, lamb=1.0, eps=0.1,
            ms_mining=True):
    """model/losses.py:76-122 (ms_det :139-185 is the same with ms_mining=False)."""
    emb = l2_normalize(embeddings, axis=1)
    labels = np.asarray(labels).reshape(-1, 1)
    batch = emb.shape[0]
    adjacency = labels == labels.T
    mask_pos = adjacency.astype(F32) - np.eye(batch, dtype=F32)
    mask_neg = (~adjacency).astype(F32)
    sim_mat = (emb @ emb.T).astype(F32)
    return _ms_core(sim_mat, mask_pos, mask_neg, alpha, beta, lamb, eps, ms_mining)


def ms_det(labels, embeddings, alpha=2.0, beta=50.0, lamb=1.0, eps=0.1,
           ms_mining=False):
    """model/losses.py:139-185: byte-for-byte ms_loss with a different default."""
    return ms_loss(labels, embeddings, alpha, beta, lamb, eps, ms_mining)


def trainer_ms_labels(tuples_per_batch, positives_per_tuple, negatives_per_tuple):
    """Label vector the trainer builds for --loss ms_loss (train/train.py:822-826)."""
    one = np.concatenate((np.zeros(1 + positives_per_tuple),
                          np.arange(negatives_per_tuple) + 1))
    out = one
    for t in range(1, tuples_per_batch):
        out = np.concatenate((out, one + t * (negatives_per_tuple + 1)))
    return out


def logratio_loss(a_feature, pos_features, neg_features, squared_pos_dists,
                  squared_neg_dists):
    """model/losses.py:125-135, rank-for-rank.

    a [T,1,E], pos [T,P,E], neg [T,N,E], squared_*_dists [T,P,1] / [T,N,1]
    (train/train.py:687-691).  The literal broadcasting is only valid for T == 1
    and P == N."""
    a, p, n = _f32(a_feature), _f32(pos_features), _f32(neg_features)
    spd, snd = _f32(squared_pos_dists), _f32(squared_neg_dists)
    pos_res = np.sum((a - p) ** 2, axis=2, dtype=F32)          # [T,P]
    neg_res = np.sum((a - n) ** 2, axis=2, dtype=F32)          # [T,N]
    feat_ratio = np.log(pos_res / np.transpose(neg_res))       # [T,P]/[N,T]
    dist_ratio = np.log(spd / np.transpose(snd))               # full axis reversal
    sq = ((feat_ratio - dist_ratio) ** 2).astype(F32)
    per = np.mean(np.mean(sq, axis=1, dtype=F32), axis=1, dtype=F32)
    return F32(np.mean(per, axis=0, dtype=F32))


# ---- tuple losses: pointnetvlad_cls (external, restated) and the in-tree twins ----

def _sq_dists_to(anchor, vecs):
    """sum(squared_difference(vecs, tile(anchor)), 2): [T,1,E] x [T,R,E] -> [T,R]."""
    a, v = _f32(anchor), _f32(vecs)
    return np.sum((v - a) ** 2, axis=2, dtype=F32)


def best_pos_distance(query, pos_vecs):
    """pointnetvlad_cls.best_pos_distance: min over positives."""
    return np.min(_sq_dists_to(query, pos_vecs), axis=1)


def worst_pos_distance(query, pos_vecs):
    """model/losses.py:217-222: max over positives."""
    return np.max(_sq_dists_to(query, pos_vecs), axis=1)


def _hinge(margin, ref_pos, anchor, neg_vecs):
    d = _sq_dists_to(anchor, neg_vecs)                              # [T,N]
    return np.maximum(F32(margin) + (ref_pos.reshape(-1, 1) - d), F32(0.0)).astype(F32)


def triplet_loss(q_vec, pos_vecs, neg_vecs, margin):
    """pointnetvlad_cls.triplet_loss (call site train/train.py:700-701)."""
    h = _hinge(margin, best_pos_distance(q_vec, pos_vecs), q_vec, neg_vecs)
    return F32(np.mean(np.sum(h, axis=1, dtype=F32), dtype=F32))


def lazy_triplet_loss(q_vec, pos_vecs, neg_vecs, margin):
    """pointnetvlad_cls.lazy_triplet_loss (train/train.py:702-703)."""
    h = _hinge(margin, best_pos_distance(q_vec, pos_vecs), q_vec, neg_vecs)
    return F32(np.mean(np.max(h, axis=1), dtype=F32))


def evil_triplet_loss(q_vec, pos_vecs, neg_vecs, margin):
    """model/losses.py:63-73."""
    h = _hinge(margin, worst_pos_distance(q_vec, pos_vecs), q_vec, neg_vecs)
    return F32(np.mean(np.sum(h, axis=1, dtype=F32), dtype=F32))


def quadruplet_loss(q_vec, pos_vecs, neg_vecs, other_neg, m1, m2):
    """pointnetvlad_cls.quadruplet_loss (train/train.py:706-708)."""
    first = triplet_loss(q_vec, pos_vecs, neg_vecs, m1)
    h2 = _hinge(m2, best_pos_distance(q_vec, pos_vecs), other_neg, neg_vecs)
    return F32(first + F32(np.mean(np.sum(h2, axis=1, dtype=F32), dtype=F32)))


def lazy_quadruplet_loss(q_vec, pos_vecs, neg_vecs, other_neg, m1, m2):
    """pointnetvlad_cls.lazy_quadruplet_loss (train/train.py:709-712)."""
    first = lazy_triplet_loss(q_vec, pos_vecs, neg_vecs, m1)
    h2 = _hinge(m2, best_pos_distance(q_vec, pos_vecs), other_neg, neg_vecs)
    return F32(first + F32(np.mean(np.max(h2, axis=1), dtype=F32)))


def evil_quadruplet_loss(q_vec, pos_vecs, neg_vecs, other_neg, m1, m2):
    """model/losses.py:197-214."""
    first = evil_triplet_loss(q_vec, pos_vecs, neg_vecs, m1)
    h2 = _hinge(m2, worst_pos_distance(q_vec, pos_vecs), other_neg, neg_vecs)
    return F32(first + F32(np.mean(np.sum(h2, axis=1, dtype=F32), dtype=F32)))


# ---- distance-term losses (model/losses.py:225-307, 664-690) ----

def _scale_distances(a_feature, pos_feature, squared_d_dists, d_max_squared, f_max_squared):
    """model/losses.py:678-690: both squared distances divided by their maxima, [T,P]."""
    sf = _sq_dists_to(a_feature, pos_feature)
    sd = _f32(squared_d_dists).reshape(sf.shape)
    return (sd / F32(d_max_squared)).astype(F32), (sf / F32(f_max_squared)).astype(F32)


def _huber(labels, predictions, delta=1.0):
    """tf.losses.huber_loss element-wise part (TF 1.10 losses_impl.py)."""
    err = (predictions - labels).astype(F32)
    a = np.abs(err)
    quad = np.minimum(a, F32(delta))
    lin = a - quad
    return (F32(0.5) * quad * quad + F32(delta) * lin).astype(F32)


def distance_loss(a_feature, pos_feature, squared_d_dists, d_max_squared, f_max_squared):
    """model/losses.py:225-230."""
    sd, sf = _scale_distances(a_feature, pos_feature, squared_d_dists, d_max_squared, f_max_squared)
    return F32(np.mean(np.mean((sf - sd) ** 2, axis=1, dtype=F32), dtype=F32))


def huber_distance_loss(a_feature, pos_feature, squared_d_dists, d_max_squared, f_max_squared):
    """model/losses.py:233-236: tf.losses.huber_loss(labels=scaled_d, predictions=scaled_f),
    delta 1, SUM_BY_NONZERO_WEIGHTS = mean over all elements."""
    sd, sf = _scale_distances(a_feature, pos_feature, squared_d_dists, d_max_squared, f_max_squared)
    return F32(np.mean(_huber(sd, sf), dtype=F32))


def distance_triplet_loss(a_feature, pos_features, neg_features, margin, lam, squared_d_dists,
                          d_max_squared, f_max_squared, triplet_loss_name='triplet_loss',
                          distance_loss_name='huber_distance_loss'):
    """model/losses.py:239-264."""
    trip = {'triplet_loss': triplet_loss, 'lazy_triplet_loss': lazy_triplet_loss}[triplet_loss_name]
    dist = huber_distance_loss if 'huber' in distance_loss_name else distance_loss
    return F32(trip(a_feature, pos_features, neg_features, margin) +
               F32(lam) * dist(a_feature, pos_features, squared_d_dists, d_max_squared,
                               f_max_squared))


def distance_quadruplet_loss(a_feature, pos_features, neg_features, other_neg, m1, m2, lam,
                             squared_d_dists, d_max_squared, f_max_squared,
                             triplet_loss_name='triplet_loss',
                             distance_loss_name='huber_distance_loss'):
    """model/losses.py:267-307: the second term always takes the max over negatives."""
    trip = distance_triplet_loss(a_feature, pos_features, neg_features, m1, lam, squared_d_dists,
                                 d_max_squared, f_max_squared, triplet_loss_name,
                                 distance_loss_name)
    sd, sf = _scale_distances(a_feature, pos_features, squared_d_dists, d_max_squared,
                              f_max_squared)
    per_pos = _huber(sf, sd) if 'huber' in distance_loss_name else ((sf - sd) ** 2).astype(F32)
    best_pos = np.min(per_pos, axis=1).reshape(-1, 1)                     # :664-675
    on = (_sq_dists_to(other_neg, neg_features) / F32(f_max_squared)).astype(F32)
    second = np.mean(np.max(np.maximum(F32(m2) + (best_pos - on), F32(0.0)), axis=1), dtype=F32)
    return F32(trip + second)


def pairwise_squared_distances(features):
    """model/losses.py:656-661: [T,S,E] -> [T,S,S] = r_i - 2 F F^T + r_j."""
    f = _f32(features)
    r = np.einsum('aij,aij->ai', f, f).astype(F32).reshape(f.shape[0], -1, 1)
    prod = np.einsum('aij,ajk->aik', f, np.transpose(f, (0, 2, 1))).astype(F32)
    return (r - F32(2.0) * prod + np.transpose(r, (0, 2, 1))).astype(F32)


def pairwise_distance_loss(anchor, positives, pairwise_squared_d_dists, d_max_squared,
                           f_max_squared, distance_loss_name='distance_loss'):
    """model/losses.py:627-646: all pairs among [anchor | positives] of every tuple; the huber
    variant is tf.losses.huber_loss(labels=scaled_f, predictions=scaled_d) (argument order as
    written there), then three nested means."""
    feats = np.concatenate([_f32(anchor), _f32(positives)], axis=1)
    sf = (pairwise_squared_distances(feats) / F32(f_max_squared)).astype(F32)
    sd = (_f32(pairwise_squared_d_dists) / F32(d_max_squared)).astype(F32)
    if 'huber' in distance_loss_name:
        sq = _huber(sf, sd)
    else:
        sq = ((sf - sd) ** 2).astype(F32)
    m1 = np.mean(sq, axis=2, dtype=F32)
    m2 = np.mean(m1, axis=1, dtype=F32)
    return F32(np.mean(m2, axis=0, dtype=F32))


def split_tuples(output, tuples_per_batch, tuple_shape):
    """Trainer glue (train/train.py:654): [T*S,E] -> list of [T,n_i,E]."""
    out = _f32(output).reshape(tuples_per_batch, sum(tuple_shape), -1)
    idx = np.cumsum(tuple_shape)[:-1]
    return np.split(out, idx, axis=1)
