"""torch-autograd twin of the oracle, for GRADIENT checks.  TEST INFRASTRUCTURE ONLY
(see ``oracle/__init__.py``; parity unpinned by the reference).

The reference never writes a backward pass: TF1 autodiff differentiates the graph
(train/train.py:877-878).  This twin re-expresses the same forward ops with torch
CPU tensors (float64 by default) so autograd gives the gradients the HIP backward
kernels are compared against.  Tie / mask conventions follow TF1:
  * tf.maximum(x, 0): gradient flows to x where x >= 0;
  * tf.where(cond, a, 0): no gradient through cond (mining masks are constants);
  * reduce_min / reduce_max: gradient split evenly among ties (torch.amin/amax do
    the same).
"""
import torch


def _t(x, dtype):
    if isinstance(x, torch.Tensor):
        return x.to(dtype)
    return torch.as_tensor(x, dtype=dtype)


def l2_normalize(x, axis, epsilon=1e-12):
    ss = (x * x).sum(dim=axis, keepdim=True)
    return x * torch.rsqrt(torch.clamp_min(ss, epsilon))


def _relu_tf(x):
    # forward max(x, 0); backward passes where x >= 0 (tf.maximum convention)
    return torch.where(x >= 0, x, torch.zeros_like(x))


def _ms_core(sim_mat, mask_pos, mask_neg, alpha, beta, lamb, eps, ms_mining,
             sumfunction='ms'):
    sim_mat = _relu_tf(sim_mat)
    pos_mat = sim_mat * mask_pos
    neg_mat = sim_mat * mask_neg
    if ms_mining:
        with torch.no_grad():
            max_val = neg_mat.amax(dim=1, keepdim=True)
            tmp = pos_mat.amax(dim=1, keepdim=True)
            min_val = ((sim_mat - tmp) * mask_pos).amin(dim=1, keepdim=True) + tmp
            keep_p = pos_mat < max_val + eps
            keep_n = neg_mat > min_val - eps
        mask_pos = torch.where(keep_p, mask_pos, torch.zeros_like(mask_pos))
        mask_neg = torch.where(keep_n, mask_neg, torch.zeros_like(mask_neg))
    sel_p = mask_pos > 0
    sel_n = mask_neg > 0
    if sumfunction == 'plain':
        pos_term = torch.where(sel_p, pos_mat, torch.zeros_like(pos_mat)).sum(dim=1)
        neg_term = torch.where(sel_n, neg_mat, torch.zeros_like(neg_mat)).sum(dim=1)
        return (neg_term - pos_term).mean()
    pos_exp = torch.where(sel_p, torch.exp(-alpha * (pos_mat - lamb)),
                          torch.zeros_like(pos_mat))
    neg_exp = torch.where(sel_n, torch.exp(beta * (neg_mat - lamb)),
                          torch.zeros_like(neg_mat))
    pos_term = torch.log(1.0 + pos_exp.sum(dim=1)) / alpha
    neg_term = torch.log(1.0 + neg_exp.sum(dim=1)) / beta
    return (pos_term + neg_term).mean()


def wms_loss(distances, embeddings, d_alpha, d_beta, alpha=2.0, beta=50.0, lamb=1.0,
             eps=0.1, ms_mining=True, wfunction='exp', sumfunction='ms',
             dtype=torch.float64):
    # The geographic masks are data, not differentiated, and their float32 evaluation is
    # part of the reference semantics: exp() overflows to +inf for far pairs and makes the
    # mask EXACTLY 0 / 1, which decides `mask > 0`.  So the twin takes them
    # from the float32 oracle and only widens the differentiable part.
    from .losses_np import wms_masks
    import numpy as _np
    mp32, mn32 = wms_masks(_np.asarray(distances, dtype=_np.float32), d_alpha, d_beta, wfunction)
    mask_pos = torch.as_tensor(mp32).to(dtype)
    mask_neg = torch.as_tensor(mn32).to(dtype)
    emb = l2_normalize(_t(embeddings, dtype), 1)
    batch = emb.shape[0]
    mask_pos = mask_pos - torch.eye(batch, dtype=dtype)
    sim = emb @ emb.T
    return _ms_core(sim, mask_pos, mask_neg, alpha, beta, lamb, eps, ms_mining,
                    sumfunction)


def ms_loss(labels, embeddings, alpha=2.0, beta=50.0, lamb=1.0, eps=0.1,
            ms_mining=True, dtype=torch.float64):
    emb = l2_normalize(_t(embeddings, dtype), 1)
    lab = torch.as_tensor(labels).reshape(-1, 1)
    batch = emb.shape[0]
    adj = lab == lab.T
    mask_pos = adj.to(dtype) - torch.eye(batch, dtype=dtype)
    mask_neg = (~adj).to(dtype)
    sim = emb @ emb.T
    return _ms_core(sim, mask_pos, mask_neg, alpha, beta, lamb, eps, ms_mining)


def logratio_loss(a_feature, pos_features, neg_features, squared_pos_dists,
                  squared_neg_dists, dtype=torch.float64):
    a, p, n = _t(a_feature, dtype), _t(pos_features, dtype), _t(neg_features, dtype)
    spd, snd = _t(squared_pos_dists, dtype), _t(squared_neg_dists, dtype)
    pr = ((a - p) ** 2).sum(dim=2)
    nr = ((a - n) ** 2).sum(dim=2)
    fr = torch.log(pr / nr.permute(*reversed(range(nr.dim()))))
    dr = torch.log(spd / snd.permute(*reversed(range(snd.dim()))))
    sq = (fr - dr) ** 2
    return sq.mean(dim=1).mean(dim=1).mean(dim=0)


def _sq_dists_to(anchor, vecs):
    return ((vecs - anchor) ** 2).sum(dim=2)


def _tuple_loss(q, pos, neg, other, m1, m2, pos_red, neg_red, dtype):
    q, pos, neg = _t(q, dtype), _t(pos, dtype), _t(neg, dtype)
    dp = _sq_dists_to(q, pos)
    ref = dp.amin(dim=1) if pos_red == 'min' else dp.amax(dim=1)

    def part(anchor, margin):
        h = margin + (ref.reshape(-1, 1) - _sq_dists_to(anchor, neg))
        h = _relu_tf(h)
        per = h.sum(dim=1) if neg_red == 'sum' else h.amax(dim=1)
        return per.mean()

    loss = part(q, m1)
    if other is not None:
        loss = loss + part(_t(other, dtype), m2)
    return loss


def triplet_loss(q, pos, neg, margin, dtype=torch.float64):
    return _tuple_loss(q, pos, neg, None, margin, None, 'min', 'sum', dtype)


def lazy_triplet_loss(q, pos, neg, margin, dtype=torch.float64):
    return _tuple_loss(q, pos, neg, None, margin, None, 'min', 'max', dtype)


def evil_triplet_loss(q, pos, neg, margin, dtype=torch.float64):
    return _tuple_loss(q, pos, neg, None, margin, None, 'max', 'sum', dtype)


def quadruplet_loss(q, pos, neg, other, m1, m2, dtype=torch.float64):
    return _tuple_loss(q, pos, neg, other, m1, m2, 'min', 'sum', dtype)


def lazy_quadruplet_loss(q, pos, neg, other, m1, m2, dtype=torch.float64):
    return _tuple_loss(q, pos, neg, other, m1, m2, 'min', 'max', dtype)


def evil_quadruplet_loss(q, pos, neg, other, m1, m2, dtype=torch.float64):
    return _tuple_loss(q, pos, neg, other, m1, m2, 'max', 'sum', dtype)


def _huber_t(labels, predictions, delta=1.0):
    err = predictions - labels
    a = err.abs()
    quad = torch.minimum(a, torch.full_like(a, delta))
    return 0.5 * quad * quad + delta * (a - quad)


def distance_tuple_loss(q, pos, neg, other, m1, m2, lam, squared_d_dists, d_max_squared,
                        f_max_squared, lazy=False, huber=True, dtype=torch.float64):
    """distance_triplet_loss (other=None) / distance_quadruplet_loss, model/losses.py:239-307."""
    q, pos, neg = _t(q, dtype), _t(pos, dtype), _t(neg, dtype)
    sf = _sq_dists_to(q, pos) / f_max_squared
    sd = _t(squared_d_dists, dtype).reshape(sf.shape) / d_max_squared
    loss = _tuple_loss(q, pos, neg, None, m1, None, 'min', 'max' if lazy else 'sum', dtype)
    term = _huber_t(sd, sf) if huber else (sf - sd) ** 2
    loss = loss + lam * term.mean()
    if other is not None:
        per_pos = _huber_t(sf, sd) if huber else (sf - sd) ** 2
        best = per_pos.amin(dim=1).reshape(-1, 1)
        on = _sq_dists_to(_t(other, dtype), neg) / f_max_squared
        loss = loss + _relu_tf(m2 + (best - on)).amax(dim=1).mean()
    return loss


def pairwise_distance_loss(anchor, positives, pairwise_squared_d_dists, d_max_squared,
                           f_max_squared, huber=False, dtype=torch.float64):
    """model/losses.py:627-646 (see oracle.losses_np.pairwise_distance_loss)."""
    f = torch.cat([_t(anchor, dtype), _t(positives, dtype)], dim=1)
    r = (f * f).sum(dim=2, keepdim=True)
    sf = (r - 2.0 * f @ f.transpose(1, 2) + r.transpose(1, 2)) / f_max_squared
    sd = _t(pairwise_squared_d_dists, dtype) / d_max_squared
    sq = _huber_t(sf, sd) if huber else (sf - sd) ** 2
    return sq.mean(dim=2).mean(dim=1).mean(dim=0)


def netvlad(x, assign_w, centers, pre_l2=True, dtype=torch.float64):
    """x [B,N,D], assign_w [D,K], centers [D,K] -> [B, D*K] (see oracle.netvlad_np)."""
    x, w, c = _t(x, dtype), _t(assign_w, dtype), _t(centers, dtype)
    if pre_l2:
        x = l2_normalize(x, -1)
    a = torch.softmax(x @ w, dim=-1)
    v = torch.einsum('bnd,bnk->bdk', x, a) + c[None] * a.sum(dim=1)[:, None, :]
    v = v / torch.sqrt((v * v).sum(dim=1, keepdim=True) + 1e-12)     # over D per k
    v = v.reshape(v.shape[0], -1)
    return v / torch.sqrt((v * v).sum(dim=1, keepdim=True) + 1e-12)
