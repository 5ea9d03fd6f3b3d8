"""Drop-in counterpart of the reference's ``model/losses.py`` hot functions.

Same function names, positional order, keyword names and defaults as the reference
(`/root/reference/model/losses.py`); tensors are ``torch.Tensor`` on a HIP device
instead of ``tf.Tensor`` and each loss is a ``torch.autograd.Function`` whose forward
and backward are the hand-written gfx950 kernels behind ``include/scl_hip.h``.

  wms_loss               model/losses.py:5-60      (the paper's soft contrastive loss)
  ms_loss / ms_det       model/losses.py:76-122 / :139-185
  logratio_loss          model/losses.py:125-135
  evil_triplet_loss      model/losses.py:63-73
  evil_quadruplet_loss   model/losses.py:197-214
  worst_pos_distance     model/losses.py:217-222
  distance_loss / huber_distance_loss         model/losses.py:225-236
  distance_triplet_loss / distance_quadruplet_loss   model/losses.py:239-307
  _pairwise_squared_distances  model/losses.py:656-661
  pairwise_distance_loss       model/losses.py:627-646
The pointnetvlad_cls losses the trainer imports beside them (train/train.py:25)
live in ``soft_contrastive_learning_amd.pointnetvlad_cls``.
"""
import torch

from .. import _lib as L

__all__ = ['wms_loss', 'ms_loss', 'ms_det', 'logratio_loss', 'evil_triplet_loss',
           'evil_quadruplet_loss', 'worst_pos_distance', '_pairwise_squared_distances',
           'distance_loss', 'huber_distance_loss', 'distance_triplet_loss',
           'distance_quadruplet_loss', 'pairwise_distance_loss']


def _as_f32(t):
    if t.dtype != torch.float32:
        t = t.float()
    return t


class _GramLoss(torch.autograd.Function):
    """Pairwise-similarity loss on the full batch; see csrc/gram_loss.hip."""

    @staticmethod
    def forward(ctx, emb, distances, labels, cfg):
        lib = L.load()
        L.require_device(emb, distances, labels)
        emb = _as_f32(emb)
        if emb.dim() != 2:
            raise ValueError("embeddings must be rank 2 [B, E], got %s" % (tuple(emb.shape),))
        if emb.stride(1) != 1:
            emb = emb.contiguous()
        b, e = emb.shape
        need_grad = ctx.needs_input_grad[0]
        loss = torch.empty((), dtype=torch.float32, device=emb.device)
        coef = torch.empty((b, b), dtype=torch.float32, device=emb.device) if need_grad else None
        nbytes = lib.scl_gram_loss_workspace_bytes(b, e)
        if nbytes == 0:
            raise ValueError("unsupported batch / embedding size B=%d E=%d" % (b, e))
        ws = L.workspace(nbytes, emb.device)
        # With the stream's zeroed sync block the forward is ONE launch: B <= 32 the finish runs in the
        # Gram kernel's last workgroup; 32 < B <= 208 a persistent kernel with grid barriers between
        # its phases (csrc/gram_loss.hip, persist_tail).  Bit-identical to the multi-launch forms.
        sync = L.sync_words(emb.device) if b <= 256 else None
        L.check(lib.scl_gram_loss_fwd_s(
            L.ptr(emb), emb.stride(0), b, e, cfg['mask_kind'], L.ptr(distances),
            cfg['dist_rank3'], cfg['d_alpha'], cfg['d_beta'], L.ptr(labels), cfg['alpha'],
            cfg['beta'], cfg['lamb'], cfg['eps'], int(bool(cfg['ms_mining'])), cfg['sum_kind'],
            L.ptr(loss), L.ptr(coef), L.ptr(ws), ws.numel(), L.ptr(sync), L.stream_of(emb)))
        ctx.rows = cfg.get('rows')
        if need_grad:
            ctx.save_for_backward(emb, coef)
        return loss

    @staticmethod
    def backward(ctx, grad_loss):
        lib = L.load()
        emb, coef = ctx.saved_tensors
        b, e = emb.shape
        row_begin, row_count = ctx.rows if ctx.rows is not None else (0, b)
        g = _as_f32(grad_loss).contiguous()
        if ctx.rows is None:
            grad = torch.empty((b, e), dtype=torch.float32, device=emb.device)
            target = grad
        else:
            # a data-parallel rank only needs its own rows; the rest stay zero
            grad = torch.zeros((b, e), dtype=torch.float32, device=emb.device)
            target = grad[row_begin:row_begin + row_count]
        # (with a workspace the many-row case runs on bf16 planes: scl_gram_loss_bwd_w)
        ws = L.workspace(lib.scl_gram_loss_bwd_workspace_bytes(b, row_count), emb.device)
        L.check(lib.scl_gram_loss_bwd_w(L.ptr(emb), emb.stride(0), b, e, L.ptr(coef), L.ptr(g),
                                        row_begin, row_count, L.ptr(target), grad.stride(0),
                                        L.ptr(ws), ws.numel(), L.stream_of(emb)))
        return grad, None, None, None


_WFUNCTIONS = {'lin': L.MASK_WMS_LIN, 'tanh': L.MASK_WMS_TANH}
_SUMFUNCTIONS = {'ms': L.SUM_MS, 'plain': L.SUM_PLAIN}


def wms_loss(distances, embeddings, d_alpha, d_beta, alpha=2.0, beta=50.0, lamb=1.0, eps=0.1,
             ms_mining=True, wfunction='exp', sumfunction='ms', _rows=None):
    """Soft (weighted) multi-similarity loss, model/losses.py:5-60.

    ``distances``: geographic distances, [B,B] or the trainer's rank-3 [1,B,B]
    placeholder (train/train.py:684-686) — with rank 3 the reference's ``axis=1``
    reductions run over the row index, which is reproduced.  ``embeddings``: [B,E].
    """
    if sumfunction not in _SUMFUNCTIONS:
        # the reference leaves `loss` unbound for any other value (UnboundLocalError)
        raise ValueError("sumfunction must be 'ms' or 'plain', got %r" % (sumfunction,))
    L.require_device(distances, embeddings)
    b = embeddings.shape[0]
    d = _as_f32(distances)
    if d.dim() == 3:
        if d.shape[0] != 1:
            # [T,S,S] * [T*S,T*S] does not broadcast for T > 1
            raise ValueError("rank-3 distances need tuples_per_batch == 1, got %s"
                             % (tuple(d.shape),))
        rank3, d2 = 1, d[0]
    elif d.dim() == 2:
        rank3, d2 = 0, d
    else:
        raise ValueError("distances must be rank 2 or 3, got %s" % (tuple(d.shape),))
    if tuple(d2.shape) != (b, b):
        raise ValueError("distances %s do not match batch %d" % (tuple(d.shape), b))
    cfg = dict(mask_kind=_WFUNCTIONS.get(wfunction, L.MASK_WMS_EXP), dist_rank3=rank3,
               d_alpha=float(d_alpha), d_beta=float(d_beta), alpha=float(alpha), beta=float(beta),
               lamb=float(lamb), eps=float(eps), ms_mining=ms_mining,
               sum_kind=_SUMFUNCTIONS[sumfunction], rows=_rows)
    return _GramLoss.apply(embeddings, d2.contiguous(), None, cfg)


def _label_ids(labels, device):
    """Any numeric label vector -> int64 ids with identical equality structure."""
    lab = torch.as_tensor(labels)
    lab = lab.reshape(-1)
    if lab.is_floating_point():
        _, lab = torch.unique(lab, return_inverse=True)
    return lab.to(device=device, dtype=torch.int64).contiguous()


def ms_loss(labels, embeddings, alpha=2.0, beta=50.0, lamb=1.0, eps=0.1, ms_mining=True,
            _rows=None):
    """Multi-similarity loss, model/losses.py:76-122."""
    L.require_device(embeddings)
    lab = _label_ids(labels, embeddings.device)
    if lab.numel() != embeddings.shape[0]:
        raise ValueError("labels (%d) do not match batch %d" % (lab.numel(), embeddings.shape[0]))
    cfg = dict(mask_kind=L.MASK_LABELS, dist_rank3=0, d_alpha=0.0, d_beta=0.0, alpha=float(alpha),
               beta=float(beta), lamb=float(lamb), eps=float(eps), ms_mining=ms_mining,
               sum_kind=L.SUM_MS, rows=_rows)
    return _GramLoss.apply(embeddings, None, lab, cfg)


def ms_det(labels, embeddings, alpha=2.0, beta=50.0, lamb=1.0, eps=0.1, ms_mining=False):
    """model/losses.py:139-185: ms_loss with mining off by default."""
    return ms_loss(labels, embeddings, alpha, beta, lamb, eps, ms_mining)


# --------------------------------------------------------------------- tuple losses
def _rows_view(t, name):
    """[T,R,E] view with unit feature stride and rows E apart -> (tensor, tuple stride)."""
    if t.dim() != 3:
        raise ValueError("%s must be rank 3 [T,R,E], got %s" % (name, tuple(t.shape)))
    t = _as_f32(t)
    e = t.shape[2]
    if t.stride(2) != 1 or (t.shape[1] > 1 and t.stride(1) != e):
        t = t.contiguous()
    ts = t.stride(0) if t.shape[0] > 1 else t.shape[1] * e
    return t, ts


class _TupleLoss(torch.autograd.Function):
    """Triplet / quadruplet family; see csrc/tuple_loss.hip."""

    @staticmethod
    def forward(ctx, q, pos, neg, other, kind, m1, m2):
        lib = L.load()
        L.require_device(q, pos, neg, other)
        q, q_ts = _rows_view(q, 'q_vec')
        pos, p_ts = _rows_view(pos, 'pos_vecs')
        neg, n_ts = _rows_view(neg, 'neg_vecs')
        o_ts = 0
        if other is not None:
            other, o_ts = _rows_view(other, 'other_neg')
        t, p, e = pos.shape
        n = neg.shape[1]
        if q.shape != (t, 1, e) or neg.shape[0] != t or neg.shape[2] != e or (
                other is not None and other.shape != (t, 1, e)):
            raise ValueError("inconsistent tuple shapes q%s pos%s neg%s" % (
                tuple(q.shape), tuple(pos.shape), tuple(neg.shape)))
        width = p + 2 * n
        loss = torch.empty((), dtype=torch.float32, device=q.device)
        sqd = torch.empty((t, width), dtype=torch.float32, device=q.device)
        coef = torch.empty((t, width), dtype=torch.float32, device=q.device)
        L.check(lib.scl_tuple_loss_fwd(kind, L.ptr(q), q_ts, L.ptr(pos), p_ts, L.ptr(neg), n_ts,
                                       L.ptr(other), o_ts, t, p, n, e, float(m1), float(m2),
                                       L.ptr(loss), L.ptr(sqd), L.ptr(coef), L.stream_of(q)))
        ctx.save_for_backward(q, pos, neg, other, coef)
        ctx.strides = (q_ts, p_ts, n_ts, o_ts)
        return loss

    @staticmethod
    def backward(ctx, grad_loss):
        lib = L.load()
        q, pos, neg, other, coef = ctx.saved_tensors
        q_ts, p_ts, n_ts, o_ts = ctx.strides
        t, p, e = pos.shape
        n = neg.shape[1]
        g = _as_f32(grad_loss).contiguous()
        # gradients use the input strides: allocate with the same geometry
        gq = torch.empty_strided(q.shape, q.stride(), dtype=torch.float32, device=q.device)
        gp = torch.empty_strided(pos.shape, pos.stride(), dtype=torch.float32, device=q.device)
        gn = torch.empty_strided(neg.shape, neg.stride(), dtype=torch.float32, device=q.device)
        go = None
        if other is not None:
            go = torch.empty_strided(other.shape, other.stride(), dtype=torch.float32,
                                     device=q.device)
        L.check(lib.scl_tuple_loss_bwd(L.ptr(q), q_ts, L.ptr(pos), p_ts, L.ptr(neg), n_ts,
                                       L.ptr(other), o_ts, t, p, n, e, L.ptr(coef), L.ptr(g),
                                       L.ptr(gq), L.ptr(gp), L.ptr(gn), L.ptr(go),
                                       L.stream_of(q)))
        return gq, gp, gn, go, None, None, None


def _tuple(kind, q, pos, neg, other, m1, m2):
    # views of one [T,S,E] tensor keep the parent's strides; make each dense so the
    # gradient buffers (empty_strided) are compact
    return _TupleLoss.apply(q.contiguous(), pos.contiguous(), neg.contiguous(),
                            None if other is None else other.contiguous(), kind, m1, m2)


class _DistanceTupleLoss(torch.autograd.Function):
    """(lazy) triplet + lam * distance term (+ the quadruplet second term); the backward is
    the tuple backward kernel on the summed d loss / d sqd coefficients."""

    @staticmethod
    def forward(ctx, q, pos, neg, other, d_dists, kind, huber, m1, m2, lam, d_max, f_max):
        lib = L.load()
        L.require_device(q, pos, neg, other, d_dists)
        q, q_ts = _rows_view(q, 'a_feature')
        pos, p_ts = _rows_view(pos, 'pos_features')
        neg, n_ts = _rows_view(neg, 'neg_features')
        o_ts = 0
        if other is not None:
            other, o_ts = _rows_view(other, 'other_neg')
        t, p, e = pos.shape
        n = neg.shape[1]
        if q.shape != (t, 1, e) or neg.shape[0] != t or neg.shape[2] != e or (
                other is not None and other.shape != (t, 1, e)):
            raise ValueError("inconsistent tuple shapes a%s pos%s neg%s" % (
                tuple(q.shape), tuple(pos.shape), tuple(neg.shape)))
        dd = _as_f32(d_dists).reshape(-1).contiguous()
        if dd.numel() != t * p:
            raise ValueError("squared_d_dists must hold T*P = %d values, got %d"
                             % (t * p, dd.numel()))
        width = p + 2 * n
        loss = torch.empty((), dtype=torch.float32, device=q.device)
        sqd = torch.empty((t, width), dtype=torch.float32, device=q.device)
        coef = torch.empty((t, width), dtype=torch.float32, device=q.device)
        L.check(lib.scl_distance_tuple_loss_fwd(
            kind, 0 if other is None else 1, 1 if huber else 0, L.ptr(q), q_ts, L.ptr(pos), p_ts,
            L.ptr(neg), n_ts, L.ptr(other), o_ts, t, p, n, e, float(m1), float(m2), float(lam),
            L.ptr(dd), float(d_max), float(f_max), L.ptr(loss), L.ptr(sqd), L.ptr(coef),
            L.stream_of(q)))
        ctx.save_for_backward(q, pos, neg, other, coef)
        ctx.strides = (q_ts, p_ts, n_ts, o_ts)
        return loss

    @staticmethod
    def backward(ctx, grad_loss):
        return _TupleLoss.backward(ctx, grad_loss)[:4] + (None,) * 8


_TRIPLET_KINDS = {'triplet_loss': L.TUPLE_TRIPLET, 'lazy_triplet_loss': L.TUPLE_LAZY_TRIPLET}


def _distance_tuple(a, pos, neg, other, m1, m2, lam, d_dists, d_max, f_max, triplet_loss_name,
                    distance_loss_name):
    if triplet_loss_name not in _TRIPLET_KINDS:
        # the reference resolves the name with getattr(pointnetvlad_cls, ...)
        raise AttributeError("pointnetvlad_cls has no loss %r" % (triplet_loss_name,))
    return _DistanceTupleLoss.apply(
        a.contiguous(), pos.contiguous(), neg.contiguous(),
        None if other is None else other.contiguous(), d_dists,
        _TRIPLET_KINDS[triplet_loss_name], 'huber' in distance_loss_name, m1, m2, lam, d_max, f_max)


def distance_triplet_loss(a_feature, pos_features, neg_features, margin, lam, squared_d_dists,
                          d_max_squared, f_max_squared, triplet_loss_name='triplet_loss',
                          distance_loss_name='huber_distance_loss'):
    """model/losses.py:239-264: triplet + lam * (Huber) distance loss."""
    return _distance_tuple(a_feature, pos_features, neg_features, None, margin, 0.0, lam,
                           squared_d_dists, d_max_squared, f_max_squared, triplet_loss_name,
                           distance_loss_name)


def distance_quadruplet_loss(a_feature, pos_features, neg_features, other_neg, m1, m2, lam,
                             squared_d_dists, d_max_squared, f_max_squared,
                             triplet_loss_name='triplet_loss',
                             distance_loss_name='huber_distance_loss'):
    """model/losses.py:267-307."""
    return _distance_tuple(a_feature, pos_features, neg_features, other_neg, m1, m2, lam,
                           squared_d_dists, d_max_squared, f_max_squared, triplet_loss_name,
                           distance_loss_name)


def _distance_only(a_feature, pos_feature, squared_d_dists, d_max_squared, f_max_squared, huber):
    # the distance term alone: lam = 1 and a triplet whose hinge can never be active
    # (margin -inf would poison the sum; use the positives as negatives with margin -1e30)
    return _distance_tuple(a_feature, pos_feature, pos_feature, None, -1.0e30, 0.0, 1.0,
                           squared_d_dists, d_max_squared, f_max_squared, 'triplet_loss',
                           'huber_distance_loss' if huber else 'distance_loss')


def distance_loss(a_feature, pos_feature, squared_d_dists, d_max_squared, f_max_squared):
    """model/losses.py:225-230."""
    return _distance_only(a_feature, pos_feature, squared_d_dists, d_max_squared, f_max_squared,
                          False)


def huber_distance_loss(a_feature, pos_feature, squared_d_dists, d_max_squared, f_max_squared):
    """model/losses.py:233-236."""
    return _distance_only(a_feature, pos_feature, squared_d_dists, d_max_squared, f_max_squared,
                          True)


def evil_triplet_loss(q_vec, pos_vecs, neg_vecs, margin):
    """model/losses.py:63-73."""
    return _tuple(L.TUPLE_EVIL_TRIPLET, q_vec, pos_vecs, neg_vecs, None, margin, 0.0)


def evil_quadruplet_loss(q_vec, pos_vecs, neg_vecs, other_neg, m1, m2):
    """model/losses.py:197-214."""
    return _tuple(L.TUPLE_EVIL_QUADRUPLET, q_vec, pos_vecs, neg_vecs, other_neg, m1, m2)


def worst_pos_distance(query, pos_vecs):
    """model/losses.py:217-222: max over positives of the squared distance, [T].
    (Forward-only helper; the losses differentiate through their own kernels.)"""
    sqd = _anchor_sqdists(query, pos_vecs)
    return sqd.max(dim=1).values


def _anchor_sqdists(query, vecs):
    """[T,1,E] x [T,R,E] -> [T,R] squared distances via the tuple kernel."""
    lib = L.load()
    L.require_device(query, vecs)
    q, q_ts = _rows_view(query.contiguous(), 'query')
    v, v_ts = _rows_view(vecs.contiguous(), 'vecs')
    t, r, e = v.shape
    loss = torch.empty((), dtype=torch.float32, device=q.device)
    sqd = torch.empty((t, 3 * r), dtype=torch.float32, device=q.device)
    coef = torch.empty((t, 3 * r), dtype=torch.float32, device=q.device)
    # P = N = R with the same rows: the first R columns are the anchor distances
    L.check(lib.scl_tuple_loss_fwd(L.TUPLE_TRIPLET, L.ptr(q), q_ts, L.ptr(v), v_ts, L.ptr(v), v_ts,
                                   None, 0, t, r, r, e, 0.0, 0.0, L.ptr(loss), L.ptr(sqd),
                                   L.ptr(coef), L.stream_of(q)))
    return sqd[:, :r]


class _LogRatio(torch.autograd.Function):
    @staticmethod
    def forward(ctx, a, pos, neg, spd, snd):
        lib = L.load()
        L.require_device(a, pos, neg, spd, snd)
        a, pos, neg = (_as_f32(x).contiguous() for x in (a, pos, neg))
        if a.dim() != 3 or pos.dim() != 3 or neg.dim() != 3:
            raise ValueError("logratio_loss features must be rank 3")
        if a.shape[0] != 1 or pos.shape[0] != 1 or neg.shape[0] != 1:
            # log(pos_residuals / transpose(neg_residuals)) only broadcasts for T == 1
            raise ValueError("logratio_loss needs tuples_per_batch == 1 (model/losses.py:130)")
        p, n, e = pos.shape[1], neg.shape[1], pos.shape[2]
        if p != n:
            raise ValueError("logratio_loss needs as many positives as negatives "
                             "(model/losses.py:131-133), got P=%d N=%d" % (p, n))
        spd = _as_f32(spd).reshape(-1).contiguous()
        snd = _as_f32(snd).reshape(-1).contiguous()
        if spd.numel() != p or snd.numel() != n:
            raise ValueError("squared distance tensors must hold P and N values")
        loss = torch.empty((), dtype=torch.float32, device=a.device)
        sqd = torch.empty((1, p + 2 * n), dtype=torch.float32, device=a.device)
        coef = torch.empty((1, p + 2 * n), dtype=torch.float32, device=a.device)
        L.check(lib.scl_logratio_fwd(L.ptr(a), L.ptr(pos), L.ptr(neg), p, n, e, L.ptr(spd),
                                     L.ptr(snd), L.ptr(loss), L.ptr(sqd), L.ptr(coef),
                                     L.stream_of(a)))
        ctx.save_for_backward(a, pos, neg, coef)
        return loss

    @staticmethod
    def backward(ctx, grad_loss):
        lib = L.load()
        a, pos, neg, coef = ctx.saved_tensors
        p, n, e = pos.shape[1], neg.shape[1], pos.shape[2]
        g = _as_f32(grad_loss).contiguous()
        ga, gp, gn = torch.empty_like(a), torch.empty_like(pos), torch.empty_like(neg)
        L.check(lib.scl_tuple_loss_bwd(L.ptr(a), e, L.ptr(pos), p * e, L.ptr(neg), n * e, None, 0,
                                       1, p, n, e, L.ptr(coef), L.ptr(g), L.ptr(ga), L.ptr(gp),
                                       L.ptr(gn), None, L.stream_of(a)))
        return ga, gp, gn, None, None


def logratio_loss(a_feature, pos_features, neg_features, squared_pos_dists, squared_neg_dists):
    """model/losses.py:125-135 (literal broadcasting: T == 1 and P == N)."""
    return _LogRatio.apply(a_feature, pos_features, neg_features, squared_pos_dists,
                           squared_neg_dists)


def _pairwise_squared_distances(features):
    """model/losses.py:656-661: [T,S,E] -> [T,S,S] (forward only, like its one caller's
    use on detached positions; gradients are not provided)."""
    lib = L.load()
    L.require_device(features)
    f = _as_f32(features).contiguous()
    if f.dim() != 3:
        raise ValueError("features must be rank 3 [T,S,E]")
    t, s, e = f.shape
    out = torch.empty((t, s, s), dtype=torch.float32, device=f.device)
    nbytes = lib.scl_pairwise_sqdist_workspace_bytes(t, s, e)
    ws = L.workspace(nbytes, f.device)
    L.check(lib.scl_pairwise_sqdist(L.ptr(f), t, s, e, L.ptr(out), L.ptr(ws), ws.numel(),
                                    L.stream_of(f)))
    return out


class _PairwiseDistanceLoss(torch.autograd.Function):
    """mean over all (t, i, j) of phi(F_tij / f_max, D_tij / d_max), F = pairwise squared feature
    distances (``scl_pairwise_sqdist``).  With g = d loss / d F (an [S,S] map per tuple) the
    feature gradient is 2 (diag(rowsum(g + g^T)) - (g + g^T)) . features — a [S,S] x [S,E]
    product per tuple, which is what ``scl_gram_loss_bwd`` computes from a coefficient matrix."""

    @staticmethod
    def forward(ctx, feats, d_dists, d_max_squared, f_max_squared, huber):
        lib = L.load()
        L.require_device(feats, d_dists)
        f = _as_f32(feats).contiguous()
        t, s, e = f.shape
        dd = _as_f32(d_dists)
        if tuple(dd.shape) != (t, s, s):
            raise ValueError("pairwise_squared_d_dists %s must be [T,1+P,1+P] = %s"
                             % (tuple(dd.shape), (t, s, s)))
        sqf = _pairwise_squared_distances(f)
        sf = sqf / f_max_squared
        sd = dd / d_max_squared
        err = sd - sf                                   # huber: predictions - labels
        if huber:
            a = err.abs()
            quad = torch.clamp(a, max=1.0)
            sq = 0.5 * quad * quad + (a - quad)
            dphi = -torch.clamp(err, -1.0, 1.0)         # d phi / d sf
        else:
            sq = err * err
            dphi = -2.0 * err
        loss = sq.mean(dim=2).mean(dim=1).mean(dim=0)
        g = dphi / (f_max_squared * t * s * s)          # d loss / d F
        c = g + g.transpose(1, 2)
        coef = 2.0 * (torch.diag_embed(c.sum(dim=2)) - c)
        ctx.save_for_backward(f, coef.contiguous())
        return loss

    @staticmethod
    def backward(ctx, grad_loss):
        lib = L.load()
        f, coef = ctx.saved_tensors
        t, s, e = f.shape
        g = _as_f32(grad_loss).contiguous()
        grad = torch.empty_like(f)
        for k in range(t):
            L.check(lib.scl_gram_loss_bwd(L.ptr(f[k]), e, s, e, L.ptr(coef[k]), L.ptr(g), 0, s,
                                          L.ptr(grad[k]), e, L.stream_of(f)))
        return grad, None, None, None, None


def pairwise_distance_loss(anchor, positives, pairwise_squared_d_dists, d_max_squared, f_max_squared,
                           distance_loss_name='distance_loss'):
    """model/losses.py:627-646: squared feature distances between ALL pairs of
    [anchor | positives] against the squared geographic ones (the trainer's 'pairwise'
    distance tensor [T,1+P,1+P], train/train.py:535-537, 665-667), plain or Huber."""
    L.require_device(anchor, positives, pairwise_squared_d_dists)
    if anchor.dim() != 3 or positives.dim() != 3 or anchor.shape[0] != positives.shape[0] \
            or anchor.shape[2] != positives.shape[2]:
        raise ValueError("anchor %s / positives %s must be [T,1,E] / [T,P,E]"
                         % (tuple(anchor.shape), tuple(positives.shape)))
    feats = torch.cat([_as_f32(anchor), _as_f32(positives)], dim=1)
    return _PairwiseDistanceLoss.apply(feats, pairwise_squared_d_dists, float(d_max_squared),
                                       float(f_max_squared), 'huber' in distance_loss_name)
