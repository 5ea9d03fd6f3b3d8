"""Exact L2 top-n retrieval, the replacement for the KDTree query of
``evaluation/top-n.py:103-108`` (also ``train/train.py:1181-1182``).

``topn_l2`` runs csrc/topn.hip on one device; ``merge_topn`` combines per-shard results
when the reference set is split over ranks (shard the reference set,
replicate the queries, all-gather the [Q,n] candidates, merge).
"""
import torch

from .. import _lib as L

MAX_N = 25


SCORE_MODES = {'f32': L.TOPN_SCORE_F32, 'bf16x3': L.TOPN_SCORE_BF16X3}


def topn_l2(ref, query, n, idx_offset=0, score='f32', certify=True, stats=None):
    """ref [R,d], query [Q,d] float32 on a HIP device -> (dists [Q,n] float64 ascending,
    idx [Q,n] int64), like ``KDTree(ref).query(query, k=n, sort_results=True)``.

    ``score`` picks how the 32 candidates per (query, reference split) are nominated before
    the float64 re-rank: 'f32' (exact-float32 matrix instructions, score error ~1e-7) or
    'bf16x3' (three bf16 matrix products of the high/low operand halves: |score error| <=
    1.2e-5 |q||r|, about 1.8x faster).  The emitted distances are float64-exact either way.

    ``certify`` (default): the kernel proves per query that no reference outside the
    nominated 32 can reach the n-th exact distance (``scl_topn_l2_cert``); the queries it
    cannot prove this for — near-duplicate references around the n-th neighbour — are
    resolved by an exact float64 pass over every reference (``scl_topn_exact_filter``), so the
    index lists are exact with no assumption on the data.  ``stats`` (a dict) receives
    ``uncertified`` = how many queries took that path."""
    lib = L.load()
    if score not in SCORE_MODES:
        raise ValueError("score must be one of %s, got %r" % (sorted(SCORE_MODES), score))
    L.require_device(ref, query)
    ref = ref.float().contiguous()
    query = query.float().contiguous()
    if ref.dim() != 2 or query.dim() != 2 or ref.shape[1] != query.shape[1]:
        raise ValueError("ref %s / query %s must be [R,d] / [Q,d]" % (tuple(ref.shape),
                                                                     tuple(query.shape)))
    r, d = ref.shape
    q = query.shape[0]
    if n > MAX_N or n > r or n < 1:
        raise ValueError("n must be in [1, min(%d, R)], got n=%d R=%d" % (MAX_N, n, r))
    if d > 256:
        return _topn_wide(ref, query, n, idx_offset)
    if d not in (32, 64, 128, 256):
        # zero-padding the feature axis leaves every distance unchanged
        pad = next(c for c in (32, 64, 128, 256) if d <= c)
        ref = torch.nn.functional.pad(ref, (0, pad - d))
        query = torch.nn.functional.pad(query, (0, pad - d))
        d = pad
    flags = SCORE_MODES[score]
    nbytes = lib.scl_topn_l2_ex_workspace_bytes(r, q, d, n, flags)
    if nbytes == 0:
        raise ValueError("unsupported retrieval shape R=%d Q=%d d=%d n=%d "
                         "(d in {32,64,128,256}, n <= %d, n <= R)" % (r, q, d, n, MAX_N))
    idx = torch.empty((q, n), dtype=torch.int64, device=ref.device)
    dist = torch.empty((q, n), dtype=torch.float64, device=ref.device)
    ws = L.workspace(nbytes, ref.device)
    if not certify:
        L.check(lib.scl_topn_l2_ex(L.ptr(ref), r, L.ptr(query), q, d, n, int(idx_offset),
                                   L.ptr(idx), L.ptr(dist), L.ptr(ws), ws.numel(), flags,
                                   L.stream_of(ref)))
        return dist, idx
    flag = torch.empty(q, dtype=torch.uint8, device=ref.device)
    bound = torch.empty(q, dtype=torch.float64, device=ref.device)
    L.check(lib.scl_topn_l2_cert(L.ptr(ref), r, L.ptr(query), q, d, n, int(idx_offset), L.ptr(idx),
                                 L.ptr(dist), L.ptr(flag), L.ptr(bound), L.ptr(ws), ws.numel(),
                                 flags, L.stream_of(ref)))
    bad = torch.nonzero(flag).reshape(-1)                      # (synchronises; usually empty)
    if stats is not None:
        stats['uncertified'] = int(bad.numel())
    if bad.numel():
        _resolve_exactly(ref, query, n, int(idx_offset), bad, bound, dist, idx)
    return dist, idx


_FILTER_CAP = 2048      # candidates within the bound kept per uncertified query


def _resolve_exactly(ref, query, n, idx_offset, bad, bound, dist, idx):
    """Rows ``bad`` of (dist, idx) recomputed from the exact float64 distances to every
    reference that lies within the query's bound."""
    lib = L.load()
    r, d = ref.shape
    dev = ref.device
    for s in range(0, bad.numel(), 4096):
        ql = bad[s:s + 4096].to(torch.int32).contiguous()
        nq = ql.numel()
        count = torch.empty(nq, dtype=torch.int32, device=dev)
        cd = torch.full((nq, _FILTER_CAP), float('inf'), dtype=torch.float64, device=dev)
        ci = torch.full((nq, _FILTER_CAP), 2 ** 31 - 1, dtype=torch.int32, device=dev)
        L.check(lib.scl_topn_exact_filter(L.ptr(ref), r, L.ptr(query), d, L.ptr(ql), nq,
                                          L.ptr(bound), _FILTER_CAP, L.ptr(count), L.ptr(cd),
                                          L.ptr(ci), L.stream_of(ref)))
        # order by (distance, index): stable sort by index first, then by distance
        o = torch.argsort(ci, dim=1, stable=True)
        cd, ci = torch.gather(cd, 1, o), torch.gather(ci, 1, o)
        o = torch.argsort(cd, dim=1, stable=True)[:, :n]
        rows = ql.long()
        dist[rows] = torch.gather(cd, 1, o).sqrt()
        idx[rows] = torch.gather(ci, 1, o).long() + idx_offset
        over = torch.nonzero(count > _FILTER_CAP).reshape(-1)
        for k in over.tolist():
            # more than _FILTER_CAP references within the n-th distance (thousands of exact
            # duplicates): plain float64 brute force for that query, in reference chunks
            qi = int(ql[k])
            qv = query[qi].double()
            best_d = torch.empty(0, dtype=torch.float64, device=dev)
            best_i = torch.empty(0, dtype=torch.int64, device=dev)
            for a in range(0, r, 65536):
                dd = ((ref[a:a + 65536].double() - qv) ** 2).sum(1)
                ii = torch.arange(a, a + dd.numel(), device=dev)
                best_d, best_i = torch.cat([best_d, dd]), torch.cat([best_i, ii])
                o2 = torch.argsort(best_d, stable=True)[:n]      # index-ascending input: ties by index
                best_d, best_i = best_d[o2], best_i[o2]
            dist[qi] = best_d.sqrt()
            idx[qi] = best_i + idx_offset


_KEEP = 32          # candidates nominated in float32 before the float64 re-rank


def _topn_wide(ref, query, n, idx_offset):
    """Descriptors wider than 256 (the in-training localisation check runs the KDTree on the
    raw 32768-d vectors, train/train.py:1181-1182; evaluation/top-n.py sweeps d up to 4096):
    blocks of queries x references go through the exact-f32 pairwise-distance kernel
    (``scl_pairwise_sqdist``), the best 32 per query are kept across blocks and re-ranked in
    float64 with the direct (q - r)^2 form, like the fused kernel does for d <= 256."""
    from ..model import losses
    r, d = ref.shape
    q = query.shape[0]
    qb_max, cap = 512, 4096
    out_d = torch.empty((q, n), dtype=torch.float64, device=ref.device)
    out_i = torch.empty((q, n), dtype=torch.int64, device=ref.device)
    for qs in range(0, q, qb_max):
        qq = query[qs:qs + qb_max]
        qb = qq.shape[0]
        cand_s, cand_i = [], []
        for rs in range(0, r, cap - qb):
            rr = ref[rs:rs + cap - qb]
            d2 = losses._pairwise_squared_distances(torch.cat([qq, rr], 0)[None])[0][:qb, qb:]
            k = min(_KEEP, rr.shape[0])
            s, i = torch.topk(d2, k, dim=1, largest=False)
            cand_s.append(s)
            cand_i.append(i + rs)
        s, i = torch.cat(cand_s, 1), torch.cat(cand_i, 1)
        k = min(_KEEP, s.shape[1])
        _, pick = torch.topk(s, k, dim=1, largest=False)
        cand = torch.gather(i, 1, pick)                                   # [qb, k]
        exact = torch.empty((qb, k), dtype=torch.float64, device=ref.device)
        step = max(1, (1 << 25) // (k * d))                               # <= 256 MB of float64
        for a in range(0, qb, step):
            diff = qq[a:a + step].double()[:, None, :] - ref[cand[a:a + step]].double()
            exact[a:a + step] = (diff * diff).sum(-1)
        # order by (distance, index): stable sort by index first, then by distance
        o = torch.argsort(cand, dim=1, stable=True)
        exact, cand = torch.gather(exact, 1, o), torch.gather(cand, 1, o)
        o = torch.argsort(exact, dim=1, stable=True)[:, :n]
        out_d[qs:qs + qb] = torch.gather(exact, 1, o).sqrt()
        out_i[qs:qs + qb] = torch.gather(cand, 1, o) + int(idx_offset)
    return out_d, out_i


def merge_topn(dists, idxs, n):
    """Merge per-shard (dists, idx) lists into the global top-n, ordered by
    (distance, index) like the single-device kernel."""
    d = torch.cat(list(dists), dim=1)
    i = torch.cat(list(idxs), dim=1)
    # stable two-key sort: by index first, then (stably) by distance
    o = torch.argsort(i, dim=1, stable=True)
    d, i = torch.gather(d, 1, o), torch.gather(i, 1, o)
    o = torch.argsort(d, dim=1, stable=True)[:, :n]
    return torch.gather(d, 1, o), torch.gather(i, 1, o)
