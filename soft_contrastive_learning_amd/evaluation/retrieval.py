"""Exact L2 top-n retrieval, the replacement for the KDTree query of
``evaluation/top-n.py:103-108`` (also ``train/train.py:1181-1182``).

``topn_l2`` runs csrc/topn.hip on one device; ``merge_topn`` combines per-shard results
when the reference set is split over ranks (shard the reference set,
replicate the queries, all-gather the [Q,n] candidates, merge).
"""
import torch

from .. import _lib as L

MAX_N = 25          # what ONE kernel pass returns per query (32 nominated, 25 certified: csrc/topn.hip)


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
    ``uncertified`` = how many queries took that path.  Descriptors wider than 256 take
    ``_topn_wide`` (nomination from the inner products of ``scl_topn_dots``, its own certificate
    of the same kind; ``score`` does not apply there)."""
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
    if n > r or n < 1:
        raise ValueError("n must be in [1, R], got n=%d R=%d" % (n, r))
    if d > 256:
        return _topn_wide(ref.contiguous(), query.contiguous(), n, idx_offset, certify, stats)
    if n > MAX_N:
        # evaluation/top-n.py:135 leaves --N free; one kernel pass certifies 25 per query
        return _topn_many(ref, query, n, idx_offset, score, stats)
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
            best_d, best_i = _brute_force_f64(ref, query[qi], n)
            dist[qi] = best_d.sqrt()
            idx[qi] = best_i + idx_offset


def _topn_many(ref, query, n, idx_offset=0, score='f32', stats=None):
    """Exact top-n for n > 25 (``KDTree.query(k=N)`` with the reference's free ``--N``,
    evaluation/top-n.py:103-108, 135) from the exact, certified top-25 primitive.

    The references are dealt into S interleaved shards (``ref[s::S]``: neighbours that are
    adjacent in list order — consecutive frames of a traverse — spread over all shards); every
    shard returns its exact top-25 per query; the merged pool, ordered by (distance, index), gives
    the candidate n-th distance D_n.  A shard whose 25th distance is STRICTLY beyond D_n cannot
    hold anything else within D_n, so its contribution is complete.  For the (query, shard) pairs
    where that fails — more than 25 of the query's n nearest fell into one shard, or ties at D_n —
    the shard is split in two and those queries alone are asked again, until every pair is
    certified (a shard of <= 25 references returns all of them).  Exact for any data; the expected
    case (S chosen so that a shard holds ~n/S << 25 of the top n) is one pass and no refinement.
    ``stats['refined']`` counts the (query, shard) pairs that needed a split."""
    dev = ref.device
    r = ref.shape[0]
    q = query.shape[0]
    shards = max(2, -(-n // 10))                      # expected share of the top n per shard: <= 10
    inf = float('inf')
    pool_d = torch.full((q, 0), inf, dtype=torch.float64, device=dev)
    pool_i = torch.zeros((q, 0), dtype=torch.int64, device=dev)
    refined = 0
    # work list: (reference rows of the shard as a LongTensor of positions, query rows or None = all)
    work = [(torch.arange(s, r, shards, device=dev), None) for s in range(shards)]
    passes = []                                      # (positions, query rows, dists, local idx)
    while work:
        pos, qrows = work.pop()
        sub = ref[pos].contiguous()
        qq = query if qrows is None else query[qrows].contiguous()
        k = min(MAX_N, sub.shape[0])
        dd, ii = topn_l2(sub, qq, k, 0, score)
        passes.append((pos, qrows, dd, pos[ii]))
        # fold into the pool: rows of the asked queries only
        rows = torch.arange(q, device=dev) if qrows is None else qrows
        add_d = torch.full((q, k), inf, dtype=torch.float64, device=dev)
        add_i = torch.full((q, k), 2 ** 62, dtype=torch.int64, device=dev)
        add_d[rows], add_i[rows] = dd, pos[ii]
        pool_d, pool_i = torch.cat([pool_d, add_d], 1), torch.cat([pool_i, add_i], 1)
        if work:
            continue
        # every outstanding pass is in: merge by (distance, index), drop what a refined shard
        # returned a second time (same index: its parent already gave it), then the candidate
        # n-th distance per query and the test
        o = torch.argsort(pool_i, dim=1, stable=True)
        pd, pi = torch.gather(pool_d, 1, o), torch.gather(pool_i, 1, o)
        dup = torch.zeros_like(pi, dtype=torch.bool)
        dup[:, 1:] = pi[:, 1:] == pi[:, :-1]
        pd = pd.masked_fill(dup, inf)
        o = torch.argsort(pd, dim=1, stable=True)
        pd, pi = torch.gather(pd, 1, o), torch.gather(pi, 1, o)
        keep = min(pd.shape[1], 4 * n)
        pool_d, pool_i = pd[:, :keep].contiguous(), pi[:, :keep].contiguous()
        d_n = pool_d[:, n - 1]
        for pos, qrows, dd, _ in passes:
            if pos.numel() <= MAX_N:
                continue                              # the shard returned all it has
            rows = torch.arange(q, device=dev) if qrows is None else qrows
            bad = rows[~(dd[:, -1] > d_n[rows])]
            if bad.numel():
                refined += int(bad.numel())
                work.append((pos[0::2], bad))
                work.append((pos[1::2], bad))
        passes = []
    if stats is not None:
        stats['refined'] = refined
    return pool_d[:, :n].contiguous(), (pool_i[:, :n] + int(idx_offset)).contiguous()


_KEEP = 32          # candidates nominated per query before the exact re-rank (wide path: >= n + 8)


def _brute_force_f64(ref, qv, n, chunk=65536):
    """Exact float64 sum((q - r)^2) of one query against every reference, best n by
    (distance, index): the arithmetic of the tree the reference builds."""
    dev = ref.device
    qv = qv.double()
    best_d = torch.empty(0, dtype=torch.float64, device=dev)
    best_i = torch.empty(0, dtype=torch.int64, device=dev)
    step = max(1, min(chunk, (1 << 25) // max(ref.shape[1], 1)))       # <= 256 MB of float64
    for a in range(0, ref.shape[0], step):
        dd = ((ref[a:a + step].double() - qv) ** 2).sum(1)
        ii = torch.arange(a, a + dd.numel(), device=dev)
        best_d, best_i = torch.cat([best_d, dd]), torch.cat([best_i, ii])
        o2 = torch.argsort(best_d, stable=True)[:n]          # index-ascending input: ties by index
        best_d, best_i = best_d[o2], best_i[o2]
    return best_d, best_i


def _topn_wide(ref, query, n, idx_offset, certify=True, stats=None, dots_fn=None):
    """Descriptors wider than 256 (the in-training localisation check runs the KDTree on the
    raw 32768-d vectors, train/train.py:1181-1182; evaluation/top-n.py sweeps d up to 4096).

    Nomination: s(q, r) = |q|^2 + |r|^2 - 2 q.r with the inner products from the library's own
    kernel ``scl_topn_dots`` (csrc/topn.hip: float32 matrix instructions inside chunks of 256
    features, float64 across chunks; round 3 used the float64 library GEMM here), norms in
    float64, blocks of queries x references, the best 32 per query kept across blocks.  Those 32
    are re-ranked with the direct float64 sum((q - r)^2) — the tree's arithmetic — and ordered
    by (distance, index).

    Certificate (no assumption on the data): every reference OUTSIDE the nominated set has
    s >= tau (the largest nominated score), and |s - D| <= eps for the exact squared distance D,
    with eps = 2 [2 gamma |q| R_max + (d + 8) 2^-53 (|q| + R_max)^2]: gamma = 256 u / (1 - 256 u),
    u = 2^-24, bounds a 256-long float32 FMA chain in ANY order, chunks combine by Cauchy-Schwarz
    (so the bound does not grow with d), the second term covers the float64 chunk sums, the two
    norms and the final additions, and the leading 2 is safety.  So if the n-th exact distance
    among the nominated is strictly below tau - eps (inflated by the direct form's own
    2 (d + 2) 2^-53 relative error), no outsider can reach it.  Queries that fail the test — near
    duplicates around the n-th neighbour within eps (6e-5 in squared distance for unit
    descriptors), or more than 32 references that close — are resolved by ``_brute_force_f64``
    over every reference.  ``stats['uncertified']`` counts them.

    ``dots_fn(ref_block, query_block) -> [Q, R] float64`` replaces the HIP kernel in the CPU
    tests of the logic around it (tests/test_retrieval_wide.py: the same chunked arithmetic in
    NumPy); the product path has no such argument and no CPU fallback."""
    lib = L.load() if dots_fn is None else None
    r, d = ref.shape
    q = query.shape[0]
    dev = ref.device
    keep = max(_KEEP, n + 8)                  # n > 25 (evaluation/top-n.py:135): nominate more
    u = 2.0 ** -53
    g32 = 256.0 * 2.0 ** -24 / (1.0 - 256.0 * 2.0 ** -24)
    qb_max = 512
    rb = 8192
    out_d = torch.empty((q, n), dtype=torch.float64, device=dev)
    out_i = torch.empty((q, n), dtype=torch.int64, device=dev)
    sb = max(256, (1 << 27) // d)                             # <= 1 GB of float64 per norm block
    rn_all = torch.cat([(ref[a:a + sb].double() ** 2).sum(1) for a in range(0, r, sb)])
    r_max = float(rn_all.max().sqrt())
    uncertified = 0
    # feature-axis splits: enough grid layers to put a workgroup on every CU for small blocks
    chunks = -(-d // 256)
    tiles = -(-min(q, qb_max) // 64) * -(-min(r, rb) // 64)
    splits = max(1, min(chunks, 16, 512 // max(tiles, 1)))
    while splits > 1 and (splits - 1) * -(-chunks // splits) >= chunks:
        splits -= 1
    dots = torch.empty(splits * min(q, qb_max) * min(r, rb), dtype=torch.float64, device=dev)
    for qs in range(0, q, qb_max):
        qblk = query[qs:qs + qb_max]
        qb = qblk.shape[0]
        qn = torch.cat([(qblk[a:a + sb].double() ** 2).sum(1) for a in range(0, qb, sb)])
        cand_s, cand_i = [], []
        for rs in range(0, r, rb):
            rblk = ref[rs:rs + rb]
            nr = rblk.shape[0]
            if dots_fn is None:
                dv = dots[:splits * qb * nr].view(splits, qb, nr)
                L.check(lib.scl_topn_dots(L.ptr(rblk), nr, L.ptr(qblk), qb, d, splits, L.ptr(dv),
                                          L.stream_of(ref)))
                dv = dv.sum(0) if splits > 1 else dv[0]
            else:
                dv = dots_fn(rblk, qblk)
            sc = qn[:, None] + rn_all[None, rs:rs + nr] - 2.0 * dv
            k = min(keep, nr)
            sv, iv = torch.topk(sc, k, dim=1, largest=False)
            cand_s.append(sv)
            cand_i.append(iv + rs)
        sv, iv = torch.cat(cand_s, 1), torch.cat(cand_i, 1)
        k = min(keep, sv.shape[1])
        top_s, pick = torch.topk(sv, k, dim=1, largest=False)
        cand = torch.gather(iv, 1, pick)                                  # [qb, k]
        exact = torch.empty((qb, k), dtype=torch.float64, device=dev)
        step = max(1, (1 << 25) // (k * d))                               # <= 256 MB of float64
        for a in range(0, qb, step):
            diff = qblk[a:a + step].double()[:, None, :] - ref[cand[a:a + step]].double()
            exact[a:a + step] = (diff * diff).sum(-1)
        # order by (distance, index): stable sort by index first, then by distance
        o = torch.argsort(cand, dim=1, stable=True)
        exact, cand = torch.gather(exact, 1, o), torch.gather(cand, 1, o)
        o = torch.argsort(exact, dim=1, stable=True)[:, :n]
        dn = torch.gather(exact, 1, o)
        out_d[qs:qs + qb] = dn.sqrt()
        out_i[qs:qs + qb] = torch.gather(cand, 1, o) + int(idx_offset)
        if certify and r > k:                    # (r <= keep: every reference was re-ranked)
            tau = top_s.max(dim=1).values
            qnorm = qn.sqrt()
            eps = 2.0 * (2.0 * g32 * qnorm * r_max + (d + 8) * u * (qnorm + r_max) ** 2)
            ok = dn[:, n - 1] * (1.0 + 2.0 * (d + 2) * u) < tau - eps
            for qi in torch.nonzero(~ok).reshape(-1).tolist():
                bd, bi = _brute_force_f64(ref, query[qs + qi], n)
                out_d[qs + qi] = bd.sqrt()
                out_i[qs + qi] = bi + int(idx_offset)
                uncertified += 1
    if stats is not None:
        stats['uncertified'] = uncertified if certify else None
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
