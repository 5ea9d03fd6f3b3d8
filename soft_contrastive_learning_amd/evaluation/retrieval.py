"""Exact L2 top-n retrieval, the replacement for the KDTree query of
``evaluation/top-n.py:103-108`` (also ``train/train.py:1181-1182``).

``topn_l2`` runs csrc/topn.hip on one device; ``merge_topn`` combines per-shard results
when the reference set is split over ranks (shard the reference set,
replicate the queries, all-gather the [Q,n] candidates, merge).
"""
import torch

from .. import _lib as L

MAX_N = 25


def topn_l2(ref, query, n, idx_offset=0):
    """ref [R,d], query [Q,d] float32 on a HIP device -> (dists [Q,n] float64 ascending,
    idx [Q,n] int64), like ``KDTree(ref).query(query, k=n, sort_results=True)``."""
    lib = L.load()
    L.require_device(ref, query)
    ref = ref.float().contiguous()
    query = query.float().contiguous()
    if ref.dim() != 2 or query.dim() != 2 or ref.shape[1] != query.shape[1]:
        raise ValueError("ref %s / query %s must be [R,d] / [Q,d]" % (tuple(ref.shape),
                                                                     tuple(query.shape)))
    r, d = ref.shape
    q = query.shape[0]
    nbytes = lib.scl_topn_l2_workspace_bytes(r, q, d, n)
    if nbytes == 0:
        raise ValueError("unsupported retrieval shape R=%d Q=%d d=%d n=%d "
                         "(d in {32,64,128,256}, n <= %d, n <= R)" % (r, q, d, n, MAX_N))
    idx = torch.empty((q, n), dtype=torch.int64, device=ref.device)
    dist = torch.empty((q, n), dtype=torch.float64, device=ref.device)
    ws = L.workspace(nbytes, ref.device)
    L.check(lib.scl_topn_l2(L.ptr(ref), r, L.ptr(query), q, d, n, int(idx_offset), L.ptr(idx),
                            L.ptr(dist), L.ptr(ws), ws.numel(), L.stream_of(ref)))
    return dist, idx


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
