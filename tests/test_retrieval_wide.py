"""The d > 256 retrieval path (evaluation/retrieval._topn_wide: the in-training localisation
check on raw 32768-d descriptors, train/train.py:1181-1182, and top-n.py's d sweep above 256):
nomination from the inner products of the library's own kernel (scl_topn_dots: float32 matrix
instructions inside chunks of 256 features, float64 across them) + a PROVEN certificate + exact
fallback.  The logic around the kernel is tested on the CPU with a NumPy stand-in of the same
chunked arithmetic (oracle side: test infrastructure), the kernel itself at d = 300 / 320 / 32768 on
the GPU.

The adversarial set: more references than the 32 nominated candidates can hold whose exact
squared distances to a query differ by ~1e-14 relative — below what the float64 Gram form
resolves (its error is ~1e-13 here) but not below what the direct sum((q - r)^2) form the
reference's KDTree uses resolves: variants of one row that differ from it by a few float32
ulps in a coordinate where the row EQUALS the query, so D(q, variant) = D(q, row) + delta^2.
"""
import numpy as np
import pytest
import torch

from oracle import topn_np as TN


def adversarial_sets(r, q, d, n_variants, seed):
    rng = np.random.default_rng(seed)
    ref = rng.standard_normal((r, d)).astype(np.float32)
    qry = rng.standard_normal((q, d)).astype(np.float32)
    base = (qry[2] + 0.01 * rng.standard_normal(d)).astype(np.float32)
    # coordinates where the query is of ordinary size (delta = k ulps of THAT value) ...
    usable = np.where((np.abs(qry[2]) >= 0.5) & (np.abs(qry[2]) < 4.0))[0]
    coords = rng.permutation(usable)[:n_variants]
    assert len(coords) == n_variants
    base[coords] = qry[2][coords]                    # delta enters the distance as delta^2 only
    rows = np.sort(rng.permutation(r)[:n_variants + 1])
    ref[rows[0]] = base
    for v, (row, j) in enumerate(zip(rows[1:], coords)):
        var = base.copy()
        x = var[j]
        # ... k = 9, 11, 13, ... ulps: odd and distinct, so no two variants share a |delta|
        # whatever binade their coordinate lies in, and every delta^2 gap (>= 1e-13 absolute) is
        # far above the direct form's own rounding (1e-17) yet below the Gram form's (1e-12)
        for _ in range(9 + 2 * v):
            x = np.nextafter(x, np.float32(np.inf), dtype=np.float32)
        var[j] = x
        ref[row] = var
    # and a block of EXACT duplicates around another query (ties: lower index first)
    dup_rows = np.setdiff1d(rng.permutation(r)[:60], rows)[:40]
    ref[dup_rows] = (qry[5] + 0.02 * rng.standard_normal(d)).astype(np.float32)
    return ref, qry, rows, np.sort(dup_rows)


def check_against_tree(got_d, got_i, ref, qry, n, dup_rows, dup_query=5):
    want_d, want_i = TN.topn_kdtree(ref, qry, n)
    for qi in range(len(qry)):
        if qi == dup_query:       # exact ties may come in any order from the tree; ours: by index
            assert sorted(got_i[qi]) == sorted(dup_rows[:n].tolist())
            assert list(got_i[qi]) == sorted(got_i[qi])
        else:
            np.testing.assert_array_equal(got_i[qi], want_i[qi])
    np.testing.assert_allclose(got_d, want_d, rtol=1e-12, atol=1e-300)


def chunked_dots(rblk, qblk):
    """What scl_topn_dots computes, in NumPy: float32 products and sums inside chunks of 256
    features, the chunk sums added in float64 (the summation order inside a chunk differs from the
    kernel's; the certificate's bound holds for any order)."""
    r32, q32 = rblk.numpy().astype(np.float32), qblk.numpy().astype(np.float32)
    out = np.zeros((q32.shape[0], r32.shape[0]), np.float64)
    for c in range(0, r32.shape[1], 256):
        out += (q32[:, c:c + 256] @ r32[:, c:c + 256].T).astype(np.float64)
    return torch.from_numpy(out)


def test_wide_path_certificate_on_cpu():
    from soft_contrastive_learning_amd.evaluation import retrieval
    ref, qry, rows, dup_rows = adversarial_sets(400, 12, 300, 44, seed=5)
    st = {}
    d, i = retrieval._topn_wide(torch.tensor(ref), torch.tensor(qry), 25, 0, True, st, dots_fn=chunked_dots)
    check_against_tree(d.numpy(), i.numpy(), ref, qry, 25, dup_rows)
    assert 2 <= st['uncertified'] <= 4, st              # the two adversarial queries
    # the nomination alone does get the adversarial query wrong here (or the test is too easy)
    d0, i0 = retrieval._topn_wide(torch.tensor(ref), torch.tensor(qry), 25, 0, False, st, dots_fn=chunked_dots)
    assert st['uncertified'] is None
    _, want_i = TN.topn_kdtree(ref, qry, 25)
    ok = np.ones(len(qry), bool)
    ok[[2, 5]] = False
    np.testing.assert_array_equal(i0.numpy()[ok], want_i[ok])


def test_wide_path_separated_data_stays_certified_on_cpu():
    from soft_contrastive_learning_amd.evaluation import retrieval
    rng = np.random.default_rng(9)
    ref = rng.standard_normal((700, 320)).astype(np.float32)
    qry = rng.standard_normal((30, 320)).astype(np.float32)
    st = {}
    d, i = retrieval._topn_wide(torch.tensor(ref), torch.tensor(qry), 5, 100, True, st, dots_fn=chunked_dots)
    want_d, want_i = TN.topn_kdtree(ref, qry, 5)
    np.testing.assert_array_equal(i.numpy(), want_i + 100)
    np.testing.assert_allclose(d.numpy(), want_d, rtol=1e-12)
    assert st['uncertified'] == 0
    # fewer references than candidates: everything is re-ranked, nothing to certify
    d, i = retrieval._topn_wide(torch.tensor(ref[:20]), torch.tensor(qry), 5, 0, True, st, dots_fn=chunked_dots)
    np.testing.assert_array_equal(i.numpy(), TN.topn_kdtree(ref[:20], qry, 5)[1])
    assert st['uncertified'] == 0


@pytest.mark.gpu
def test_wide_path_certificate_at_the_localisation_width():
    """d = 32768 through the public entry point, as train/evaluate.py calls it."""
    from soft_contrastive_learning_amd.evaluation import retrieval
    assert torch.cuda.is_available()
    dev = torch.device('cuda:0')
    ref, qry, rows, dup_rows = adversarial_sets(300, 8, 32768, 40, seed=11)
    st = {}
    d, i = retrieval.topn_l2(torch.tensor(ref, device=dev), torch.tensor(qry, device=dev), 5, stats=st)
    check_against_tree(d.cpu().numpy(), i.cpu().numpy(), ref, qry, 5, dup_rows)
    assert 2 <= st['uncertified'] <= 4, st


@pytest.mark.gpu
@pytest.mark.parametrize("r,q,d", [(400, 12, 300), (700, 30, 320), (3000, 70, 1025), (200, 3, 4096)])
def test_wide_dots_kernel_within_its_stated_bound(r, q, d):
    """scl_topn_dots against float64 inner products: |error| <= gamma_256 |q||r| (+ the float64
    chunk sums) — the bound _topn_wide's certificate is built on — at widths that are not
    multiples of the chunk or of the 16-byte loads, and tiles that are not full."""
    from soft_contrastive_learning_amd import _lib as L
    dev = torch.device('cuda:0')
    rng = np.random.default_rng(d)
    ref = rng.standard_normal((r, d)).astype(np.float32)
    qry = rng.standard_normal((q, d)).astype(np.float32)
    splits = 1 if d < 1024 else 3
    out = torch.empty((splits, q, r), dtype=torch.float64, device=dev)
    rt, qt = torch.tensor(ref, device=dev), torch.tensor(qry, device=dev)
    L.check(L.load().scl_topn_dots(L.ptr(rt), r, L.ptr(qt), q, d, splits, L.ptr(out), L.stream_of(rt)))
    out = out.sum(0)
    want = qry.astype(np.float64) @ ref.astype(np.float64).T
    g32 = 256 * 2.0 ** -24 / (1 - 256 * 2.0 ** -24)
    bound = g32 * np.linalg.norm(qry.astype(np.float64), axis=1)[:, None] * np.linalg.norm(
        ref.astype(np.float64), axis=1)[None] + 2.0 ** -50 * np.abs(want)
    err = np.abs(out.cpu().numpy() - want)
    assert (err <= bound).all(), float((err / bound).max())
    # typical error sits two orders below the worst-case bound
    assert float((err / bound).max()) < 0.2


@pytest.mark.gpu
def test_wide_path_on_the_gpu_at_the_cpu_test_shapes():
    from soft_contrastive_learning_amd.evaluation import retrieval
    dev = torch.device('cuda:0')
    ref, qry, rows, dup_rows = adversarial_sets(400, 12, 300, 44, seed=5)
    st = {}
    d, i = retrieval.topn_l2(torch.tensor(ref, device=dev), torch.tensor(qry, device=dev), 25, stats=st)
    check_against_tree(d.cpu().numpy(), i.cpu().numpy(), ref, qry, 25, dup_rows)
    assert 2 <= st['uncertified'] <= 4, st


# ---- n > 25 (evaluation/top-n.py:135 leaves --N free): retrieval._topn_many --------------------
def _exact_topk_cpu(ref, query, k, idx_offset=0, score='f32', certify=True, stats=None):
    """CPU stand-in of the certified top-25 primitive (float64 direct form, (distance, index)
    order) for the logic test of the layer above it: test infrastructure, never the product."""
    r64, q64 = ref.double(), query.double()
    d = ((q64[:, None, :] - r64[None]) ** 2).sum(-1)
    idx = torch.arange(ref.shape[0])[None].expand_as(d)
    o = torch.argsort(d, dim=1, stable=True)[:, :k]              # stable: ties by index
    return torch.gather(d, 1, o).sqrt(), torch.gather(idx, 1, o) + idx_offset


@pytest.mark.parametrize("n", [26, 50, 100])
def test_topn_many_is_exact_from_the_top25_primitive(monkeypatch, n):
    """Interleaved shards + per-(query, shard) certification + splitting: index lists equal the
    reference's KDTree.query(k=N) on tie-free data AND on a set where one query's 80 nearest
    references are consecutive multiples of the shard count (all in ONE shard: the refinement
    must run), incl. exact duplicates (ties go by index, like the kernel's order)."""
    from soft_contrastive_learning_amd.evaluation import retrieval
    monkeypatch.setattr(retrieval, 'topn_l2', _exact_topk_cpu)
    rng = np.random.default_rng(n)
    r, q, d = 1500, 40, 24
    ref = rng.standard_normal((r, d)).astype(np.float32)
    qry = rng.standard_normal((q, d)).astype(np.float32)
    shards = max(2, -(-n // 10))
    rows = np.arange(0, 80 * shards, shards)                    # every one in shard 0
    ref[rows] = qry[3] + 1e-3 * rng.standard_normal((80, d)).astype(np.float32)
    ref[rows[5]] = ref[rows[4]]                                  # exact duplicates: a tie
    ref[rows[7]] = ref[rows[4]]
    st = {}
    got_d, got_i = retrieval._topn_many(torch.tensor(ref), torch.tensor(qry), n, 7, 'f32', st)
    want_d, want_i = _exact_topk_cpu(torch.tensor(ref), torch.tensor(qry), n, 7)
    assert torch.equal(got_i, want_i)
    np.testing.assert_allclose(got_d.numpy(), want_d.numpy(), rtol=1e-12)
    assert st['refined'] >= 1                                    # query 3 overflowed its shard
    kd_d, kd_i = TN.topn_kdtree(ref, qry[10:20], n)              # tie-free queries: the tree itself
    np.testing.assert_array_equal(got_i[10:20].numpy() - 7, kd_i)
