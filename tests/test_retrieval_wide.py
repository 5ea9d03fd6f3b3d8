"""The d > 256 retrieval path (evaluation/retrieval._topn_wide: the in-training localisation
check on raw 32768-d descriptors, train/train.py:1181-1182, and top-n.py's d sweep above 256):
float64 nomination + a PROVEN certificate + exact fallback.  The path is plain torch above a
library GEMM, so its logic is tested on the CPU here and at d = 32768 on the GPU.

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


def test_wide_path_certificate_on_cpu():
    from soft_contrastive_learning_amd.evaluation import retrieval
    ref, qry, rows, dup_rows = adversarial_sets(400, 12, 300, 44, seed=5)
    st = {}
    d, i = retrieval._topn_wide(torch.tensor(ref), torch.tensor(qry), 25, 0, True, st)
    check_against_tree(d.numpy(), i.numpy(), ref, qry, 25, dup_rows)
    assert 2 <= st['uncertified'] <= 4, st              # the two adversarial queries
    # the nomination alone does get the adversarial query wrong here (or the test is too easy)
    d0, i0 = retrieval._topn_wide(torch.tensor(ref), torch.tensor(qry), 25, 0, False, st)
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
    d, i = retrieval._topn_wide(torch.tensor(ref), torch.tensor(qry), 5, 100, True, st)
    want_d, want_i = TN.topn_kdtree(ref, qry, 5)
    np.testing.assert_array_equal(i.numpy(), want_i + 100)
    np.testing.assert_allclose(d.numpy(), want_d, rtol=1e-12)
    assert st['uncertified'] == 0
    # fewer references than candidates: everything is re-ranked, nothing to certify
    d, i = retrieval._topn_wide(torch.tensor(ref[:20]), torch.tensor(qry), 5, 0, True, st)
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
