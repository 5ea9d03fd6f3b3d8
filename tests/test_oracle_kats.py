"""Known-answer tests that pin the CPU oracle.

K1 is the only constant the reference itself holds (model/losses.py:708-711, the
un-asserted __main__ smoke of _pairwise_squared_distances).  K2..K8 are derived by
pencil from the reference formulas; each test states the derivation.
"""
import math

import numpy as np
import pytest

from oracle import losses_np as L
from oracle import netvlad_np as NV
from oracle import topn_np as TN
from oracle import twin_torch as TT

F32 = np.float32


def test_k1_pairwise_squared_distances_reference_smoke():
    # model/losses.py:709-710: B = [[[1,1],[2,2],[3,3]], [[1,1],[2,2],[4,4]]]
    c = np.array([[[1.0, 1], [2, 2], [3, 3]], [[1, 1], [2, 2], [4, 4]]], dtype=F32)
    d = L.pairwise_squared_distances(c)
    want = np.array([[[0, 2, 8], [2, 0, 2], [8, 2, 0]],
                     [[0, 2, 18], [2, 0, 8], [18, 8, 0]]], dtype=F32)
    np.testing.assert_array_equal(d, want)


def _k2():
    q = np.array([[[0.0, 0.0]]], dtype=F32)
    pos = np.array([[[1.0, 0.0], [0.0, 2.0]]], dtype=F32)
    neg = np.array([[[3.0, 0.0], [0.0, 1.0]]], dtype=F32)
    return q, pos, neg


def test_k2_triplet_family():
    # best_pos = min(1, 4) = 1; neg d2 = {9, 1}; hinges = max(0.5+1-{9,1}, 0) = {0, .5}
    q, pos, neg = _k2()
    assert L.triplet_loss(q, pos, neg, 0.5) == F32(0.5)
    assert L.lazy_triplet_loss(q, pos, neg, 0.5) == F32(0.5)
    # worst_pos = 4 -> hinges {0, 3.5}
    assert L.evil_triplet_loss(q, pos, neg, 0.5) == F32(3.5)


def test_k3_quadruplet_family():
    # other = (3,1): |neg-other|^2 = {1, 9}; second hinges max(0.2+1-{1,9},0) = {.2, 0}
    q, pos, neg = _k2()
    other = np.array([[[3.0, 1.0]]], dtype=F32)
    assert L.quadruplet_loss(q, pos, neg, other, 0.5, 0.2) == pytest.approx(0.7, rel=1e-6)
    assert L.lazy_quadruplet_loss(q, pos, neg, other, 0.5, 0.2) == pytest.approx(0.7, rel=1e-6)
    # evil: worst_pos = 4: first {0,3.5} -> 3.5 ; second max(0.2+4-{1,9},0) = {3.2,0}
    assert L.evil_quadruplet_loss(q, pos, neg, other, 0.5, 0.2) == pytest.approx(6.7, rel=1e-6)


def test_k4_ms_same_label_identical_no_mining():
    e = np.array([[1.0, 0.0], [1.0, 0.0]], dtype=F32)
    got = L.ms_loss([7, 7], e, ms_mining=False)
    assert got == pytest.approx(math.log(2.0) / 2.0, rel=1e-6)      # 0.346574


def test_k4b_ms_same_label_identical_mining_removes_the_positive():
    e = np.array([[1.0, 0.0], [1.0, 0.0]], dtype=F32)
    assert L.ms_loss([7, 7], e) == F32(0.0)


def test_k5_ms_different_label_identical():
    e = np.array([[1.0, 0.0], [1.0, 0.0]], dtype=F32)
    for mining in (True, False):
        got = L.ms_loss([0, 1], e, ms_mining=mining)
        assert got == pytest.approx(math.log(2.0) / 50.0, rel=1e-6)  # 0.0138629


def test_k6_ms_different_label_orthogonal_rounds_to_zero_in_fp32():
    e = np.array([[1.0, 0.0], [0.0, 1.0]], dtype=F32)
    assert L.ms_loss([0, 1], e) == F32(0.0)


def test_k7_wms_exp_mask_identities():
    d = np.array([[0.0, 5.0, 2000.0], [5.0, 0.0, 30.0], [2000.0, 30.0, 0.0]], dtype=F32)
    mp, mn = L.wms_masks(d, 0.8, 15.0, 'exp')
    off = ~np.eye(3, dtype=bool)
    np.testing.assert_allclose((mp + mn)[off], 1.0, rtol=0, atol=2e-7)
    # diagonal after "- eye": sigma(d_alpha*d_beta) - 1 = -6.144e-6 at (0.8, 15)
    diag = (mp - np.eye(3, dtype=F32))[0, 0]
    assert diag == pytest.approx(-6.144e-6, rel=2e-2)
    assert diag < 0
    # exp overflow on a far pair gives exactly 0 / 1
    assert mp[0, 2] == F32(0.0) and mn[0, 2] == F32(1.0)


def test_k7b_wms_lin_and_tanh_masks():
    d = np.array([[0.0, 7.5], [20.0, 15.0]], dtype=F32)
    mp, mn = L.wms_masks(d, 0.8, 15.0, 'lin')
    np.testing.assert_allclose(mp, [[1.0, 0.5], [0.0, 0.0]], atol=1e-7)
    np.testing.assert_allclose(mn, [[0.0, 0.5], [1.0, 1.0]], atol=1e-7)
    mp, mn = L.wms_masks(d, 0.8, 15.0, 'tanh')
    np.testing.assert_allclose(mp + mn, 1.0, atol=1e-7)
    assert mn[0, 1] == pytest.approx(math.tanh(0.5), rel=1e-6)


def test_k8_netvlad_zero_assignment_zero_centres():
    # W = 0 -> a = 1/K everywhere; C = 0 -> every cluster column is sum_n xhat / K;
    # after intra-norm each column is unit(sum_n xhat); after global norm / sqrt(K).
    rng = np.random.default_rng(3)
    b, n, d, k = 2, 5, 8, 4
    x = rng.standard_normal((b, n, d)).astype(F32)
    w = np.zeros((d, k), F32)
    c = np.zeros((d, k), F32)
    for fn in (NV.netvlad_literal, NV.netvlad_fused):
        out = fn(x, w, c).reshape(b, d, k)
        xs = L.l2_normalize(x, -1).sum(axis=1)
        unit = xs / np.linalg.norm(xs, axis=1, keepdims=True)
        want = np.repeat(unit[:, :, None], k, axis=2) / math.sqrt(k)
        np.testing.assert_allclose(out, want, rtol=2e-6, atol=2e-7)


def test_k8b_netvlad_centres_term():
    # W = 0, C != 0: V[d,k] = (sum_n xhat[n,d])/K + (N/K) C[d,k] before the norms.
    rng = np.random.default_rng(4)
    b, n, d, k = 1, 6, 4, 2
    x = rng.standard_normal((b, n, d)).astype(F32)
    c = rng.standard_normal((d, k)).astype(F32)
    _, a, v = NV.netvlad_fused(x, np.zeros((d, k), F32), c, return_aux=True)
    xs = L.l2_normalize(x, -1).sum(axis=1)[0]
    np.testing.assert_allclose(v[0], xs[:, None] / k + (n / k) * c, rtol=1e-5, atol=1e-6)


def test_netvlad_literal_equals_fused_and_is_unit_norm():
    rng = np.random.default_rng(5)
    x = rng.standard_normal((3, 20, 16)).astype(F32)
    w = (rng.standard_normal((16, 8)) * 0.3).astype(F32)
    c = (rng.standard_normal((16, 8)) * 0.1).astype(F32)
    a = NV.netvlad_literal(x, w, c)
    f = NV.netvlad_fused(x, w, c)
    np.testing.assert_allclose(a, f, rtol=2e-5, atol=2e-6)
    np.testing.assert_allclose(np.linalg.norm(a, axis=1), 1.0, rtol=1e-5)
    t = TT.netvlad(x, w, c).numpy()
    np.testing.assert_allclose(a, t, rtol=2e-5, atol=2e-6)


def _rand_batch(b=12, e=64, seed=0):
    rng = np.random.default_rng(seed)
    emb = rng.standard_normal((b, e)).astype(F32)
    xy = rng.uniform(0, 60, size=(b, 2))
    dist = np.linalg.norm(xy[:, None] - xy[None], axis=2).astype(F32)
    return emb, dist


def test_wms_rank3_equals_rank2_for_symmetric_inputs_and_scale_invariance():
    emb, dist = _rand_batch()
    a = L.wms_loss(dist, emb, 0.8, 15.0)
    b = L.wms_loss(dist[None], emb, 0.8, 15.0)
    assert a == pytest.approx(b, rel=2e-6)
    scaled = emb * np.linspace(0.5, 3.0, emb.shape[0], dtype=F32)[:, None]
    assert L.wms_loss(dist, scaled, 0.8, 15.0) == pytest.approx(a, rel=2e-5)


def test_wms_rank3_axis1_is_literal_for_asymmetric_distances():
    # With an asymmetric distance matrix the rank-3 call reduces over rows (axis=1 of
    # [1,B,B]); that equals the rank-2 call on the transposed matrix.
    emb, dist = _rand_batch(seed=2)
    dist = dist + np.triu(np.ones_like(dist), 1) * 9.0
    r3 = L.wms_loss(dist[None], emb, 0.8, 15.0)
    r2t = L.wms_loss(dist.T.copy(), emb, 0.8, 15.0)
    assert r3 == pytest.approx(r2t, rel=2e-6)


@pytest.mark.parametrize('wf', ['exp', 'lin', 'tanh'])
@pytest.mark.parametrize('sf', ['ms', 'plain'])
def test_wms_f32_oracle_matches_f64_twin(wf, sf):
    emb, dist = _rand_batch(b=16, e=128, seed=7)
    got = L.wms_loss(dist[None], emb, 0.8, 15.0, wfunction=wf, sumfunction=sf)
    want = float(TT.wms_loss(dist[None], emb, 0.8, 15.0, wfunction=wf, sumfunction=sf))
    assert got == pytest.approx(want, rel=1e-4, abs=1e-6)


def test_ms_permutation_invariance_and_twin():
    emb, _ = _rand_batch(b=10, e=32, seed=9)
    labels = L.trainer_ms_labels(2, 2, 2)
    base = L.ms_loss(labels, emb)
    perm = np.random.default_rng(1).permutation(10)
    assert L.ms_loss(labels[perm], emb[perm]) == pytest.approx(base, rel=1e-5)
    assert L.ms_det(labels, emb) == L.ms_loss(labels, emb, ms_mining=False)
    assert base == pytest.approx(float(TT.ms_loss(labels, emb)), rel=1e-4, abs=1e-7)


def test_trainer_ms_labels_layout():
    # train/train.py:822-826, T=2, P=2, N=3
    got = L.trainer_ms_labels(2, 2, 3)
    np.testing.assert_array_equal(got, [0, 0, 0, 1, 2, 3, 4, 4, 4, 5, 6, 7])


def test_logratio_literal_broadcast_matches_elementwise_definition():
    rng = np.random.default_rng(11)
    p = n = 4
    e = 16
    a = rng.standard_normal((1, 1, e)).astype(F32)
    pos = rng.standard_normal((1, p, e)).astype(F32)
    neg = rng.standard_normal((1, n, e)).astype(F32)
    spd = rng.uniform(1, 100, (1, p, 1)).astype(F32)
    snd = rng.uniform(200, 900, (1, n, 1)).astype(F32)
    got = L.logratio_loss(a, pos, neg, spd, snd)
    pr = ((a - pos) ** 2).sum(2)[0].astype(np.float64)
    nr = ((a - neg) ** 2).sum(2)[0].astype(np.float64)
    want = 0.0
    for i in range(n):          # out[0,i,j] = (log(pr[j]/nr[i]) - log(spd[i]/snd[i]))^2
        for j in range(p):
            want += (math.log(pr[j] / nr[i]) - math.log(spd[0, i, 0] / snd[0, i, 0])) ** 2
    want /= n * p
    assert got == pytest.approx(want, rel=1e-5)
    assert got == pytest.approx(float(TT.logratio_loss(a, pos, neg, spd, snd)), rel=1e-5)


def test_split_tuples_layout():
    out = np.arange(2 * 5 * 3, dtype=F32).reshape(10, 3)
    q, pos, neg = L.split_tuples(out, 2, [1, 2, 2])
    assert q.shape == (2, 1, 3) and pos.shape == (2, 2, 3) and neg.shape == (2, 2, 3)
    np.testing.assert_array_equal(q[1, 0], out[5])
    np.testing.assert_array_equal(neg[0, 1], out[4])


def test_topn_bruteforce_equals_reference_kdtree_call():
    rng = np.random.default_rng(12)
    ref = rng.standard_normal((500, 16)).astype(F32)
    qry = rng.standard_normal((40, 16)).astype(F32)
    d0, i0 = TN.topn_bruteforce(ref, qry, 7)
    d1, i1 = TN.topn_kdtree(ref, qry, 7)
    np.testing.assert_array_equal(i0, i1)
    np.testing.assert_allclose(d0, d1, rtol=1e-12)
    g = rng.uniform(0, 30, size=(40, 7))
    r = TN.recall_at_threshold(g, [5.0, 10.0, 1e9], n=1)
    assert r[2] == 1.0 and r[0] <= r[1]


def test_tuple_losses_twin_agreement_random():
    rng = np.random.default_rng(13)
    t, p, n, e = 3, 4, 5, 24
    q = rng.standard_normal((t, 1, e)).astype(F32) * 0.2
    pos = rng.standard_normal((t, p, e)).astype(F32) * 0.2
    neg = rng.standard_normal((t, n, e)).astype(F32) * 0.2
    oth = rng.standard_normal((t, 1, e)).astype(F32) * 0.2
    for name in ('triplet_loss', 'lazy_triplet_loss', 'evil_triplet_loss'):
        got = getattr(L, name)(q, pos, neg, 0.5)
        assert got == pytest.approx(float(getattr(TT, name)(q, pos, neg, 0.5)), rel=1e-5)
    for name in ('quadruplet_loss', 'lazy_quadruplet_loss', 'evil_quadruplet_loss'):
        got = getattr(L, name)(q, pos, neg, oth, 0.5, 0.2)
        assert got == pytest.approx(float(getattr(TT, name)(q, pos, neg, oth, 0.5, 0.2)), rel=1e-5)


def test_k7c_tanh_is_the_eigen_rational_approximation():
    """tf.tanh on float32 CPU tensors (model/losses.py:14-16) is Eigen's clamp-to-9 rational
    approximation, not libm: accurate to ~1e-7 in the bulk, exactly 1.0 from 9 upwards, and
    NOT monotone in the last bit just below — which is what decides ``mask_pos > 0``."""
    x = np.array([0.0, 0.1, 0.5, 1.0, 2.0, 4.0, 6.0, -0.5, -3.0], np.float32)
    got = L.eigen_fast_tanh_f32(x)
    assert got.dtype == np.float32
    np.testing.assert_allclose(got, np.tanh(x.astype(np.float64)), atol=2e-7)
    sat = L.eigen_fast_tanh_f32(np.array([8.5, 8.7, 9.0, 9.5, 19.0, -30.0], np.float32))
    assert sat.tolist() == [1.0, np.float32(0.99999994), 1.0, 1.0, 1.0, -1.0]
    # masks: far pairs drop out exactly, a pair at 8.7 d_beta stays in with weight 2^-24
    d = np.array([[0.0, 8.7 * 15.0], [9.5 * 15.0, 300.0]], np.float32)
    mp, mn = L.wms_masks(d, 0.8, 15.0, 'tanh')
    assert mp[0, 1] == np.float32(2.0 ** -24) and mp[1, 0] == 0.0 and mp[1, 1] == 0.0
    assert mn[1, 1] == 1.0 and mp[0, 0] == 1.0


def test_pairwise_distance_loss_kat():
    """model/losses.py:627-646 by pencil on the reference's own K1 matrices
    (model/losses.py:708-711): tuple 0 = rows (1,1),(2,2),(3,3) -> squared feature distances
    [[0,2,8],[2,0,2],[8,2,0]].  With f_max = 2, d_max = 4 and squared geographic distances
    [[0,4,16],[4,0,4],[16,4,0]]: scaled f = [[0,1,4],[1,0,1],[4,1,0]], scaled d the same ->
    loss 0.  With d = 0 everywhere: squared differences sum 2 * (1 + 16 + 1) = 36 over 9
    entries -> 4; Huber (delta 1): 2 * (0.5 + 3.5 + 0.5) = 9 over 9 entries -> 1."""
    from oracle import losses_np as O
    f = np.array([[[1.0, 1], [2, 2], [3, 3]]], np.float32)
    d_same = np.array([[[0, 4, 16], [4, 0, 4], [16, 4, 0]]], np.float32)
    assert float(O.pairwise_distance_loss(f[:, :1], f[:, 1:], d_same, 4.0, 2.0)) == 0.0
    zero = np.zeros((1, 3, 3), np.float32)
    assert float(O.pairwise_distance_loss(f[:, :1], f[:, 1:], zero, 4.0, 2.0)) == 4.0
    assert float(O.pairwise_distance_loss(f[:, :1], f[:, 1:], zero, 4.0, 2.0,
                                          'huber_distance_loss')) == 1.0
