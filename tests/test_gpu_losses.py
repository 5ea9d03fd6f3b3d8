"""GPU parity: HIP loss kernels (through the C-ABI) vs the CPU oracle.

Tolerance: BASELINE.json's north_star asks for loss values within 1e-4 relative fp32 of
the CPU reference; gradients are compared to the float64 autograd twin with a
norm-relative bound of 2e-4 (fp32 Gram of 32768-term dot products feeding exp(50 x)).
"""
import numpy as np
import pytest
import torch

from oracle import losses_np as O
from oracle import twin_torch as TT
from tests import util_data as U

pytestmark = pytest.mark.gpu

REL = 1e-4
GRAD_REL = 2e-4


@pytest.fixture(scope="module")
def dev():
    assert torch.cuda.is_available(), "GPU tests need a HIP device"
    return torch.device("cuda:0")


def _rel(a, b):
    a = np.asarray(a, dtype=np.float64)
    b = np.asarray(b, dtype=np.float64)
    return float(np.linalg.norm(a - b) / max(np.linalg.norm(b), 1e-30))


def _close(got, want, rel=REL, abs_=1e-7):
    assert abs(float(got) - float(want)) <= rel * abs(float(want)) + abs_, (float(got), float(want))


def _twin_grad(fn, emb_np):
    e = torch.tensor(emb_np, dtype=torch.float64, requires_grad=True)
    loss = fn(e)
    loss.backward()
    return float(loss), e.grad.numpy()


# ------------------------------------------------------------------ KATs on the GPU
def test_kat_ms_two_rows(dev):
    from soft_contrastive_learning_amd.model import losses as M
    e = torch.tensor([[1.0, 0.0], [1.0, 0.0]], device=dev)
    _close(M.ms_loss([7, 7], e, ms_mining=False), np.log(2.0) / 2.0)      # K4
    assert float(M.ms_loss([7, 7], e)) == 0.0                              # K4b
    _close(M.ms_loss([0, 1], e), np.log(2.0) / 50.0)                       # K5
    eo = torch.tensor([[1.0, 0.0], [0.0, 1.0]], device=dev)
    assert float(M.ms_loss([0, 1], eo)) == 0.0                             # K6


def test_kat_pairwise_sqdist(dev):
    from soft_contrastive_learning_amd.model import losses as M
    c = torch.tensor([[[1.0, 1], [2, 2], [3, 3]], [[1, 1], [2, 2], [4, 4]]], device=dev)
    want = np.array([[[0, 2, 8], [2, 0, 2], [8, 2, 0]], [[0, 2, 18], [2, 0, 8], [18, 8, 0]]],
                    dtype=np.float32)
    np.testing.assert_array_equal(M._pairwise_squared_distances(c).cpu().numpy(), want)


def test_kat_tuple_losses(dev):
    from soft_contrastive_learning_amd import pointnetvlad_cls as P
    from soft_contrastive_learning_amd.model import losses as M
    q = torch.tensor([[[0.0, 0.0]]], device=dev)
    pos = torch.tensor([[[1.0, 0.0], [0.0, 2.0]]], device=dev)
    neg = torch.tensor([[[3.0, 0.0], [0.0, 1.0]]], device=dev)
    oth = torch.tensor([[[3.0, 1.0]]], device=dev)
    assert float(P.triplet_loss(q, pos, neg, 0.5)) == 0.5                  # K2
    assert float(P.lazy_triplet_loss(q, pos, neg, 0.5)) == 0.5
    assert float(M.evil_triplet_loss(q, pos, neg, 0.5)) == 3.5
    _close(P.quadruplet_loss(q, pos, neg, oth, 0.5, 0.2), 0.7, rel=1e-6)   # K3
    _close(P.lazy_quadruplet_loss(q, pos, neg, oth, 0.5, 0.2), 0.7, rel=1e-6)
    _close(M.evil_quadruplet_loss(q, pos, neg, oth, 0.5, 0.2), 6.7, rel=1e-6)
    assert float(P.best_pos_distance(q, pos)[0]) == 1.0
    assert float(M.worst_pos_distance(q, pos)[0]) == 4.0


# --------------------------------------------------------------- wms / ms parity sweep
@pytest.mark.parametrize("b,e", [(24, 32768), (25, 32768), (7, 100), (64, 4096), (192, 32768)])
@pytest.mark.parametrize("wf,sf", [("exp", "ms"), ("lin", "ms"), ("tanh", "plain"), ("exp", "plain"),
                                   ("tanh", "ms")])
def test_wms_loss_and_grad(dev, b, e, wf, sf):
    # (192, tanh, ms) at 200 m is the case where the last bit of tanh near saturation decides
    # pair membership: oracle and kernel evaluate the same rational approximation op for op
    from soft_contrastive_learning_amd.model import losses as M
    emb = U.embeddings(b, e)
    dist = U.positions_distances(b, side=60.0 if b < 64 else 200.0)
    want = O.wms_loss(dist[None], emb, 0.8, 15.0, wfunction=wf, sumfunction=sf)
    want64, grad64 = _twin_grad(
        lambda t: TT.wms_loss(dist[None], t, 0.8, 15.0, wfunction=wf, sumfunction=sf), emb)
    et = torch.tensor(emb, device=dev, requires_grad=True)
    dt = torch.tensor(dist[None], device=dev)
    loss = M.wms_loss(dt, et, d_alpha=0.8, d_beta=15.0, wfunction=wf, sumfunction=sf)
    loss.backward()
    _close(loss, want)
    _close(loss, want64)
    assert _rel(et.grad.cpu().numpy(), grad64) < GRAD_REL
    # rank-2 distances give the same value for symmetric inputs
    _close(M.wms_loss(dt[0], et.detach(), 0.8, 15.0, wfunction=wf, sumfunction=sf), want)


def test_wms_rank3_asymmetric_is_literal(dev):
    from soft_contrastive_learning_amd.model import losses as M
    emb = U.embeddings(12, 256, seed=3, mix=3.0)
    dist = U.positions_distances(12, side=20.0)
    dist = dist + np.triu(np.ones_like(dist), 1) * 9.0
    want3, want2 = O.wms_loss(dist[None], emb, 0.8, 15.0), O.wms_loss(dist, emb, 0.8, 15.0)
    assert abs(float(want3) - float(want2)) > 1e-3     # the two readings really differ here
    et = torch.tensor(emb, device=dev)
    got3 = M.wms_loss(torch.tensor(dist[None], device=dev), et, 0.8, 15.0)
    got2 = M.wms_loss(torch.tensor(dist, device=dev), et, 0.8, 15.0)
    _close(got3, want3)
    _close(got2, want2)


def test_wms_no_mining_and_upstream_grad_scale(dev):
    from soft_contrastive_learning_amd.model import losses as M
    emb = U.embeddings(24, 2048, seed=4)
    dist = U.positions_distances(24, side=80.0)
    want64, grad64 = _twin_grad(lambda t: 3.0 * TT.wms_loss(dist, t, 0.8, 15.0, ms_mining=False), emb)
    et = torch.tensor(emb, device=dev, requires_grad=True)
    loss = 3.0 * M.wms_loss(torch.tensor(dist, device=dev), et, 0.8, 15.0, ms_mining=False)
    loss.backward()
    _close(loss, want64)
    assert _rel(et.grad.cpu().numpy(), grad64) < GRAD_REL


@pytest.mark.parametrize("t,p,n", [(1, 12, 12), (2, 2, 3), (8, 12, 12)])
@pytest.mark.parametrize("mining", [True, False])
def test_ms_loss_and_grad(dev, t, p, n, mining):
    from soft_contrastive_learning_amd.model import losses as M
    b = t * (1 + p + n)
    emb = U.embeddings(b, 32768 if t == 1 else 1024, seed=17)
    labels = O.trainer_ms_labels(t, p, n)
    want = O.ms_loss(labels, emb, ms_mining=mining)
    want64, grad64 = _twin_grad(lambda x: TT.ms_loss(labels, x, ms_mining=mining), emb)
    et = torch.tensor(emb, device=dev, requires_grad=True)
    loss = M.ms_loss(labels, et, ms_mining=mining)
    loss.backward()
    _close(loss, want)
    _close(loss, want64)
    assert _rel(et.grad.cpu().numpy(), grad64) < GRAD_REL
    _close(M.ms_det(labels, et.detach()), O.ms_det(labels, emb))


def test_wms_data_parallel_rows_match_full_gradient(dev):
    from soft_contrastive_learning_amd.model import losses as M
    emb = U.embeddings(48, 4096, seed=8)
    dist = U.positions_distances(48)
    dt = torch.tensor(dist[None], device=dev)
    full = torch.tensor(emb, device=dev, requires_grad=True)
    M.wms_loss(dt, full, 0.8, 15.0).backward()
    part = torch.tensor(emb, device=dev, requires_grad=True)
    M.wms_loss(dt, part, 0.8, 15.0, _rows=(24, 24)).backward()
    np.testing.assert_array_equal(part.grad[24:].cpu().numpy(), full.grad[24:].cpu().numpy())
    assert float(part.grad[:24].abs().max()) == 0.0


def test_wms_bitwise_reproducible(dev):
    from soft_contrastive_learning_amd.model import losses as M
    emb = torch.tensor(U.embeddings(24, 32768), device=dev)
    dt = torch.tensor(U.positions_distances(24)[None], device=dev)
    vals = {float(M.wms_loss(dt, emb, 0.8, 15.0)) for _ in range(5)}
    assert len(vals) == 1


# ------------------------------------------------------------------- tuple losses
TUPLE_CASES = [("triplet_loss", False), ("lazy_triplet_loss", False), ("evil_triplet_loss", False),
               ("quadruplet_loss", True), ("lazy_quadruplet_loss", True),
               ("evil_quadruplet_loss", True)]


@pytest.mark.parametrize("name,quad", TUPLE_CASES)
@pytest.mark.parametrize("t,p,n,e", [(1, 12, 12, 32768), (2, 1, 2, 32768), (3, 4, 5, 100)])
def test_tuple_losses_and_grads(dev, name, quad, t, p, n, e):
    from soft_contrastive_learning_amd import pointnetvlad_cls as P
    from soft_contrastive_learning_amd.model import losses as M
    fn = getattr(P, name, None) or getattr(M, name)
    shape = [1, p, n] + ([1] if quad else [])
    out = U.tuple_batch(t, p, n, e, quad=quad)
    flat = out.reshape(t * sum(shape), e)
    parts = O.split_tuples(flat, t, shape)
    margins = (0.5, 0.2) if quad else (0.5,)
    want = getattr(O, name)(*parts, *margins)

    x64 = torch.tensor(flat, dtype=torch.float64, requires_grad=True)
    parts64 = torch.split(x64.reshape(t, sum(shape), e), shape, dim=1)
    l64 = getattr(TT, name)(*parts64, *margins)
    l64.backward()

    xt = torch.tensor(flat, device=dev, requires_grad=True)
    # the trainer's glue: reshape + split (train/train.py:654)
    parts_t = torch.split(xt.reshape(t, sum(shape), e), shape, dim=1)
    loss = fn(*parts_t, *margins)
    loss.backward()
    _close(loss, want)
    _close(loss, float(l64))
    assert _rel(xt.grad.cpu().numpy(), x64.grad.numpy()) < GRAD_REL


@pytest.mark.parametrize('huber', [True, False])
@pytest.mark.parametrize('lazy', [False, True])
@pytest.mark.parametrize('quad,t,p,n,e', [(False, 1, 12, 12, 32768), (True, 1, 12, 11, 32768),
                                          (False, 3, 4, 5, 300), (True, 4, 2, 6, 257)])
def test_distance_tuple_losses_and_grads(dev, huber, lazy, quad, t, p, n, e):
    """[huber_]distance_[lazy_]{triplet,quadruplet} (model/losses.py:239-307)."""
    from soft_contrastive_learning_amd.model import losses as M
    shape = [1, p, n] + ([1] if quad else [])
    out = U.tuple_batch(t, p, n, e, quad=quad, seed=50 + p)
    rng = np.random.default_rng(51 + t)
    # scaled geographic distances straddle the scaled feature distances so that both Huber
    # branches and both signs occur
    dd = rng.uniform(0.0, 15.0 ** 2, (t, p)).astype(np.float32)
    dmax, fmax, lam, m1, m2 = 15.0 ** 2, 2.0, 0.5, 0.5, 0.2
    if not huber:
        fmax = 0.05                      # |err| > 1 somewhere: the squared term keeps growing
    flat = out.reshape(t * sum(shape), e)
    parts = O.split_tuples(flat, t, shape)
    tl = 'lazy_triplet_loss' if lazy else 'triplet_loss'
    dl = 'huber_distance_loss' if huber else 'distance_loss'
    if quad:
        want = O.distance_quadruplet_loss(parts[0], parts[1], parts[2], parts[3], m1, m2, lam, dd,
                                          dmax, fmax, tl, dl)
    else:
        want = O.distance_triplet_loss(parts[0], parts[1], parts[2], m1, lam, dd, dmax, fmax, tl, dl)
    x64 = torch.tensor(flat, dtype=torch.float64, requires_grad=True)
    p64 = torch.split(x64.reshape(t, sum(shape), e), shape, dim=1)
    l64 = TT.distance_tuple_loss(p64[0], p64[1], p64[2], p64[3] if quad else None, m1, m2, lam, dd,
                                 dmax, fmax, lazy=lazy, huber=huber)
    l64.backward()
    xt = torch.tensor(flat, device=dev, requires_grad=True)
    pt = torch.split(xt.reshape(t, sum(shape), e), shape, dim=1)
    ddt = torch.tensor(dd, device=dev)
    if quad:
        loss = M.distance_quadruplet_loss(pt[0], pt[1], pt[2], pt[3], m1, m2, lam, ddt, dmax, fmax,
                                          tl, dl)
    else:
        loss = M.distance_triplet_loss(pt[0], pt[1], pt[2], m1, lam, ddt, dmax, fmax, tl, dl)
    loss.backward()
    _close(loss, want)
    _close(loss, float(l64))
    assert _rel(xt.grad.cpu().numpy(), x64.grad.numpy()) < GRAD_REL


def test_distance_terms_alone_and_their_kat(dev):
    from soft_contrastive_learning_amd.model import losses as M
    # hand-derivable: E=2, a=(0,0), pos={(1,0),(0,2)}: sqd = (1,4); f_max=2 -> sf=(0.5,2);
    # d=(50,25), d_max=100 -> sd=(0.5,0.25); err=(0,1.75)
    a = np.zeros((1, 1, 2), np.float32)
    pos = np.array([[[1.0, 0.0], [0.0, 2.0]]], np.float32)
    dd = np.array([[50.0, 25.0]], np.float32)
    want_sq = (0.0 + 1.75 ** 2) / 2
    want_hub = (0.0 + (1.75 - 0.5)) / 2
    assert np.isclose(O.distance_loss(a, pos, dd, 100.0, 2.0), want_sq)
    assert np.isclose(O.huber_distance_loss(a, pos, dd, 100.0, 2.0), want_hub)
    at = torch.tensor(a, device=dev, requires_grad=True)
    pt = torch.tensor(pos, device=dev, requires_grad=True)
    ddt = torch.tensor(dd, device=dev)
    got_sq = M.distance_loss(at, pt, ddt, 100.0, 2.0)
    got_hub = M.huber_distance_loss(at, pt, ddt, 100.0, 2.0)
    _close(got_sq, want_sq)
    _close(got_hub, want_hub)
    got_hub.backward()
    # d/d pos_2 = clamp(err,-1,1)/(P f_max) * 2 (pos_2 - a) = 1/4 * (0,4) = (0,1); pos_1: err=0
    np.testing.assert_allclose(pt.grad.cpu().numpy(), [[[0.0, 0.0], [0.0, 1.0]]], atol=1e-6)
    np.testing.assert_allclose(at.grad.cpu().numpy(), [[[0.0, -1.0]]], atol=1e-6)
    with pytest.raises(AttributeError):
        M.distance_triplet_loss(at, pt, pt, 0.1, 0.5, ddt, 100.0, 2.0, 'evil_triplet_loss')
    with pytest.raises(ValueError):
        M.distance_triplet_loss(at, pt, pt, 0.1, 0.5, ddt[:, :1], 100.0, 2.0)


def test_logratio_loss_and_grad(dev):
    from soft_contrastive_learning_amd.model import losses as M
    p = n = 12
    e = 32768
    out = U.tuple_batch(1, p, n, e, seed=33)
    rng = np.random.default_rng(34)
    spd = rng.uniform(1, 200, (1, p, 1)).astype(np.float32)
    snd = rng.uniform(300, 4000, (1, n, 1)).astype(np.float32)
    a, pos, neg = O.split_tuples(out.reshape(-1, e), 1, [1, p, n])
    want = O.logratio_loss(a, pos, neg, spd, snd)
    x64 = torch.tensor(out, dtype=torch.float64, requires_grad=True)
    a64, p64, n64 = torch.split(x64, [1, p, n], dim=1)
    l64 = TT.logratio_loss(a64, p64, n64, spd, snd)
    l64.backward()
    xt = torch.tensor(out, device=dev, requires_grad=True)
    at, pt, nt = torch.split(xt, [1, p, n], dim=1)
    loss = M.logratio_loss(at, pt, nt, torch.tensor(spd, device=dev), torch.tensor(snd, device=dev))
    loss.backward()
    _close(loss, want)
    _close(loss, float(l64))
    assert _rel(xt.grad.cpu().numpy(), x64.grad.numpy()) < GRAD_REL


def test_pairwise_sqdist_random(dev):
    from soft_contrastive_learning_amd.model import losses as M
    f = U.tuple_batch(3, 6, 6, 4096, seed=40, scale=1.0)
    want = O.pairwise_squared_distances(f)
    got = M._pairwise_squared_distances(torch.tensor(f, device=dev)).cpu().numpy()
    np.testing.assert_allclose(got, want, rtol=1e-4, atol=1e-3 * float(np.abs(want).max()) * 1e-2)


# ------------------------------------------------------------------ error behaviour
def test_shape_errors_raise_like_the_reference(dev):
    from soft_contrastive_learning_amd.model import losses as M
    emb = torch.zeros(8, 64, device=dev)
    with pytest.raises(ValueError):
        M.wms_loss(torch.zeros(2, 4, 4, device=dev), emb, 0.8, 15.0)      # T > 1 rank-3
    with pytest.raises(ValueError):
        M.wms_loss(torch.zeros(7, 7, device=dev), emb, 0.8, 15.0)
    with pytest.raises(ValueError):
        M.ms_loss([0, 1, 2], emb)
    with pytest.raises(ValueError):
        M.logratio_loss(torch.zeros(1, 1, 8, device=dev), torch.zeros(1, 3, 8, device=dev),
                        torch.zeros(1, 4, 8, device=dev), torch.ones(1, 3, 1, device=dev),
                        torch.ones(1, 4, 1, device=dev))


@pytest.mark.parametrize("huber", [False, True])
@pytest.mark.parametrize("t,p,e", [(1, 12, 32768), (3, 5, 4096)])
def test_pairwise_distance_loss_and_grad(dev, t, p, e, huber):
    """model/losses.py:627-646 — the caller of _pairwise_squared_distances: loss vs the float32
    oracle (1e-4), gradients vs the float64 twin (2e-4 norm-relative)."""
    from soft_contrastive_learning_amd.model import losses
    tb = U.tuple_batch(t, p, 0, e, seed=50 + p, scale=0.02)
    a, pos = tb[:, :1], tb[:, 1:]
    rng = np.random.default_rng(p)
    xy = rng.uniform(0, 25, (t, 1 + p, 2))
    dd = ((xy[:, :, None] - xy[:, None]) ** 2).sum(-1).astype(np.float32)   # squared metres
    name = 'huber_distance_loss' if huber else 'distance_loss'
    want = float(O.pairwise_distance_loss(a, pos, dd, 225.0, 2.0, name))
    at = torch.tensor(a, device=dev, requires_grad=True)
    pt = torch.tensor(pos, device=dev, requires_grad=True)
    got = losses.pairwise_distance_loss(at, pt, torch.tensor(dd, device=dev), 225.0, 2.0, name)
    assert abs(float(got) - want) <= 1e-4 * abs(want) + 1e-8, (float(got), want)
    got.backward()
    a64 = torch.tensor(a, dtype=torch.float64, requires_grad=True)
    p64 = torch.tensor(pos, dtype=torch.float64, requires_grad=True)
    TT.pairwise_distance_loss(a64, p64, dd, 225.0, 2.0, huber=huber).backward()
    assert _rel(at.grad.cpu().numpy(), a64.grad.numpy()) < 2e-4
    assert _rel(pt.grad.cpu().numpy(), p64.grad.numpy()) < 2e-4


def test_fused_finish_with_a_dirty_sync_word_is_loud_until_the_host_clears_it(dev):
    """ADVICE rounds 4 and 5: the one-launch forward relies on the caller's sync word being zero on
    entry.  A word left non-zero (a contract violation) makes a middle workgroup take itself for the
    last: the call cannot be repaired and must not pass for sound.  The workgroups that draw a ticket
    past the grid size set a STICKY error word: this call and every later one on the block return
    NaN until the host zeroes the block — the library does not pretend to heal it."""
    from soft_contrastive_learning_amd import _lib as L
    from soft_contrastive_learning_amd.model import losses
    b, e = 24, 32768
    emb = torch.tensor(U.embeddings(b, e), device=dev)
    dm = torch.tensor(U.positions_distances(b)[None], device=dev)
    good = float(losses.wms_loss(dm, emb, 0.8, 15.0))
    word = L.sync_words(dev)
    assert int(word.view(torch.int32)[0]) == 0                # left zero by the call above
    word.view(torch.int32)[0] = 3                             # the violation
    losses.wms_loss(dm, emb, 0.8, 15.0)
    torch.cuda.synchronize()
    assert int(word.view(torch.int32)[2]) == 1                # recorded for good
    for _ in range(3):                                        # ... and loud from here on
        assert np.isnan(float(losses.wms_loss(dm, emb, 0.8, 15.0)))
    word.zero_()                                              # the host's repair
    assert float(losses.wms_loss(dm, emb, 0.8, 15.0)) == good
    assert int(word.view(torch.int32).abs().sum()) == 0


# ---- round 4: the B <= 32 forward in one launch ---------------------------------------------------
@pytest.mark.parametrize("b,e", [(2, 64), (7, 200), (16, 4096), (24, 32768), (25, 32768), (32, 32768),
                                 (33, 32768), (40, 4096), (48, 32768), (49, 32768), (56, 200), (64, 32768)])
@pytest.mark.parametrize("kind", ["wms_exp", "wms_tanh_plain", "ms"])
def test_fused_finish_gives_the_bits_of_the_two_launch_forward(dev, b, e, kind):
    """scl_gram_loss_fwd_s with the stream's sync block: the Gram kernel's last workgroup runs the
    finish (same sums, same order).  scl_debug_set_variant(32) = the two-launch forward: loss and
    d loss / d embeddings must be bit-identical, and the sync block must be zero afterwards.
    Round 5: 32 < B <= 64 as well (final64_body against gram_reduce + gram_rows_wave + gram_coef;
    there the one-launch form is variant 36 of the diagnostic build — measured slower than the four
    launches, so not the product path — and the product path is the comparison partner)."""
    from soft_contrastive_learning_amd import _lib as L
    from soft_contrastive_learning_amd.model import losses
    emb = torch.tensor(U.embeddings(b, e), device=dev)
    dm = torch.tensor(U.positions_distances(b)[None], device=dev)
    lab = torch.tensor(np.arange(b) // 3, device=dev)

    def run(variant):
        x = emb.clone().requires_grad_(True)
        with L.maybe_variant(variant):        # 0: the product library; 32: the diagnostic build
            if kind == "wms_exp":
                loss = losses.wms_loss(dm, x, 0.8, 15.0)
            elif kind == "wms_tanh_plain":
                loss = losses.wms_loss(dm, x, 0.8, 15.0, wfunction='tanh', sumfunction='plain')
            else:
                loss = losses.ms_loss(lab, x)
            loss.backward()
            torch.cuda.synchronize()
        return loss.detach().cpu().numpy(), x.grad.cpu().numpy()
    l1, g1 = run(0)
    l2, g2 = run(32 if b <= 32 else 36)
    assert l1.view(np.uint32) == l2.view(np.uint32)
    assert np.array_equal(g1.view(np.uint32), g2.view(np.uint32))
    assert int(L.sync_words(dev).sum()) == 0
    for _ in range(20):                                       # the counter keeps returning to zero
        l3, _ = run(0)
        assert l3.view(np.uint32) == l1.view(np.uint32)


# ---- round 6: the strip-scheduled Gram kernel; the one-launch persistent forward (diagnostic) ------
@pytest.mark.parametrize("b,e", [(33, 32768), (48, 32768), (64, 32768), (64, 4096), (65, 32768), (100, 4096),
                                 (130, 32768), (160, 2048), (192, 32768), (192, 8192), (200, 32768),
                                 (208, 32768), (209, 32768), (256, 32768)])
@pytest.mark.parametrize("kind", ["wms_exp", "wms_tanh_plain", "ms"])
def test_forward_above_32_gives_the_bits_of_round_5_in_every_form(dev, b, e, kind):
    """Three forms of the forward for 32 < B <= 256, loss and d loss / d embeddings bit-identical:
      0   the product path: 64 < B <= 208 the strip-scheduled Gram kernel (gram16x6p_kernel: two tile
          rows per strip, second pass finishing pair by pair) + the three finishing launches;
      37  round 5's four launches (gram16x6_kernel / gram16_kernel in front);
      41  ONE persistent kernel (the Gram, then slab sums / rows / M behind grid barriers; B <= 208) —
          built, measured slower, kept in the diagnostic build (profiles/r06) — and 39, the same with
          no patience at the barriers: the run aborts and the last workgroup out redoes the phases
          alone (the repair path that makes the spinning kernel safe without co-residency).
    The sync block is zero after every form."""
    from soft_contrastive_learning_amd import _lib as L
    from soft_contrastive_learning_amd.model import losses
    emb = torch.tensor(U.embeddings(b, e), device=dev)
    dm = torch.tensor(U.positions_distances(b)[None], device=dev)
    lab = torch.tensor(np.arange(b) // 3, device=dev)

    def run(variant):
        x = emb.clone().requires_grad_(True)
        with L.maybe_variant(variant):
            with L.KernelTimer(capacity=64) as kt:
                if kind == "wms_exp":
                    loss = losses.wms_loss(dm, x, 0.8, 15.0)
                elif kind == "wms_tanh_plain":
                    loss = losses.wms_loss(dm, x, 0.8, 15.0, wfunction='tanh', sumfunction='plain')
                else:
                    loss = losses.ms_loss(lab, x)
                torch.cuda.synchronize()
            names = set(kt.summary())
            loss.backward()
            torch.cuda.synchronize()
            assert int(L.sync_words(dev).sum()) == 0
        return loss.detach().cpu().numpy(), x.grad.cpu().numpy(), names
    l0, g0, n0 = run(0)
    l1, g1, n1 = run(37)
    persistent = {'gram16x6_persist_kernel', 'gram16_persist_kernel'}
    assert ('gram16x6p_kernel' in n0) == (64 < b <= 208 and e % 128 == 0), n0
    assert 'gram16x6p_kernel' not in n1 and 'gram_reduce_kernel' in n1 and not ((n0 | n1) & persistent)
    assert np.isfinite(l0) and l0.view(np.uint32) == l1.view(np.uint32), (l0, l1)
    assert np.array_equal(g0.view(np.uint32), g1.view(np.uint32))
    if b <= 208:
        for v in (41, 39):
            l2, g2, n2 = run(v)
            assert n2 <= persistent and len(n2) == 1, (v, n2)
            assert l2.view(np.uint32) == l0.view(np.uint32), v
            assert np.array_equal(g2.view(np.uint32), g0.view(np.uint32)), v


def test_two_streams_of_persistent_forwards_do_not_deadlock(dev):
    """(Diagnostic variant 41.)  Two threads, each on its own stream with its own sync block, launch
    the persistent B = 192 forward back to back: 2 x 256 workgroups of 145 KB LDS cannot all be
    resident, so whenever the two grids split the chip between them both spin — the bounded spin +
    repair path must bring every call home with the bits of an undisturbed call."""
    import threading
    from soft_contrastive_learning_amd import _lib as L
    from soft_contrastive_learning_amd.model import losses
    b, e, reps = 192, 32768, 150
    emb = torch.tensor(U.embeddings(b, e), device=dev)
    dm = torch.tensor(U.positions_distances(b)[None], device=dev)
    want = losses.wms_loss(dm, emb, 0.8, 15.0).cpu().numpy()
    torch.cuda.synchronize()
    out, errs = {}, []

    def work(k):
        try:
            st = torch.cuda.Stream(device=dev)
            with torch.cuda.stream(st):
                vals = [losses.wms_loss(dm, emb, 0.8, 15.0) for _ in range(reps)]
                st.synchronize()
            out[k] = torch.stack(vals).cpu().numpy()
        except Exception as ex:                                # noqa: BLE001 (reported below)
            errs.append(ex)
    with L.variant(41):
        ths = [threading.Thread(target=work, args=(k,)) for k in range(2)]
        for t in ths:
            t.start()
        for t in ths:
            t.join(timeout=300)
        assert not any(t.is_alive() for t in ths), 'a stream of persistent forwards did not finish'
    assert not errs, errs
    for k in range(2):
        assert np.array_equal(out[k].view(np.uint32), np.broadcast_to(want.view(np.uint32), (reps,))), k


@pytest.mark.gpu
@pytest.mark.parametrize('b,row_begin,row_count,e', [(192, 0, 192, 4096), (96, 0, 96, 2048), (200, 40, 100, 1024),
                                                    (256, 0, 200, 1024), (256, 0, 256, 512), (130, 7, 33, 640), (192, 0, 192, 448), (160, 16, 130, 256)])
def test_gram_backward_on_bf16_planes(dev, b, row_begin, row_count, e):
    """scl_gram_loss_bwd_w: grad = g M E for many rows on three bf16 planes per operand (six
    products) against float64, and against the float32-MFMA kernels it replaces (variant 34) —
    float32-equivalent: both within 2e-6 of the float64 result relative to its largest entry."""
    from soft_contrastive_learning_amd import _lib as L
    lib = L.load()
    g = torch.Generator().manual_seed(b * 1000 + e)
    emb = torch.randn(b, e, generator=g).to(dev)
    coef = (torch.randn(b, b, generator=g) * torch.rand(b, b, generator=g).pow(4)).to(dev)
    gl = torch.tensor([0.37], device=dev)
    want = 0.37 * (coef.double()[row_begin:row_begin + row_count] @ emb.double())
    ws = L.workspace(lib.scl_gram_loss_bwd_workspace_bytes(b, row_count), dev)
    outs = {}
    for variant in (0, 34):
        out = torch.full((row_count, e), 7.0, device=dev)
        with L.maybe_variant(variant) as vlib:
            with L.KernelTimer(capacity=16) as kt:
                L.check(vlib.scl_gram_loss_bwd_w(L.ptr(emb), e, b, e, L.ptr(coef), L.ptr(gl), row_begin,
                                                 row_count, L.ptr(out), e, L.ptr(ws), ws.numel(),
                                                 L.stream_of(emb)))
                torch.cuda.synchronize()
        outs[variant] = (out, set(kt.summary()))
        err = float((out.double() - want).abs().max() / want.abs().max())
        assert err < 2e-6, (variant, err)
    planes = 'gram_bwd_planes_kernel' in outs[0][1]
    assert planes == (64 < b <= 256 and b % 4 == 0 and row_count >= 96 and e % 128 == 0), outs[0][1]
    assert 'gram_bwd_planes_kernel' not in outs[34][1]
