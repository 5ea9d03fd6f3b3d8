"""GPU parity of exactly what bench.py runs (BASELINE.json configs[1]): the bf16x3 NetVLAD
kernels at (24, 1200) with all three gradients, one whole 24 x 640x480 bf16 train step against
the float32-mode HIP step and against the CPU oracle on the HIP conv5_3 map, and the
`vgg16` (no-VLAD) embedder.

Tolerances (north_star): 1e-4 relative on descriptors and loss values for the same inputs;
2e-4 norm-relative on float32 gradients against the float64 autograd twin; quantities that
are STORED in bf16 (grad_x, anything downstream of the bf16 backbone) carry the bf16 bounds
written at the assertion.
"""
import numpy as np
import pytest
import torch

from oracle import losses_np as O
from oracle import netvlad_np as NV
from oracle import twin_torch as TT
from tests import util_data as U

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def dev():
    assert torch.cuda.is_available()
    return torch.device("cuda:0")


def _maxrel(got, want):
    got = np.asarray(got, np.float64)
    want = np.asarray(want, np.float64)
    return float(np.abs(got - want).max() / np.abs(want).max())


def _nrel(got, want):
    got = np.asarray(got, np.float64)
    want = np.asarray(want, np.float64)
    return float(np.linalg.norm(got - want) / max(np.linalg.norm(want), 1e-30))


def test_bf16x3_netvlad_at_bench_size_forward_and_all_gradients(dev):
    """rowtile16b / aggregate16b / dx16b at B=24, N=1200 — the launch shapes of the bench.
    Input = bf16 values; oracle and float64 twin are fed the same (rounded) values, so only
    the kernel arithmetic is under test."""
    from soft_contrastive_learning_amd.model import nets
    b, n = 24, 1200
    x = U.feature_map(b, n, seed=2400)
    xb = torch.tensor(x).to(torch.bfloat16)
    xr = xb.float().numpy()
    w, c = U.vlad_params(seed=8)
    g = np.random.default_rng(5).standard_normal((b, 32768)).astype(np.float32)

    xt = xb.to(dev).reshape(b, 30, 40, 512).requires_grad_(True)
    wt = torch.tensor(w, device=dev).reshape(1, 1, 512, 64).requires_grad_(True)
    ct = torch.tensor(c, device=dev).reshape(1, 1, 1, 512, 64).requires_grad_(True)
    out = nets.netvlad(xt, wt, ct, True)
    out.backward(torch.tensor(g, device=dev))
    got = out.detach().cpu().numpy()

    assert _maxrel(got, NV.netvlad_fused(xr, w, c)) < 1e-4          # float32 oracle
    x64 = torch.tensor(xr, dtype=torch.float64, requires_grad=True)
    w64 = torch.tensor(w, dtype=torch.float64, requires_grad=True)
    c64 = torch.tensor(c, dtype=torch.float64, requires_grad=True)
    o64 = TT.netvlad(x64, w64, c64)
    o64.backward(torch.tensor(g, dtype=torch.float64))
    assert _maxrel(got, o64.detach().numpy()) < 1e-4
    # float32 outputs: same bound as the float32-input kernels
    assert _nrel(wt.grad.cpu().numpy().reshape(512, 64), w64.grad.numpy()) < 2e-4
    assert _nrel(ct.grad.cpu().numpy().reshape(512, 64), c64.grad.numpy()) < 2e-4
    # grad_x is STORED as bf16 (round to nearest: relative error <= 2^-9 per element, rms
    # 2^-9 / sqrt(3) = 1.1e-3 norm-relative); arithmetic error is three orders below that
    gx = xt.grad.float().cpu().numpy().reshape(b, n, 512)
    assert _nrel(gx, x64.grad.numpy()) < 2.5e-3
    # ... and elementwise: grad_x[n, :] = rn (dxhat - xhat <dxhat, xhat>) is a difference, so an
    # element's error is bounded by the bf16 rounding of the LARGEST entries of its location
    # (2^-9 of the row maximum), not by its own magnitude
    want = x64.grad.numpy()
    rowmax = np.abs(want).max(axis=2, keepdims=True)
    assert np.all(np.abs(gx - want) <= 2.0 ** -8 * rowmax)


@pytest.fixture(scope="module")
def config1(dev):
    """One configs[1] step in bf16 (what bench.py times) and the same step in float32 mode
    (library float32 convolutions + float32-MFMA NetVLAD kernels), same weights and inputs."""
    from soft_contrastive_learning_amd import parallel
    from soft_contrastive_learning_amd.model import losses, nets
    b, h, w = 24, 480, 640
    img = torch.randint(0, 256, (b, h, w, 3), generator=torch.Generator().manual_seed(42)).float().to(dev)
    dmat = U.positions_distances(b, side=200.0, seed=7)
    dist = torch.tensor(dmat[None], device=dev)
    res = {}
    for name, cdt in (('bf16', torch.bfloat16), ('f32', torch.float32)):
        model = nets.VGG16NetVLAD(compute_dtype=cdt, seed=1234).to(dev)
        buckets = parallel.GradBuckets(list(model.parameters()))
        nets.GRAD_SINK = buckets if cdt == torch.bfloat16 else None     # bench.py's setting
        try:
            buckets.zero()
            fmap = model.features(img)
            fmap.retain_grad()
            emb = nets.netvlad(fmap, model.assignment_kernel, model.cluster_centers, True)
            emb.retain_grad()
            loss = losses.wms_loss(dist, emb, d_alpha=0.8, d_beta=15.0)
            loss.backward()
            torch.cuda.synchronize()
        finally:
            nets.GRAD_SINK = None
        res[name] = dict(loss=float(loss), emb=emb.detach().float().cpu().numpy(),
                         fmap=fmap.detach().float().cpu().numpy().reshape(b, -1, 512),
                         gemb=emb.grad.float().cpu().numpy(),
                         grads={k: p.grad.detach().float().cpu().numpy().copy()
                                for k, p in model.named_parameters()},
                         w=model.assignment_kernel.detach().cpu().numpy().reshape(512, 64),
                         c=model.cluster_centers.detach().cpu().numpy().reshape(512, 64))
        del model, buckets, fmap, emb, loss
        torch.cuda.empty_cache()
    res['dmat'] = dmat
    return res


@pytest.mark.parametrize("mode", ['bf16', 'f32'])
def test_config1_head_matches_oracle_on_the_hip_conv5_3_map(config1, mode):
    """NetVLAD + wms of the full-size step against the CPU oracle fed the SAME conv5_3 map
    (as the HIP backbone produced it): descriptors and loss within 1e-4."""
    r = config1[mode]
    want_emb = NV.netvlad_fused(r['fmap'], r['w'], r['c'])
    assert _maxrel(r['emb'], want_emb) < 1e-4
    want_loss = float(O.wms_loss(config1['dmat'][None], want_emb, 0.8, 15.0))
    assert abs(r['loss'] - want_loss) <= 1e-4 * abs(want_loss), (r['loss'], want_loss)
    # d loss / d embeddings against the float64 twin on the oracle's descriptors, at the DEFAULT
    # initialisation — the regime bench.py runs.  The uniform-noise images of this fixture give 24
    # nearly parallel descriptors (all similarities > 0.99), so every gradient row M_i . E is a
    # difference of nearly equal terms: the float32 error, ~1e-7 of sum_j |M_ij| |E_j|, is amplified
    # by that cancellation (measured 2.3e-4 norm-relative on MI355X), hence the looser bound here;
    # test_config1_head_gradient_on_spread_descriptors holds 2e-4 on descriptors that spread out.
    if 'gemb' in r:
        e64 = torch.tensor(want_emb, dtype=torch.float64, requires_grad=True)
        TT.wms_loss(config1['dmat'][None], e64, 0.8, 15.0).backward()
        assert _nrel(r['gemb'], e64.grad.numpy()) < 1e-3


def test_config1_head_gradient_on_spread_descriptors(dev):
    """The full-size bf16 step on images whose content differs (tests/util_data.pose_images) and
    sharp VLAD assignments, so that the 24 descriptors spread out: descriptors and loss within 1e-4 of the oracle on the HIP
    conv5_3 map, and d loss / d embeddings within 2e-4 (norm-relative) of the float64 twin — the
    bound tests/test_gpu_losses.py holds on its synthetic embeddings."""
    from soft_contrastive_learning_amd import parallel
    from soft_contrastive_learning_amd.model import losses, nets
    b, h, w = 24, 480, 640
    img = torch.tensor(U.pose_images(b, h, w, seed=42), device=dev)
    dmat = U.positions_distances(b, side=200.0, seed=7)
    model = nets.VGG16NetVLAD(compute_dtype=torch.bfloat16, seed=1234).to(dev)
    # A random-init VGG16 maps ANY image to nearly the same mean direction (|mean xhat| = 0.93,
    # cosine 0.97-0.99 between images: measured), and with the near-uniform soft-assignment of
    # random-init VLAD variables every descriptor is that mean + the same centres.  Sharp
    # assignments and small centres — what a trained NetVLAD has — let the per-cluster residual
    # sums, which do differ between images, through: median cosine 0.83 instead of 0.99.
    w, c = U.vlad_params(logit_scale=10.0)
    with torch.no_grad():
        model.assignment_kernel.copy_(torch.tensor(w).reshape(1, 1, 512, 64))
        model.cluster_centers.copy_(torch.tensor(0.1 * c).reshape(1, 1, 1, 512, 64))
    buckets = parallel.GradBuckets(list(model.parameters()))
    nets.GRAD_SINK = buckets
    try:
        buckets.zero()
        fmap = model.features(img)
        emb = nets.netvlad(fmap, model.assignment_kernel, model.cluster_centers, True)
        emb.retain_grad()
        loss = losses.wms_loss(torch.tensor(dmat[None], device=dev), emb, d_alpha=0.8, d_beta=15.0)
        loss.backward()
        buckets.finish()
        torch.cuda.synchronize()
    finally:
        nets.GRAD_SINK = None
    e = emb.detach().float().cpu().numpy()
    sim = e @ e.T
    off = sim[~np.eye(b, dtype=bool)]
    print('descriptor cosine similarities: min %.3f median %.3f max %.3f'
          % (off.min(), np.median(off), off.max()))
    assert np.median(off) < 0.9, 'the synthetic images no longer spread the descriptors'
    want_emb = NV.netvlad_fused(fmap.detach().float().cpu().numpy().reshape(b, -1, 512),
                                model.assignment_kernel.detach().cpu().numpy().reshape(512, 64),
                                model.cluster_centers.detach().cpu().numpy().reshape(512, 64))
    assert _maxrel(e, want_emb) < 1e-4
    want_loss = float(O.wms_loss(dmat[None], want_emb, 0.8, 15.0))
    assert abs(float(loss) - want_loss) <= 1e-4 * abs(want_loss), (float(loss), want_loss)
    e64 = torch.tensor(want_emb, dtype=torch.float64, requires_grad=True)
    TT.wms_loss(dmat[None], e64, 0.8, 15.0).backward()
    err = _nrel(emb.grad.float().cpu().numpy(), e64.grad.numpy())
    print('d loss / d embeddings, norm-relative error vs float64 twin: %.3g' % err)
    assert err < 2e-4


def test_config1_bf16_step_against_float32_step(config1):
    """The bf16 backbone against the float32 one, 13 layers deep at 24 x 640x480.  bf16 keeps
    8 significant bits per stored activation (2^-9 = 2e-3 relative per rounding).  Forward
    quantities stay within a few bf16 roundings.  Gradients near the head (conv5_3, the VLAD
    variables) differ by 0.5-3 %; further down every ReLU / max-pool whose pre-activation sits
    within the bf16 error of its decision point takes the other branch, and those flips
    compound: measured on MI355X 18 % at conv4_1, 22 % at conv2_2, 30 % at conv1_1 and the
    trainable mean (norm-relative; cosine similarity 0.95).  This is a property of bf16
    training of this network, not of the kernels — every backward kernel is compared with a
    float32 reference ON THE SAME saved activations in tests/test_gpu_backbone.py (6e-3)."""
    b, f = config1['bf16'], config1['f32']
    assert _nrel(b['fmap'], f['fmap']) < 2e-2
    assert _nrel(b['emb'], f['emb']) < 3e-2
    assert abs(b['loss'] - f['loss']) <= 5e-3 * abs(f['loss']), (b['loss'], f['loss'])
    bound = {'conv5_3_kernel': 0.05, 'conv5_3_bias': 0.05, 'assignment_kernel': 0.06,
             'cluster_centers': 0.02, 'conv4_1_kernel': 0.35, 'conv3_1_bias': 0.35,
             'conv2_2_kernel': 0.4, 'conv1_1_kernel': 0.5, 'average_rgb': 0.5}
    worst = {k: _nrel(b['grads'][k], f['grads'][k]) for k in bound}
    cos = {k: float(np.vdot(b['grads'][k], f['grads'][k]) /
                    (np.linalg.norm(b['grads'][k]) * np.linalg.norm(f['grads'][k])))
           for k in bound}
    print('bf16 vs f32 gradient norm-relative differences:', worst)
    print('bf16 vs f32 gradient cosine similarities:', cos)
    for k, lim in bound.items():
        assert worst[k] < lim, (k, worst[k])
        assert cos[k] > 0.9, (k, cos[k])


def test_vgg16_without_vlad_matches_cpu(dev):
    """A5 `vgg16` (model/nets.py:72-131): backbone + channel L2 norm, no VLAD."""
    from soft_contrastive_learning_amd.model import nets
    model = nets.VGG16NetVLAD(seed=3)
    img = torch.randint(0, 256, (2, 64, 96, 3), generator=torch.Generator().manual_seed(9)).float()
    with torch.no_grad():
        model = model.double()
        model.compute_dtype = torch.float64
        x64 = model.features(img.double())
        model.compute_dtype = torch.float32
        want = (x64 * torch.rsqrt(torch.clamp_min((x64 * x64).sum(-1, keepdim=True), 1e-12))).numpy()
        got = nets.vgg16(img.to(dev), model=model.float().to(dev)).cpu().numpy()
    assert got.shape == (2, 4, 6, 512)
    np.testing.assert_allclose(np.linalg.norm(got, axis=-1), 1.0, rtol=1e-5)
    assert _maxrel(got, want) < 1e-4
    # flattened like the callers do (train/train.py:611, evaluation/inference.py:92)
    assert got.reshape(2, -1).shape == (2, 4 * 6 * 512)


def test_a1_float32_end_to_end_error_against_float64(dev):
    """A1 in float32 mode against a float64 CPU evaluation of the same network: the error of
    the HIP path must be within 1e-4 (north_star), measured, not assumed.  The float32 CPU
    evaluation is reported beside it (its own distance from float64)."""
    from soft_contrastive_learning_amd.model import nets
    model = nets.VGG16NetVLAD(seed=1234)
    img = torch.randint(0, 256, (4, 224, 224, 3), generator=torch.Generator().manual_seed(42)).float()
    with torch.no_grad():
        f32 = model.features(img).reshape(4, -1, 512).numpy()
        w = model.assignment_kernel.reshape(512, 64).numpy()
        c = model.cluster_centers.reshape(512, 64).numpy()
        got = nets.vgg16Netvlad(img.to(dev), model=model.to(dev)).cpu().numpy()
        model = model.cpu().double()
        model.compute_dtype = torch.float64
        f64 = model.features(img.double()).reshape(4, -1, 512)
        want = TT.netvlad(f64, model.assignment_kernel.reshape(512, 64),
                          model.cluster_centers.reshape(512, 64)).numpy()
    cpu32 = NV.netvlad_fused(f32, w, c)
    e_hip, e_cpu = _maxrel(got, want), _maxrel(cpu32, want)
    print('A1 float32 end-to-end max-relative error vs float64: HIP %.3g, torch-CPU f32 %.3g'
          % (e_hip, e_cpu))
    assert e_hip < 1e-4, (e_hip, e_cpu)
