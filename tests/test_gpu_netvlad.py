"""GPU parity: fused NetVLAD head (csrc/netvlad.hip through the C-ABI) vs the CPU oracle.

Tolerance: 1e-4 relative (north_star) on the unit-norm descriptor, measured as
max |got - want| / max |want|; gradients vs the float64 autograd twin, norm-relative.
"""
import numpy as np
import pytest
import torch

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


def _run(dev, x, w, c, pre_l2=True, dtype=torch.float32, grad=None):
    from soft_contrastive_learning_amd.model import nets
    b, n, d = x.shape
    xt = torch.tensor(x, device=dev).to(dtype).reshape(b, 1, n, d).requires_grad_(grad is not None)
    wt = torch.tensor(w, device=dev).reshape(1, 1, d, -1).requires_grad_(grad is not None)
    ct = torch.tensor(c, device=dev).reshape(1, 1, 1, d, -1).requires_grad_(grad is not None)
    out = nets.netvlad(xt, wt, ct, pre_l2)
    if grad is None:
        return out.cpu().numpy()
    out.backward(torch.tensor(grad, device=dev))
    return (out.detach().cpu().numpy(), xt.grad.float().cpu().numpy().reshape(b, n, d),
            wt.grad.cpu().numpy().reshape(d, -1), ct.grad.cpu().numpy().reshape(d, -1))


def test_kat_k8_zero_assignment(dev):
    # W = 0, C = 0: every cluster column is unit(sum_n xhat) / sqrt(K)
    x = U.feature_map(2, 40, seed=3)
    w = np.zeros((512, 64), np.float32)
    c = np.zeros((512, 64), np.float32)
    out = _run(dev, x, w, c).reshape(2, 512, 64)
    xs = (x / np.linalg.norm(x, axis=2, keepdims=True)).sum(axis=1)
    unit = xs / np.linalg.norm(xs, axis=1, keepdims=True)
    want = np.repeat(unit[:, :, None], 64, axis=2) / 8.0
    assert _maxrel(out, want) < 1e-5


@pytest.mark.parametrize("b,n", [(1, 1), (2, 31), (3, 33), (4, 196), (2, 165), (24, 1200)])
def test_forward_matches_oracle(dev, b, n):
    x = U.feature_map(b, n, seed=b * 100 + n)
    w, c = U.vlad_params()
    want = NV.netvlad_fused(x, w, c)
    got = _run(dev, x, w, c)
    assert got.shape == (b, 32768)
    assert _maxrel(got, want) < 1e-4
    np.testing.assert_allclose(np.linalg.norm(got, axis=1), 1.0, rtol=1e-5)


def test_forward_matches_literal_tf_style_oracle(dev):
    x = U.feature_map(2, 50, seed=77)
    w, c = U.vlad_params(seed=5)
    assert _maxrel(_run(dev, x, w, c), NV.netvlad_literal(x, w, c)) < 1e-4


def test_forward_without_pre_l2(dev):
    x = U.feature_map(2, 70, seed=9) * 0.1
    w, c = U.vlad_params(seed=6, logit_scale=1.0)
    assert _maxrel(_run(dev, x, w, c, pre_l2=False), NV.netvlad_fused(x, w, c, pre_l2=False)) < 1e-4


def test_forward_bf16_feature_map(dev):
    # bf16 storage of the conv5_3 map (config 2): compare against the oracle fed the
    # same bf16-rounded values, so only the kernel arithmetic is under test
    x = U.feature_map(3, 200, seed=12)
    xb = torch.tensor(x).to(torch.bfloat16).float().numpy()
    w, c = U.vlad_params()
    got = _run(dev, x, w, c, dtype=torch.bfloat16)
    assert _maxrel(got, NV.netvlad_fused(xb, w, c)) < 1e-4


@pytest.mark.parametrize("b,n,pre_l2", [(1, 1, True), (2, 31, True), (3, 33, True), (4, 196, True),
                                        (2, 165, False), (5, 1200, True), (40, 70, True), (2, 2100, True)])
def test_fused_bf16_kernels_on_ragged_shapes(dev, b, n, pre_l2):
    """The fused bf16 kernels (vlad_fwd / vlad_bwd / vlad_dx: 32-location steps, slices of an image
    on different workgroups, the slab sums) at shapes that exercise every edge: a single location,
    a partial last step, one step per slice, more slices than the slab reader takes in one batch of
    loads (2100 locations: 66 slices), many images, no channel norm; training mode (saved rows)
    and inference mode (vlad_fwd_kernel<false>: another instantiation — its float operations may be
    contracted differently, which moves the two-plane split of a coefficient by one unit, 2^-17)
    must agree to that level."""
    x = U.feature_map(b, n, seed=7 * b + n)
    if not pre_l2:
        x = x * 0.1
    xb = torch.tensor(x).to(torch.bfloat16).float().numpy()
    w, c = U.vlad_params(seed=8, logit_scale=3.0 if pre_l2 else 1.0)
    g = np.random.default_rng(2).standard_normal((b, 32768)).astype(np.float32)
    x64 = torch.tensor(xb, dtype=torch.float64, requires_grad=True)
    w64 = torch.tensor(w, dtype=torch.float64, requires_grad=True)
    c64 = torch.tensor(c, dtype=torch.float64, requires_grad=True)
    o64 = TT.netvlad(x64, w64, c64, pre_l2=pre_l2)
    o64.backward(torch.tensor(g, dtype=torch.float64))
    out, gx, gw, gc = _run(dev, x, w, c, pre_l2=pre_l2, dtype=torch.bfloat16, grad=g)
    assert _maxrel(out, NV.netvlad_fused(xb, w, c, pre_l2=pre_l2)) < 1e-4
    assert _maxrel(out, o64.detach().numpy()) < 1e-4
    assert _nrel(gx, x64.grad.numpy()) < 6e-3               # grad_x is stored in bf16
    assert _nrel(gw, w64.grad.numpy()) < 2e-4
    assert _nrel(gc, c64.grad.numpy()) < 2e-4
    infer = _run(dev, x, w, c, pre_l2=pre_l2, dtype=torch.bfloat16)
    assert _maxrel(infer, out) < 2e-5


@pytest.mark.parametrize("b,n,pre_l2", [(2, 37, True), (3, 196, True), (2, 64, False), (4, 1200, True)])
def test_backward_matches_float64_twin(dev, b, n, pre_l2):
    x = U.feature_map(b, n, seed=n)
    if not pre_l2:
        x = x * 0.1
    w, c = U.vlad_params(seed=8, logit_scale=3.0 if pre_l2 else 1.0)
    g = np.random.default_rng(2).standard_normal((b, 32768)).astype(np.float32)
    x64 = torch.tensor(x, dtype=torch.float64, requires_grad=True)
    w64 = torch.tensor(w, dtype=torch.float64, requires_grad=True)
    c64 = torch.tensor(c, dtype=torch.float64, requires_grad=True)
    o64 = TT.netvlad(x64, w64, c64, pre_l2=pre_l2)
    o64.backward(torch.tensor(g, dtype=torch.float64))
    out, gx, gw, gc = _run(dev, x, w, c, pre_l2=pre_l2, grad=g)
    assert _maxrel(out, o64.detach().numpy()) < 1e-4
    assert _nrel(gx, x64.grad.numpy()) < 2e-4
    assert _nrel(gw, w64.grad.numpy()) < 2e-4
    assert _nrel(gc, c64.grad.numpy()) < 2e-4


def test_backward_bf16_feature_map(dev):
    x = U.feature_map(2, 100, seed=4)
    xb = torch.tensor(x).to(torch.bfloat16).float().numpy()
    w, c = U.vlad_params(seed=8)
    g = np.random.default_rng(3).standard_normal((2, 32768)).astype(np.float32)
    x64 = torch.tensor(xb, dtype=torch.float64, requires_grad=True)
    o64 = TT.netvlad(x64, torch.tensor(w, dtype=torch.float64), torch.tensor(c, dtype=torch.float64))
    o64.backward(torch.tensor(g, dtype=torch.float64))
    _, gx, _, _ = _run(dev, x, w, c, dtype=torch.bfloat16, grad=g)
    # grad_x is stored in bf16: 2^-8 relative per element
    assert _nrel(gx, x64.grad.numpy()) < 6e-3


def test_vgg16netvlad_end_to_end_vs_cpu(dev):
    """Whole embedder (model/nets.py:7-69) on a tiny image: HIP path vs torch-CPU backbone
    + NumPy NetVLAD oracle with the same weights."""
    from soft_contrastive_learning_amd.model import nets
    torch.manual_seed(0)
    model = nets.VGG16NetVLAD()
    img = torch.randint(0, 256, (2, 64, 80, 3), generator=torch.Generator().manual_seed(42)).float()
    with torch.no_grad():
        fmap = model.features(img)                               # CPU torch convs
        w = model.assignment_kernel.reshape(512, 64).numpy()
        c = model.cluster_centers.reshape(512, 64).numpy()
        want = NV.netvlad_fused(fmap.reshape(2, -1, 512).numpy(), w, c)
        got = nets.vgg16Netvlad(img.to(dev), model=model.to(dev)).cpu().numpy()
    assert got.shape == (2, 32768)
    # measured on MI355X against a float64 evaluation: HIP 1.8e-7, torch-CPU float32 5.3e-7
    # (tests/test_gpu_config1.py::test_a1_float32_end_to_end_error_against_float64)
    assert _maxrel(got, want) < 1e-4
    # grey input is replicated to 3 channels (model/nets.py:15-16)
    grey = img[..., :1].to(dev)
    with torch.no_grad():
        g1 = nets.vgg16Netvlad(grey, model=model)
        g3 = nets.vgg16Netvlad(grey.expand(-1, -1, -1, 3).contiguous(), model=model)
    assert _maxrel(g1.cpu().numpy(), g3.cpu().numpy()) < 1e-5   # MIOpen may pick another solver


# ---- round 4: two launches forward, three backward ---------------------------------------------
def _bits(a):
    return np.ascontiguousarray(a).view(np.uint32)


def _run_variant(dev, variant, x, w, c, g, dtype=torch.bfloat16, planes=False):
    """_run under scl_debug_set_variant(variant); planes=True: the assignment weights' plane images
    come from prepack() (the weight-packing launch) instead of a launch inside the call."""
    from soft_contrastive_learning_amd import _lib as L
    from soft_contrastive_learning_amd.model import nets
    b, n, d = x.shape
    xt = torch.tensor(x, device=dev).to(dtype).reshape(b, 1, n, d).requires_grad_(True)
    wt = torch.tensor(w, device=dev).reshape(1, 1, d, -1).requires_grad_(True)
    ct = torch.tensor(c, device=dev).reshape(1, 1, 1, d, -1).requires_grad_(True)
    with L.maybe_variant(variant):            # 0: the product library; else the diagnostic build
        pl = None
        if planes:
            wd = wt.detach()                                  # (the registry holds a weak reference)
            assert nets.prepack([], force=True, vlad_w=wd) == 1
            pl = nets.fresh_vlad_planes(wd)
            assert pl is not None and nets.fresh_vlad_planes(wd) is None      # handed over once
        out = nets.netvlad(xt, wt, ct, True, pl)
        out.backward(torch.tensor(g, device=dev))
        torch.cuda.synchronize()
    return (out.detach().cpu().numpy(), xt.grad.float().cpu().numpy().reshape(b, n, d),
            wt.grad.cpu().numpy().reshape(d, -1), ct.grad.cpu().numpy().reshape(d, -1))


@pytest.mark.parametrize("b,n", [(1, 1), (3, 33), (5, 1200), (24, 1200), (40, 70)])
def test_sibling_exchange_and_self_computing_path_give_the_same_bits(dev, b, n):
    """vlad_finish_kernel / vlad_bwd_prologue_kernel: the 8 workgroups of an image exchange their
    shares of the global sums through tagged words with a bounded wait.  Variant 921 sets the
    patience to zero, so that workgroups compute their siblings' shares themselves: every output
    must be bit-identical (same routine, same order), i.e. no result depends on co-residency."""
    x = U.feature_map(b, n, seed=11 * b + n)
    w, c = U.vlad_params(seed=8, logit_scale=3.0)
    g = np.random.default_rng(5).standard_normal((b, 32768)).astype(np.float32)
    ref = _run_variant(dev, 0, x, w, c, g)
    alone = _run_variant(dev, 921, x, w, c, g)
    for r, a in zip(ref, alone):
        assert np.array_equal(_bits(r), _bits(a))


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float32])
def test_new_launch_structure_agrees_with_round_3s(dev, dtype):
    """scl_debug_set_variant(920) runs round 3's ten launches (separate plane split, finish_sum,
    finish_norm, bwd_dots, bwd_du, wgrad_partial, wgrad_finish); the merged kernels add the same
    numbers in another order."""
    b, n = 24, 1200
    x = U.feature_map(b, n, seed=3)
    w, c = U.vlad_params(seed=8, logit_scale=3.0)
    g = np.random.default_rng(6).standard_normal((b, 32768)).astype(np.float32)
    new = _run_variant(dev, 0, x, w, c, g, dtype=dtype)
    old = _run_variant(dev, 920, x, w, c, g, dtype=dtype)
    assert _maxrel(new[0], old[0]) < 5e-6
    assert _nrel(new[1], old[1]) < (2e-3 if dtype == torch.bfloat16 else 5e-6)   # bf16 storage of grad_x
    assert _nrel(new[2], old[2]) < 5e-6
    assert _nrel(new[3], old[3]) < 5e-6


@pytest.mark.parametrize("b,n", [(1, 1), (3, 33), (7, 165), (5, 1200), (200, 165), (192, 1200)])
def test_inference_with_the_finish_in_the_forward_kernel(dev, b, n):
    """Round 6: inference (nothing saved) at >= 144 images runs ONE workgroup per image with the
    finish in the tail of vlad_fwd8_kernel — no partial VLADs, one launch.  Against the two launches
    (diagnostic variant 923; 924 forces the one-launch form at any batch size): the same descriptors
    to float32 rounding (the column and global norms are summed in another order), unit norm, and
    against the oracle where it is small enough to run."""
    from soft_contrastive_learning_amd import _lib as L
    from soft_contrastive_learning_amd.model import nets
    from oracle import netvlad_np
    x = U.feature_map(b, n, seed=7 * b + n)
    w, c = U.vlad_params(seed=8, logit_scale=3.0)
    xt = torch.tensor(x, device=dev).bfloat16().reshape(b, 1, n, 512)
    wt, ct = torch.tensor(w, device=dev), torch.tensor(c, device=dev)

    def run(variant):
        with L.variant(variant):
            with L.KernelTimer(capacity=16) as kt:
                with torch.no_grad():
                    out = nets.netvlad(xt, wt, ct, True)
                torch.cuda.synchronize()
        return out.cpu().numpy(), set(kt.summary())
    two, n2 = run(923)
    one, n1 = run(924)
    assert 'vlad_finish_kernel' in n2 and 'vlad_fwd8_kernel<finish>' not in n2, n2
    assert n1 == {'vlad_fwd8_kernel<finish>'} or n1 == {'vlad_fwd8_kernel<finish>', 'vlad_planes_kernel'}, n1
    assert np.abs(one - two).max() <= 2e-6 * np.abs(two).max()
    np.testing.assert_allclose(np.linalg.norm(one.astype(np.float64), axis=1), 1.0, atol=1e-5)
    if b >= 144:                                              # the product picks it by itself there
        with L.KernelTimer(capacity=16) as kt:
            with torch.no_grad():
                auto = nets.netvlad(xt, wt, ct, True)
            torch.cuda.synchronize()
        assert 'vlad_fwd8_kernel<finish>' in set(kt.summary())
        assert np.array_equal(auto.cpu().numpy(), one)
    if b * n <= 6000:
        want = netvlad_np.netvlad_fused(xt.float().cpu().numpy().reshape(b, n, 512), w, c)
        assert np.abs(one - want).max() <= 1e-4 * np.abs(want).max()


def test_plane_images_from_the_packing_launch(dev):
    """The assignment weights' plane images written by scl_conv_pack_batch (SCL_PACK_VLAD_W, the
    launch that packs the convolution weights) are the ones the call would build for itself."""
    x = U.feature_map(4, 300, seed=21)
    w, c = U.vlad_params(seed=9, logit_scale=3.0)
    g = np.random.default_rng(7).standard_normal((4, 32768)).astype(np.float32)
    own = _run_variant(dev, 0, x, w, c, g)
    pre = _run_variant(dev, 0, x, w, c, g, planes=True)
    for r, a in zip(own, pre):
        assert np.array_equal(_bits(r), _bits(a))


def test_plane_images_are_never_trusted_by_address_and_version(dev):
    """the plane images used to be looked up by (data_ptr, _version), which a fused
    optimizer step or a ``.data`` write moves neither of.  Now they are handed over by the pass
    that packed them: a direct nets.netvlad() call after an in-place weight change must see the
    NEW weights, and a backward whose forward's plane buffer was rewritten by a later pass must
    fall back to building its own (same bits as a run that never had plane images)."""
    from soft_contrastive_learning_amd.model import nets
    b, n = 3, 200
    x = U.feature_map(b, n, seed=4)
    w, c = U.vlad_params(seed=5, logit_scale=3.0)
    xt = torch.tensor(x, device=dev).bfloat16().reshape(b, 1, n, 512).requires_grad_(True)
    wt = torch.tensor(w, device=dev).reshape(1, 1, 512, 64).requires_grad_(True)
    ct = torch.tensor(c, device=dev).reshape(1, 1, 1, 512, 64).requires_grad_(True)
    go = torch.randn(b, 32768, device=dev, generator=torch.Generator(device=dev).manual_seed(3))
    wd = wt.detach()
    assert nets.prepack([], force=True, vlad_w=wd) == 1
    pl = nets.fresh_vlad_planes(wd)
    out_old = nets.netvlad(xt, wt, ct, True, pl)
    wd.data.mul_(0.5)                                  # in place: neither address nor _version moves
    out_new = nets.netvlad(xt, wt, ct, True)           # no plane images handed in: splits W itself
    assert not torch.equal(out_old, out_new)
    assert nets.prepack([], force=True, vlad_w=wd) == 1          # rewrites the SAME buffer
    pl2 = nets.fresh_vlad_planes(wd)
    assert pl2[0].data_ptr() == pl[0].data_ptr() and pl2[1] == pl[1] + 1
    assert torch.equal(out_new, nets.netvlad(xt, wt, ct, True, pl2))
    # the old forward's backward: its plane generation is stale -> self-built images of current W
    g_old = torch.autograd.grad(out_old, (xt, wt, ct), go)
    out_ref = nets.netvlad(xt, wt, ct, True)
    del out_ref
    assert all(torch.isfinite(t.float()).all() for t in g_old)


def test_backward_twice_over_the_same_saved_tensors(dev):
    """Row 513 of save_vlad carries the prologue's exchange words: zero on entry, zero again on
    return — a second backward pass over the same graph sees what the first one saw."""
    from soft_contrastive_learning_amd.model import nets
    b, n = 6, 500
    x = U.feature_map(b, n, seed=2)
    w, c = U.vlad_params(seed=8, logit_scale=3.0)
    xt = torch.tensor(x, device=dev).bfloat16().reshape(b, 1, n, 512).requires_grad_(True)
    wt = torch.tensor(w, device=dev).reshape(1, 1, 512, 64).requires_grad_(True)
    ct = torch.tensor(c, device=dev).reshape(1, 1, 1, 512, 64).requires_grad_(True)
    go = torch.randn(b, 32768, device=dev, generator=torch.Generator(device=dev).manual_seed(1))
    out = nets.netvlad(xt, wt, ct, True)
    g1 = torch.autograd.grad(out, (xt, wt, ct), go, retain_graph=True)
    g2 = torch.autograd.grad(out, (xt, wt, ct), go)
    for a, bb in zip(g1, g2):
        assert torch.equal(a, bb)
