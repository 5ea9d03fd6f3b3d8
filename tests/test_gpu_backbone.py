"""GPU checks of the fused backbone glue (csrc/vgg_glue.hip): the fused path must equal
the plain PyTorch composition of conv -> bias -> [pool] -> ReLU (model/nets.py:27-63) in
forward and in every gradient, and the raw kernels must match torch ops elementwise."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def dev():
    assert torch.cuda.is_available()
    return torch.device("cuda:0")


def _pair(dev, dtype):
    from soft_contrastive_learning_amd.model import nets
    a = nets.VGG16NetVLAD(compute_dtype=dtype, seed=11, fused_relu=True).to(dev)
    b = nets.VGG16NetVLAD(compute_dtype=dtype, seed=11, fused_relu=False).to(dev)
    return a, b


def _nrel(x, y):
    x, y = x.double(), y.double()
    return float((x - y).norm() / y.norm().clamp_min(1e-30))


@pytest.mark.parametrize("shape", [(2, 64, 80), (1, 48, 32)])
def test_fused_backbone_equals_plain_composition_f32(dev, shape):
    fused, plain = _pair(dev, torch.float32)
    b, h, w = shape
    img = torch.randint(0, 256, (b, h, w, 3), generator=torch.Generator().manual_seed(1)).float().to(dev)
    gseed = torch.Generator().manual_seed(2)
    xf = fused.features(img)
    xp = plain.features(img)
    assert xf.shape == xp.shape == (b, h // 16, w // 16, 512)
    assert _nrel(xf, xp) < 1e-5
    g = torch.randn(xf.shape, generator=gseed).to(dev)
    xf.backward(g)
    xp.backward(g)
    for (n1, p1), (_, p2) in zip(fused.named_parameters(), plain.named_parameters()):
        if p2.grad is None:
            assert p1.grad is None, n1
            continue
        assert _nrel(p1.grad, p2.grad) < 2e-4, n1


def test_fused_backbone_bf16_close_to_plain(dev):
    fused, plain = _pair(dev, torch.bfloat16)
    img = torch.randint(0, 256, (2, 64, 80, 3), generator=torch.Generator().manual_seed(3)).float().to(dev)
    xf, xp = fused.features(img).float(), plain.features(img).float()
    assert _nrel(xf, xp) < 5e-2            # bf16 storage; the fused path rounds once less
    g = torch.randn(xf.shape, generator=torch.Generator().manual_seed(4)).to(dev)
    fused.features(img).backward(g.bfloat16())
    plain.features(img).backward(g.bfloat16())
    # 13 bf16 layers deep, two different rounding sequences (and MIOpen may pick different
    # solvers run to run): compare loosely; exact agreement is asserted in f32 above
    for (n1, p1), (_, p2) in zip(fused.named_parameters(), plain.named_parameters()):
        if p2.grad is not None:
            assert _nrel(p1.grad, p2.grad) < 0.3, n1


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("c", [64, 128, 512])
def test_raw_glue_kernels_match_torch(dev, dtype, c):
    from soft_contrastive_learning_amd import _lib as L
    lib = L.load()
    code = L.DT_F32 if dtype == torch.float32 else L.DT_BF16
    gen = torch.Generator().manual_seed(c)
    b, h, w = 2, 6, 10
    z = torch.randn(b, h, w, c, generator=gen).to(dev).to(dtype)            # channels-last storage
    bias = torch.randn(c, generator=gen).to(dev)
    st = L.stream_of(z)
    # bias + relu in place
    y = z.clone()
    L.check(lib.scl_vgg_bias_act(L.ptr(y), code, L.ptr(bias), b * h * w, c, 1, st))
    want = F.relu(z.float() + bias).to(dtype)
    assert torch.equal(y, want)
    # pool forward
    a = torch.empty(b, h // 2, w // 2, c, dtype=dtype, device=dev)
    L.check(lib.scl_vgg_pool_fwd(L.ptr(z), code, L.ptr(bias), b, h, w, c, L.ptr(a), st))
    zp = z.float().permute(0, 3, 1, 2)
    want_a = F.relu(F.max_pool2d(zp, 2, 2) + bias.view(1, -1, 1, 1)).permute(0, 2, 3, 1).to(dtype)
    assert torch.equal(a, want_a)
    # pool backward vs autograd of the same composition (ties are measure-zero for randn f32;
    # for bf16 compare the bias gradient and the total mass instead of positions)
    g = torch.randn(a.shape, generator=gen).to(dev).to(dtype)
    gz = torch.empty_like(z)
    gb = torch.empty(c, device=dev)
    ws = L.workspace(lib.scl_vgg_workspace_bytes(c), dev)
    L.check(lib.scl_vgg_pool_bwd(L.ptr(g), L.ptr(a), L.ptr(z), code, b, h, w, c, L.ptr(gz), L.ptr(gb),
                                 L.ptr(ws), ws.numel(), st))
    zr = z.float().permute(0, 3, 1, 2).clone().requires_grad_(True)
    br = bias.clone().requires_grad_(True)
    out = F.relu(F.max_pool2d(zr, 2, 2) + br.view(1, -1, 1, 1))
    out.backward(g.float().permute(0, 3, 1, 2))
    np.testing.assert_allclose(gb.cpu().numpy(), br.grad.cpu().numpy(), rtol=1e-4, atol=1e-4)
    if dtype == torch.float32:
        assert torch.equal(gz.permute(0, 3, 1, 2), zr.grad)
    else:
        np.testing.assert_allclose(gz.float().sum().item(), zr.grad.sum().item(), rtol=1e-2, atol=1e-2)
    # relu backward + bias gradient
    gy = torch.randn(b, h, w, c, generator=gen).to(dev).to(dtype)
    gz2 = torch.empty_like(gy)
    L.check(lib.scl_vgg_act_bwd(L.ptr(gy), L.ptr(y), code, b * h * w, c, L.ptr(gz2), L.ptr(gb),
                                L.ptr(ws), ws.numel(), st))
    want_g = torch.where(y.float() > 0, gy.float(), torch.zeros_like(gy.float()))
    assert torch.equal(gz2.float(), want_g)
    np.testing.assert_allclose(gb.cpu().numpy(), want_g.sum(dim=(0, 1, 2)).cpu().numpy(),
                               rtol=1e-4, atol=1e-3)
    # bias gradient only (no activation)
    L.check(lib.scl_vgg_act_bwd(L.ptr(gy), None, code, b * h * w, c, None, L.ptr(gb), L.ptr(ws),
                                ws.numel(), st))
    np.testing.assert_allclose(gb.cpu().numpy(), gy.float().sum(dim=(0, 1, 2)).cpu().numpy(),
                               rtol=1e-4, atol=1e-3)


def test_pool_bwd_odd_sizes_zero_the_uncovered_border(dev):
    from soft_contrastive_learning_amd import _lib as L
    lib = L.load()
    b, h, w, c = 1, 5, 7, 64
    z = torch.randn(b, h, w, c, device=dev)
    bias = torch.zeros(c, device=dev)
    a = torch.empty(b, h // 2, w // 2, c, device=dev)
    st = L.stream_of(z)
    L.check(lib.scl_vgg_pool_fwd(L.ptr(z), L.DT_F32, L.ptr(bias), b, h, w, c, L.ptr(a), st))
    g = torch.ones_like(a)
    gz = torch.full_like(z, 7.0)
    gb = torch.empty(c, device=dev)
    ws = L.workspace(lib.scl_vgg_workspace_bytes(c), dev)
    L.check(lib.scl_vgg_pool_bwd(L.ptr(g), L.ptr(a), L.ptr(z), L.DT_F32, b, h, w, c, L.ptr(gz),
                                 L.ptr(gb), L.ptr(ws), ws.numel(), st))
    assert float(gz[:, 4].abs().max()) == 0.0 and float(gz[:, :, 6].abs().max()) == 0.0
    zr = z.permute(0, 3, 1, 2).clone().requires_grad_(True)
    F.relu(F.max_pool2d(zr, 2, 2)).backward(g.permute(0, 3, 1, 2))
    assert torch.equal(gz.permute(0, 3, 1, 2), zr.grad)


@pytest.mark.parametrize('shape', [(2, 20, 40), (1, 13, 37), (3, 8, 32), (1, 33, 70)])
def test_conv64_forward_and_backward_data(dev, shape):
    """csrc/conv64.hip (conv1_2's kernel) against a float32 convolution of the same bf16 data."""
    from soft_contrastive_learning_amd.model import nets
    b, h, w = shape
    g = torch.Generator().manual_seed(11)
    x = torch.randn(b, 64, h, w, generator=g).to(dev).bfloat16().contiguous(memory_format=torch.channels_last)
    wt = (torch.randn(64, 64, 3, 3, generator=g) * 0.05).to(dev).bfloat16()
    for wv in (wt, wt.contiguous(memory_format=torch.channels_last)):     # any weight strides
        got = nets.conv64(x, wv, False)
        want = torch.nn.functional.conv2d(x.float(), wv.float(), padding=1)
        assert got.shape == want.shape and got.is_contiguous(memory_format=torch.channels_last)
        err = (got.float() - want).abs().max() / want.abs().max()
        assert float(err) < 6e-3, float(err)                      # bf16 output rounding
        gx = nets.conv64(x, wv, True)
        want_gx = torch.nn.functional.conv_transpose2d(x.float(), wv.float(), padding=1)
        err = (gx.float() - want_gx).abs().max() / want_gx.abs().max()
        assert float(err) < 6e-3, float(err)
    # agrees with the library kernel to bf16 rounding as well
    lib_out = torch.ops.aten.convolution(x, wt, None, [1, 1], [1, 1], [1, 1], False, [0, 0], 1)
    assert float((got.float() - lib_out.float()).abs().max() / lib_out.float().abs().max()) < 1.2e-2


@pytest.mark.parametrize('shape', [(2, 20, 40), (1, 13, 37), (3, 8, 32), (2, 33, 70)])
def test_wrw64_weight_gradient(dev, shape):
    from soft_contrastive_learning_amd.model import nets
    b, h, w = shape
    g = torch.Generator().manual_seed(13)
    x = torch.randn(b, 64, h, w, generator=g).to(dev).bfloat16().contiguous(memory_format=torch.channels_last)
    gz = torch.randn(b, 64, h, w, generator=g).to(dev).bfloat16().contiguous(memory_format=torch.channels_last)
    xf = x.float().requires_grad_(False)
    wf = torch.zeros(64, 64, 3, 3, device=dev, requires_grad=True)
    torch.nn.functional.conv2d(xf, wf, padding=1).backward(gz.float())
    want = wf.grad
    for like in (torch.empty(64, 64, 3, 3, device=dev, dtype=torch.bfloat16),
                 torch.empty(64, 64, 3, 3, device=dev, dtype=torch.bfloat16).contiguous(
                     memory_format=torch.channels_last)):
        got = nets.wrw64(x, gz, like)
        assert got.stride() == like.stride()
        err = (got.float() - want).abs().max() / want.abs().max()
        assert float(err) < 6e-3, float(err)
    again = nets.wrw64(x, gz, like)
    assert torch.equal(got, again)                       # fixed-order slab reduction


@pytest.mark.parametrize('cin,cout', [(64, 128), (128, 128), (64, 64)])
@pytest.mark.parametrize('shape', [(2, 12, 40), (1, 13, 37)])
def test_own_conv_other_shapes(dev, shape, cin, cout):
    """conv2_1 / conv2_2 shapes of csrc/conv64.hip, forward and backward-data."""
    from soft_contrastive_learning_amd.model import nets
    b, h, w = shape
    g = torch.Generator().manual_seed(17)
    x = torch.randn(b, cin, h, w, generator=g).to(dev).bfloat16().contiguous(memory_format=torch.channels_last)
    gy = torch.randn(b, cout, h, w, generator=g).to(dev).bfloat16().contiguous(memory_format=torch.channels_last)
    wt = (torch.randn(cout, cin, 3, 3, generator=g) * 0.05).to(dev).bfloat16().contiguous(
        memory_format=torch.channels_last)
    assert nets._conv64_ok(x, wt) and nets._conv64_ok(gy, wt, True)
    got = nets.conv64(x, wt, False)
    want = torch.nn.functional.conv2d(x.float(), wt.float(), padding=1)
    assert got.shape == want.shape
    assert float((got.float() - want).abs().max() / want.abs().max()) < 6e-3
    gx = nets.conv64(gy, wt, True)
    want_gx = torch.nn.functional.conv_transpose2d(gy.float(), wt.float(), padding=1)
    assert gx.shape == want_gx.shape
    assert float((gx.float() - want_gx).abs().max() / want_gx.abs().max()) < 6e-3


@pytest.mark.parametrize('cin,cout,shape', [(64, 64, (2, 16, 40)), (128, 128, (1, 10, 37)),
                                            (64, 128, (1, 9, 33))])
def test_own_conv_fused_tails(dev, cin, cout, shape):
    """bias + ReLU and bias + max-pool + ReLU fused into the convolution epilogue."""
    from soft_contrastive_learning_amd.model import nets
    b, h, w = shape
    g = torch.Generator().manual_seed(19)
    x = torch.randn(b, cin, h, w, generator=g).to(dev).bfloat16().contiguous(memory_format=torch.channels_last)
    wt = (torch.randn(cout, cin, 3, 3, generator=g) * 0.05).to(dev).bfloat16()
    bias = torch.randn(cout, generator=g).to(dev)
    z32 = torch.nn.functional.conv2d(x.float(), wt.float(), padding=1)
    scale = float(z32.abs().max())
    y = nets.conv64(x, wt, False, bias=bias, relu=True)
    want = torch.relu(z32 + bias[None, :, None, None])
    assert float((y.float() - want).abs().max()) < 6e-3 * scale
    y = nets.conv64(x, wt, False, bias=bias, relu=False)
    assert float((y.float() - (z32 + bias[None, :, None, None])).abs().max()) < 6e-3 * scale
    z, a = nets.conv64(x, wt, False, bias=bias, pool=True)
    assert float((z.float() - z32).abs().max()) < 6e-3 * scale
    assert a.shape == (b, cout, h // 2, w // 2)
    # the pooled map is computed from the float32 accumulators, the library path from the bf16 z
    want_a = torch.relu(torch.nn.functional.max_pool2d(z32, 2) + bias[None, :, None, None])
    assert float((a.float() - want_a).abs().max()) < 6e-3 * scale


@pytest.fixture(params=['mfma32x32x16', 'mfma16x16x32'])
def lds_kernel(request):
    """Pins which of the two LDS-weights convolution kernels runs (csrc/convg.hip: 40000 + v,
    csrc/convh.hip: 50000 + v); yields the variant base for tests that add their own v."""
    from soft_contrastive_learning_amd import _lib as L
    base = 40000 if request.param == 'mfma32x32x16' else 50000
    with L.variant(base):                      # (the diagnostic build for the test's duration)
        yield base


@pytest.fixture
def block_height(request, lds_kernel):
    """Pins the LDS-weights kernel's block height (12 or 8 rows) for one test."""
    from soft_contrastive_learning_amd import _lib as L
    with L.variant(lds_kernel + 3000 + request.param):
        yield request.param


@pytest.mark.parametrize('block_height', [12, 13, 8, 6], indirect=True)
@pytest.mark.parametrize('cin,cout,shape', [(128, 256, (2, 12, 40)), (256, 256, (1, 30, 40)),
                                            (256, 512, (1, 15, 80)), (512, 512, (1, 7, 23))])
def test_lds_weight_conv_deeper_layers(dev, cin, cout, shape, block_height):
    """csrc/convg.hip: conv3_x .. conv5_x shapes, forward (+ bias / ReLU) and backward-data."""
    from soft_contrastive_learning_amd.model import nets
    b, h, w = shape
    g = torch.Generator().manual_seed(23)
    x = torch.randn(b, cin, h, w, generator=g).to(dev).bfloat16().contiguous(memory_format=torch.channels_last)
    gy = torch.randn(b, cout, h, w, generator=g).to(dev).bfloat16().contiguous(memory_format=torch.channels_last)
    wt = (torch.randn(cout, cin, 3, 3, generator=g) * 0.03).to(dev).bfloat16().contiguous(
        memory_format=torch.channels_last)
    bias = torch.randn(cout, generator=g).to(dev)
    assert nets._own_conv_kind(x, wt) == 'lds' and nets._own_conv_kind(gy, wt, True) == 'lds'
    z32 = torch.nn.functional.conv2d(x.float(), wt.float(), padding=1)
    scale = float(z32.abs().max())
    got = nets.conv64(x, wt, False)
    assert got.shape == z32.shape and got.is_contiguous(memory_format=torch.channels_last)
    assert float((got.float() - z32).abs().max()) < 6e-3 * scale
    y = nets.conv64(x, wt, False, bias=bias, relu=True)
    assert float((y.float() - torch.relu(z32 + bias[None, :, None, None])).abs().max()) < 6e-3 * scale
    gx = nets.conv64(gy, wt, True)
    want_gx = torch.nn.functional.conv_transpose2d(gy.float(), wt.float(), padding=1)
    assert gx.shape == want_gx.shape
    assert float((gx.float() - want_gx).abs().max()) < 6e-3 * float(want_gx.abs().max())


def test_lds_weight_conv_odd_chunk_count(dev, lds_kernel):
    """cin = 160 is five 32-channel chunks: csrc/convh.hip walks chunks in pairs and leaves such
    shapes to csrc/convg.hip — the forward must be right whichever kernel is pinned."""
    from soft_contrastive_learning_amd.model import nets
    g = torch.Generator().manual_seed(71)
    x = torch.randn(1, 160, 12, 40, generator=g).to(dev).bfloat16().contiguous(memory_format=torch.channels_last)
    wt = (torch.randn(128, 160, 3, 3, generator=g) * 0.03).to(dev).bfloat16().contiguous(
        memory_format=torch.channels_last)
    assert nets._own_conv_kind(x, wt) == 'lds'
    z32 = torch.nn.functional.conv2d(x.float(), wt.float(), padding=1)
    got = nets.conv64(x, wt, False)
    assert float((got.float() - z32).abs().max()) < 6e-3 * float(z32.abs().max())


@pytest.mark.parametrize('cin,cout,shape', [(64, 128, (2, 12, 40)), (128, 128, (1, 13, 37)),
                                            (256, 256, (1, 16, 40)), (512, 512, (1, 7, 23)),
                                            # narrow maps: the 32 x 8 tile shape
                                            (64, 64, (1, 30, 40)), (128, 64, (2, 60, 80)),
                                            (64, 64, (1, 33, 9))])
def test_wrw_other_shapes(dev, cin, cout, shape):
    from soft_contrastive_learning_amd.model import nets
    b, h, w = shape
    g = torch.Generator().manual_seed(29)
    x = torch.randn(b, cin, h, w, generator=g).to(dev).bfloat16().contiguous(memory_format=torch.channels_last)
    gz = torch.randn(b, cout, h, w, generator=g).to(dev).bfloat16().contiguous(memory_format=torch.channels_last)
    wf = torch.zeros(cout, cin, 3, 3, device=dev, requires_grad=True)
    torch.nn.functional.conv2d(x.float(), wf, padding=1).backward(gz.float())
    like = torch.empty(cout, cin, 3, 3, device=dev, dtype=torch.bfloat16).contiguous(
        memory_format=torch.channels_last)
    assert nets._own_wrw_ok(x, gz, like)
    got = nets.wrw64(x, gz, like)
    assert got.shape == wf.grad.shape and got.stride() == like.stride()
    assert float((got.float() - wf.grad).abs().max() / wf.grad.abs().max()) < 6e-3
    assert torch.equal(got, nets.wrw64(x, gz, like))
    # the bias gradient from the same pass (scl_wrw3x3_bias): column sums of gz in float32
    gb = torch.full((cout,), 7.0, device=dev)
    assert torch.equal(nets.wrw64(x, gz, like, gb), got)
    want_gb = gz.float().sum(dim=(0, 2, 3))
    assert float((gb - want_gb).abs().max()) < 1e-4 * float(gz.float().abs().sum(dim=(0, 2, 3)).max())
    gb2 = torch.empty_like(gb)
    nets.wrw64(x, gz, like, gb2)
    assert torch.equal(gb, gb2)                          # fixed summation order


@pytest.mark.parametrize('shape', [(2, 16, 40), (1, 13, 37), (1, 480, 640)])
def test_first_layer_kernel(dev, shape):
    """scl_conv_first: mean subtraction + cast + conv1_1 + bias + ReLU in one pass."""
    from soft_contrastive_learning_amd import _lib as L
    b, h, w = shape
    lib = L.load()
    g = torch.Generator().manual_seed(31)
    img = torch.randint(0, 256, (b, h, w, 3), generator=g).float().to(dev)
    avg = torch.tensor([123.68, 116.78, 103.94], device=dev)
    wt = (torch.randn(64, 3, 3, 3, generator=g) * 0.1).to(dev).bfloat16().contiguous(
        memory_format=torch.channels_last)
    bias = torch.randn(64, generator=g).to(dev)
    x0 = torch.empty((b, h, w, 3), dtype=torch.bfloat16, device=dev)
    y = torch.empty((b, 64, h, w), dtype=torch.bfloat16, device=dev, memory_format=torch.channels_last)
    sk, sc, sh, sw = wt.stride()
    L.check(lib.scl_conv_first(L.ptr(img), L.ptr(avg), L.ptr(wt), sk, sc, sh, sw, 0, L.ptr(bias), b, h, w,
                               L.ptr(x0), L.ptr(y), L.stream_of(img)))
    want_x0 = (img - avg).bfloat16()
    assert torch.equal(x0, want_x0)
    want = torch.relu(torch.nn.functional.conv2d(want_x0.float().permute(0, 3, 1, 2), wt.float(),
                                                 padding=1) + bias[None, :, None, None])
    assert float((y.float() - want).abs().max()) < 6e-3 * float(want.abs().max())


@pytest.mark.parametrize('shape', [(2, 16, 64), (1, 14, 38), (3, 40, 96), (1, 480, 640)])
@pytest.mark.parametrize('w_f32', [False, True])
def test_first_two_layers_in_one_kernel(dev, shape, w_f32):
    """scl_conv_first_pool_idx (conv1_1 + conv1_2 forward, the y1 halo windows computed in LDS)
    against scl_conv_first followed by scl_conv3x3_pool_idx: x0, y1, pooled map and window index
    bit for bit — ragged tiles, image borders, bf16 and float32 master weights."""
    from soft_contrastive_learning_amd import _lib as L
    b, h, w = shape
    lib = L.load()
    g = torch.Generator().manual_seed(131)
    img = torch.randint(0, 256, (b, h, w, 3), generator=g).float().to(dev)
    avg = torch.tensor([123.68, 116.78, 103.94], device=dev)
    cl = torch.channels_last
    w1 = (torch.randn(64, 3, 3, 3, generator=g) * 0.1).to(dev)
    w2 = (torch.randn(64, 64, 3, 3, generator=g) * 0.05).to(dev)
    if not w_f32:
        w1 = w1.bfloat16().contiguous(memory_format=cl)
        w2 = w2.bfloat16().contiguous(memory_format=cl)
    b1 = torch.randn(64, generator=g).to(dev)
    b2 = torch.randn(64, generator=g).to(dev)
    ws = L.workspace(lib.scl_conv3x3_workspace_bytes(), dev)

    def outputs():
        return (torch.full((b, h, w, 3), 7, dtype=torch.bfloat16, device=dev),
                torch.full((b, h, w, 64), 7, dtype=torch.bfloat16, device=dev),
                torch.full((b, h // 2, w // 2, 64), 7, dtype=torch.bfloat16, device=dev),
                torch.full((b, h // 2, w // 2, 64), 9, dtype=torch.uint8, device=dev))

    s1, s2 = w1.stride(), w2.stride()
    f2 = L.W_F32 if w_f32 else 0
    x0, y1, a, idx = outputs()
    L.check(lib.scl_conv_first(L.ptr(img), L.ptr(avg), L.ptr(w1), *s1, int(w_f32), L.ptr(b1), b, h, w,
                               L.ptr(x0), L.ptr(y1), L.stream_of(img)))
    L.check(lib.scl_conv3x3_pool_idx(L.ptr(y1), L.ptr(w2), *s2, f2, b, h, w, 64, 64, L.ptr(b2),
                                     L.ptr(a), L.ptr(idx), L.ptr(ws), ws.numel(), L.stream_of(img)))
    fx0, fy1, fa, fidx = outputs()
    L.check(lib.scl_conv_first_pool_idx(L.ptr(img), L.ptr(avg), L.ptr(w1), *s1, int(w_f32), L.ptr(b1),
                                        L.ptr(w2), *s2, f2, L.ptr(b2), b, h, w, L.ptr(fx0), L.ptr(fy1),
                                        L.ptr(fa), L.ptr(fidx), L.ptr(ws), ws.numel(), L.stream_of(img)))
    torch.cuda.synchronize()
    assert torch.equal(fx0, x0)
    assert torch.equal(fy1, y1)
    assert torch.equal(fa, a)
    assert torch.equal(fidx, idx)
    # and the pair against float32 torch (the fused kernel is not only equal to its sibling)
    want_y1 = torch.relu(torch.nn.functional.conv2d(x0.float().permute(0, 3, 1, 2), w1.float(), padding=1)
                         + b1[None, :, None, None])
    assert float((fy1.float().permute(0, 3, 1, 2) - want_y1).abs().max()) < 6e-3 * float(want_y1.abs().max())
    z = torch.nn.functional.conv2d(fy1.float().permute(0, 3, 1, 2), w2.bfloat16().float(), padding=1)
    want_a = torch.relu(torch.nn.functional.max_pool2d(z, 2) + b2[None, :, None, None])
    assert float((fa.float().permute(0, 3, 1, 2) - want_a).abs().max()) < 6e-3 * float(want_a.abs().max() + 1)


def test_model_with_and_without_the_fused_first_block(dev):
    """nets.USE_FUSED12: features and every gradient of a backbone step are bit-identical with the
    first two layers in one kernel and in two."""
    from soft_contrastive_learning_amd.model import nets
    # (at 480 x 640 every layer runs on the library's own, deterministic kernels)
    img = torch.randint(0, 256, (2, 480, 640, 3), generator=torch.Generator().manual_seed(71)).float().to(dev)
    g = torch.randn(2, 30, 40, 512, generator=torch.Generator().manual_seed(72)).to(dev).bfloat16()
    res = {}
    old = nets.USE_FUSED12
    try:
        for fused in (False, True):
            nets.USE_FUSED12 = fused
            model = nets.VGG16NetVLAD(compute_dtype=torch.bfloat16, seed=9, fused_relu=True).to(dev)
            f = model.features(img)
            f.backward(g)
            res[fused] = (f.detach().clone(), {n: p.grad.clone() for n, p in model.named_parameters()
                                               if p.grad is not None})
    finally:
        nets.USE_FUSED12 = old
    assert torch.equal(res[True][0], res[False][0])
    assert set(res[True][1]) == set(res[False][1])
    for n in res[True][1]:
        assert torch.equal(res[True][1][n], res[False][1][n]), n


@pytest.mark.parametrize('block_height', [12, 13, 8, 6], indirect=True)
@pytest.mark.parametrize('cin,cout,shape', [(64, 64, (2, 16, 40)), (128, 128, (1, 13, 37)),
                                            (128, 64, (1, 9, 33)), (256, 128, (2, 12, 40)),
                                            (512, 256, (1, 15, 80))])
def test_backward_data_with_relu_mask_in_the_epilogue(dev, cin, cout, shape, block_height):
    """scl_conv3x3_masked / scl_convg_masked: gx * [y > 0] must be the plain kernel's gx with
    the mask applied afterwards, bit for bit (zeros, negative zeros and negatives in y cut)."""
    from soft_contrastive_learning_amd.model import nets
    b, h, w = shape
    g = torch.Generator().manual_seed(37)
    # gradient w.r.t. the input of conv(cout <- cin ... ) seen from above: gz has `cin` channels
    gz = torch.randn(b, cin, h, w, generator=g).to(dev).bfloat16().contiguous(memory_format=torch.channels_last)
    wt = (torch.randn(cin, cout, 3, 3, generator=g) * 0.05).to(dev).bfloat16().contiguous(
        memory_format=torch.channels_last)
    y = torch.randn(b, cout, h, w, generator=g)
    y[y.abs() < 0.3] = 0.0
    y[(y > 1.0)] = -0.0
    y = y.to(dev).bfloat16().contiguous(memory_format=torch.channels_last)
    plain = nets.conv64(gz, wt, True)
    got = nets.conv64(gz, wt, True, mask=y)
    want = torch.where(y > 0, plain, torch.zeros_like(plain))
    assert got.shape == want.shape and got.is_contiguous(memory_format=torch.channels_last)
    assert torch.equal(got, want)
    assert float((want != 0).float().mean()) > 0.1


def test_masked_backward_chain_equals_separate_masking_pass(dev):
    """The hand-off of ReLU' to the layer above (nets._GradLink) must not change a gradient:
    a three-layer chain of own kernels gives bit-identical input and weight gradients with and
    without it, and the lower layers really skip their masking pass.  The bias gradients of the
    handed-off layers come out of the weight-gradient kernel instead of the masking pass: the
    same float32 column sums in another order."""
    from soft_contrastive_learning_amd.model import nets
    g = torch.Generator().manual_seed(41)
    cl = torch.channels_last
    x0 = torch.randn(1, 64, 60, 80, generator=g).to(dev).bfloat16().contiguous(memory_format=cl)
    ws = [(torch.randn(co, ci, 3, 3, generator=g) * (2.0 / (9 * ci)) ** 0.5).to(dev).bfloat16()
          .contiguous(memory_format=cl) for ci, co in ((64, 64), (64, 128), (128, 128))]
    bs = [(torch.randn(co, generator=g) * 0.1).to(dev) for co in (64, 128, 128)]
    gout = torch.randn(1, 128, 30, 40, generator=g).to(dev).bfloat16().contiguous(memory_format=cl)

    def run(linked):
        x = x0.clone().requires_grad_(True)
        w = [t.clone().requires_grad_(True) for t in ws]
        b = [t.clone().requires_grad_(True) for t in bs]
        l1 = nets._GradLink() if linked else None
        l2 = nets._GradLink() if linked else None
        y = nets._ConvBiasAct.apply(x, w[0], b[0], True, None, l1)
        y = nets._ConvBiasAct.apply(y, w[1], b[1], True, l1, l2)
        y = nets._ConvBiasPoolReLU.apply(y, w[2], b[2], l2)
        y.backward(gout)
        return [x.grad] + [t.grad for t in w] + [t.grad for t in b]

    takes = []
    orig = nets._GradLink.take

    def counting_take(self, gy):
        hit = orig(self, gy)
        takes.append(hit)
        return hit
    nets._GradLink.take = counting_take
    try:
        linked = run(True)
    finally:
        nets._GradLink.take = orig
    assert takes == [True, True]
    plain = run(False)
    for a, b_ in zip(linked[:4], plain[:4]):
        assert torch.equal(a, b_)
    for a, b_ in zip(linked[4:], plain[4:]):
        assert float((a - b_).abs().max()) <= 1e-5 * float(b_.abs().max()) + 1e-6
    assert torch.equal(linked[6], plain[6])              # pooled layer: same kernel either way


@pytest.mark.parametrize('shape', [(2, 16, 40), (1, 13, 37), (2, 120, 160)])
def test_first_layer_weight_and_bias_gradient(dev, shape):
    """scl_conv_first_wrw: conv1_1's weight gradient and bias gradient from one pass over gz."""
    from soft_contrastive_learning_amd.model import nets
    b, h, w = shape
    g = torch.Generator().manual_seed(43)
    x0 = torch.randn(b, h, w, 3, generator=g).to(dev).bfloat16().permute(0, 3, 1, 2)
    gz = torch.randn(b, 64, h, w, generator=g).to(dev).bfloat16().contiguous(memory_format=torch.channels_last)
    gz[gz.abs() < 0.5] = 0
    wf = torch.zeros(64, 3, 3, 3, device=dev, requires_grad=True)
    bf = torch.zeros(64, device=dev, requires_grad=True)
    torch.nn.functional.conv2d(x0.float(), wf, bf, padding=1).backward(gz.float())
    like = torch.empty(64, 3, 3, 3, device=dev, dtype=torch.bfloat16).contiguous(memory_format=torch.channels_last)
    gb = torch.empty(64, device=dev)
    gw = nets.first_wrw(x0, gz, like, gb)
    assert gw.shape == wf.grad.shape and gw.stride() == like.stride()
    assert float((gw.float() - wf.grad).abs().max() / wf.grad.abs().max()) < 6e-3
    assert float((gb - bf.grad).abs().max() / bf.grad.abs().max()) < 1e-5
    gb2 = torch.empty(64, device=dev)
    assert torch.equal(gw, nets.first_wrw(x0, gz, like, gb2)) and torch.equal(gb, gb2)
    # the gradient of the trainable mean from the same pass == the closed form in torch ==
    # minus the spatial sum of the first layer's input gradient
    wt = (torch.randn(64, 3, 3, 3, generator=g) * 0.1).to(dev).bfloat16().contiguous(memory_format=torch.channels_last)
    gw3, davg = nets.first_wrw(x0, gz, like, gb2, wt)
    assert torch.equal(gw3, gw)
    want = nets.avg_rgb_grad(gz, wt, gb)
    gx = torch.nn.functional.conv_transpose2d(gz.float(), wt.float(), padding=1)
    direct = -gx.sum(dim=(0, 2, 3))
    scale = float(direct.abs().max())
    assert float((davg - want).abs().max()) < 2e-4 * scale + 1e-3
    assert float((davg - direct).abs().max()) < 2e-3 * scale + 1e-2


@pytest.mark.parametrize('cin,shape', [(64, (2, 16, 40)), (128, (1, 10, 38)), (64, (1, 13, 37)),
                                       (256, (1, 24, 80)), (512, (1, 13, 37)), (256, (2, 8, 40))])
@pytest.mark.parametrize('block_height', [12, 13, 8, 6], indirect=True)
def test_pool_index_epilogue_and_its_backward(dev, cin, shape, block_height):
    """scl_conv3x3_pool_idx + scl_vgg_pool_bwd_idx: pooled map as the fused-tail kernel gives
    it, every stored position points at a maximum of its window, and the backward routes
    g * [a > 0] there (and only there)."""
    from soft_contrastive_learning_amd import _lib as L
    from soft_contrastive_learning_amd.model import nets
    lib = L.load()
    b, h, w = shape
    g = torch.Generator().manual_seed(47)
    x = torch.randn(b, cin, h, w, generator=g).to(dev).bfloat16().contiguous(memory_format=torch.channels_last)
    wt = (torch.randn(cin, cin, 3, 3, generator=g) * 0.05).to(dev).bfloat16()
    bias = torch.randn(cin, generator=g).to(dev) * 0.2
    a, idx = nets.conv_pool_idx(x, wt, bias)
    assert idx.dtype == torch.uint8 and int(idx.max()) <= 3
    ho, wo = h // 2, w // 2
    if cin <= 128:
        z, a_ref = nets.conv64(x, wt, False, bias=bias, pool=True)
        assert torch.equal(a, a_ref)
    z32 = torch.nn.functional.conv2d(x.float(), wt.float(), padding=1)[:, :, :2 * ho, :2 * wo]
    want_a = torch.relu(torch.nn.functional.max_pool2d(z32, 2) + bias[None, :, None, None])
    assert float((a.float() - want_a).abs().max()) < 6e-3 * float(z32.abs().max())
    win = z32.reshape(b, cin, ho, 2, wo, 2).permute(0, 1, 2, 4, 3, 5).reshape(b, cin, ho, wo, 4)
    picked = torch.gather(win, 4, idx.long().unsqueeze(-1)).squeeze(-1)
    scale = float(z32.abs().max())
    assert float((win.max(dim=4).values - picked).abs().max()) < 1e-3 * scale
    # backward
    ga = torch.randn(b, cin, ho, wo, generator=g).to(dev).bfloat16().contiguous(memory_format=torch.channels_last)
    gz = torch.empty((b, cin, h, w), dtype=torch.bfloat16, device=dev, memory_format=torch.channels_last)
    gz.fill_(7.0)                                     # uncovered borders must be zeroed
    gb = torch.empty(cin, device=dev)
    ws = L.workspace(lib.scl_vgg_workspace_bytes(cin), dev)
    L.check(lib.scl_vgg_pool_bwd_idx(L.ptr(ga), L.ptr(a), L.ptr(idx), L.DT_BF16, b, h, w, cin,
                                     L.ptr(gz), L.ptr(gb), L.ptr(ws), ws.numel(), L.stream_of(ga)))
    gg = torch.where(a > 0, ga, torch.zeros_like(ga)).float()
    want = torch.zeros(b, cin, h, w, device=dev)
    onehot = torch.nn.functional.one_hot(idx.long(), 4).float() * gg.unsqueeze(-1)   # [b,c,ho,wo,4]
    want[:, :, :2 * ho, :2 * wo] = onehot.reshape(b, cin, ho, wo, 2, 2).permute(0, 1, 2, 4, 3, 5).reshape(
        b, cin, 2 * ho, 2 * wo)
    assert torch.equal(gz.float(), want)
    assert float((gb - gg.sum(dim=(0, 2, 3))).abs().max()) < 1e-3 * float(gg.abs().sum(dim=(0, 2, 3)).max())


def _unpooled(ga, idx, h, w):
    """scl_vgg_pool_bwd_idx on an already masked pooled gradient: the full-size map + bias gradient."""
    from soft_contrastive_learning_amd import _lib as L
    lib = L.load()
    b, c = ga.shape[0], ga.shape[1]
    gz = torch.empty((b, c, h, w), dtype=torch.bfloat16, device=ga.device, memory_format=torch.channels_last)
    gb = torch.empty(c, device=ga.device)
    ws = L.workspace(lib.scl_vgg_workspace_bytes(c), ga.device)
    L.check(lib.scl_vgg_pool_bwd_idx(L.ptr(ga), None, L.ptr(idx), L.DT_BF16, b, h, w, c,
                                     L.ptr(gz), L.ptr(gb), L.ptr(ws), ws.numel(), L.stream_of(ga)))
    return gz, gb


@pytest.mark.parametrize('cin,cout,shape', [(64, 64, (2, 16, 40)), (64, 64, (1, 10, 38)),
                                            (128, 128, (1, 14, 70)), (256, 256, (1, 24, 80)),
                                            (512, 512, (2, 12, 16)), (128, 256, (1, 30, 40)),
                                            (64, 128, (3, 36, 80))])
def test_weight_gradient_unpools_the_pooled_gradient_in_staging(dev, cin, cout, shape):
    """scl_wrw3x3_pooled(x, g_pooled, idx) == scl_wrw3x3_bias(x, un-pooled g) bit for bit (weights
    and bias gradient): the kernel builds the same LDS tile from a quarter of the bytes."""
    from soft_contrastive_learning_amd.model import nets
    b, h, w = shape
    g = torch.Generator().manual_seed(61)
    x = torch.randn(b, cin, h, w, generator=g).to(dev).bfloat16().contiguous(memory_format=torch.channels_last)
    ga = torch.randn(b, cout, h // 2, w // 2, generator=g).to(dev).bfloat16().contiguous(
        memory_format=torch.channels_last)
    idx = torch.randint(0, 4, (b, cout, h // 2, w // 2), generator=g, dtype=torch.uint8).to(dev).contiguous(
        memory_format=torch.channels_last)
    gz, gb_pass = _unpooled(ga, idx, h, w)
    for wdtype in (torch.float32, torch.bfloat16):
        w_like = torch.empty(cout, cin, 3, 3, device=dev, dtype=wdtype)
        gb0, gb1 = torch.empty(cout, device=dev), torch.empty(cout, device=dev)
        want = nets.wrw64(x, gz, w_like, gb0)
        got = nets.wrw64(x, ga, w_like, gb1, pool_idx=idx)
        assert torch.equal(got, want)
        assert torch.equal(gb1, gb0)
    assert float((gb1 - gb_pass).abs().max()) <= 1e-3 * float(gb_pass.abs().max())
    with pytest.raises(ValueError):
        nets.wrw64(x, ga[:, :, :-1], w_like, None, pool_idx=idx[:, :, :-1])


@pytest.mark.parametrize('c,shape', [(64, (2, 16, 40)), (64, (1, 10, 38)), (64, (3, 36, 80)),
                                     (64, (1, 2, 2)), (128, (1, 14, 70)), (128, (2, 8, 64)),
                                     (128, (1, 30, 40))])
def test_backward_data_unpools_the_pooled_gradient_in_staging(dev, c, shape):
    """scl_conv3x3_masked_pooled(g_pooled, idx) == scl_conv3x3_masked(un-pooled g) bit for bit:
    the threads build the same halo window (zero borders, odd window origin) from the pooled map."""
    from soft_contrastive_learning_amd.model import nets
    b, h, w = shape
    g = torch.Generator().manual_seed(67)
    ga = torch.randn(b, c, h // 2, w // 2, generator=g).to(dev).bfloat16().contiguous(
        memory_format=torch.channels_last)
    idx = torch.randint(0, 4, (b, c, h // 2, w // 2), generator=g, dtype=torch.uint8).to(dev).contiguous(
        memory_format=torch.channels_last)
    mask = torch.relu(torch.randn(b, c, h, w, generator=g)).to(dev).bfloat16().contiguous(
        memory_format=torch.channels_last)
    wt = (torch.randn(c, c, 3, 3, generator=g) * 0.05).to(dev)
    gz, _ = _unpooled(ga, idx, h, w)
    want = nets.conv64(gz, wt, True, mask=mask)
    got = nets.conv64(ga, wt, True, mask=mask, pool_idx=idx)
    assert torch.equal(got, want)
    assert float(want.float().abs().max()) > 0
    with pytest.raises(ValueError):
        nets.conv64(ga, wt, True, pool_idx=idx)              # un-masked form does not exist


@pytest.mark.parametrize('c,shape', [(64, (1, 30, 40)), (64, (2, 32, 48)), (128, (1, 30, 40)), (256, (2, 30, 40)),
                                     (512, (1, 30, 40))])
def test_pooling_layer_backward_is_the_same_with_and_without_the_unpooling_pass(dev, c, shape):
    """SCL_POOLED_BWD: the layer's backward through autograd — input, weight and bias gradient —
    with the consumers un-pooling while they stage (conv1_2 / conv2_2: no full-size gradient at
    all; conv3_3 / conv4_3: the weight gradient alone) equals the two-pass form bit for bit (the
    bias gradient, where another kernel takes it: to summation order)."""
    from soft_contrastive_learning_amd.model import nets
    b, h, w = shape
    g = torch.Generator().manual_seed(71)
    x0 = torch.relu(torch.randn(b, c, h, w, generator=g)).to(dev).bfloat16().contiguous(
        memory_format=torch.channels_last)
    wt0 = (torch.randn(c, c, 3, 3, generator=g) * (2.0 / (9 * c)) ** 0.5).to(dev)
    b0 = (torch.randn(c, generator=g) * 0.1).to(dev)
    gy0 = torch.randn(b, c, h // 2, w // 2, generator=g).to(dev).bfloat16().contiguous(
        memory_format=torch.channels_last)
    from soft_contrastive_learning_amd import _lib as L
    results, kernels = [], []
    keep = nets.USE_POOLED_BWD
    try:
        for mode in (True, False):
            nets.USE_POOLED_BWD = mode
            x = x0.clone().requires_grad_(True)
            wt, bias = torch.nn.Parameter(wt0.clone()), torch.nn.Parameter(b0.clone())
            link_in, link_out = nets._GradLink(), nets._GradLink()
            y = nets._ConvBiasPoolReLU.apply(x, wt, bias, link_in, link_out)
            assert y.grad_fn.by_idx
            gy = torch.where(y.detach() > 0, gy0, torch.zeros_like(gy0)).contiguous(
                memory_format=torch.channels_last)
            link_out.mark(gy)                           # "the layer above masked it"
            with L.KernelTimer(capacity=64) as kt:
                y.backward(gy)
                torch.cuda.synchronize()
            assert link_in.ptr is not None
            results.append((x.grad.clone(), wt.grad.clone(), bias.grad.clone()))
            kernels.append(set(kt.summary()))
    finally:
        nets.USE_POOLED_BWD = keep
    # what ran: the pooled weight gradient in the first pass only; no un-pooling pass where the
    # backward-data kernel un-pools too (the register kernels: 64 and 128 channels)
    assert 'wrw64_kernel<pooled>' in kernels[0] and 'wrw64_kernel<pooled>' not in kernels[1]
    assert 'pool_bwd_idx_kernel' in kernels[1]
    assert ('pool_bwd_idx_kernel' in kernels[0]) == (c > 128)
    assert ('conv3x3_kernel<pooled>' in kernels[0]) == (c <= 128)
    (gx0, gw0, gb0), (gx1, gw1, gb1) = results
    assert torch.equal(gx0, gx1) and torch.equal(gw0, gw1)
    assert float(gw0.abs().max()) > 0
    # the bias gradient is the same sum taken by another kernel (the weight-gradient kernel's
    # side product instead of the un-pooling pass's column sums): another order of additions
    if c > 128:
        assert torch.equal(gb0, gb1)
    else:
        assert float((gb0 - gb1).abs().max()) <= 2e-5 * float(gy0.float().abs().sum(dim=(0, 2, 3)).max())


@pytest.mark.parametrize('cin,cout,shape', [(64, 64, (1, 16, 40)), (128, 128, (1, 9, 33)),
                                            (256, 256, (1, 12, 40)), (128, 256, (1, 30, 40))])
def test_float32_master_weights_equal_the_bf16_cast(dev, cin, cout, shape, lds_kernel):
    """SCL_W_F32: the kernels round the float32 weight to bf16 while packing — results must be
    bit-identical to passing the bf16 cast, and the float32 weight gradient must round to the
    bf16 one."""
    from soft_contrastive_learning_amd.model import nets
    b, h, w = shape
    g = torch.Generator().manual_seed(53)
    x = torch.randn(b, cin, h, w, generator=g).to(dev).bfloat16().contiguous(memory_format=torch.channels_last)
    gy = torch.randn(b, cout, h, w, generator=g).to(dev).bfloat16().contiguous(memory_format=torch.channels_last)
    w32 = (torch.randn(cout, cin, 3, 3, generator=g) * 0.05).to(dev)          # contiguous OIHW master
    w16 = w32.to(dtype=torch.bfloat16, memory_format=torch.channels_last)
    bias = torch.randn(cout, generator=g).to(dev)
    assert torch.equal(nets.conv64(x, w32, False), nets.conv64(x, w16, False))
    assert torch.equal(nets.conv64(x, w32, False, bias=bias, relu=True),
                       nets.conv64(x, w16, False, bias=bias, relu=True))
    assert torch.equal(nets.conv64(gy, w32, True), nets.conv64(gy, w16, True))
    y = torch.relu(torch.randn(b, cin, h, w, generator=g)).to(dev).bfloat16().contiguous(
        memory_format=torch.channels_last)
    assert torch.equal(nets.conv64(gy, w32, True, mask=y), nets.conv64(gy, w16, True, mask=y))
    if cin == cout and cin <= 128:
        a32, i32 = nets.conv_pool_idx(x, w32, bias)
        a16, i16 = nets.conv_pool_idx(x, w16, bias)
        assert torch.equal(a32, a16) and torch.equal(i32, i16)
    g32 = nets.wrw64(x, gy, w32)
    g16 = nets.wrw64(x, gy, w16)
    assert g32.dtype == torch.float32 and g32.stride() == w32.stride()
    assert torch.equal(g32.to(torch.bfloat16), g16)


def test_first_layer_with_float32_master_weights(dev):
    from soft_contrastive_learning_amd.model import nets
    g = torch.Generator().manual_seed(59)
    img = torch.randint(0, 256, (1, 32, 48, 3), generator=g).float().to(dev)
    avg = torch.tensor([123.68, 116.78, 103.94], device=dev)
    w32 = (torch.randn(64, 3, 3, 3, generator=g) * 0.1).to(dev)
    w16 = w32.to(dtype=torch.bfloat16, memory_format=torch.channels_last)
    bias = torch.randn(64, generator=g).to(dev)
    gy = torch.randn(1, 64, 32, 48, generator=g).to(dev).bfloat16().contiguous(memory_format=torch.channels_last)
    outs = []
    for wt in (w32, w16):
        a_ = avg.clone().requires_grad_(True)
        w_ = wt.clone().requires_grad_(True)
        b_ = bias.clone().requires_grad_(True)
        y = nets._FirstConv.apply(img, a_, w_, b_, torch.bfloat16, None)
        y.backward(gy)
        outs.append((y.detach(), a_.grad, w_.grad, b_.grad))
    assert torch.equal(outs[0][0], outs[1][0])
    assert outs[0][2].dtype == torch.float32 and torch.equal(outs[0][2].to(torch.bfloat16), outs[1][2])
    assert torch.equal(outs[0][3], outs[1][3])
    assert float((outs[0][1] - outs[1][1]).abs().max()) <= 1e-4 * float(outs[1][1].abs().max()) + 1e-5


def test_gradient_sink_gives_the_same_gradients(dev):
    """nets.GRAD_SINK (parallel.GradBuckets): conv weight / bias gradients written straight
    into the flat buffer must equal the ones that travel through autograd — bit for bit where
    every kernel on the way is deterministic (at 480 x 640 all of conv1 .. conv4 are own
    kernels; the library's conv5_x forward differs in the last bits from run to run)."""
    from soft_contrastive_learning_amd import parallel
    from soft_contrastive_learning_amd.model import nets
    img = torch.randint(0, 256, (1, 480, 640, 3), generator=torch.Generator().manual_seed(61)).float().to(dev)
    g = torch.randn(1, 30, 40, 512, generator=torch.Generator().manual_seed(62)).to(dev).bfloat16()
    grads = {}
    # (the fused first-layer gradients follow the sink + second stream by default and sum conv1_1's
    # gradients in another order: pinned off here, compared on their own further down)
    old_ffw, nets.USE_FUSED_FIRST_WRW = nets.USE_FUSED_FIRST_WRW, False
    try:
        for sink in (False, True):
            model = nets.VGG16NetVLAD(compute_dtype=torch.bfloat16, seed=9, fused_relu=True).to(dev)
            buckets = parallel.GradBuckets(list(model.parameters()))
            nets.GRAD_SINK = buckets if sink else None
            try:
                buckets.zero()
                model.features(img).backward(g)
            finally:
                nets.GRAD_SINK = None
            grads[sink] = {n: p.grad.clone() for n, p in model.named_parameters() if p.grad is not None}
    finally:
        nets.USE_FUSED_FIRST_WRW = old_ffw
    assert set(grads[True]) == set(grads[False])
    for n in grads[True]:
        if n.startswith(('conv1_', 'conv2_', 'conv3_', 'conv4_', 'average_rgb')):
            assert torch.equal(grads[True][n], grads[False][n]), n
        else:
            assert _nrel(grads[True][n], grads[False][n]) < 1e-3, n


def test_lds_weight_conv_persistent_workgroups_equal_one_tile_per_workgroup(dev, lds_kernel):
    """csrc/convg.hip and csrc/convh.hip run persistent workgroups that prefetch the next tile's first stage under
    the epilogue of the current one.  Every epilogue variant must give bit-identical results
    with several tiles per workgroup (grid pinned to one group of 8 x kout/128, padding tiles
    included) and with one tile per workgroup."""
    from soft_contrastive_learning_amd import _lib as L
    from soft_contrastive_learning_amd.model import nets
    lib = L.load()
    g = torch.Generator().manual_seed(41)
    b, h, w, cin, cout = 3, 36, 80, 128, 256
    x = torch.randn(b, cin, h, w, generator=g).to(dev).bfloat16().contiguous(memory_format=torch.channels_last)
    gy = torch.randn(b, cout, h, w, generator=g).to(dev).bfloat16().contiguous(memory_format=torch.channels_last)
    wt = (torch.randn(cout, cin, 3, 3, generator=g) * 0.03).to(dev).bfloat16().contiguous(
        memory_format=torch.channels_last)
    bias = torch.randn(cout, generator=g).to(dev)
    mask = torch.randn(b, cin, h, w, generator=g).to(dev).bfloat16().contiguous(memory_format=torch.channels_last)

    def run_all():
        return (nets.conv64(x, wt, False), nets.conv64(x, wt, False, bias=bias, relu=True),
                nets.conv64(gy, wt, True), nets.conv64(gy, wt, True, mask=mask),
                *nets.conv_pool_idx(x, wt, bias))
    outs = {}
    for variant in (3099, 3100, 3101, 0):
        with L.variant(lds_kernel + variant):
            outs[variant] = [t.clone() for t in run_all()]
    for variant in (3100, 3101, 0):
        for a, bb in zip(outs[3099], outs[variant]):
            assert torch.equal(a, bb), variant
    z32 = torch.nn.functional.conv2d(x.float(), wt.float(), padding=1)
    assert float((outs[0][0].float() - z32).abs().max()) < 6e-3 * float(z32.abs().max())


def test_prepacked_weight_images_equal_self_packing(dev):
    """nets.prepack (scl_conv_pack_batch: every packed weight image of a step in one launch)
    + SCL_W_PACKED must give the same bits as each convolution packing for itself — register
    and LDS-weights kernels, forward (all tails) and backward-data, float32 master and bf16
    weights — and a weight changed in place must not be served from a stale image."""
    from soft_contrastive_learning_amd.model import nets
    g = torch.Generator().manual_seed(83)
    cl = torch.channels_last
    cases = [(64, 64, torch.float32), (64, 128, torch.bfloat16), (128, 128, torch.float32),
             (128, 256, torch.float32), (256, 256, torch.bfloat16), (512, 512, torch.float32)]
    for cin, cout, wdt in cases:
        b, h, w = 1, 14, 40
        x = torch.randn(b, cin, h, w, generator=g).to(dev).bfloat16().contiguous(memory_format=cl)
        gy = torch.randn(b, cout, h, w, generator=g).to(dev).bfloat16().contiguous(memory_format=cl)
        wt = (torch.randn(cout, cin, 3, 3, generator=g) * 0.03).to(dev)
        wt = wt.bfloat16().contiguous(memory_format=cl) if wdt == torch.bfloat16 else wt
        bias = torch.randn(cout, generator=g).to(dev)
        mask = torch.randn(b, cin, h, w, generator=g).to(dev).bfloat16().contiguous(memory_format=cl)

        def run_all():
            outs = [nets.conv64(x, wt, False), nets.conv64(x, wt, False, bias=bias, relu=True),
                    nets.conv64(gy, wt, True), nets.conv64(gy, wt, True, mask=mask)]
            if cin == cout:
                outs += list(nets.conv_pool_idx(x, wt, bias))
            return [t.clone() for t in outs]
        nets._PACKED.clear()
        plain = run_all()
        assert nets.prepack([wt]) == 2 and nets.prepack([wt]) == 0
        # more jobs than one launch takes (32): the batch entry chunks them
        many = [(torch.randn(128, 64, 3, 3, generator=g) * 0.03).to(dev) for _ in range(18)]
        assert nets.prepack(many) == 36
        ref = nets.conv64(x[:, :64].contiguous(memory_format=cl), many[17], False) if cin >= 64 else None
        if ref is not None:
            nets._PACKED.pop((many[17].data_ptr(), False, nets._pack_slot()))
            assert torch.equal(ref, nets.conv64(x[:, :64].contiguous(memory_format=cl), many[17], False))
        assert nets._packed_for(wt, False) is not None and nets._packed_for(wt, True) is not None
        packed = run_all()
        for a, bb in zip(plain, packed):
            assert torch.equal(a, bb), (cin, cout, wdt)
        wt.mul_(0.5)                                       # in place: the images are stale now
        assert nets._packed_for(wt, False) is None
        halved = run_all()
        assert not torch.equal(halved[0], plain[0])
        assert nets.prepack([wt]) == 2
        for a, bb in zip(halved, run_all()):
            assert torch.equal(a, bb), (cin, cout, wdt)
    nets._PACKED.clear()


def test_weight_gradients_on_the_second_stream_equal_the_one_stream_run(dev):
    """nets.USE_SIDE_WRW: with the gradient sink, the weight-gradient kernels run on a second
    stream next to the backward-data kernels and GradBuckets joins that stream in finish().
    Same kernels, same inputs: every gradient in the flat buffer must be bit-identical to the
    one-stream run (the library's conv5_x kernels are not involved at 480 x 640), three times
    in a row (stale events, early reuse of a gradient map by the allocator)."""
    from soft_contrastive_learning_amd import parallel
    from soft_contrastive_learning_amd.model import nets
    img = torch.randint(0, 256, (2, 480, 640, 3), generator=torch.Generator().manual_seed(91)).float().to(dev)
    g = torch.randn(2, 30, 40, 512, generator=torch.Generator().manual_seed(92)).to(dev).bfloat16()
    model = nets.VGG16NetVLAD(compute_dtype=torch.bfloat16, seed=11, fused_relu=True).to(dev)
    buckets = parallel.GradBuckets(list(model.parameters()))
    flats = {}
    old = nets.USE_SIDE_WRW
    old_ffw, nets.USE_FUSED_FIRST_WRW = nets.USE_FUSED_FIRST_WRW, False      # (as above)
    try:
        for side in (False, True, True, True):
            nets.USE_SIDE_WRW = side
            nets.GRAD_SINK = buckets
            try:
                buckets.zero()
                model.features(img).backward(g)
                buckets.finish()
            finally:
                nets.GRAD_SINK = None
            torch.cuda.synchronize()
            if side:
                assert len(buckets._streams) == 1
                assert torch.equal(buckets.flat, flats[False])
            else:
                flats[False] = buckets.flat.clone()
    finally:
        nets.USE_SIDE_WRW = old
        nets.USE_FUSED_FIRST_WRW = old_ffw
    assert float(flats[False].abs().max()) > 0


@pytest.mark.parametrize('shape', [(4, 480, 640), (3, 480, 640), (3, 96, 160), (2, 64, 80)])
def test_half_batches_on_two_streams_equal_the_one_stream_forward(dev, shape):
    """nets.USE_SPLIT_FWD: features() pipelines the two halves of the batch on two streams (own
    kernels split, library / glue ops joined around).  Same kernels per image: descriptors and
    every gradient must be bit-identical to the one-stream run — with autograd and under
    no_grad (where activations are freed while the half-batch streams may still read them),
    for an odd batch, and at sizes where the last layers fall back to the library."""
    from soft_contrastive_learning_amd.model import nets
    b, h, w = shape
    img = torch.randint(0, 256, (b, h, w, 3), generator=torch.Generator().manual_seed(101)).float().to(dev)
    model = nets.VGG16NetVLAD(compute_dtype=torch.bfloat16, seed=13, fused_relu=True).to(dev)
    g = torch.randn(b, h // 16, w // 16, 512, generator=torch.Generator().manual_seed(102)).to(dev).bfloat16()
    old = nets.USE_SPLIT_FWD
    res = {}
    try:
        for split in (False, True):
            nets.USE_SPLIT_FWD = split
            model.zero_grad(set_to_none=True)
            y = model.features(img)
            y.backward(g)
            with torch.no_grad():
                outs = [model.features(img).clone() for _ in range(3)]
            torch.cuda.synchronize()
            res[split] = (y.detach().clone(), {n: p.grad.clone() for n, p in model.named_parameters()
                                               if p.grad is not None}, outs)
    finally:
        nets.USE_SPLIT_FWD = old
    lib_tail = h * w < 480 * 640       # conv5_x (and more) on the library: not bit-reproducible
    for o in res[True][2] + [res[True][0]]:
        if lib_tail:
            assert _nrel(o.float(), res[False][0].float()) < 2e-2
        else:
            assert torch.equal(o, res[False][0])
    assert set(res[True][1]) == set(res[False][1])
    for n in res[True][1]:
        if not lib_tail:
            assert torch.equal(res[True][1][n], res[False][1][n]), n
        elif n != 'average_rgb':      # (a sum with heavy cancellation: the library's noise shows)
            assert _nrel(res[True][1][n], res[False][1][n]) < 0.15, n


# ---- round 5: conv1_2's backward-data pass computes conv1_1's parameter gradients -------------------
def _first_block_grads(dev, img, seed, fused):
    """conv1_1 -> ReLU -> conv1_2 -> pool -> ReLU as the model builds it; backward from a fixed
    gradient at the pooled map.  Returns (gw1, gb1, davg, gw2, gb2)."""
    from soft_contrastive_learning_amd import parallel
    from soft_contrastive_learning_amd.model import nets
    g = torch.Generator().manual_seed(seed)
    avg = torch.nn.Parameter(torch.tensor([123.68, 116.78, 103.94], device=dev))
    w1 = torch.nn.Parameter((torch.randn(64, 3, 3, 3, generator=g) * (2.0 / 27) ** 0.5).to(dev))
    b1 = torch.nn.Parameter((torch.randn(64, generator=g) * 0.1).to(dev))
    w2 = torch.nn.Parameter((torch.randn(64, 64, 3, 3, generator=g) * (2.0 / 576) ** 0.5).to(dev))
    b2 = torch.nn.Parameter((torch.randn(64, generator=g) * 0.1).to(dev))
    params = [avg, w1, b1, w2, b2]
    buckets = parallel.GradBuckets(params)
    old = nets.USE_FUSED_FIRST_WRW
    nets.USE_FUSED_FIRST_WRW = fused
    nets.GRAD_SINK = buckets
    try:
        assert nets.prepack([w2], force=True) == 2
        l1, l2 = nets._GradLink(), nets._GradLink()
        y1 = nets._FirstConv.apply(img, avg, w1, b1, torch.bfloat16, l1)
        a = nets._ConvBiasPoolReLU.apply(y1, w2, b2, l1, l2)
        ga = torch.randn(a.shape, generator=g).to(dev).bfloat16().contiguous(memory_format=torch.channels_last)
        ga = torch.where(a.detach() > 0, ga, torch.zeros_like(ga)).contiguous(memory_format=torch.channels_last)
        l2.mark(ga)
        buckets.zero()
        a.backward(ga)
        buckets.finish()
        torch.cuda.synchronize()
    finally:
        nets.GRAD_SINK = None
        nets.USE_FUSED_FIRST_WRW = old
    return [p.grad.detach().clone() for p in (w1, b1, avg, w2, b2)]


@pytest.mark.parametrize('shape', [(24, 480, 640), (3, 36, 70), (2, 32, 64), (5, 180, 240), (40, 2, 2)])
def test_first_layer_gradients_from_conv1_2_backward_kernel(dev, shape):
    """scl_conv3x3_masked_pooled_first_wrw: conv1_2's backward-data kernel keeps the masked tile of
    the gradient at conv1_1's pre-activation in LDS and multiplies it with the im2col of x0 right
    there (weight, bias and the five columns of the mean gradient's closed form); the 944 MB map
    is neither written nor read.  Against the two-kernel path (scl_conv3x3_masked_pooled +
    scl_conv_first_wrw, validated against float32 torch in test_gpu_conv_bench_shapes.py) at the
    bench launch shape and at ragged ones (partial tiles, one tile, the reference's 240 x 180, a
    2 x 2 image whose four pixels are all corners — 40 of them, so that the own weight-gradient
    kernel and with it the fused path is taken): the same sums in another order."""
    b, h, w = shape
    img = torch.randint(0, 256, (b, h, w, 3), generator=torch.Generator().manual_seed(5)).float().to(dev)
    two = _first_block_grads(dev, img, 17, False)
    one = _first_block_grads(dev, img, 17, True)
    names = ('conv1_1 weight', 'conv1_1 bias', 'average_rgb', 'conv1_2 weight', 'conv1_2 bias')
    for n, a, bb in zip(names, one, two):
        if n.startswith('conv1_2'):
            assert torch.equal(a, bb), n                     # untouched by the fusion
            continue
        rel = float((a.double() - bb.double()).norm() / bb.double().norm().clamp_min(1e-30))
        assert rel < (2e-5 if n != 'average_rgb' else 2e-4), (n, rel, shape)
        assert float((a - bb).abs().max()) <= 1e-4 * float(bb.abs().max()) + 1e-6, (n, shape)


def test_model_gradients_with_and_without_the_fused_first_layer_gradients(dev):
    """The whole backbone step with nets.USE_FUSED_FIRST_WRW on and off: every other gradient
    bit-identical, conv1_1's and the mean's within float32 summation-order noise."""
    from soft_contrastive_learning_amd import parallel
    from soft_contrastive_learning_amd.model import nets
    img = torch.randint(0, 256, (2, 480, 640, 3), generator=torch.Generator().manual_seed(81)).float().to(dev)
    g = torch.randn(2, 30, 40, 512, generator=torch.Generator().manual_seed(82)).to(dev).bfloat16()
    grads = {}
    old = nets.USE_FUSED_FIRST_WRW
    try:
        for fused in (False, True):
            nets.USE_FUSED_FIRST_WRW = fused
            model = nets.VGG16NetVLAD(compute_dtype=torch.bfloat16, seed=13, fused_relu=True).to(dev)
            buckets = parallel.GradBuckets(list(model.parameters()))
            nets.GRAD_SINK = buckets
            try:
                buckets.zero()
                model.features(img).backward(g)
                buckets.finish()
            finally:
                nets.GRAD_SINK = None
            torch.cuda.synchronize()
            grads[fused] = {n: p.grad.clone() for n, p in model.named_parameters() if p.grad is not None}
    finally:
        nets.USE_FUSED_FIRST_WRW = old
    assert set(grads[True]) == set(grads[False])
    for n in grads[True]:
        a, bb = grads[True][n], grads[False][n]
        if n in ('conv1_1_kernel', 'conv1_1_bias', 'average_rgb'):
            rel = float((a.double() - bb.double()).norm() / bb.double().norm())
            assert rel < 2e-4, (n, rel)
        else:
            assert torch.equal(a, bb), n
