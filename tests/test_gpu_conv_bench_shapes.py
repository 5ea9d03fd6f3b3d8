"""Parity of the convolution kernels AT THE LAUNCH SHAPES OF THE BENCH STEP (configs[1]: 24 images,
640x480): one test per VGG16 layer (model/nets.py:27-63), batch 24 at the layer's resolution,
through exactly the dispatch ``VGG16NetVLAD.features`` takes in the timed step — float32 master
weights, packed weight images written by one ``scl_conv_pack_batch`` launch, persistent grids
(3.75 tile rounds per workgroup, XCD tile order, next-tile prefetch under the epilogue), the
pooling epilogue with its one-byte window index, the ReLU' mask in the backward-data epilogue,
weight AND bias gradient from the weight-gradient kernel on the second stream, written straight
into the gradient sink (split grids, 16-slab-group reduce); the pooling layers' backward straight
from the pooled gradient + window index (conv1_2 / conv2_2: backward-data and weight-gradient
kernels un-pool while they stage, no full-size gradient; conv3_3 / conv4_3: the weight gradient).

Reference: float32 ``torch`` convolutions ON THE SAME bf16 INPUTS (bf16-rounded weights,
activations and incoming gradient), evaluated a few images at a time.  Three gates per result
(round 5; the first alone lets a systematically wrong LOW-MAGNITUDE region through):
  * max-abs error <= 6e-3 of the reference's scale — the gate the toy shapes of
    tests/test_gpu_backbone.py use (bf16 output rounding is 2e-3 of a value);
  * norm-relative: ||got - want||_2 <= 3e-3 ||want||_2 over the whole tensor (bf16 rounding alone
    is 2^-9 / sqrt 3 = 1.1e-3 RMS);
  * per 16 x 16-pixel tile of every image (all channels): max error <= 6e-3 of THE TILE'S OWN
    maximum + 2e-4 of the global scale — a tile that should be small or zero must be, whatever the
    rest of the map looks like.
The measured values go to gpurun_out/conv_parity_r05.json (copied to profiles/).

Two decision points are taken FROM the kernel's own output and validated separately, because an
accumulation-order difference of 1e-6 can legitimately flip them and each flip moves a whole
gradient term: the ReLU mask [y > 0] (y is checked against the reference), and the position of
a window's maximum (checked to point at an element within 1e-4 of the reference window's
maximum).  Given those, every backward pass is LINEAR in its inputs and is compared exactly
like the forward.
"""
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

B = 24
CL = torch.channels_last
GATE = 6e-3

# (name, cin, cout, H, W of the layer's input, ReLU directly after, pool after) at 640x480
LAYERS = [
    ('1_2', 64, 64, 480, 640, False, True),
    ('2_1', 64, 128, 240, 320, True, False),
    ('2_2', 128, 128, 240, 320, False, True),
    ('3_1', 128, 256, 120, 160, True, False),
    ('3_2', 256, 256, 120, 160, True, False),
    ('3_3', 256, 256, 120, 160, False, True),
    ('4_1', 256, 512, 60, 80, True, False),
    ('4_2', 512, 512, 60, 80, True, False),
    ('4_3', 512, 512, 60, 80, False, True),
    ('5_1', 512, 512, 30, 40, True, False),
    ('5_2', 512, 512, 30, 40, True, False),
    ('5_3', 512, 512, 30, 40, False, False),
]


@pytest.fixture(scope="module")
def dev():
    assert torch.cuda.is_available()
    torch.backends.cudnn.benchmark = False
    return torch.device("cuda:0")


@pytest.fixture
def sink():
    """Installs a gradient sink over the layer's parameters like bench.py / train.py do."""
    from soft_contrastive_learning_amd import parallel
    from soft_contrastive_learning_amd.model import nets
    made = []

    def make(params):
        b = parallel.GradBuckets(params)
        nets.GRAD_SINK = b
        made.append(b)
        return b
    yield make
    nets.GRAD_SINK = None


@pytest.fixture
def takes():
    """Records whether each _GradLink.take() of the test found the marked buffer."""
    from soft_contrastive_learning_amd.model import nets
    hits, orig = [], nets._GradLink.take

    def counting(self, gy):
        hit = orig(self, gy)
        hits.append(hit)
        return hit
    nets._GradLink.take = counting
    yield hits
    nets._GradLink.take = orig


NORM_GATE = 3e-3
TILE_GATE = (6e-3, 2e-4)          # of the tile's own maximum, + of the global scale
MEASURED = {}


def _maxerr(got, want, what=None, scale=None):
    """max |got - want| / max |want| evaluated in image chunks (float32 copies stay small); with
    ``what`` also asserts the norm-relative and the per-tile gate and records what it measured.
    ``scale``: the scale the per-tile floor refers to (default max |want|)."""
    err, wmax, e2, w2 = 0.0, 0.0, 0.0, 0.0
    tiles = []
    for lo in range(0, got.shape[0], 4):
        g, w = got[lo:lo + 4].float(), want[lo:lo + 4].float()
        d = (g - w).abs()
        err = max(err, float(d.max()))
        wmax = max(wmax, float(w.abs().max()))
        e2 += float((d.double() ** 2).sum())
        w2 += float((w.double() ** 2).sum())
        if what is not None and g.dim() == 4:
            tiles.append((F.max_pool2d(d.amax(dim=1, keepdim=True), 16, ceil_mode=True),
                          F.max_pool2d(w.abs().amax(dim=1, keepdim=True), 16, ceil_mode=True)))
    rel = err / max(wmax, 1e-30)
    if what is not None:
        nrel = (e2 / max(w2, 1e-300)) ** 0.5
        te = torch.cat([t[0] for t in tiles]) if tiles else None
        tw = torch.cat([t[1] for t in tiles]) if tiles else None
        floor = TILE_GATE[1] * (scale if scale is not None else wmax)
        worst = float((te / (TILE_GATE[0] * tw + floor)).max()) if tiles else 0.0
        MEASURED[what] = dict(max_over_scale=rel, norm_relative=nrel, worst_tile_over_its_gate=worst)
        assert nrel < NORM_GATE, (what, 'norm-relative', nrel)
        assert worst <= 1.0, (what, 'per-tile: error / (6e-3 tile max + 2e-4 scale)', worst)
    return rel, wmax


def _rel_f32(got, want, what, gate=1e-3):
    """Float32 parameter gradients (one [cout, cin, 3, 3] block): max over scale (returned) and
    the norm-relative error, gated at ``gate`` — nothing is rounded to bf16 on this path."""
    got, want = got.double(), want.double()
    rel = float((got - want).abs().max() / want.abs().max())
    nrel = float((got - want).norm() / want.norm())
    # per output channel: a wrong 32-channel block of a small-magnitude filter must not hide
    ch = ((got - want).flatten(1).norm(dim=1) / want.flatten(1).norm(dim=1).clamp_min(1e-3 * float(want.norm()))).max()
    MEASURED[what] = dict(max_over_scale=rel, norm_relative=nrel, worst_channel_norm_relative=float(ch))
    assert nrel < gate, (what, 'norm-relative', nrel)
    assert float(ch) < 3 * gate, (what, 'per-output-channel norm-relative', float(ch))
    return rel


@pytest.fixture(scope="module", autouse=True)
def _dump_measured():
    yield
    import json
    import os
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    out = os.path.join(root, 'gpurun_out')
    if MEASURED and os.path.isdir(out):
        with open(os.path.join(out, 'conv_parity_r05.json'), 'w') as f:
            json.dump(MEASURED, f, indent=1, sort_keys=True)


def _ref_conv(x, wq, chunk=4):
    """float32 conv2d of bf16 x with the bf16-rounded weights, as a float32 NCHW tensor."""
    return torch.cat([F.conv2d(x[lo:lo + chunk].float(), wq, padding=1)
                      for lo in range(0, x.shape[0], chunk)], 0)


def _ref_backward(x, gz, wq, chunk=4):
    """(gx float32, gw float32) of conv2d(x, wq) for the upstream gradient gz (bf16)."""
    gx = torch.empty(x.shape, dtype=torch.float32, device=x.device)
    gw = torch.zeros(wq.shape, dtype=torch.float64, device=x.device)
    for lo in range(0, x.shape[0], chunk):
        g = gz[lo:lo + chunk].float()
        gx[lo:lo + chunk] = F.conv_transpose2d(g, wq, padding=1)
        gw += torch.nn.grad.conv2d_weight(x[lo:lo + chunk].float(), wq.shape, g, padding=1).double()
    return gx, gw.float()


@pytest.fixture
def reserve():
    """scl_set_reserve_cus(n) for the duration of a test (what bench.py may choose at N > 1: every
    persistent grid shrinks to 256 - n workgroups, the weight-gradient splits change)."""
    from soft_contrastive_learning_amd import _lib as L
    lib = L.load()

    def set_(n):
        lib.scl_set_reserve_cus(int(n))
    yield set_
    lib.scl_set_reserve_cus(0)


# (layer, CUs left free): every layer with the whole chip; one layer of every kernel family and
# resolution again with 8 CUs reserved — 248 workgroups: other tile rounds, weight-gradient pixel
# splits that are not multiples of 8 (the workgroup -> (split, block) map takes its other branch)
CASES = [(l, 0) for l in LAYERS] + [(l, 8) for l in LAYERS if l[0] in ('1_2', '2_1', '3_3', '4_2', '5_1')]

# Round 5: the LDS-weights layers at the launch shapes of the reference's OWN training resolution
# (240 x 180, 25 images per batch: train/train.py:423-428, util/cv.py:7-9 -> conv4_x maps 22 x 30,
# conv5_x 11 x 15) and of configs[0]'s image size at a full batch (224 x 224 -> 28 x 28 / 14 x 14):
# the tile-count rule of nets._lds_conv_pays / _wrw_pays sends them to the own kernels (round 4's
# map-size gate handed them to the library).  Maps narrower than a 40-pixel tile, fewer rows than
# a block, more images than tile rounds.
SMALL = [(('4_1', 256, 512, 22, 30, True, False), 25), (('4_2', 512, 512, 22, 30, True, False), 25),
         (('4_3', 512, 512, 22, 30, False, True), 25), (('5_1', 512, 512, 11, 15, True, False), 25),
         (('5_3', 512, 512, 11, 15, False, False), 25), (('4_2', 512, 512, 28, 28, True, False), 24),
         (('4_3', 512, 512, 28, 28, False, True), 24), (('5_2', 512, 512, 14, 14, True, False), 24)]


@pytest.mark.parametrize('layer,batch', SMALL, ids=['%s-%dx%dx%d' % (l[0], b, l[3], l[4]) for l, b in SMALL])
def test_layer_at_small_map_launch_shapes(dev, sink, takes, reserve, layer, batch):
    from soft_contrastive_learning_amd.model import nets
    x_like = torch.empty(batch, layer[1], layer[3], layer[4], device='meta')
    assert nets._lds_conv_pays(x_like, False, kout=layer[2]) and nets._lds_conv_pays(x_like, True, kout=layer[1])
    assert nets._wrw_pays(x_like)
    _layer_case(dev, sink, takes, reserve, layer, 0, batch)


@pytest.mark.parametrize('layer,free_cus', CASES, ids=['%s%s' % (l[0], '-reserve%d' % r if r else '') for l, r in CASES])
def test_layer_at_bench_shape(dev, sink, takes, reserve, layer, free_cus):
    _layer_case(dev, sink, takes, reserve, layer, free_cus, B)


def _layer_case(dev, sink, takes, reserve, layer, free_cus, B):
    from soft_contrastive_learning_amd.model import nets
    reserve(free_cus)
    name, cin, cout, h, w, relu, pool = layer
    if B != 24 or h * w < 1200:
        name = '%s@%dx%dx%d' % (name, B, h, w)
    assert nets.USE_PREPACK and nets.USE_SIDE_WRW and nets.USE_POOL_IDX and nets.USE_MASKED_BWD
    assert nets.USE_POOLED_BWD
    g = torch.Generator().manual_seed(1000 + 17 * cin + cout + h)
    # a post-ReLU input (about half of it exactly zero, like the step's activations)
    x = torch.relu(torch.randn(B, cin, h, w, generator=g)).to(dev).bfloat16().contiguous(memory_format=CL)
    x.requires_grad_(True)
    wt = torch.nn.Parameter((torch.randn(cout, cin, 3, 3, generator=g) * (2.0 / (9 * cin)) ** 0.5).to(dev))
    bias = torch.nn.Parameter((torch.randn(cout, generator=g) * 0.1).to(dev))
    wq = wt.detach().bfloat16().float()                     # what the packed image holds
    buckets = sink([wt, bias])
    assert nets.prepack([wt], force=True) == 2                # both directions, one launch
    link_in, link_out = nets._GradLink(), nets._GradLink()
    if pool:
        y = nets._ConvBiasPoolReLU.apply(x, wt, bias, link_in, link_out)
    else:
        y = nets._ConvBiasAct.apply(x, wt, bias, relu, link_in, link_out if relu else None)
    z32 = _ref_conv(x.detach(), wq) + bias.detach()[None, :, None, None]
    zscale = float(z32.abs().max())
    if pool:
        assert y.grad_fn.by_idx                               # the index epilogue, no full-size z
        assert tuple(y.shape) == (B, cout, h // 2, w // 2)
        idx = y.grad_fn.saved_tensors[2]
        want = torch.relu(F.max_pool2d(z32, 2))
        tag = '%s%s ' % (name, '-reserve%d' % free_cus if free_cus else '')
        err, _ = _maxerr(y.detach(), want, tag + 'pooled forward', zscale)
        assert err * float(want.abs().max()) < GATE * zscale, (name, 'pooled forward', err)
        # every stored position points at (numerically) a maximum of its window
        win = torch.stack([z32[:, :, dy::2, dx::2] for dy in (0, 1) for dx in (0, 1)], -1)
        assert int(idx.max()) <= 3
        picked = torch.gather(win, -1, idx.long()[..., None])[..., 0]
        assert float((win.max(-1).values - picked).max()) <= 1e-4 * zscale, (name, 'pool index')
        del win, picked
    else:
        tag = '%s%s ' % (name, '-reserve%d' % free_cus if free_cus else '')
        want = torch.relu(z32) if relu else z32
        err, _ = _maxerr(y.detach(), want, tag + 'forward', zscale)
        assert err < GATE, (name, 'forward', err)
    del z32, want

    # incoming gradient, masked by THIS layer's ReLU' the way the layer above hands it down
    gy = torch.randn(y.shape, generator=g).to(dev).bfloat16().contiguous(memory_format=CL)
    if relu or pool:
        gy = torch.where(y.detach() > 0, gy, torch.zeros_like(gy)).contiguous(memory_format=CL)
        link_out.mark(gy)
    buckets.zero()
    y.backward(gy)
    buckets.finish()
    torch.cuda.synchronize()
    if relu or pool:     # the layer found its gradient already masked: bias gradient from wrw64
        assert takes == [True], "the layer did not take the masked gradient (autograd copied it?)"
    assert link_in.ptr is not None                            # ReLU' of the layer below applied

    if pool:   # un-pool by the kernel's own (validated) index: exact routing
        gz = torch.zeros(B, cout, h, w, dtype=torch.bfloat16, device=dev).contiguous(memory_format=CL)
        for k in range(4):
            gz[:, :, (k >> 1)::2, (k & 1)::2] = torch.where(idx == k, gy, torch.zeros_like(gy))
    else:
        gz = gy
    gx_ref, gw_ref = _ref_backward(x.detach(), gz, wq)
    gx_ref = torch.where(x.detach() > 0, gx_ref, torch.zeros_like(gx_ref))
    err, _ = _maxerr(x.grad, gx_ref, tag + 'masked backward-data')
    assert err < GATE, (name, 'masked backward-data', err)
    assert wt.grad.data_ptr() == buckets.view(wt).data_ptr()  # written into the sink
    err = _rel_f32(wt.grad, gw_ref, tag + 'weight gradient')
    assert err < GATE, (name, 'weight gradient', err)
    gb_ref = torch.stack([gz[lo:lo + 4].float().sum(dim=(0, 2, 3)) for lo in range(0, B, 4)]).sum(0)
    gb_scale = float(torch.stack([gz[lo:lo + 4].float().abs().sum(dim=(0, 2, 3))
                                  for lo in range(0, B, 4)]).sum(0).max())
    assert float((bias.grad - gb_ref).abs().max()) < 1e-4 * gb_scale, (name, 'bias gradient')


def test_first_layer_at_bench_shape(dev, sink, takes):
    """conv1_1: mean subtraction + cast + convolution + bias + ReLU in one kernel; weight, bias and
    mean gradient from one pass over the incoming gradient (model/nets.py:22-24, :39)."""
    from soft_contrastive_learning_amd.model import nets
    h, w = 480, 640
    g = torch.Generator().manual_seed(77)
    img = torch.randint(0, 256, (B, h, w, 3), generator=g).float().to(dev)
    avg = torch.nn.Parameter(torch.tensor([123.68, 116.78, 103.94], device=dev))
    wt = torch.nn.Parameter((torch.randn(64, 3, 3, 3, generator=g) * (2.0 / 27) ** 0.5).to(dev))
    bias = torch.nn.Parameter((torch.randn(64, generator=g) * 0.1).to(dev))
    wq = wt.detach().bfloat16().float()
    buckets = sink([avg, wt, bias])
    link_out = nets._GradLink()
    y = nets._FirstConv.apply(img, avg, wt, bias, torch.bfloat16, link_out)
    x0 = (img - avg.detach()).bfloat16().permute(0, 3, 1, 2)
    want = torch.relu(_ref_conv(x0, wq) + bias.detach()[None, :, None, None])
    err, _ = _maxerr(y.detach(), want, '1_1 forward')
    assert err < GATE, ('forward', err)
    del want
    gy = torch.randn(y.shape, generator=g).to(dev).bfloat16().contiguous(memory_format=CL)
    gy = torch.where(y.detach() > 0, gy, torch.zeros_like(gy)).contiguous(memory_format=CL)
    link_out.mark(gy)
    buckets.zero()
    y.backward(gy)
    buckets.finish()
    torch.cuda.synchronize()
    assert takes == [True]
    gx_ref, gw_ref = _ref_backward(x0, gy, wq)
    err = _rel_f32(wt.grad, gw_ref, '1_1 weight gradient')
    assert err < GATE, ('weight gradient', err)
    gb_ref = torch.stack([gy[lo:lo + 4].float().sum(dim=(0, 2, 3)) for lo in range(0, B, 4)]).sum(0)
    gb_scale = float(torch.stack([gy[lo:lo + 4].float().abs().sum(dim=(0, 2, 3))
                                  for lo in range(0, B, 4)]).sum(0).max())
    assert float((bias.grad - gb_ref).abs().max()) < 1e-4 * gb_scale
    # d loss / d average_rgb = - the spatial sum of the first layer's input gradient (the kernel
    # takes the closed form of nets.avg_rgb_grad with the same bf16-rounded weights): a sum of
    # 7.4 M signed terms per channel, compared at 5e-3 of its own magnitude
    davg_ref = -torch.stack([gx_ref[lo:lo + 4].double().sum(dim=(0, 2, 3)) for lo in range(0, B, 4)]).sum(0)
    scale = float(davg_ref.abs().max())
    assert float((avg.grad.double() - davg_ref).abs().max()) < 5e-3 * scale + 1e-2


def test_first_block_backward_at_bench_shape(dev, sink, takes):
    """Round 5: conv1_2's backward-data kernel computes conv1_1's weight / bias / mean gradient on
    its LDS tile (scl_conv3x3_masked_pooled_first_wrw; the 944 MB gradient map at conv1_1's
    pre-activation is never written).  The chain conv1_1 -> ReLU -> conv1_2 -> pool -> ReLU as the
    model builds it, 24 x 640x480, against float32 torch on the same bf16 operands: the kernel's own
    y1 (validated in test_first_layer_at_bench_shape) as conv1_2's input and ReLU' mask, its own
    window index (validated in test_layer_at_bench_shape[1_2]) for the un-pooling, the gradient at
    conv1_1's pre-activation rounded to bf16 where the kernel rounds it."""
    from soft_contrastive_learning_amd.model import nets
    assert nets.USE_FUSED_FIRST_WRW is None and nets.USE_SIDE_WRW     # 'auto': on with the sink + second stream
    h, w = 480, 640
    g = torch.Generator().manual_seed(177)
    img = torch.randint(0, 256, (B, h, w, 3), generator=g).float().to(dev)
    avg = torch.nn.Parameter(torch.tensor([123.68, 116.78, 103.94], device=dev))
    w1 = torch.nn.Parameter((torch.randn(64, 3, 3, 3, generator=g) * (2.0 / 27) ** 0.5).to(dev))
    b1 = torch.nn.Parameter((torch.randn(64, generator=g) * 0.1).to(dev))
    w2 = torch.nn.Parameter((torch.randn(64, 64, 3, 3, generator=g) * (2.0 / 576) ** 0.5).to(dev))
    b2 = torch.nn.Parameter((torch.randn(64, generator=g) * 0.1).to(dev))
    w1q, w2q = w1.detach().bfloat16().float(), w2.detach().bfloat16().float()
    buckets = sink([avg, w1, b1, w2, b2])
    assert nets.prepack([w2], force=True) == 2
    l1, l2 = nets._GradLink(), nets._GradLink()
    y1 = nets._FirstConv.apply(img, avg, w1, b1, torch.bfloat16, l1)
    a = nets._ConvBiasPoolReLU.apply(y1, w2, b2, l1, l2)
    assert a.grad_fn.by_idx and l1.first is not None
    idx = a.grad_fn.saved_tensors[2]
    ga = torch.randn(a.shape, generator=g).to(dev).bfloat16().contiguous(memory_format=CL)
    ga = torch.where(a.detach() > 0, ga, torch.zeros_like(ga)).contiguous(memory_format=CL)
    l2.mark(ga)
    buckets.zero()
    a.backward(ga)
    buckets.finish()
    torch.cuda.synchronize()
    assert takes == [True, True]                    # both layers took what the layer above handed down
    x0 = (img - avg.detach()).bfloat16().permute(0, 3, 1, 2)
    gw1 = torch.zeros(w1.shape, dtype=torch.float64, device=dev)
    gb1 = torch.zeros(64, dtype=torch.float64, device=dev)
    davg = torch.zeros(3, dtype=torch.float64, device=dev)
    gb1_abs = torch.zeros(64, dtype=torch.float64, device=dev)
    for lo in range(0, B, 2):
        sl = slice(lo, lo + 2)
        gz2 = torch.zeros(2, 64, h, w, dtype=torch.float32, device=dev)
        for k in range(4):
            gz2[:, :, (k >> 1)::2, (k & 1)::2] = torch.where(idx[sl] == k, ga[sl], torch.zeros_like(ga[sl])).float()
        gz1 = F.conv_transpose2d(gz2, w2q, padding=1)
        gz1 = torch.where(y1.detach()[sl] > 0, gz1, torch.zeros_like(gz1)).bfloat16().float()
        gw1 += torch.nn.grad.conv2d_weight(x0[sl].float(), w1q.shape, gz1, padding=1).double()
        gb1 += gz1.double().sum(dim=(0, 2, 3))
        gb1_abs += gz1.double().abs().sum(dim=(0, 2, 3))
        davg -= F.conv_transpose2d(gz1, w1q, padding=1).double().sum(dim=(0, 2, 3))
        del gz2, gz1
    err = _rel_f32(w1.grad, gw1.float(), '1_1 weight gradient out of the conv1_2 backward kernel', gate=2e-3)
    assert err < GATE, err
    assert float((b1.grad.double() - gb1).abs().max()) < 2e-4 * float(gb1_abs.max())
    assert float((avg.grad.double() - davg).abs().max()) < 5e-3 * float(davg.abs().max()) + 1e-2
