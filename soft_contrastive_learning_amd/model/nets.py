"""Drop-in counterpart of the reference's ``model/nets.py``.

``vgg16Netvlad(image_batch)`` and ``vgg16(image_batch)`` keep the reference names and
the NHWC / raw-0..255-RGB input convention (model/nets.py:7-69, :72-131).  In bf16 mode every
VGG16 convolution pass (forward, backward-data, weight gradient, every layer) runs on the
hand-written implicit-GEMM kernels of csrc/conv64.hip, csrc/convh.hip and csrc/convg.hip (bias /
ReLU / max-pool / ReLU' fused into their epilogues, float32 master weights read directly,
weight gradients written straight into the flat gradient buffer); only the float32 mode and
maps below 30x40 use the library (MIOpen / CK) with the fused glue passes of
csrc/vgg_glue.hip.  The channel L2 norm + NetVLAD head (model/nets.py:66-67) is one autograd op
over the kernels of csrc/netvlad.hip.

TF1 keeps variables in the graph scope ``vgg16_netvlad_pca``; here they live in a
``VGG16NetVLAD`` module.  ``state_dict_tf`` / ``load_state_dict_tf`` expose them under
the reference's checkpoint names and shapes (HWIO kernels, [1,1,512,64] assignment,
[1,1,1,512,64] centres) so checkpoints stay layout-compatible.
"""
import math

import contextlib
import os
import threading

import torch
import torch.nn.functional as F

from .. import _lib as L

SCOPE = 'vgg16_netvlad_pca'

# (name, out_channels, relu directly after the conv) — model/nets.py:39-63.
# A pool (and the ReLU that follows it) comes after the last conv of blocks 1-4.
VGG_LAYERS = [
    ('1_1', 64, True), ('1_2', 64, False), 'pool',
    ('2_1', 128, True), ('2_2', 128, False), 'pool',
    ('3_1', 256, True), ('3_2', 256, True), ('3_3', 256, False), 'pool',
    ('4_1', 512, True), ('4_2', 512, True), ('4_3', 512, False), 'pool',
    ('5_1', 512, True), ('5_2', 512, True), ('5_3', 512, False),
]


class _NetVLADFn(torch.autograd.Function):
    """l2_normalize(axis=-1) + netVLAD(x, 64) on a channels-last feature map."""

    @staticmethod
    def forward(ctx, x, assign_w, centers, pre_l2, planes=None):
        lib = L.load()
        L.require_device(x, assign_w, centers)
        if x.dim() != 4 or x.shape[3] != L.VLAD_D:
            raise ValueError("feature map must be [B,H,W,512] channels-last, got %s"
                             % (tuple(x.shape),))
        if x.dtype == torch.float32:
            dt = L.DT_F32
        elif x.dtype == torch.bfloat16:
            dt = L.DT_BF16
        else:
            raise ValueError("feature map dtype must be float32 or bfloat16, got %s" % x.dtype)
        x = x.contiguous()
        w = assign_w.reshape(L.VLAD_D, L.VLAD_K).float().contiguous()
        c = centers.reshape(L.VLAD_D, L.VLAD_K).float().contiguous()
        b, n = x.shape[0], x.shape[1] * x.shape[2]
        dev = x.device
        out = torch.empty((b, L.VLAD_D * L.VLAD_K), dtype=torch.float32, device=dev)
        train = any(ctx.needs_input_grad[:3])
        sa = sl = sr = sv = None
        if train:
            sa = torch.empty((b, n, L.VLAD_K), dtype=torch.float32, device=dev)
            # the logits: float32 feature maps only (and the four-wave kernels of the diagnostic build);
            # the bf16 path's backward takes log a instead (round 6: 7.4 MB less written and read
            # at 24 x 1200 locations)
            if dt == L.DT_F32 or lib.scl_build_is_diag():
                sl = torch.empty((b, n, L.VLAD_K), dtype=torch.float32, device=dev)
            sr = torch.empty((b, n), dtype=torch.float32, device=dev)
            sv = torch.empty((b, L.VLAD_SAVE_ROWS, L.VLAD_K), dtype=torch.float32, device=dev)
        ws = L.workspace(lib.scl_netvlad_fwd_workspace_bytes(b, n), dev)
        # the plane images of the assignment weights: handed in by the caller whose forward pass
        # wrote them in its weight-packing launch (VGG16NetVLAD.forward: a forced prepack() of THIS
        # pass — never found by address + version, which in-place optimizers do not move); without
        # them the call builds its own
        pgen = None
        if planes is not None and dt == L.DT_BF16:
            planes, pgen = planes
        else:
            planes = None
        L.check(lib.scl_netvlad_fwd_p(L.ptr(x), dt, L.ptr(w), L.ptr(c), L.ptr(planes), b, n,
                                      int(bool(pre_l2)), L.ptr(out), L.ptr(sa), L.ptr(sl), L.ptr(sr),
                                      L.ptr(sv), L.ptr(ws), ws.numel(), L.stream_of(x)))
        if train:
            ctx.save_for_backward(x, w, c, sa, sr, sv)
            ctx.logits = sl                      # (None on the product's bf16 path)
            ctx.meta = (dt, b, n, int(bool(pre_l2)), assign_w.shape, centers.shape, planes, pgen)
        return out

    @staticmethod
    def backward(ctx, grad_out):
        lib = L.load()
        x, w, c, sa, sr, sv = ctx.saved_tensors
        sl = ctx.logits
        dt, b, n, pre_l2, w_shape, c_shape, planes, pgen = ctx.meta
        if planes is not None and _PLANE_GEN.get(planes.data_ptr()) != pgen:
            planes = None        # rewritten by a later forward pass: the call builds its own from w
        go = grad_out.float().contiguous()
        gx = torch.empty_like(x)
        gw = torch.empty_like(w)
        gc = torch.empty_like(c)
        ws = L.workspace(lib.scl_netvlad_bwd_workspace_bytes(b, n), x.device)
        # (the forward's plane images, unless a later pass has rewritten the buffer since)
        L.check(lib.scl_netvlad_bwd_p(L.ptr(x), dt, L.ptr(w), L.ptr(c), L.ptr(planes), L.ptr(go),
                                      L.ptr(sa), L.ptr(sl), L.ptr(sr), L.ptr(sv), b, n, pre_l2,
                                      L.ptr(gx), L.ptr(gw), L.ptr(gc), L.ptr(ws), ws.numel(),
                                      L.stream_of(x)))
        return gx, gw.reshape(w_shape), gc.reshape(c_shape), None, None


_CL = torch.channels_last
_ONES = [1, 1]


# conv1_2 and conv2_x (64 / 128 channels at full and half resolution) run on csrc/conv64.hip:
# the library kernels take up to twice as long there as on the equal-FLOP deeper layers.
# SCL_CONV64=0 restores MIOpen everywhere.
USE_CONV64 = os.environ.get('SCL_CONV64', '1') != '0'
USE_CONVG = os.environ.get('SCL_CONVG', '1') != '0'
USE_WRW = os.environ.get('SCL_WRW', '1') != '0'
USE_FIRST = os.environ.get('SCL_FIRST', '1') != '0'
_OWN_CONV_SHAPES = {(64, 64), (64, 128), (128, 64), (128, 128)}      # (contraction, output)

# bench.py sets this to a dict for its timed steps: C-library kernel name -> [calls,
# algorithmic flops, algorithmic HBM bytes], added up by the call sites below (what a launch
# must compute / move, not what it happens to do) so that the measured durations can be priced.
WORK_LOG = None


# The trainer may install a gradient sink (parallel.GradBuckets): an object with view(p) ->
# the tensor a parameter's gradient must end up in (or None) and done(p).  The backward
# kernels of the convolution layers then write weight and bias gradients straight into it and
# return None to autograd — valid because every parameter is used once per step and the sink
# is zeroed before each backward; it removes a temporary and a `grad += new` pass per
# parameter.  None (the default): gradients are returned to autograd as usual.
GRAD_SINK = None


def _grad_out(p_like):
    """Tensor to write a parameter gradient into: the sink's slice, or a fresh one."""
    if GRAD_SINK is not None:
        v = GRAD_SINK.view(p_like)
        if v is not None and v.dtype == p_like.dtype and v.stride() == p_like.stride():
            return v
    return torch.empty_like(p_like)


def _grad_ret(g, p_like):
    """What a Function returns to autograd for that gradient: None once it sits in the sink."""
    if GRAD_SINK is not None and g is not None:
        v = GRAD_SINK.view(p_like)
        if v is not None and v.data_ptr() == g.data_ptr():
            GRAD_SINK.done(p_like)
            return None
    return g


def _lds_work(flops, nbytes):
    """Work of an LDS-weights convolution, under the names of both kernels that may serve it
    (csrc/convh.hip by default, csrc/convg.hip when pinned or for odd chunk counts): bench.py
    prices whichever one it timed."""
    _work('convh_kernel', flops, nbytes)
    _work('convg_kernel', flops, nbytes)


def _work(name, flops, nbytes):
    if WORK_LOG is not None:
        e = WORK_LOG.setdefault(name, [0, 0.0, 0.0])
        e[0] += 1
        e[1] += flops
        e[2] += nbytes


# The own kernels take the weight as bf16 or as the float32 master parameter (rounded to bf16
# while it is packed: the 13 per-step cast kernels and their 13 backward casts disappear) and
# return the weight gradient in the same type.
_W_DTYPES = (torch.bfloat16, torch.float32)


def _wflag(w):
    return L.W_F32 if w.dtype == torch.float32 else 0


# ---- packed weight images, all in one launch per step -------------------------------------
# Each own convolution otherwise packs its weights in a small kernel of its own, 24 of them per
# training step on the critical path.  prepack() writes every image a step will need with ONE
# launch (scl_conv_pack_batch); conv64 / conv_pool_idx find them here by (storage address,
# direction) and a version check, and fall back to packing for themselves on a miss.
USE_PREPACK = os.environ.get('SCL_PREPACK', '1') != '0'
# (data_ptr, transposed, calling thread) -> (weakref to the weight, _version, shape, image).
# Per THREAD: the reference drives one session from three threads (train/train.py:967-975); the
# images are rewritten by every forward pass, and a forward on the evaluation thread's stream
# must not rewrite the images a backward-data kernel of the training thread is reading (nor land
# its — possibly older — copy of the weights after the training thread's own repack).
_PACKED = {}
_PACKED_LOCK = threading.Lock()
# plane-image buffer address -> how many times prepack() has written it: a NetVLAD backward trusts
# the forward's plane images only while the count is the one its forward saw
_PLANE_GEN = {}


class _PackSlot(threading.local):
    """Which thread's images the calling thread reads: its own, except inside a backward pass —
    autograd runs that on an engine thread, with the slot of the thread that ran the forward
    (saved in the node by _pack_slot(), restored by _in_slot())."""
    slot = None


_SLOT = _PackSlot()


def _pack_slot():
    return _SLOT.slot if _SLOT.slot is not None else threading.get_ident()


@contextlib.contextmanager
def _in_slot(slot):
    old, _SLOT.slot = _SLOT.slot, slot
    try:
        yield
    finally:
        _SLOT.slot = old


def prepack(weights, force=False, vlad_w=None):
    """Bring the packed images of ``weights`` (3x3 convolution weights on a HIP device, both
    directions) up to date; returns the number of images written.  Without ``force`` an image
    is rewritten only when the tensor's autograd version moved — which in-place optimizers do
    NOT guarantee (torch's fused Adam updates the parameters without touching ``_version``), so
    the model's forward pass forces: one 40 us launch per forward, never a stale weight.
    ``vlad_w``: the NetVLAD assignment weights (float32, contiguous, 512 * 64 elements): their bf16
    plane images ride in the same launch (SCL_PACK_VLAD_W); fresh_vlad_planes() hands them to the
    caller that asked for them (and to nobody else)."""
    import weakref
    lib = L.load()
    jobs, keep = [], []
    me = _pack_slot()

    def slot_buffer(key, w, nbytes):
        """The image buffer registered under ``key`` (None = up to date, nothing to write)."""
        with _PACKED_LOCK:
            ent = _PACKED.get(key)
            if (not force and ent is not None and ent[0]() is not None and ent[1] == w._version
                    and ent[2] == tuple(w.shape)):
                return None
            buf = (ent[3] if ent is not None and ent[3].numel() == nbytes and ent[3].device == w.device
                   else torch.empty(nbytes, dtype=torch.uint8, device=w.device))
            _PACKED[key] = (weakref.ref(w), w._version, tuple(w.shape), buf)
            return buf

    with _PACKED_LOCK:
        alive = {t.ident for t in threading.enumerate()}
        for key in [k for k, ent in _PACKED.items() if ent[0]() is None or k[2] not in alive]:
            del _PACKED[key]                               # the weight, or the thread, is gone
    # the backward directions are only ever read by a backward pass
    directions = (False, True) if torch.is_grad_enabled() else (False,)
    for w in weights:
        if not (w.is_cuda and w.dtype in _W_DTYPES and w.dim() == 4 and tuple(w.shape[2:]) == (3, 3)):
            continue
        for transposed in directions:
            cin, kout = (w.shape[0], w.shape[1]) if transposed else (w.shape[1], w.shape[0])
            nbytes = lib.scl_conv_packed_bytes(cin, kout)
            if nbytes == 0:
                continue
            buf = slot_buffer((w.data_ptr(), transposed, me), w, nbytes)
            if buf is None:
                continue
            sk, sc, sh, sw = w.stride()
            jobs.append(L.PackJob(L.ptr(w), sk, sc, sh, sw, int(transposed) | _wflag(w), cin, kout,
                                  L.ptr(buf)))
            keep.append(w)
    if (vlad_w is not None and vlad_w.is_cuda and vlad_w.dtype == torch.float32 and vlad_w.is_contiguous()
            and vlad_w.numel() == L.VLAD_D * L.VLAD_K):
        buf = slot_buffer((vlad_w.data_ptr(), 'vlad', me), vlad_w, lib.scl_netvlad_planes_bytes())
        if buf is not None:
            jobs.append(L.PackJob(L.ptr(vlad_w), 0, 0, 0, 0, L.PACK_VLAD_W, L.VLAD_D, L.VLAD_K, L.ptr(buf)))
            keep.append(vlad_w)
            with _PACKED_LOCK:
                gen = _PLANE_GEN[buf.data_ptr()] = _PLANE_GEN.get(buf.data_ptr(), 0) + 1
            _FWD.planes = (vlad_w.data_ptr(), buf, gen)
    if jobs:
        L.require_device(*keep)
        arr = (L.PackJob * len(jobs))(*jobs)
        L.check(lib.scl_conv_pack_batch(arr, len(jobs), L.stream_of(keep[0])))
    return len(jobs)


def fresh_vlad_planes(w):
    """(plane images, generation) of the assignment weights ``w`` if the LAST prepack() of this
    thread wrote them for exactly this tensor, else None — consumed by the call: the images are
    valid for the forward pass that packed them and are handed over once.  (Round 4 looked them up
    by storage address + ``_version``; an in-place optimizer step or a ``.data`` write moves
    neither, so a direct nets.netvlad() call after one silently used the old W.)"""
    got, _FWD.planes = _FWD.planes, None
    if got is None or not USE_PREPACK or got[0] != w.data_ptr() or got[1].device != w.device:
        return None
    return got[1], got[2]


def _packed_for(w, transposed):
    """The up-to-date packed image of ``w`` for this direction, or None."""
    if not USE_PREPACK:
        return None
    with _PACKED_LOCK:
        ent = _PACKED.get((w.data_ptr(), bool(transposed), _pack_slot()))
    if ent is None or ent[0]() is None or ent[1] != w._version or ent[2] != tuple(w.shape):
        return None
    return ent[3]


def _lib_weight(w, x):
    """The weight as the library convolution wants it: activation dtype, channels-last."""
    if w.dtype == x.dtype and w.is_contiguous(memory_format=_CL):
        return w
    return w.to(dtype=x.dtype, memory_format=_CL)


def _own_conv_kind(x, w, transposed=False):
    """'reg' (weights in registers, csrc/conv64.hip), 'lds' (weights streamed through LDS,
    csrc/convg.hip) or None (library)."""
    if not (USE_CONV64 and x.is_cuda and x.dtype == torch.bfloat16 and w.dtype in _W_DTYPES
            and x.dim() == 4 and w.dim() == 4 and tuple(w.shape[2:]) == (3, 3)):
        return None
    cin, kout = (w.shape[0], w.shape[1]) if transposed else (w.shape[1], w.shape[0])
    if x.shape[1] != cin:
        return None
    if (cin, kout) in _OWN_CONV_SHAPES:
        return 'reg'
    if USE_CONVG and cin % 32 == 0 and kout % 128 == 0 and cin >= 128 and max(cin, kout) <= 1024:
        return 'lds'
    return None


def _lds_conv_pays(x, transposed=False, fused_tail=False, kout=512):
    """Own LDS-weights kernel (csrc/convh.hip) or the library?  Measured on MI355X
    (scripts/conv_layers.py; profiles/r02 for the bench shapes, profiles/r05/conv_layers_*.txt for
    the small maps): what decides is how many workgroup tiles ([8 rows x 40 px] x 128 output
    channels) the launch has for the chip's 256 CUs, not the map size —
      * 24 x 30x40 (conv5_x of the bench step, 384 tiles): own 1070-1110 vs 910 / 570-600 TFLOP/s;
      * 25 x 22x30 and 25 x 11x15 (conv4_x / conv5_x at the reference's own training resolution,
        240 x 180, train/train.py:423-428; 1200 / 200 tiles): forward 61-104 vs 67-120 us and
        67 vs 64 us, backward-data 64-103 vs 131-188 and 67 vs 88 us;
      * 24 x 14x14 (192 tiles): forward 72 vs 65 us, backward-data 71 vs 88;
      * 4 x 28x28 / 4 x 14x14 (configs[0]; 64 / 32 tiles): forward 59 vs 49 and 58 vs 36 us — the
        library wins when most CUs would get no tile at all; backward-data 60 vs 67 / 58 vs 39.
    Round 4 gated on the map size (>= 30 x 40) and so handed conv4_x / conv5_x of the reference's
    real training shape to the library: 39 % of its convolution FLOPs."""
    b, _, h, w = x.shape
    if h * w >= 30 * 40:
        return True                  # rounds 2-4: measured at the bench batch; small batches of large
                                     # maps (inference passes of 1-4 images) keep the own, deterministic kernels
    tiles = b * -(-h // 8) * -(-w // 40) * max(int(kout) // 128, 1)
    return tiles >= (48 if transposed else 192)


def _conv64_ok(x, w, transposed=False):
    return _own_conv_kind(x, w, transposed) is not None


# The forward pass of the backbone as two half-batches on two streams.  Every kernel of the
# backbone treats images independently, and the one-workgroup-per-CU convolution kernels end in a
# partial round (960 tiles on 256 CUs = 3.75): with the halves of the batch pipelined on two
# streams, the next layer's kernel of one half fills the CUs the other half's kernel leaves idle.
# Autograd sees full-batch tensors and one node per layer as before — only the launches inside a
# forward are split; VGG16NetVLAD.features opens the window (_FWD.split) and joins both streams
# before it returns.  Measured (scripts/fwd_split_probe.py, 24 x 640x480): the forward alone
# 4.29 -> 3.97 ms, but a whole training step only 12.81 -> 12.70 ms (same-box bench.py A/B:
# within noise) — the backward pass, whose two streams already fill the chip, gives part of it
# back.  Hence: on for inference (feature extraction: torch.no_grad), off for training unless
# SCL_SPLIT_FWD=1; SCL_SPLIT_FWD=0 turns it off everywhere.
_SPLIT_FWD_ENV = os.environ.get('SCL_SPLIT_FWD', 'infer')
USE_SPLIT_FWD = None if _SPLIT_FWD_ENV == 'infer' else _SPLIT_FWD_ENV != '0'


def _split_fwd_wanted():
    return (not torch.is_grad_enabled()) if USE_SPLIT_FWD is None else bool(USE_SPLIT_FWD)


_FWD_STREAMS = {}


class _FwdState(threading.local):
    """Per calling thread (the reference drives one session from three threads,
    train/train.py:967-975): the window features() opens, and what must stay alive in it."""
    split = None         # (stream A, stream B) while features() pipelines the halves
    keep = None
    planes = None        # (weight address, plane images, generation) of this thread's last prepack()


_FWD = _FwdState()


@contextlib.contextmanager
def _whole_batch_op():
    """An op of the forward pass that is NOT split into halves (a library convolution, a glue
    pass): it runs on the caller's stream, joined with both half-batch streams on either side."""
    sp = _FWD.split
    if sp is None:
        yield
        return
    cur = torch.cuda.current_stream(sp[0].device)
    cur.wait_stream(sp[0])
    cur.wait_stream(sp[1])
    # nothing inside splits again: a nested conv64 would otherwise launch its halves on the
    # half-batch streams while the glue pass that follows reads the result on `cur`
    _FWD.split = None
    try:
        yield
    finally:
        _FWD.split = sp
        sp[0].wait_stream(cur)
        sp[1].wait_stream(cur)


def _on_half(stream, *tensors):
    """The tensors are used by kernels on `stream`, not on the stream they were allocated on.
    Under no_grad an activation would be freed — and its block handed to the next layer's
    output — while the half-batch streams may still be reading it: keep a reference until
    features() has joined the streams.  (Tensor.record_stream would do, but it makes the
    caching allocator defer every reuse behind events: measured 12.9 -> 18 ms per step.)"""
    _FWD.keep.extend(t for t in tensors if t is not None)


def _split_halves(b):
    """[(stream, lo, hi)] of the two half-batches, or None outside a split forward."""
    sp = _FWD.split
    if sp is None or b < 2:
        return None
    return [(sp[0], 0, b // 2), (sp[1], b // 2, b)]


def conv64(x, w, transposed=False, bias=None, relu=False, pool=False, mask=None, out=None,
           pooled_out=None, pool_idx=None):
    """3x3 same-padding convolution on bf16 channels-last activations (``scl_conv3x3_fused``)
    for the shapes of ``_OWN_CONV_SHAPES``; ``transposed`` gives the gradient with respect to
    the input of ``conv(., w)``.  ``bias`` (float32 [kout]) and ``relu`` fuse the layer's tail
    into the epilogue; ``pool=True`` returns ``(raw conv, relu(maxpool2x2(raw) + bias))``;
    ``mask`` (bf16, the output's shape) multiplies the result by ``[mask > 0]`` — the ReLU' of
    the layer below fused into a backward-data pass.  ``pool_idx`` (with ``transposed`` and
    ``mask``, register kernels with cin == kout): ``x`` is the gradient at the POOLED map and the
    kernel un-pools it by the forward pass's window positions while staging
    (``scl_conv3x3_masked_pooled``)."""
    lib = L.load()
    L.require_device(x, w, bias, mask, pool_idx)
    x = x.contiguous(memory_format=_CL)
    b, _, h, wd = x.shape
    if pool_idx is not None:
        if mask is None or not transposed:
            raise ValueError("pool_idx goes with the masked backward-data pass")
        h, wd = mask.shape[2], mask.shape[3]
        if (h % 2 or wd % 2 or tuple(x.shape[2:]) != (h // 2, wd // 2)
                or tuple(pool_idx.shape) != tuple(x.shape) or pool_idx.dtype != torch.uint8):
            raise ValueError("pooled gradient / index must be [B,C,H/2,W/2] (uint8 index), H and W even")
    cin, kout = (w.shape[0], w.shape[1]) if transposed else (w.shape[1], w.shape[0])
    inner = out is not None            # one half of a split forward: no further splitting
    if out is None:
        out = torch.empty((b, kout, h, wd), dtype=x.dtype, device=x.device, memory_format=_CL)
    pooled = pooled_out
    if pool:
        if bias is None:
            raise ValueError("pool=True needs the bias (the pooled map is relu(pool + bias))")
        if pooled is None:
            pooled = torch.empty((b, kout, h // 2, wd // 2), dtype=x.dtype, device=x.device,
                                 memory_format=_CL)
    if bias is not None:
        bias = bias.float().contiguous()
    halves = None if (inner or transposed or mask is not None) else _split_halves(b)
    if halves is not None:
        for stream, lo, hi in halves:
            _on_half(stream, x, out, pooled, bias, w)
            with torch.cuda.stream(stream):
                conv64(x[lo:hi], w, False, bias, relu, pool, None, out[lo:hi],
                       pooled[lo:hi] if pool else None)
        return (out, pooled) if pool else out
    sk, sc, sh, sw = w.stride()
    px = b * h * wd
    if pool_idx is not None:                # pooled gradient bf16 / 4 + index / 4; mask; output
        _work('conv3x3_kernel<pooled>', 2.0 * px * cin * kout * 9, px * (0.75 * cin + 4.0 * kout))
    elif (cin, kout) in _OWN_CONV_SHAPES:
        _work('conv3x3_kernel', 2.0 * px * cin * kout * 9,
              2.0 * px * (cin + kout * (1 + (mask is not None) + 0.25 * bool(pool))))
    else:
        _lds_work(2.0 * px * cin * kout * 9, 2.0 * px * (cin + kout * (1 + (mask is not None))))
    pk = _packed_for(w, transposed)
    wp = L.ptr(w) if pk is None else L.ptr(pk)            # the weight, or its packed image
    wflags = (int(bool(transposed)) | _wflag(w)) if pk is None else (int(bool(transposed)) | L.W_PACKED)
    if mask is not None:
        if bias is not None or pool or relu:
            raise ValueError("mask excludes the forward tails (bias / relu / pool)")
        if tuple(mask.shape) != tuple(out.shape) or mask.dtype != x.dtype:
            raise ValueError("mask must have the output's shape and dtype")
        mask = mask.contiguous(memory_format=_CL)
        own = (cin, kout) in _OWN_CONV_SHAPES
        if pool_idx is not None:
            if not (own and cin == kout):
                raise ValueError("un-pooling window staging exists for the 64->64 and 128->128 kernels")
            ws = L.workspace(lib.scl_conv3x3_workspace_bytes(), x.device)
            L.check(lib.scl_conv3x3_masked_pooled(
                L.ptr(x), L.ptr(pool_idx.contiguous(memory_format=_CL)), wp, sk, sc, sh, sw, wflags,
                b, h, wd, cin, kout, L.ptr(out), L.ptr(mask), L.ptr(ws), ws.numel(), L.stream_of(x)))
            return out
        ws = L.workspace(lib.scl_conv3x3_workspace_bytes() if own
                         else lib.scl_convg_workspace_bytes(cin, kout), x.device)
        fn = lib.scl_conv3x3_masked if own else lib.scl_convg_masked
        L.check(fn(L.ptr(x), wp, sk, sc, sh, sw, wflags, b, h, wd,
                   cin, kout, L.ptr(out), L.ptr(mask), L.ptr(ws), ws.numel(), L.stream_of(x)))
        return out
    if (cin, kout) not in _OWN_CONV_SHAPES:
        if pool:
            raise ValueError("the fused pooling epilogue exists for the register kernels only")
        ws = L.workspace(lib.scl_convg_workspace_bytes(cin, kout), x.device)
        L.check(lib.scl_convg(L.ptr(x), wp, sk, sc, sh, sw, wflags,
                              b, h, wd, cin, kout, L.ptr(out), L.ptr(bias), int(bool(relu)),
                              L.ptr(ws), ws.numel(), L.stream_of(x)))
        return out
    ws = L.workspace(lib.scl_conv3x3_workspace_bytes(), x.device)
    L.check(lib.scl_conv3x3_fused(L.ptr(x), wp, sk, sc, sh, sw,
                                  wflags, b, h, wd, cin, kout,
                                  L.ptr(out), L.ptr(bias), int(bool(relu)), L.ptr(pooled),
                                  L.ptr(ws), ws.numel(), L.stream_of(x)))
    return (out, pooled) if pool else out


def _own_wrw_ok(x, gz, w):
    return (USE_CONV64 and USE_WRW and x.is_cuda and x.dtype == torch.bfloat16
            and gz.dtype == torch.bfloat16 and w.dtype in _W_DTYPES and w.dim() == 4
            and tuple(w.shape[2:]) == (3, 3) and w.shape[0] % 64 == 0 and w.shape[1] % 64 == 0
            and max(w.shape[0], w.shape[1]) <= 1024)


def wrw64(x, gz, w_like, bias_grad=None, pool_idx=None):
    """Weight gradient of a 3x3 same-padding convolution whose channel counts are multiples
    of 64 (``scl_wrw3x3_bias``): [kout,cin,3,3] with the dtype (bf16 or float32) and strides of
    ``w_like``.  ``bias_grad`` (float32 [kout]) receives the column sums of ``gz`` — the bias
    gradient — from the same pass.  With ``pool_idx`` (uint8 [B,kout,H/2,W/2], conv_pool_idx's
    second result) ``gz`` is the gradient at the POOLED map of the same shape and the kernel
    un-pools it while staging (``scl_wrw3x3_pooled``)."""
    lib = L.load()
    L.require_device(x, gz, pool_idx)
    x = x.contiguous(memory_format=_CL)
    gz = gz.contiguous(memory_format=_CL)
    b, cin, h, wd = x.shape
    kout = gz.shape[1]
    gw = _grad_out(w_like)
    ws = L.workspace(lib.scl_wrw3x3_workspace_bytes(cin, kout), x.device)
    sk, sc, sh, sw = gw.stride()
    if bias_grad is not None and not (bias_grad.dtype == torch.float32 and bias_grad.is_contiguous()
                                      and bias_grad.numel() == kout):
        raise ValueError("bias_grad must be a contiguous float32 vector of kout elements")
    if pool_idx is not None:
        if (h % 2 or wd % 2 or tuple(gz.shape) != (b, kout, h // 2, wd // 2)
                or tuple(pool_idx.shape) != tuple(gz.shape) or pool_idx.dtype != torch.uint8):
            raise ValueError("pooled gradient / index must be [B,kout,H/2,W/2] (uint8 index), H and W even")
        pool_idx = pool_idx.contiguous(memory_format=_CL)
        _work('wrw64_kernel<pooled>', 2.0 * b * h * wd * cin * kout * 9,
              b * h * wd * (2.0 * cin + 0.75 * kout))
        L.check(lib.scl_wrw3x3_pooled(L.ptr(x), L.ptr(gz), L.ptr(pool_idx), b, h, wd, cin, kout,
                                      L.ptr(gw), sk, sc, sh, sw, int(gw.dtype == torch.float32),
                                      L.ptr(bias_grad), L.ptr(ws), ws.numel(), L.stream_of(x)))
        return gw
    _work('wrw64_kernel', 2.0 * b * h * wd * cin * kout * 9, 2.0 * b * h * wd * (cin + kout))
    L.check(lib.scl_wrw3x3_bias(L.ptr(x), L.ptr(gz), b, h, wd, cin, kout, L.ptr(gw), sk, sc, sh, sw,
                                int(gw.dtype == torch.float32), L.ptr(bias_grad), L.ptr(ws),
                                ws.numel(), L.stream_of(x)))
    return gw


def _conv3x3(x, w):
    """3x3 / stride 1 / same-padding convolution without bias (MIOpen, or conv64)."""
    kind = _own_conv_kind(x, w)
    if kind == 'reg' or (kind == 'lds' and _lds_conv_pays(x, kout=w.shape[0])):
        return conv64(x, w, False)
    return torch.ops.aten.convolution(x, _lib_weight(w, x), None, _ONES, _ONES, _ONES, False,
                                      [0, 0], 1)


def _wrw_pays(x):
    """scripts/conv_layers.py: own 850-1070 vs library 370-750 TFLOP/s on conv1_2 .. conv4_x,
    900 vs 600 at 30 x 40 (conv5_x) with the 32 x 8 tile shape for narrow maps.  Round 5
    (profiles/r05/conv_layers_*.txt): by pixel tiles of the launch, not by map size — 25 x 11x15
    (50 tiles of 128 pixels) own 69 vs 83 us, 24 x 14x14 (48) 64 vs 82, 4 x 28x28 (28) 44-56 vs
    42-62, 4 x 14x14 (8 tiles) 44 vs 27: the library below 32 tiles."""
    b, _, h, w = x.shape
    if h * w >= 30 * 40:
        return True
    tiles = b * min(-(-h // 16) * -(-w // 8), -(-h // 4) * -(-w // 32))
    return tiles >= 32


class _GradLink:
    """Hand-off between two ADJACENT layers of a purely sequential chain (the producer's output
    has this one consumer): the upper layer's backward-data pass applies the lower layer's
    ReLU' in its epilogue and records the buffer here; the lower layer's backward, which runs
    next, takes it and skips its own masking pass.  Anything else (a copied or re-accumulated
    gradient, a library backward) leaves the link empty and the lower layer masks itself —
    masking twice would be harmless, skipping it is only done on this exact buffer."""
    __slots__ = ('ptr', 'fused', 'first', 'first_done')

    def __init__(self):
        self.ptr = None
        self.fused = None      # (data_ptr of y1, pooled map, window index): conv1_2's forward
                               # results, computed by the kernel that produced y1 (_FirstConv)
        self.first = None      # (x0, w1, bias1, avg): set by _FirstConv.forward — the layer above may
                               # compute conv1_1's weight / bias / mean gradient in its own
                               # backward-data kernel instead of writing the gradient map
        self.first_done = None  # (gw1, gb1, davg) once it has

    def mark(self, gx):
        self.ptr = gx.data_ptr()

    def take(self, gy):
        hit = self.ptr is not None and self.ptr == gy.data_ptr()
        self.ptr = None
        return hit


USE_BIAS_IN_WRW = os.environ.get('SCL_BIAS_IN_WRW', '1') != '0'
USE_MASKED_BWD = os.environ.get('SCL_MASKED_BWD', '1') != '0'
USE_POOL_IDX = os.environ.get('SCL_POOL_IDX', '1') != '0'
USE_POOLED_BWD = os.environ.get('SCL_POOLED_BWD', '1') != '0'     # un-pool inside the consumers
USE_F32_WEIGHTS = os.environ.get('SCL_F32_WEIGHTS', '1') != '0'
# Round 5: conv1_2's backward-data kernel keeps its output tile in LDS and multiplies it with the
# im2col of x0 right there (scl_conv3x3_masked_pooled_first_wrw): the 944 MB gradient map at
# conv1_1's pre-activation is neither written nor read, conv_first_wrw_kernel is not launched.
# As a kernel it is slower than the two it replaces; the step gains where the weight-gradient
# kernels run on the second stream next to it (they no longer compete with 1.9 GB of traffic):
# 'auto' (default) = with the second stream (autotune_side_wrw decides both per device), 1 / 0 pin it.
_FFW_ENV = os.environ.get('SCL_FUSED_FIRST_WRW', 'auto')
USE_FUSED_FIRST_WRW = None if _FFW_ENV == 'auto' else _FFW_ENV != '0'


def _fused_first_wrw_wanted():
    if USE_FUSED_FIRST_WRW is None:
        return bool(USE_SIDE_WRW) and GRAD_SINK is not None
    return bool(USE_FUSED_FIRST_WRW)
# conv1_1 and conv1_2 of the forward pass in one kernel (scl_conv_first_pool_idx): bit-identical
# to the two-kernel path and 1.25 GB less read per step, but SLOWER as built (776 against 243 + 480
# us on one box, profiles/r04/first_block_one_kernel_vs_two.txt): conv1_2's kernel sits at 256
# registers with two waves per SIMD, so the first layer's products, gathers and bias / ReLU /
# rounding run in a phase of their own in front of the K loop instead of under it.  Off by default.
USE_FUSED12 = os.environ.get('SCL_FUSED12', '0') != '0'


def conv_pool_idx(x, w, bias, out=None):
    """conv -> +bias -> max-pool 2x2 -> ReLU with the pooling in the convolution's epilogue and
    NO full-size output: returns (pooled bf16 [B,K,H/2,W/2], idx uint8 of the same shape: the
    window position of each maximum) — ``scl_conv3x3_pool_idx``."""
    lib = L.load()
    L.require_device(x, w, bias)
    x = x.contiguous(memory_format=_CL)
    b, cin, h, wd = x.shape
    kout = w.shape[0]
    if out is None:
        a = torch.empty((b, kout, h // 2, wd // 2), dtype=x.dtype, device=x.device, memory_format=_CL)
        idx = torch.empty((b, kout, h // 2, wd // 2), dtype=torch.uint8, device=x.device,
                          memory_format=_CL)
    else:
        a, idx = out
    bias = bias.float().contiguous()
    halves = _split_halves(b) if out is None else None
    if halves is not None:
        for stream, lo, hi in halves:
            _on_half(stream, x, a, idx, bias, w)
            with torch.cuda.stream(stream):
                conv_pool_idx(x[lo:hi], w, bias, (a[lo:hi], idx[lo:hi]))
        return a, idx
    sk, sc, sh, sw = w.stride()
    own = (cin, kout) in _OWN_CONV_SHAPES
    if own:
        _work('conv3x3_kernel', 2.0 * b * h * wd * cin * kout * 9,
              b * h * wd * (2.0 * cin + 0.75 * kout))       # in + pooled bf16 / 4 + index / 4
    else:
        _lds_work(2.0 * b * h * wd * cin * kout * 9, b * h * wd * (2.0 * cin + 0.75 * kout))
    ws = L.workspace(lib.scl_conv3x3_workspace_bytes() if own
                     else lib.scl_convg_workspace_bytes(cin, kout), x.device)
    fn = lib.scl_conv3x3_pool_idx if own else lib.scl_convg_pool_idx
    pk = _packed_for(w, False)
    L.check(fn(L.ptr(x), L.ptr(w) if pk is None else L.ptr(pk), sk, sc, sh, sw,
               _wflag(w) if pk is None else L.W_PACKED, b, h, wd, cin, kout,
               L.ptr(bias.float().contiguous()), L.ptr(a), L.ptr(idx), L.ptr(ws), ws.numel(),
               L.stream_of(x)))
    return a, idx


def _own_wrw_used(x, gz, w):
    return _own_wrw_ok(x, gz, w) and _wrw_pays(x)


# Weight gradients on a second stream.  Nothing in the backward chain reads a weight gradient:
# with the gradient sink it goes straight into the flat buffer, and only the bucket's
# all-reduce / the optimizer need it.  Run next to the backward-data kernel of the same layer
# its workgroups fill the CUs that the other kernel's last, partial round leaves idle (both are
# one-workgroup-per-CU kernels: 960 tiles on 256 CUs is 3.75 rounds).  The sink joins the
# stream before a collective or the optimizer touches the buffer (GradBuckets.note_stream).
USE_SIDE_WRW = os.environ.get('SCL_SIDE_WRW', '1') != '0'
_SIDE = {}


def autotune_side_wrw(step, steps=3, rounds=2):
    """Decide USE_SIDE_WRW for THIS device by measurement: ``step()`` (one training step, already
    warmed up) is timed ``steps`` at a time with the second stream off and on, ``rounds`` times
    alternating, and the faster setting is kept.  The second stream is worth -4 % of a step on some
    MI355X boxes and costs +1.4 % on others (profiles/r02/README.md): the clocks the two settings
    hold differ from device to device, so a fixed default loses on part of the pool.  Gradients are
    bit-identical either way.  Returns {'chosen': bool, 'ms_on': .., 'ms_off': ..}."""
    global USE_SIDE_WRW, USE_FUSED_FIRST_WRW
    if not torch.cuda.is_available():
        return {'chosen': USE_SIDE_WRW, 'ms_on': None, 'ms_off': None}
    # Round 5: the fused first-layer gradients (USE_FUSED_FIRST_WRW) pay only next to the second
    # stream; unless pinned by SCL_FUSED_FIRST_WRW the three settings (one stream / two streams /
    # two streams + fused) are timed together and the fastest kept.
    fused_free = _FFW_ENV == 'auto'
    settings = [(False, False), (True, False)] + ([(True, True)] if fused_free else [])
    if not fused_free:
        settings = [(False, bool(USE_FUSED_FIRST_WRW)), (True, bool(USE_FUSED_FIRST_WRW))]
    best = {st: float('inf') for st in settings}
    for _ in range(rounds):
        for st in settings:
            USE_SIDE_WRW, USE_FUSED_FIRST_WRW = st
            step()                                          # settle into the mode
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(steps):
                step()
            e1.record()
            torch.cuda.synchronize()
            best[st] = min(best[st], e0.elapsed_time(e1) / steps)
    pick = min(settings, key=lambda st: (best[st], st))
    USE_SIDE_WRW, USE_FUSED_FIRST_WRW = pick
    on = min(best[st] for st in settings if st[0])
    out = {'chosen': USE_SIDE_WRW, 'ms_on': round(on, 3), 'ms_off': round(best[settings[0]], 3),
           'fused_first_wrw': bool(USE_FUSED_FIRST_WRW)}
    if fused_free:
        out['ms_on_fused_first_wrw'] = round(best[(True, True)], 3)
        out['ms_on_two_kernels'] = round(best[(True, False)], 3)
    return out


def _wrw_maybe_async(x, gz, w, gb, pool_idx=None):
    sink = GRAD_SINK

    def in_sink(t):                  # written where only the sink's consumers will read it
        flat = sink.flat
        return flat.data_ptr() <= t.data_ptr() < flat.data_ptr() + flat.numel() * flat.element_size()
    if not (USE_SIDE_WRW and sink is not None and hasattr(sink, 'note_stream')
            and sink.view(w) is not None and sink.view(w).stride() == w.stride()
            and (gb is None or in_sink(gb))):
        # (a gradient that goes back to autograd is read on the current stream right away)
        return wrw64(x, gz, w, gb, pool_idx)
    dev = x.device
    side = _SIDE.get(dev)
    if side is None:
        side = _SIDE[dev] = torch.cuda.Stream(device=dev)
    side.wait_stream(torch.cuda.current_stream(dev))
    with torch.cuda.stream(side):
        gw = wrw64(x, gz, w, gb, pool_idx)
    # x and gz are read on `side`; they must not go back to the allocator before that stream
    # is joined.  Tensor.record_stream would say so, but it makes the allocator hold the blocks
    # behind events and, every few runs, hipMalloc a new one in the middle of a step (one step
    # of 0.5 s in half of the bench runs): the sink keeps the references until finish() instead.
    sink.note_stream(side, x, gz, *([] if pool_idx is None else [pool_idx]))
    return gw


def _masked_pooled_first_wrw(ga, idx, w2, y1, first):
    """conv1_2's backward-data pass fused with conv1_1's parameter gradients
    (``scl_conv3x3_masked_pooled_first_wrw``): ga / idx = gradient at the pooled map and window
    positions, w2 = conv1_2's weight, y1 = conv1_1's output (the ReLU' mask), ``first`` = (x0, w1,
    bias1) from _FirstConv.forward.  Returns (gw1, gb1, davg), written where the gradient sink
    wants them."""
    lib = L.load()
    x0, w1, bias1 = first
    L.require_device(ga, idx, w2, y1, x0, w1)
    ga = ga.contiguous(memory_format=_CL)
    idx = idx.contiguous(memory_format=_CL)
    y1 = y1.contiguous(memory_format=_CL)
    b, _, h, wd = y1.shape
    gw1 = _grad_out(w1)
    gb1 = _grad_out(bias1) if bias1.dtype == torch.float32 else torch.empty(
        64, dtype=torch.float32, device=y1.device)
    davg = torch.empty(3, dtype=torch.float32, device=y1.device)
    w1c = w1
    if w1c.stride() != gw1.stride() or w1c.dtype != gw1.dtype:
        w1c = w1.to(gw1.dtype).contiguous(memory_format=_CL) if gw1.is_contiguous(memory_format=_CL) \
            else w1.to(gw1.dtype).contiguous()
    pk = _packed_for(w2, True)
    wp = L.ptr(w2) if pk is None else L.ptr(pk)
    wflags = _wflag(w2) if pk is None else L.W_PACKED
    s2 = w2.stride()
    s1 = gw1.stride()
    ws = L.workspace(lib.scl_conv3x3_workspace_bytes(), y1.device)
    fws = L.workspace(lib.scl_conv_first_wrw_workspace_bytes(), y1.device)
    px = b * h * wd
    _work('conv3x3_kernel<pooled+first_wrw>', 2.0 * px * (64 * 64 * 9 + 64 * 28), px * (0.75 * 64 + 2.0 * 64 + 6.0))
    L.check(lib.scl_conv3x3_masked_pooled_first_wrw(
        L.ptr(ga), L.ptr(idx), wp, s2[0], s2[1], s2[2], s2[3], wflags, b, h, wd, L.ptr(y1), L.ptr(x0),
        L.ptr(gw1), s1[0], s1[1], s1[2], s1[3], int(gw1.dtype == torch.float32), L.ptr(gb1), L.ptr(w1c),
        L.ptr(davg), L.ptr(ws), ws.numel(), L.ptr(fws), fws.numel(), L.stream_of(y1)))
    return gw1, gb1, davg


def _conv3x3_backward(gz, x, w, need_x, link=None, gb=None, pooled=None):
    """(gx, gw) of a 3x3 convolution.  With ``link`` (x is a post-ReLU map whose producer
    holds the other end) an own backward-data kernel returns gx * [x > 0] and marks the
    link.  ``gb`` (only where ``_own_wrw_used``): the weight-gradient kernel also writes the
    bias gradient there.  ``pooled`` = (gradient at the pooled map, window positions) of a layer
    that ends in the 2x2 max-pooling: the own weight-gradient kernel reads that instead of the
    full-size ``gz``, and so does the backward-data kernel where it can (then ``gz`` may be
    None: see ``_pooled_consumers``)."""
    ga, idx = pooled if pooled is not None else (None, None)
    g_any = ga if gz is None else gz                      # (channels and dtype are what is looked at)
    kind = _own_conv_kind(g_any, w, True)
    own_gx = kind == 'reg' or (kind == 'lds' and _lds_conv_pays(x, True, kout=w.shape[1]))
    own_gw = _own_wrw_used(x, g_any, w)
    if gb is not None and not own_gw:
        raise RuntimeError("bias gradient requested from a weight-gradient pass that is not own")
    if pooled is not None and not own_gw:
        raise RuntimeError("pooled gradient handed to a weight-gradient pass that is not own")

    def own_wrw():
        if pooled is not None:
            return _wrw_maybe_async(x, ga, w, gb, pool_idx=idx)
        return _wrw_maybe_async(x, gz, w, gb)
    if own_gx and need_x and link is not None and USE_MASKED_BWD:
        if (gz is None and _fused_first_wrw_wanted() and link.first is not None and own_gw
                and tuple(w.shape) == (64, 64, 3, 3) and x.shape[2] % 2 == 0 and x.shape[3] % 2 == 0):
            # x is conv1_1's output: its gradient map has ONE consumer, the first layer's weight /
            # bias / mean gradient — computed here, the map itself never exists.  What autograd
            # carries down is an UNINITIALISED tensor of its shape that _FirstConv.backward must
            # recognise (same hand-off as the ReLU' mark) and never read.
            gx = torch.empty_like(x)
            link.first_done = _masked_pooled_first_wrw(ga, idx, w, x, link.first)
            link.mark(gx)
            return gx, own_wrw()
        if gz is None:
            gx = conv64(ga, w, True, mask=x, pool_idx=idx)
        else:
            gx = conv64(gz, w, True, mask=x)
        link.mark(gx)
        if own_gw:
            return gx, own_wrw()
        _, gw, _ = torch.ops.aten.convolution_backward(gz, x, _lib_weight(w, gz), None, _ONES,
                                                       _ONES, _ONES, False, [0, 0], 1,
                                                       [False, True, False])
        return gx, gw.to(w.dtype)
    if gz is None:
        if need_x:
            raise RuntimeError("the full-size gradient is needed for this backward-data pass")
        return None, own_wrw()
    if own_gx and own_gw:
        gw = own_wrw()
        return (conv64(gz, w, True) if need_x else None), gw
    if own_gx or own_gw:
        gx = conv64(gz, w, True) if (own_gx and need_x) else None
        gw = own_wrw() if own_gw else None
        if (gx is None and need_x) or gw is None:
            lx, lw, _ = torch.ops.aten.convolution_backward(
                gz, x, _lib_weight(w, gz), None, _ONES, _ONES, _ONES, False, [0, 0], 1,
                [bool(need_x) and gx is None, gw is None, False])
            gx = lx if gx is None else gx
            gw = lw.to(w.dtype) if gw is None else gw
        return gx, gw
    gx, gw, _ = torch.ops.aten.convolution_backward(gz, x, _lib_weight(w, gz), None, _ONES, _ONES,
                                                    _ONES, False, [0, 0], 1,
                                                    [bool(need_x), True, False])
    return gx, gw.to(w.dtype)


def _pooled_consumers(x, ga, w, need_x, link):
    """Which consumers of a pooling layer's gradient can un-pool while they stage:
    (weight gradient, backward-data).  Weight gradient: every own shape (csrc/conv64.hip,
    wrw64_kernel<.., 1>).  Backward-data: the register kernels with cin == kout — conv1_2 and
    conv2_2 — in their masked form; conv3_3 / conv4_3 fill their windows by LDS-DMA, which
    copies bytes as they lie."""
    h, wd = x.shape[2], x.shape[3]
    if not (USE_POOLED_BWD and h % 2 == 0 and wd % 2 == 0 and _own_wrw_used(x, ga, w)):
        return False, False
    kind = _own_conv_kind(ga, w, True)
    return True, (not need_x) or (kind == 'reg' and w.shape[0] == w.shape[1]
                                  and link is not None and USE_MASKED_BWD)


def _glue_dtype(t):
    if t.dtype == torch.float32:
        return L.DT_F32
    if t.dtype == torch.bfloat16:
        return L.DT_BF16
    raise ValueError("backbone activations must be float32 or bfloat16, got %s" % t.dtype)


class _ConvBiasAct(torch.autograd.Function):
    """conv -> (+bias, optional ReLU) with the elementwise part in ONE in-place HIP pass
    (csrc/vgg_glue.hip) instead of separate bias-add and ReLU kernels; the backward fuses
    ReLU' with the bias-gradient reduction.  Only the post-activation map is kept."""

    @staticmethod
    def forward(ctx, x, w, bias, relu, link_in=None, link_out=None):
        lib = L.load()
        ctx.link_in, ctx.link_out = link_in, (link_out if relu else None)
        ctx.slot = _pack_slot()
        kind = _own_conv_kind(x, w)
        if kind == 'reg' or (kind == 'lds' and _lds_conv_pays(x, False, True, kout=w.shape[0])):
            y = conv64(x, w, False, bias=bias, relu=relu)         # tail fused in the epilogue
        else:
            with _whole_batch_op():
                y = _conv3x3(x, w).contiguous(memory_format=_CL)
                b, c, h, wd = y.shape
                _work('bias_act_kernel', 0.0, 2.0 * y.numel() * y.element_size())
                L.check(lib.scl_vgg_bias_act(L.ptr(y), _glue_dtype(y), L.ptr(bias), b * h * wd, c,
                                             int(relu), L.stream_of(y)))
        ctx.relu = relu
        ctx.save_for_backward(x, w, y if relu else None, bias)
        return y

    @staticmethod
    def backward(ctx, gy):
        lib = L.load()
        x, w, y, bias = ctx.saved_tensors
        gy = gy.contiguous(memory_format=_CL)
        b, c, h, wd = gy.shape
        gb = _grad_out(bias) if bias.dtype == torch.float32 else torch.empty(
            c, dtype=torch.float32, device=gy.device)
        ws = L.workspace(lib.scl_vgg_workspace_bytes(c), gy.device)
        # ReLU' already applied by the layer above (its backward-data epilogue)?
        masked = ctx.link_out is not None and ctx.link_out.take(gy)
        mask_here = ctx.relu and not masked
        gz = torch.empty_like(gy) if mask_here else gy
        # nothing to mask: the bias gradient (column sums of gy) comes out of the own
        # weight-gradient kernel, which has gy in LDS anyway — no separate pass over the map
        fold = USE_BIAS_IN_WRW and not mask_here and _own_wrw_used(x, gy, w)
        if not fold:
            _work('act_bwd_kernel', 0.0, (3.0 if mask_here else 1.0) * gy.numel() * gy.element_size())
            L.check(lib.scl_vgg_act_bwd(L.ptr(gy), L.ptr(y) if mask_here else None,
                                        _glue_dtype(gy), b * h * wd, c,
                                        L.ptr(gz) if mask_here else None, L.ptr(gb), L.ptr(ws),
                                        ws.numel(), L.stream_of(gy)))
        with _in_slot(ctx.slot):
            gx, gw = _conv3x3_backward(gz, x, w, ctx.needs_input_grad[0], ctx.link_in,
                                       gb if fold else None)
        return gx, _grad_ret(gw, w), _grad_ret(gb, bias), None, None, None


class _ConvBiasPoolReLU(torch.autograd.Function):
    """conv -> +bias -> max-pool 2x2/2 -> ReLU (model/nets.py:40-42) with the three
    elementwise ops in one HIP pass; the backward recomputes the arg-max from the saved
    conv output (no int64 index tensor) and fuses ReLU' and the bias gradient."""

    @staticmethod
    def forward(ctx, x, w, bias, link_in=None, link_out=None):
        lib = L.load()
        ctx.link_in, ctx.link_out = link_in, link_out
        ctx.by_idx = False
        ctx.slot = _pack_slot()
        fused = link_in.fused if link_in is not None else None
        if fused is not None:
            link_in.fused = None
            if fused[0] == x.data_ptr() and fused[3] is w and fused[4] is bias:
                # conv1_2: the kernel that wrote x (= y1) computed this layer's forward as well
                a, idx = fused[1], fused[2]
                ctx.by_idx = True
                ctx.save_for_backward(x, w, idx, a, bias)
                return a
            raise RuntimeError("fused conv1_1 + conv1_2 results do not belong to this layer")
        kind = _own_conv_kind(x, w)
        if USE_POOL_IDX and ((kind == 'reg' and w.shape[0] == w.shape[1])
                             or (kind == 'lds' and _lds_conv_pays(x, kout=w.shape[0]))):
            # conv1_2 .. conv4_3: pooled map and the position of each maximum from the epilogue;
            # the full-size convolution output is never written
            a, idx = conv_pool_idx(x, w, bias)
            ctx.by_idx = True
            ctx.save_for_backward(x, w, idx, a, bias)
            return a
        if _own_conv_kind(x, w) == 'reg' and w.shape[0] == w.shape[1]:
            # pooled map from the epilogue (no pooling pass over z)
            z, a = conv64(x, w, False, bias=bias, pool=True)
        else:
            with _whole_batch_op():
                z = _conv3x3(x, w).contiguous(memory_format=_CL)
                b, c, h, wd = z.shape
                a = torch.empty((b, c, h // 2, wd // 2), dtype=z.dtype, device=z.device,
                                memory_format=_CL)
                _work('pool_fwd_kernel', 0.0, 1.25 * z.numel() * z.element_size())
                L.check(lib.scl_vgg_pool_fwd(L.ptr(z), _glue_dtype(z), L.ptr(bias), b, h, wd, c,
                                             L.ptr(a), L.stream_of(z)))
        ctx.save_for_backward(x, w, z, a, bias)
        return a

    @staticmethod
    def backward(ctx, ga):
        lib = L.load()
        x, w, z, a, bias = ctx.saved_tensors
        ga = ga.contiguous(memory_format=_CL)
        b, c = a.shape[0], a.shape[1]
        h, wd = x.shape[2], x.shape[3]
        gb = _grad_out(bias) if bias.dtype == torch.float32 else torch.empty(
            c, dtype=torch.float32, device=a.device)
        # ReLU' of this layer already applied by the layer above (its backward-data epilogue
        # masks with its own input, which is this layer's output a)?  Then a is not read.
        masked = ctx.by_idx and ctx.link_out is not None and ctx.link_out.take(ga)
        # The full-size gradient has one non-zero per pooling window and channel.  Where both of
        # its consumers can un-pool while they stage (conv1_2, conv2_2) it is never written: no
        # un-pooling pass, and the consumers read 0.75 instead of 2 bytes per element; the bias
        # gradient comes out of the weight-gradient kernel.  Otherwise (conv3_3, conv4_3) the
        # pass runs for the backward-data kernel and the weight gradient still reads the pooled
        # form.
        need_x = ctx.needs_input_grad[0]
        pool_w, pool_x = _pooled_consumers(x, ga, w, need_x, ctx.link_in) if masked else (False, False)
        if pool_w and pool_x:
            with _in_slot(ctx.slot):
                gx, gw = _conv3x3_backward(None, x, w, need_x, ctx.link_in, gb, pooled=(ga, z))
            return gx, _grad_ret(gw, w), _grad_ret(gb, bias), None, None
        gz = torch.empty((b, c, h, wd), dtype=a.dtype, device=a.device, memory_format=_CL)
        ws = L.workspace(lib.scl_vgg_workspace_bytes(c), a.device)
        fn = lib.scl_vgg_pool_bwd_idx if ctx.by_idx else lib.scl_vgg_pool_bwd
        # read g, a (1/4 each) and the index bytes (1/8) or z (1); write gz
        _work('pool_bwd_idx_kernel' if ctx.by_idx else 'pool_bwd_kernel', 0.0,
              ((1.375 if masked else 1.625) if ctx.by_idx else 2.5) * gz.numel() * gz.element_size())
        L.check(fn(L.ptr(ga), None if masked else L.ptr(a), L.ptr(z), _glue_dtype(a), b, h, wd, c,
                   L.ptr(gz), L.ptr(gb), L.ptr(ws), ws.numel(), L.stream_of(a)))
        with _in_slot(ctx.slot):
            gx, gw = _conv3x3_backward(gz, x, w, need_x, ctx.link_in,
                                       pooled=(ga, z) if pool_w else None)
        return gx, _grad_ret(gw, w), _grad_ret(gb, bias), None, None


def avg_rgb_grad(gz, w, gb):
    """Gradient of the loss w.r.t. ``average_rgb`` WITHOUT the conv1_1 input gradient.

    x0 = img - avg feeds a 3x3 same-padding conv; d loss / d avg[c] = -sum_{b,h,w} dx0[b,c,h,w]
    and the spatial sum of a transposed convolution only needs, per output channel o and tap
    (kh, kw), the sum of gz over the positions whose tap stays inside the image:
        S[o,kh,kw] = T[o] - R_kh[o] - C_kw[o] + X_khkw[o]
    (T total = the bias gradient, R / C the first or last row / column sums, X the corners).
    gz [B,64,H,W] is the gradient at the conv1_1 pre-activation, w [64,3,3,3], gb [64] f32.
    Exact, and replaces a full bwd-data pass over the 24x480x640x64 map plus an 88 MB
    reduction by four thin slices."""
    gzf = gz
    zero = torch.zeros_like(gb)
    r0 = gzf[:, :, 0, :].float().sum(dim=(0, 2))
    r2 = gzf[:, :, -1, :].float().sum(dim=(0, 2))
    c0 = gzf[:, :, :, 0].float().sum(dim=(0, 2))
    c2 = gzf[:, :, :, -1].float().sum(dim=(0, 2))
    rows = torch.stack([r0, zero, r2], dim=1)                      # [O,3] by kh
    cols = torch.stack([c0, zero, c2], dim=1)                      # [O,3] by kw
    corner = torch.zeros(gb.shape[0], 3, 3, dtype=torch.float32, device=gb.device)
    corner[:, 0, 0] = gzf[:, :, 0, 0].float().sum(0)
    corner[:, 0, 2] = gzf[:, :, 0, -1].float().sum(0)
    corner[:, 2, 0] = gzf[:, :, -1, 0].float().sum(0)
    corner[:, 2, 2] = gzf[:, :, -1, -1].float().sum(0)
    s = gb[:, None, None] - rows[:, :, None] - cols[:, None, :] + corner    # [O,3,3]
    return -torch.einsum('ockl,okl->c', w.float(), s)


def first_wrw(x0, gz, w_like, gb=None, w=None):
    """conv1_1's weight gradient (dtype and strides of ``w_like``: bf16 or float32) and bias gradient (float32 [64],
    written into ``gb`` when given) from x0 [B,3,H,W] (NHWC storage) and gz [B,64,H,W]
    (channels-last): ``scl_conv_first_wrw``.  With the layer's weight ``w`` the gradient of
    the trainable mean comes out of the same pass: returns (gw, davg [3])."""
    lib = L.load()
    L.require_device(x0, gz)
    x0 = x0.permute(0, 2, 3, 1).contiguous()
    gz = gz.contiguous(memory_format=_CL)
    b, h, wd, _ = x0.shape
    gw = _grad_out(w_like)
    if gb is None:
        gb = torch.empty(64, dtype=torch.float32, device=gz.device)
    _work('conv_first_wrw_kernel', 2.0 * b * h * wd * 64 * 28, b * h * wd * (128.0 + 6.0))
    ws = L.workspace(lib.scl_conv_first_wrw_workspace_bytes(), gz.device)
    sk, sc, sh, sw = gw.stride()
    davg = None
    if w is not None:
        if w.stride() != gw.stride() or w.dtype != gw.dtype:
            w = w.to(gw.dtype).contiguous(memory_format=_CL) if gw.is_contiguous(memory_format=_CL) \
                else w.to(gw.dtype).contiguous()
        if w.stride() != gw.stride():
            raise ValueError("w must have the strides of w_like")
        davg = torch.empty(3, dtype=torch.float32, device=gz.device)
    L.check(lib.scl_conv_first_wrw(L.ptr(x0), L.ptr(gz), b, h, wd, L.ptr(gw), sk, sc, sh, sw,
                                   int(gw.dtype == torch.float32), L.ptr(gb), L.ptr(w),
                                   L.ptr(davg), L.ptr(ws), ws.numel(), L.stream_of(gz)))
    return gw if davg is None else (gw, davg)


class _FirstConv(torch.autograd.Function):
    """(img - average_rgb) -> conv1_1 -> +bias -> ReLU (model/nets.py:22-24, 39) in one
    node: the only consumer of the image gradient is the trainable mean, whose gradient has
    the closed form of ``avg_rgb_grad`` — so conv1_1's bwd-data pass is never run."""

    @staticmethod
    def forward(ctx, img_nhwc, avg, w, bias, dtype, link_out=None, w2=None, bias2=None):
        lib = L.load()
        ctx.link_out = link_out
        if (USE_CONV64 and USE_FIRST and dtype == torch.bfloat16 and img_nhwc.is_cuda
                and img_nhwc.dtype == torch.float32 and w.dtype in _W_DTYPES
                and tuple(w.shape) == (64, 3, 3, 3)):
            # mean subtraction, cast, convolution, bias and ReLU in one kernel
            img = img_nhwc.contiguous()
            b, h, wd, _ = img.shape
            x0 = torch.empty((b, h, wd, 3), dtype=torch.bfloat16, device=img.device)
            y = torch.empty((b, 64, h, wd), dtype=torch.bfloat16, device=img.device,
                            memory_format=_CL)
            sk, sc, sh, sw = w.stride()
            avg_f, bias_f = avg.float().contiguous(), bias.float().contiguous()
            y_nhwc = y.permute(0, 2, 3, 1)                    # the storage order, batch-sliceable
            # ... and, when the caller hands over conv1_2's parameters, that layer's forward too
            # (pooled map + window index): its halo windows of y are computed, not read back
            fuse2 = (USE_FUSED12 and USE_POOL_IDX and w2 is not None and bias2 is not None
                     and link_out is not None and tuple(w2.shape) == (64, 64, 3, 3)
                     and w2.dtype in _W_DTYPES and h % 2 == 0 and wd % 2 == 0
                     and b * h * wd * 64 < 2 ** 31)
            if fuse2:
                a = torch.empty((b, 64, h // 2, wd // 2), dtype=torch.bfloat16, device=img.device,
                                memory_format=_CL)
                idx = torch.empty((b, 64, h // 2, wd // 2), dtype=torch.uint8, device=img.device,
                                  memory_format=_CL)
                a_nhwc, idx_nhwc = a.permute(0, 2, 3, 1), idx.permute(0, 2, 3, 1)
                bias2_f = bias2.float().contiguous()
                pk2 = _packed_for(w2, False)
                ws = L.workspace(lib.scl_conv3x3_workspace_bytes(), img.device)
                s2 = w2.stride()
            for stream, lo, hi in (_split_halves(b) or [(None, 0, b)]):
                if stream is not None:
                    _on_half(stream, img, x0, y, avg_f, bias_f, w)
                    if fuse2:
                        _on_half(stream, a, idx, bias2_f, w2)
                with (torch.cuda.stream(stream) if stream is not None else contextlib.nullcontext()):
                    n = hi - lo
                    if fuse2:
                        _work('conv12_kernel', 2.0 * n * h * wd * (27 * 64 + 64 * 64 * 9),
                              n * h * wd * (12.0 + 6.0 + 128.0 + 0.75 * 64))
                        L.check(lib.scl_conv_first_pool_idx(
                            L.ptr(img[lo:hi]), L.ptr(avg_f), L.ptr(w), sk, sc, sh, sw,
                            int(w.dtype == torch.float32), L.ptr(bias_f),
                            L.ptr(w2) if pk2 is None else L.ptr(pk2), s2[0], s2[1], s2[2], s2[3],
                            _wflag(w2) if pk2 is None else L.W_PACKED, L.ptr(bias2_f), n, h, wd,
                            L.ptr(x0[lo:hi]), L.ptr(y_nhwc[lo:hi]), L.ptr(a_nhwc[lo:hi]),
                            L.ptr(idx_nhwc[lo:hi]), L.ptr(ws), ws.numel(), L.stream_of(img)))
                        continue
                    _work('conv_first_kernel', 2.0 * n * h * wd * 27 * 64, n * h * wd * (12.0 + 6.0 + 128.0))
                    L.check(lib.scl_conv_first(L.ptr(img[lo:hi]), L.ptr(avg_f), L.ptr(w), sk, sc,
                                               sh, sw, int(w.dtype == torch.float32),
                                               L.ptr(bias_f), n, h, wd,
                                               L.ptr(x0[lo:hi]), L.ptr(y_nhwc[lo:hi]), L.stream_of(img)))
            if fuse2:
                link_out.fused = (y.data_ptr(), a, idx, w2, bias2)
            if link_out is not None and b * h * wd * 64 < 2 ** 31 and b <= 8192:
                link_out.first = (x0, w, bias)            # (x0 in its [B,H,W,3] storage order)
            x0 = x0.permute(0, 3, 1, 2)
        else:
            with _whole_batch_op():
                x0 = (img_nhwc - avg.to(img_nhwc.dtype)).to(dtype).permute(0, 3, 1, 2)
                y = _conv3x3(x0, w).contiguous(memory_format=_CL)
                b, c, h, wd = y.shape
                L.check(lib.scl_vgg_bias_act(L.ptr(y), _glue_dtype(y), L.ptr(bias), b * h * wd, c, 1,
                                             L.stream_of(y)))
        ctx.save_for_backward(x0, w, y, bias)
        return y

    @staticmethod
    def backward(ctx, gy):
        lib = L.load()
        x0, w, y, bias = ctx.saved_tensors
        gy = gy.contiguous(memory_format=_CL)
        b, c, h, wd = gy.shape
        gb = _grad_out(bias) if bias.dtype == torch.float32 else torch.empty(
            c, dtype=torch.float32, device=gy.device)
        ws = L.workspace(lib.scl_vgg_workspace_bytes(c), gy.device)
        done = ctx.link_out.first_done if ctx.link_out is not None else None
        masked = ctx.link_out is not None and ctx.link_out.take(gy)
        if done is not None:
            ctx.link_out.first_done = None
            if not masked:
                # the layer above skipped the gradient map (it lives in LDS tiles only) and what
                # arrived here is not the placeholder it handed down: nothing valid to fall back on
                raise RuntimeError("fused first-layer gradients: the placeholder gradient was replaced on "
                                   "its way down (a hook or a second consumer of conv1_1's output?); "
                                   "set SCL_FUSED_FIRST_WRW=0")
            gw, gb, davg = done
            return None, davg, _grad_ret(gw, w), _grad_ret(gb, bias), None, None, None, None
        own_wrw = (USE_CONV64 and USE_FIRST and gy.dtype == torch.bfloat16
                   and x0.dtype == torch.bfloat16 and w.dtype in _W_DTYPES
                   and tuple(w.shape) == (64, 3, 3, 3))
        gz = gy if masked else torch.empty_like(gy)
        if not (masked and own_wrw):
            _work('act_bwd_kernel', 0.0, (1.0 if masked else 3.0) * gy.numel() * gy.element_size())
            L.check(lib.scl_vgg_act_bwd(L.ptr(gy), None if masked else L.ptr(y), _glue_dtype(gy),
                                        b * h * wd, c, None if masked else L.ptr(gz), L.ptr(gb),
                                        L.ptr(ws), ws.numel(), L.stream_of(gy)))
        if own_wrw:
            # weight, bias and mean gradient in one pass over gz (the bias gradient and the
            # border sums of the mean's closed form are five more columns of the same product)
            gw, davg = first_wrw(x0, gz, w, gb, w)
        else:
            _, gw = _conv3x3_backward(gz, x0, w, False)
            davg = avg_rgb_grad(gz, w, gb)
        return None, davg, _grad_ret(gw, w), _grad_ret(gb, bias), None, None, None, None


class _SubMean(torch.autograd.Function):
    """x - average_rgb (model/nets.py:22-24) straight into the compute dtype; the gradient
    of the trainable mean is a [M,3] column sum, done as two well-shaped reductions instead
    of one reduction over a 3-wide inner dimension."""

    @staticmethod
    def forward(ctx, img_nhwc, avg, dtype):
        out = (img_nhwc - avg.to(img_nhwc.dtype)).to(dtype)
        return out.permute(0, 3, 1, 2)                 # NHWC storage == channels-last NCHW

    @staticmethod
    def backward(ctx, g):
        g = g.permute(0, 2, 3, 1).reshape(-1, 3)
        m = g.shape[0]
        if g.is_contiguous() and m % 256 == 0:
            s = g.view(m // 256, 768).sum(0, dtype=torch.float32).view(256, 3).sum(0)
        else:
            s = g.float().sum(0)
        return None, -s, None


def netvlad(x_nhwc, assign_w, centers, pre_l2=True, planes=None):
    """``tf.nn.l2_normalize(x, axis=-1)`` then ``layers.netVLAD(x, 64)``
    (model/nets.py:66-67).  x_nhwc [B,H',W',512] -> [B,32768].  ``planes``: what
    fresh_vlad_planes() returned for ``assign_w`` right after a prepack() of its current value
    (bf16 maps only; None: the call splits the weights itself)."""
    return _NetVLADFn.apply(x_nhwc, assign_w, centers, pre_l2, planes)


class VGG16NetVLAD(torch.nn.Module):
    """Variables of the reference's ``vgg16_netvlad_pca`` scope."""

    def __init__(self, compute_dtype=torch.float32, seed=1234, fused_relu=True, vlad_cores=64):
        super().__init__()
        if vlad_cores not in (0, 64):
            # the reference's callers know two heads: netVLAD(x, 64) and none (train/train.py:606-611)
            raise ValueError("vlad_cores must be 64 (NetVLAD head) or 0 (flattened conv5_3 map)")
        self.vlad_cores = vlad_cores
        self.compute_dtype = compute_dtype
        self.fused_relu = fused_relu
        g = torch.Generator().manual_seed(seed)
        self.average_rgb = torch.nn.Parameter(torch.tensor([123.68, 116.78, 103.94]))
        self.conv_names = []
        cin = 3
        for item in VGG_LAYERS:
            if item == 'pool':
                continue
            name, cout, _ = item
            w = torch.randn(cout, cin, 3, 3, generator=g) * math.sqrt(2.0 / (cin * 9))  # He
            self.register_parameter('conv%s_kernel' % name, torch.nn.Parameter(w))
            self.register_parameter('conv%s_bias' % name, torch.nn.Parameter(torch.zeros(cout)))
            self.conv_names.append(name)
            cin = cout
        self.assignment_kernel = torch.nn.Parameter(
            torch.randn(1, 1, L.VLAD_D, L.VLAD_K, generator=g) / math.sqrt(L.VLAD_D))
        self.cluster_centers = torch.nn.Parameter(
            torch.randn(1, 1, 1, L.VLAD_D, L.VLAD_K, generator=g) * 0.05)

    # ---- forward pieces -------------------------------------------------------
    def features(self, image_batch):
        """model/nets.py:10-63 -> conv5_3 map as a channels-last [B,H',W',512] tensor
        (not yet L2-normalised)."""
        if image_batch.dim() != 4:
            raise AssertionError("image batch must be rank 4 [B,H,W,C]")   # nets.py:10
        ch = image_batch.shape[3]
        if ch == 1:
            image_batch = image_batch.expand(-1, -1, -1, 3)              # nets.py:15-16
        elif ch != 3:
            raise AssertionError("last axis must be 1 or 3")               # nets.py:18
        dt = self.compute_dtype
        # On a HIP device the elementwise ops between the convolutions run as fused HIP
        # passes (csrc/vgg_glue.hip); the plain PyTorch composition below is the same math
        # and is what runs on CPU (tests, CPU baseline).
        fuse = self.fused_relu and image_batch.is_cuda
        if fuse:
            x = None                                       # mean subtraction lives in _FirstConv
        else:
            x = image_batch - self.average_rgb.to(image_batch.dtype)      # nets.py:22-24
            # NHWC storage viewed as NCHW == channels_last: no copy
            x = x.permute(0, 3, 1, 2)
            if x.dtype != dt:
                x = x.to(dt)
        if x is not None:
            x = x.contiguous(memory_format=torch.channels_last)
        if fuse and dt == torch.bfloat16 and USE_CONV64 and USE_F32_WEIGHTS and USE_PREPACK:
            # every packed weight image of this step (both directions) in one launch; forced: an
            # optimizer step in between may not have moved the tensors' versions
            # (+ the plane images of the NetVLAD assignment weights: the same launch)
            prepack([getattr(self, 'conv%s_kernel' % n) for n in self.conv_names], force=True,
                    vlad_w=self.assignment_kernel)
        split = None
        if (fuse and dt == torch.bfloat16 and USE_CONV64 and _split_fwd_wanted() and _FWD.split is None
                and image_batch.shape[0] >= 2):
            dev = image_batch.device
            key = (dev, threading.get_ident())              # a pair of streams per calling thread
            with _PACKED_LOCK:
                split = _FWD_STREAMS.get(key)
                if split is None:
                    # (streams of threads that have ended are dropped here: short-lived
                    # evaluation threads would otherwise leak a pair each)
                    alive = {t.ident for t in threading.enumerate()}
                    for k in [k for k in _FWD_STREAMS if k[1] not in alive]:
                        del _FWD_STREAMS[k]
                    split = _FWD_STREAMS[key] = (torch.cuda.Stream(device=dev), torch.cuda.Stream(device=dev))
            cur = torch.cuda.current_stream(dev)
            split[0].wait_stream(cur)
            split[1].wait_stream(cur)
            _FWD.split, _FWD.keep = split, []
        try:
            x = self._layers(image_batch, x, dt, fuse)
        finally:
            if split is not None:
                _FWD.split = None
                cur = torch.cuda.current_stream(image_batch.device)
                cur.wait_stream(split[0])
                cur.wait_stream(split[1])
                # the half-batch streams are done with everything once `cur` passes this point;
                # blocks freed from here on are reused by `cur` only after it
                _FWD.keep = None
        return x.permute(0, 2, 3, 1)                                      # [B,H',W',512] view

    def _layers(self, image_batch, x, dt, fuse):
        skip_pool = False
        link = None        # set while x is the post-ReLU output of the previous conv node
        for idx, item in enumerate(VGG_LAYERS):
            if item == 'pool':
                if not skip_pool:
                    x = F.relu(F.max_pool2d(x, 2, 2))                    # pool, then ReLU
                    link = None
                skip_pool = False                       # (fused conv + pool: its link stays)
                continue
            name, _, relu = item
            pool_next = idx + 1 < len(VGG_LAYERS) and VGG_LAYERS[idx + 1] == 'pool'
            w = getattr(self, 'conv%s_kernel' % name)
            bias = getattr(self, 'conv%s_bias' % name)
            if fuse and dt == torch.bfloat16 and USE_CONV64 and USE_F32_WEIGHTS:
                # the own kernels read the float32 master weight directly (rounded to bf16 as
                # it is packed) and return float32 gradients: no cast passes either way
                pass
            else:
                # OIHW master weights -> channels-last (and bf16) operands for MIOpen
                w = w.to(dtype=dt, memory_format=torch.channels_last)
            if fuse:
                if x is None:
                    # nets.py:22-24 + conv1_1 + ReLU; no image gradient is ever formed
                    link = _GradLink()
                    nxt = VGG_LAYERS[idx + 1] if idx + 2 < len(VGG_LAYERS) else None
                    if (nxt is not None and nxt != 'pool' and VGG_LAYERS[idx + 2] == 'pool'
                            and USE_F32_WEIGHTS):
                        # conv1_2 ends in the pooling: its forward rides in the first layer's kernel
                        x = _FirstConv.apply(image_batch, self.average_rgb, w, bias, dt, link,
                                             getattr(self, 'conv%s_kernel' % nxt[0]),
                                             getattr(self, 'conv%s_bias' % nxt[0]))
                    else:
                        x = _FirstConv.apply(image_batch, self.average_rgb, w, bias, dt, link)
                elif pool_next:
                    # conv -> bias -> pool -> ReLU in one elementwise pass (nets.py:40-42)
                    # (the pooled map is post-ReLU too: the next layer's backward-data kernel can
                    # apply this layer's ReLU' the same way)
                    pool_link = _GradLink()
                    x = _ConvBiasPoolReLU.apply(x, w, bias, link, pool_link)
                    skip_pool = True
                    link = pool_link
                else:
                    link_out = _GradLink() if relu else None
                    x = _ConvBiasAct.apply(x, w, bias, relu, link, link_out)
                    link = link_out
                continue
            if bias.dtype != dt:
                bias = bias.to(dt)
            x = F.conv2d(x, w, bias, stride=1, padding=1)
            if relu:
                x = F.relu(x)
        return x

    def forward(self, image_batch):
        _FWD.planes = None
        x = self.features(image_batch)
        # the plane images features() had written by its own (forced) packing launch, or None
        planes = fresh_vlad_planes(self.assignment_kernel)
        return netvlad(x, self.assignment_kernel, self.cluster_centers, True, planes)

    def forward_vgg16(self, image_batch):
        """model/nets.py:72-131: backbone + channel L2 norm, no VLAD."""
        x = self.features(image_batch).float()
        return x * torch.rsqrt(torch.clamp_min((x * x).sum(dim=-1, keepdim=True), 1e-12))

    # ---- checkpoint layout (TF variable names and shapes) ----------------------
    def state_dict_tf(self):
        sd = {SCOPE + '/average_rgb': self.average_rgb.detach()}
        for name in self.conv_names:
            w = getattr(self, 'conv%s_kernel' % name).detach()
            sd['%s/conv%s/kernel' % (SCOPE, name)] = w.permute(2, 3, 1, 0).contiguous()  # HWIO
            sd['%s/conv%s/bias' % (SCOPE, name)] = getattr(self, 'conv%s_bias' % name).detach()
        if self.vlad_cores == 64:      # the vgg16() graph creates no head variables (nets.py:72-131)
            sd[SCOPE + '/assignment/kernel'] = self.assignment_kernel.detach()
            sd[SCOPE + '/cluster_centers'] = self.cluster_centers.detach()
        return sd

    def load_state_dict_tf(self, sd, strict=True):
        """Restore by TF variable name like restore_weights (train/train.py:882-905):
        only names containing the scope are taken."""
        own = self.state_dict_tf()
        missing = [k for k in own if k not in sd]
        if strict and missing:
            raise KeyError("checkpoint lacks %s" % missing)
        with torch.no_grad():
            for key, val in sd.items():
                if SCOPE not in key or key not in own:
                    continue
                val = torch.as_tensor(val, dtype=torch.float32)
                if tuple(val.shape) != tuple(own[key].shape):
                    raise ValueError("%s: shape %s != %s" % (key, tuple(val.shape),
                                                             tuple(own[key].shape)))
                short = key[len(SCOPE) + 1:]
                if short == 'average_rgb':
                    self.average_rgb.copy_(val)
                elif short == 'assignment/kernel':
                    self.assignment_kernel.copy_(val)
                elif short == 'cluster_centers':
                    self.cluster_centers.copy_(val)
                else:
                    layer, kind = short.split('/')
                    name = layer[len('conv'):]
                    if kind == 'kernel':
                        getattr(self, 'conv%s_kernel' % name).copy_(val.permute(3, 2, 0, 1))
                    else:
                        getattr(self, 'conv%s_bias' % name).copy_(val)
        return missing


_DEFAULT = None


def default_model():
    """Process-wide variable store, the analogue of TF's reusable variable scope."""
    global _DEFAULT
    if _DEFAULT is None:
        _DEFAULT = VGG16NetVLAD()
        if torch.cuda.is_available():
            _DEFAULT = _DEFAULT.cuda()
    return _DEFAULT


def set_default_model(model):
    global _DEFAULT
    _DEFAULT = model
    return model


def vgg16Netvlad(image_batch, model=None):
    """model/nets.py:7-69.  image_batch [B,H,W,{1|3}] float, raw 0..255 RGB ->
    [B, 32768] unit-norm VLAD descriptors."""
    return (model or default_model())(image_batch)


def vgg16(image_batch, model=None):
    """model/nets.py:72-131 -> [B,H',W',512] channel-normalised conv5_3 map."""
    return (model or default_model()).forward_vgg16(image_batch)


def full_out(image_batch, model=None):
    """``ops['full_out']`` of both reference callers (train/train.py:606-611,
    evaluation/inference.py:89-92): ``vgg16Netvlad(input)`` when the model was built with
    ``vlad_cores == 64``, else ``tf.layers.flatten(vgg16(input))`` — the channel-normalised
    conv5_3 map flattened in NHWC order, [B, H' W' 512]."""
    model = model or default_model()
    if getattr(model, 'vlad_cores', 64) == 64:
        return model(image_batch)
    x = model.forward_vgg16(image_batch)
    return x.reshape(x.shape[0], -1)


def trainable_parameters(model):
    """The variables the chosen head creates (the reference's ``vgg16`` graph has no
    ``assignment/kernel`` / ``cluster_centers``): what the optimizer and the gradient buckets see."""
    skip = () if getattr(model, 'vlad_cores', 64) == 64 else ('assignment_kernel', 'cluster_centers')
    return [p for n, p in model.named_parameters() if n not in skip]
