"""ctypes binding of ``libscl_hip.so`` (the C-ABI in ``include/scl_hip.h``).

There is no CPU fallback: if the shared library is missing, or a tensor is not on a
HIP device, the call raises.  PyTorch is used only for device memory and streams.

Two builds of the same sources exist (csrc/Makefile): the PRODUCT library ``libscl_hip.so`` — what
``load()`` returns — has no diagnostic kernel variants compiled in and rejects
``scl_debug_set_variant(v != 0)``; ``libscl_hip_diag.so`` (``-DSCL_DIAG``) carries the A/B,
ablation and clock-stamp variants.  ``with variant(v):`` runs the enclosed calls on the
diagnostic build under variant ``v`` (equality tests, scripts/); ``SCL_DIAG=1`` in the
environment or ``use_diag()`` makes it the process's library (scripts/).
"""
import contextlib
import ctypes
import os

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libscl_hip.so")
DIAG_LIB_PATH = os.path.join(_HERE, "libscl_hip_diag.so")

# constants of include/scl_hip.h
ABI_VERSION = 12
DT_F32, DT_BF16 = 0, 1
MASK_WMS_EXP, MASK_WMS_LIN, MASK_WMS_TANH, MASK_LABELS = 0, 1, 2, 3
SUM_MS, SUM_PLAIN = 0, 1
TUPLE_TRIPLET, TUPLE_LAZY_TRIPLET, TUPLE_EVIL_TRIPLET = 0, 1, 2
TUPLE_QUADRUPLET, TUPLE_LAZY_QUADRUPLET, TUPLE_EVIL_QUADRUPLET = 3, 4, 5
VLAD_D, VLAD_K = 512, 64
VLAD_SAVE_ROWS = 514      # save_vlad rows per image: U, asum, sync words of the backward pass
TOPN_SCORE_F32, TOPN_SCORE_BF16X3 = 0, 1
CONV_TRANSPOSED, W_F32, W_PACKED = 1, 2, 4   # flag word of the convolution entry points
PACK_VLAD_W = 16           # SclPackJob.flags: the job writes the NetVLAD plane images of assign_w

_p = ctypes.c_void_p
_i = ctypes.c_int
_l = ctypes.c_int64
_f = ctypes.c_float
_z = ctypes.c_size_t



class PackJob(ctypes.Structure):
    """SclPackJob of include/scl_hip.h."""
    _fields_ = [("w", _p), ("w_stride_k", _l), ("w_stride_c", _l), ("w_stride_h", _l),
                ("w_stride_w", _l), ("flags", _i), ("cin", _i), ("kout", _i), ("packed", _p)]


# name -> (restype, argtypes); every symbol include/scl_hip.h declares
SIGNATURES = {
    "scl_conv_packed_bytes": (_z, [_i, _i]),
    "scl_conv_pack_batch": (_i, [ctypes.POINTER(PackJob), _i, _p]),
    "scl_abi_version": (_i, []),
    "scl_error_string": (ctypes.c_char_p, [_i]),
    "scl_netvlad_fwd_workspace_bytes": (_z, [_i, _i]),
    "scl_netvlad_fwd": (_i, [_p, _i, _p, _p, _i, _i, _i, _p, _p, _p, _p, _p, _p, _z, _p]),
    "scl_netvlad_planes_bytes": (_z, []),
    "scl_netvlad_planes": (_i, [_p, _p, _p]),
    "scl_netvlad_fwd_p": (_i, [_p, _i, _p, _p, _p, _i, _i, _i, _p, _p, _p, _p, _p, _p, _z, _p]),
    "scl_netvlad_bwd_p": (_i, [_p, _i, _p, _p, _p, _p, _p, _p, _p, _p, _i, _i, _i, _p, _p, _p, _p, _z,
                               _p]),
    "scl_netvlad_bwd_workspace_bytes": (_z, [_i, _i]),
    "scl_netvlad_bwd": (_i, [_p, _i, _p, _p, _p, _p, _p, _p, _p, _i, _i, _i, _p, _p, _p, _p, _z,
                             _p]),
    "scl_gram_loss_workspace_bytes": (_z, [_i, _i]),
    "scl_gram_loss_fwd": (_i, [_p, _l, _i, _i, _i, _p, _i, _f, _f, _p, _f, _f, _f, _f, _i, _i, _p,
                               _p, _p, _z, _p]),
    "scl_gram_loss_fwd_s": (_i, [_p, _l, _i, _i, _i, _p, _i, _f, _f, _p, _f, _f, _f, _f, _i, _i, _p,
                                 _p, _p, _z, _p, _p]),
    "scl_gram_loss_bwd": (_i, [_p, _l, _i, _i, _p, _p, _i, _i, _p, _l, _p]),
    "scl_gram_loss_bwd_workspace_bytes": (_z, [_i, _i]),
    "scl_gram_loss_bwd_w": (_i, [_p, _l, _i, _i, _p, _p, _i, _i, _p, _l, _p, _z, _p]),
    "scl_pairwise_sqdist_workspace_bytes": (_z, [_i, _i, _i]),
    "scl_pairwise_sqdist": (_i, [_p, _i, _i, _i, _p, _p, _z, _p]),
    "scl_tuple_loss_fwd": (_i, [_i, _p, _l, _p, _l, _p, _l, _p, _l, _i, _i, _i, _i, _f, _f, _p, _p,
                                _p, _p]),
    "scl_tuple_loss_bwd": (_i, [_p, _l, _p, _l, _p, _l, _p, _l, _i, _i, _i, _i, _p, _p, _p, _p, _p,
                                _p, _p]),
    "scl_distance_tuple_loss_fwd": (_i, [_i, _i, _i, _p, _l, _p, _l, _p, _l, _p, _l, _i, _i, _i, _i,
                                         _f, _f, _f, _p, _f, _f, _p, _p, _p, _p]),
    "scl_logratio_fwd": (_i, [_p, _p, _p, _i, _i, _i, _p, _p, _p, _p, _p, _p]),
    "scl_topn_l2_workspace_bytes": (_z, [_i, _i, _i, _i]),
    "scl_topn_l2": (_i, [_p, _i, _p, _i, _i, _i, _l, _p, _p, _p, _z, _p]),
    "scl_topn_l2_ex_workspace_bytes": (_z, [_i, _i, _i, _i, _i]),
    "scl_topn_l2_ex": (_i, [_p, _i, _p, _i, _i, _i, _l, _p, _p, _p, _z, _i, _p]),
    "scl_topn_l2_cert": (_i, [_p, _i, _p, _i, _i, _i, _l, _p, _p, _p, _p, _p, _z, _i, _p]),
    "scl_topn_exact_filter": (_i, [_p, _i, _p, _i, _p, _i, _p, _i, _p, _p, _p, _p]),
    "scl_topn_dots": (_i, [_p, _i, _p, _i, _i, _i, _p, _p]),
    "scl_vgg_workspace_bytes": (_z, [_i]),
    "scl_vgg_bias_act": (_i, [_p, _i, _p, _l, _i, _i, _p]),
    "scl_vgg_act_bwd": (_i, [_p, _p, _i, _l, _i, _p, _p, _p, _z, _p]),
    "scl_vgg_pool_fwd": (_i, [_p, _i, _p, _i, _i, _i, _i, _p, _p]),
    "scl_vgg_pool_bwd": (_i, [_p, _p, _p, _i, _i, _i, _i, _i, _p, _p, _p, _z, _p]),
    "scl_conv64_workspace_bytes": (_z, []),
    "scl_conv64": (_i, [_p, _p, _l, _l, _l, _l, _i, _i, _i, _i, _p, _p, _z, _p]),
    "scl_conv3x3_workspace_bytes": (_z, []),
    "scl_conv3x3": (_i, [_p, _p, _l, _l, _l, _l, _i, _i, _i, _i, _i, _i, _p, _p, _z, _p]),
    "scl_conv3x3_fused": (_i, [_p, _p, _l, _l, _l, _l, _i, _i, _i, _i, _i, _i, _p, _p, _i, _p, _p,
                               _z, _p]),
    "scl_convg_workspace_bytes": (_z, [_i, _i]),
    "scl_conv3x3_pool_idx": (_i, [_p, _p, _l, _l, _l, _l, _i, _i, _i, _i, _i, _i, _p, _p, _p, _p, _z,
                                  _p]),
    "scl_convg_pool_idx": (_i, [_p, _p, _l, _l, _l, _l, _i, _i, _i, _i, _i, _i, _p, _p, _p, _p, _z,
                                _p]),
    "scl_vgg_pool_bwd_idx": (_i, [_p, _p, _p, _i, _i, _i, _i, _i, _p, _p, _p, _z, _p]),
    "scl_conv3x3_masked": (_i, [_p, _p, _l, _l, _l, _l, _i, _i, _i, _i, _i, _i, _p, _p, _p, _z, _p]),
    "scl_conv3x3_masked_pooled": (_i, [_p, _p, _p, _l, _l, _l, _l, _i, _i, _i, _i, _i, _i, _p, _p, _p, _z, _p]),
    "scl_convg_masked": (_i, [_p, _p, _l, _l, _l, _l, _i, _i, _i, _i, _i, _i, _p, _p, _p, _z, _p]),
    "scl_convg": (_i, [_p, _p, _l, _l, _l, _l, _i, _i, _i, _i, _i, _i, _p, _p, _i, _p, _z, _p]),
    "scl_wrw64_workspace_bytes": (_z, []),
    "scl_wrw64": (_i, [_p, _p, _i, _i, _i, _p, _l, _l, _l, _l, _p, _z, _p]),
    "scl_wrw3x3_workspace_bytes": (_z, [_i, _i]),
    "scl_wrw3x3": (_i, [_p, _p, _i, _i, _i, _i, _i, _p, _l, _l, _l, _l, _p, _z, _p]),
    "scl_wrw3x3_ex": (_i, [_p, _p, _i, _i, _i, _i, _i, _p, _l, _l, _l, _l, _i, _p, _z, _p]),
    "scl_wrw3x3_bias": (_i, [_p, _p, _i, _i, _i, _i, _i, _p, _l, _l, _l, _l, _i, _p, _p, _z, _p]),
    "scl_wrw3x3_pooled": (_i, [_p, _p, _p, _i, _i, _i, _i, _i, _p, _l, _l, _l, _l, _i, _p, _p, _z, _p]),
    "scl_conv_first_wrw_workspace_bytes": (_z, []),
    "scl_conv_first_wrw": (_i, [_p, _p, _i, _i, _i, _p, _l, _l, _l, _l, _i, _p, _p, _p, _p, _z, _p]),
    "scl_conv_first": (_i, [_p, _p, _p, _l, _l, _l, _l, _i, _p, _i, _i, _i, _p, _p, _p]),
    "scl_conv_first_pool_idx": (_i, [_p, _p, _p, _l, _l, _l, _l, _i, _p, _p, _l, _l, _l, _l, _i, _p,
                                     _i, _i, _i, _p, _p, _p, _p, _p, _z, _p]),
    "scl_conv3x3_masked_pooled_first_wrw": (_i, [_p, _p, _p, _l, _l, _l, _l, _i, _i, _i, _i, _p, _p, _p, _l,
                                                 _l, _l, _l, _i, _p, _p, _p, _p, _z, _p, _z, _p]),
    "scl_debug_set_variant": (_i, [_i]),
    "scl_build_is_diag": (_i, []),
    "scl_calibrate_mfma_bf16": (_i, [_i, _i, _i, _p, _p, _p]),
    "scl_calibrate_mfma_bf16_flops": (ctypes.c_double, [_i, _i]),
    "scl_set_reserve_cus": (_i, [_i]),
    "scl_get_reserve_cus": (_i, []),
    "scl_prof_begin": (_i, [_i]),
    "scl_prof_count": (_i, []),
    "scl_prof_end": (_i, [_p, _p, _i]),
    "scl_prof_null": (_i, [_p]),
    "scl_crc32c": (ctypes.c_uint, [ctypes.c_uint, _p, _z]),
}

_libs = {}
_diag = os.environ.get('SCL_DIAG', '0') == '1'


class SclError(RuntimeError):
    pass


def load(diag=None):
    """The process's library (loaded once): the product build, unless the diagnostic build was
    asked for (``diag=True``, ``use_diag()``, ``SCL_DIAG=1``, or inside ``with variant(v)``).
    Raises loudly if it has not been built."""
    want = _diag if diag is None else bool(diag)
    lib = _libs.get(want)
    if lib is not None:
        return lib
    path = DIAG_LIB_PATH if want else LIB_PATH
    if not os.path.exists(path):
        raise SclError(
            "%s not found at %s: build it with "
            "`python -c 'import __graft_entry__ as g; g.build()'` or "
            "`make -C soft_contrastive_learning_amd/csrc` (there is no CPU fallback)"
            % (os.path.basename(path), path))
    lib = ctypes.CDLL(path)
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)
        fn.restype = res
        fn.argtypes = args
    if lib.scl_abi_version() != ABI_VERSION:
        raise SclError("%s ABI %d != expected %d" % (os.path.basename(path), lib.scl_abi_version(),
                                                     ABI_VERSION))
    if bool(lib.scl_build_is_diag()) != want:
        raise SclError("%s is not the %s build" % (path, 'diagnostic' if want else 'product'))
    _libs[want] = lib
    return lib


def use_diag(on=True):
    """Make the diagnostic build (libscl_hip_diag.so) the process's library from here on; returns
    the previous setting.  Diagnostics only: scripts/, bench.py --variant."""
    global _diag
    old, _diag = _diag, bool(on)
    return old


@contextlib.contextmanager
def variant(v):
    """Run the enclosed library calls on the DIAGNOSTIC build under scl_debug_set_variant(v)
    (process-wide inside the block, like the switch itself; tests and scripts only)."""
    old_diag = use_diag(True)
    lib = load()
    old = lib.scl_debug_set_variant(int(v))
    try:
        yield lib
    finally:
        lib.scl_debug_set_variant(old)
        use_diag(old_diag)


@contextlib.contextmanager
def maybe_variant(v):
    """variant(v) for v != 0; for 0 the process's own library untouched (the product build unless
    the process asked for the diagnostic one) — so that an equality test compares the PRODUCT
    kernels with a diagnostic variant."""
    if v:
        with variant(v) as lib:
            yield lib
    else:
        yield load()


def check(code):
    """Map a C-ABI return code onto the reference's error behaviour: bad shapes and
    selectors are ValueError (TF raised at graph-build time), HIP errors RuntimeError."""
    if code == 0:
        return
    msg = load().scl_error_string(code).decode()
    if code < 0:
        raise ValueError("scl: %s (code %d)" % (msg, code))
    raise SclError("scl: HIP error %d: %s" % (code, msg))


def require_device(*tensors):
    for t in tensors:
        if t is None:
            continue
        if not isinstance(t, torch.Tensor) or not t.is_cuda:
            raise SclError("soft_contrastive_learning_amd ops need tensors on a HIP device "
                           "(no CPU fallback exists); got %r" % (getattr(t, 'device', type(t)),))


def ptr(t):
    return None if t is None else ctypes.c_void_p(t.data_ptr())


def stream_of(t):
    return ctypes.c_void_p(torch.cuda.current_stream(t.device).cuda_stream)


_SYNC = {}
_SYNC_LOCK = __import__('threading').Lock()


def sync_words(device):
    """The zero-initialised sync block of the calling thread's current stream on ``device`` (64
    bytes; scl_gram_loss_fwd_s: zero on entry, zero on return).  One per (device, stream, thread):
    two calls that could be in flight at once never share one."""
    import threading
    key = (torch.device(device).index, torch.cuda.current_stream(device).cuda_stream,
           threading.get_ident())
    with _SYNC_LOCK:
        t = _SYNC.get(key)
        if t is None:
            alive = {th.ident for th in threading.enumerate()}
            for k in [k for k in _SYNC if k[2] not in alive]:
                del _SYNC[k]
            t = _SYNC[key] = torch.zeros(64, dtype=torch.uint8, device=device)
    return t


def workspace(nbytes, device):
    """Caller-owned scratch from torch's caching allocator (512-byte aligned)."""
    return torch.empty(max(int(nbytes), 256), dtype=torch.uint8, device=device)


class KernelTimer:
    """Context manager over scl_prof_begin / scl_prof_end: per-launch kernel durations
    (HIP events on the launch stream) of every library call made by this thread."""

    def __init__(self, capacity=4096):
        self.capacity = capacity
        self.records = []          # (kernel name, milliseconds)

    def __enter__(self):
        check(load().scl_prof_begin(self.capacity))
        return self

    def __exit__(self, *exc):
        ms = (ctypes.c_float * self.capacity)()
        names = (ctypes.c_char_p * self.capacity)()
        n = load().scl_prof_end(ms, names, self.capacity)
        self.records = [(names[i].decode(), float(ms[i])) for i in range(n)]
        return False

    def summary(self):
        """name -> (launches, mean ms)"""
        acc = {}
        for name, t in self.records:
            c, s = acc.get(name, (0, 0.0))
            acc[name] = (c + 1, s + t)
        return {k: (c, s / c) for k, (c, s) in acc.items()}
