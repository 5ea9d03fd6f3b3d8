"""scl_calibrate_mfma_bf16 (csrc/calibrate.hip): the bare bf16 MFMA loop bench.py times beside the
convolution kernels (`roofline.sustained`).  No counterpart in the reference; the checks are that
the entry validates its arguments, that the loop really computes (checksums finite, non-zero,
identical over workgroups and runs, both MFMA shapes agreeing with each other: they accumulate the
same [128 x 64] products in a different order) and that the FLOP count it reports is the loop's."""
import ctypes

import pytest
import torch

from soft_contrastive_learning_amd import _lib

pytestmark = pytest.mark.gpu


def _run(lib, shape, wgs, iters, ops, sink):
    return lib.scl_calibrate_mfma_bf16(shape, wgs, iters, ctypes.c_void_p(ops.data_ptr()),
                                       ctypes.c_void_p(sink.data_ptr()),
                                       ctypes.c_void_p(torch.cuda.current_stream().cuda_stream))


def test_calibration_loop_computes_and_validates():
    lib = _lib.load()
    dev = torch.device('cuda:0')
    g = torch.Generator(device=dev).manual_seed(5)
    ops = (torch.rand(32768, device=dev, generator=g) * 2 - 1).to(torch.bfloat16)
    sums = {}
    for shape in (32, 16):
        sink = torch.zeros(64, device=dev)
        assert _run(lib, shape, 64, 16, ops, sink) == 0
        torch.cuda.synchronize()
        a = sink.clone()
        assert _run(lib, shape, 64, 16, ops, sink) == 0
        torch.cuda.synchronize()
        assert torch.isfinite(a).all() and float(a.abs().min()) > 0
        assert torch.equal(a, sink)                                  # deterministic
        assert torch.equal(a, a[:1].expand_as(a))                    # same operands, same checksum
        sums[shape] = float(a[0])
    # the two shapes lay the same 64 KB out as different operand fragments: no equality between
    # them, but the magnitude of a sum of 8 x 128 x 64 x 512-term products of U(-1, 1) values agrees
    assert 0.01 < abs(sums[32]) / abs(sums[16]) < 100
    assert lib.scl_calibrate_mfma_bf16_flops(256, 2000) == 256 * 8 * 2000 * 2.0 * 128 * 64 * 32
    sink = torch.zeros(8, device=dev)
    assert _run(lib, 8, 8, 16, ops, sink) == -2                     # SCL_E_KIND: no such MFMA shape
    assert _run(lib, 32, 0, 16, ops, sink) == -1                    # SCL_E_SHAPE
    assert _run(lib, 32, 8, 0, ops, sink) == -2
    assert lib.scl_calibrate_mfma_bf16(32, 8, 16, None, ctypes.c_void_p(sink.data_ptr()), None) == -3
