"""The C-ABI library builds, loads without a GPU, and exports every symbol that
include/scl_hip.h declares (no compute calls here)."""
import ctypes
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HEADER = os.path.join(ROOT, "include", "scl_hip.h")


@pytest.fixture(scope="module")
def lib():
    import __graft_entry__ as g
    g.build()
    from soft_contrastive_learning_amd import _lib
    return _lib.load()


def _declared():
    src = open(HEADER).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(scl_[a-z0-9_]+)\s*\(", src)))


def test_header_declares_the_expected_surface():
    names = _declared()
    for want in ("scl_netvlad_fwd", "scl_netvlad_bwd", "scl_gram_loss_fwd", "scl_gram_loss_bwd",
                 "scl_tuple_loss_fwd", "scl_tuple_loss_bwd", "scl_logratio_fwd",
                 "scl_pairwise_sqdist", "scl_topn_l2", "scl_abi_version", "scl_error_string"):
        assert want in names


def test_every_declared_symbol_is_exported_and_bound(lib):
    from soft_contrastive_learning_amd import _lib
    declared = _declared()
    assert sorted(_lib.SIGNATURES) == declared
    for name in declared:
        assert getattr(lib, name) is not None


def test_product_library_has_no_variants_and_the_diagnostic_build_does(lib):
    """csrc/Makefile builds the sources twice: libscl_hip.so (what load() returns) is compiled
    without -DSCL_DIAG — scl_variant() is the constant 0 there, no variant branch survives — and
    refuses every selector but 0; libscl_hip_diag.so exports the same surface with the switch."""
    from soft_contrastive_learning_amd import _lib
    assert lib.scl_build_is_diag() == 0
    assert lib.scl_debug_set_variant(0) == 0
    assert lib.scl_debug_set_variant(921) == -2 and lib.scl_debug_set_variant(0) == 0
    diag = _lib.load(diag=True)
    assert diag is not lib and diag.scl_build_is_diag() == 1
    for name in _declared():
        assert getattr(diag, name) is not None
    assert diag.scl_debug_set_variant(921) == 0 and diag.scl_debug_set_variant(0) == 921
    with _lib.variant(32) as inside:
        assert inside is diag and _lib.load() is diag
    assert _lib.load() is lib and diag.scl_debug_set_variant(0) == 0
    # the product object files carry no reference to the switch at all
    import subprocess
    so = os.path.join(ROOT, 'soft_contrastive_learning_amd', 'libscl_hip.so')
    syms = subprocess.run(['nm', '-D', so], capture_output=True, text=True).stdout
    assert 'scl_debug_variant' not in syms.replace('scl_debug_set_variant', '')
    syms = subprocess.run(['nm', '-D', so.replace('.so', '_diag.so')], capture_output=True, text=True).stdout
    assert 'scl_debug_variant' in syms.replace('scl_debug_set_variant', '')


def test_abi_version_and_error_strings(lib):
    assert lib.scl_abi_version() == 12
    assert lib.scl_error_string(0) == b"ok"
    assert b"shape" in lib.scl_error_string(-1)
    assert b"NULL" in lib.scl_error_string(-3)


def test_workspace_queries_are_pure_host_functions(lib):
    # NetVLAD forward at the bench shape: Wt + slabs + a + rn, all 256-byte rounded
    n = lib.scl_netvlad_fwd_workspace_bytes(24, 1200)
    assert n >= 24 * 2 * 512 * 64 * 4 + 24 * 1200 * 64 * 4
    assert n % 256 == 0
    assert lib.scl_netvlad_fwd_workspace_bytes(0, 10) == 0
    assert lib.scl_netvlad_bwd_workspace_bytes(24, 1200) % 256 == 0
    assert lib.scl_gram_loss_workspace_bytes(24, 32768) > 0
    assert lib.scl_gram_loss_workspace_bytes(5000, 32768) == 0          # B above the cap
    assert lib.scl_topn_l2_workspace_bytes(100000, 10000, 256, 25) > 100000 * 4
    assert lib.scl_topn_l2_workspace_bytes(100, 10, 100, 5) == 0         # unsupported d
    assert lib.scl_topn_l2_workspace_bytes(100, 10, 64, 26) == 0         # n above the cap


def test_argument_validation_happens_before_any_launch(lib):
    # NULL pointers and bad selectors are rejected on the host: safe without a GPU
    assert lib.scl_gram_loss_fwd(None, 0, 4, 8, 0, None, 0, 0.8, 15.0, None, 2.0, 50.0, 1.0, 0.1,
                                 1, 0, None, None, None, 0, None) == -3
    assert lib.scl_tuple_loss_fwd(99, None, 0, None, 0, None, 0, None, 0, 1, 1, 1, 8, 0.1, 0.2,
                                  None, None, None, None) == -2
    assert lib.scl_netvlad_fwd(None, 0, None, None, 1, 1, 1, None, None, None, None, None, None, 0,
                               None) == -3
    assert lib.scl_topn_l2(None, 1, None, 1, 64, 1, 0, None, None, None, 0, None) == -3


def test_pooled_backward_entry_points_validate_on_the_host(lib):
    """scl_wrw3x3_pooled / scl_conv3x3_masked_pooled (round 3): NULL operands, odd map sizes and
    shapes the un-pooling staging does not exist for are refused before any launch."""
    import ctypes
    buf = ctypes.create_string_buffer(4096)
    p = ctypes.cast(buf, ctypes.c_void_p)
    ws_bytes = lib.scl_wrw3x3_workspace_bytes(64, 64)
    assert ws_bytes > 0 and lib.scl_wrw3x3_workspace_bytes(48, 64) == 0
    # NULL index / NULL gradient
    assert lib.scl_wrw3x3_pooled(p, p, None, 1, 8, 8, 64, 64, p, 576, 9, 3, 1, 1, None, p, ws_bytes,
                                 None) == -3
    assert lib.scl_wrw3x3_pooled(p, None, p, 1, 8, 8, 64, 64, p, 576, 9, 3, 1, 1, None, p, ws_bytes,
                                 None) == -3
    # channel counts without a kernel (-1 = shape), odd height (a 2x2 window would straddle the edge)
    assert lib.scl_wrw3x3_pooled(p, p, p, 1, 8, 8, 48, 64, p, 432, 9, 3, 1, 1, None, p, 1 << 20,
                                 None) == -1
    assert lib.scl_wrw3x3_pooled(p, p, p, 1, 7, 8, 64, 64, p, 576, 9, 3, 1, 1, None, p, ws_bytes,
                                 None) == -1
    cw = lib.scl_conv3x3_workspace_bytes()
    assert lib.scl_conv3x3_masked_pooled(p, None, p, 576, 9, 3, 1, 1, 1, 8, 8, 64, 64, p, p, p, cw,
                                         None) == -3
    assert lib.scl_conv3x3_masked_pooled(p, p, p, 576, 9, 3, 1, 1, 1, 8, 8, 64, 64, p, None, p, cw,
                                         None) == -3
    # cin != kout has no un-pooling window staging; odd width
    assert lib.scl_conv3x3_masked_pooled(p, p, p, 576, 9, 3, 1, 1, 1, 8, 8, 128, 64, p, p, p, cw,
                                         None) == -1
    assert lib.scl_conv3x3_masked_pooled(p, p, p, 576, 9, 3, 1, 1, 1, 8, 9, 64, 64, p, p, p, cw,
                                         None) == -1
