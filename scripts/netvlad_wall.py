#!/usr/bin/env python
"""Wall time of the NetVLAD head forward + backward (bf16 map, 24 x 1200, plane images from the
packing launch) as the stream sees it — kernel durations AND the gaps between dependent launches —
for round 4's launch structure (5 launches) against round 3's (scl_debug_set_variant(920): 10
launches, four-wave kernels).  Device events around 200 back-to-back iterations, no per-kernel
instrumentation; three rounds, alternating.  The C-ABI is called directly (through autograd the Python
overhead of ~100 us per iteration starves the device and the figure measures the host)."""
import os

os.environ.setdefault('SCL_DIAG', '1')   # the diagnostic build carries the variants (csrc/Makefile)
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import torch  # noqa: E402

from soft_contrastive_learning_amd import _lib as L  # noqa: E402
from soft_contrastive_learning_amd.model import nets  # noqa: E402
from tests import util_data as U  # noqa: E402


def main():
    dev = torch.device('cuda:0')
    b, n, iters = 24, 1200, 200
    x = torch.tensor(U.feature_map(b, n, seed=5), device=dev).bfloat16()
    w, c = U.vlad_params()
    wt, ct = torch.tensor(w, device=dev), torch.tensor(c, device=dev)
    go = torch.randn(b, 32768, device=dev)
    lib = L.load()
    out = torch.empty(b, 32768, device=dev)
    sa, sl = torch.empty(b, n, 64, device=dev), torch.empty(b, n, 64, device=dev)
    sr, sv = torch.empty(b, n, device=dev), torch.empty(b, L.VLAD_SAVE_ROWS, 64, device=dev)
    gx, gw, gc = torch.empty_like(x), torch.empty_like(wt), torch.empty_like(ct)
    wsf = L.workspace(lib.scl_netvlad_fwd_workspace_bytes(b, n), dev)
    wsb = L.workspace(lib.scl_netvlad_bwd_workspace_bytes(b, n), dev)
    planes = torch.empty(lib.scl_netvlad_planes_bytes(), dtype=torch.uint8, device=dev)
    st = L.stream_of(x)
    L.check(lib.scl_netvlad_planes(L.ptr(wt), L.ptr(planes), st))

    def fwd():
        L.check(lib.scl_netvlad_fwd_p(L.ptr(x), L.DT_BF16, L.ptr(wt), L.ptr(ct), L.ptr(planes), b, n, 1, L.ptr(out),
                                      L.ptr(sa), L.ptr(sl), L.ptr(sr), L.ptr(sv), L.ptr(wsf), wsf.numel(), st))

    def bwd():
        L.check(lib.scl_netvlad_bwd_p(L.ptr(x), L.DT_BF16, L.ptr(wt), L.ptr(ct), L.ptr(planes), L.ptr(go), L.ptr(sa),
                                      L.ptr(sl), L.ptr(sr), L.ptr(sv), b, n, 1, L.ptr(gx), L.ptr(gw), L.ptr(gc),
                                      L.ptr(wsb), wsb.numel(), st))

    def run(variant, fwd_only):
        lib.scl_debug_set_variant(variant)
        try:
            for _ in range(10):
                fwd()
                if not fwd_only:
                    bwd()
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(iters):
                fwd()
                if not fwd_only:
                    bwd()
            e1.record()
            torch.cuda.synchronize()
            return e0.elapsed_time(e1) / iters * 1e3
        finally:
            lib.scl_debug_set_variant(0)
    for rnd in range(3):
        for name, var in (('round 4 (2 + 3 launches)', 0), ('round 3 (4 + 6 launches, variant 920)', 920)):
            f = run(var, True)
            fb = run(var, False)
            print('round %d  %-40s forward %6.1f us   forward + backward %6.1f us' % (rnd, name, f, fb))


if __name__ == '__main__':
    main()
