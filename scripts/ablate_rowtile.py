#!/usr/bin/env python
"""Where does the forward row-tile kernel (x·W + softmax, csrc/netvlad.hip) spend its time?

Runs the production kernel and its seven diagnostic variants interleaved in ONE process
(cdna_hip_programming.md §5.4 rule 24) at the bench shape and prints the median per-launch
duration of each.  Variant bits: 1 = no epilogue stores, 2 = no x loads, 4 = no operand
staging / barriers.  The variants compute garbage; only their timing is meaningful.

    python scripts/ablate_rowtile.py [--batch 24] [--rounds 15]
"""
import argparse
import os

os.environ.setdefault('SCL_DIAG', '1')   # the diagnostic build carries the variants (csrc/Makefile)
import statistics
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import torch  # noqa: E402

from soft_contrastive_learning_amd import _lib  # noqa: E402
from soft_contrastive_learning_amd.model import nets  # noqa: E402
from tests import util_data as U  # noqa: E402

NAMES = {0: 'production', 1: 'no stores', 2: 'no x loads', 3: 'no stores, no x loads',
         4: 'no staging/barriers', 5: 'no stores, no staging', 6: 'no x loads, no staging',
         7: 'MFMA + LDS reads only'}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--batch', type=int, default=24)
    ap.add_argument('--rounds', type=int, default=15)
    args = ap.parse_args()
    lib = _lib.load()
    dev = torch.device('cuda:0')
    b, n = args.batch, 1200
    x = torch.tensor(U.feature_map(b, n, seed=5), device=dev).bfloat16().reshape(b, 1, n, 512)
    w, c = U.vlad_params()
    wt, ct = torch.tensor(w, device=dev), torch.tensor(c, device=dev)
    times = {v: [] for v in NAMES}
    try:
        for rnd in range(args.rounds + 2):
            for v in NAMES:
                lib.scl_debug_set_variant(v)
                with _lib.KernelTimer(capacity=16) as kt:
                    nets.netvlad(x, wt, ct, True)
                    torch.cuda.synchronize()
                if rnd >= 2:
                    times[v].append(dict(kt.records)['rowtile16_kernel<ASSIGN>'] * 1e3)
    finally:
        lib.scl_debug_set_variant(0)
    flops = 2.0 * b * n * 512 * 64
    for v, name in NAMES.items():
        med = statistics.median(times[v])
        print('variant %d  %-26s median %6.1f us  min %6.1f us  %6.1f TF'
              % (v, name, med, min(times[v]), flops / (med * 1e-6) / 1e12))


if __name__ == '__main__':
    main()
