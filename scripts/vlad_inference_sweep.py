#!/usr/bin/env python
"""NetVLAD INFERENCE forward (nothing saved) over the batch size: the sliced kernel + finish launch
(diagnostic variant 923) against ONE workgroup per image with the finish in its tail (924), bf16 maps
of 1200 locations.  us per call (30 calls between two events, best of 3) and the fraction of the HBM
bound (x read once + descriptors written once at 8 TB/s).

    python scripts/vlad_inference_sweep.py [--batches 24,48,96,128,160,192,256]
"""
import argparse
import json
import os
import sys

os.environ.setdefault('SCL_DIAG', '1')
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import torch  # noqa: E402

from soft_contrastive_learning_amd import _lib as L  # noqa: E402
from soft_contrastive_learning_amd.model import nets  # noqa: E402
from tests import util_data as U  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--batches', default='24,48,96,128,160,192,256')
    ap.add_argument('--locations', type=int, default=1200)
    args = ap.parse_args()
    dev = torch.device('cuda:0')
    L.use_diag()
    lib = L.load()
    w, c = U.vlad_params()
    wt, ct = torch.tensor(w, device=dev), torch.tensor(c, device=dev)
    n = args.locations
    for b in [int(v) for v in args.batches.split(',')]:
        g = torch.Generator().manual_seed(b)
        x = torch.randn(b, 1, n, 512, generator=g).to(dev).bfloat16()
        rec = {'images': b, 'locations': n}
        bound_us = (b * n * 512 * 2 + b * 32768 * 4 + 2 * 512 * 64 * 4) / 8e12 * 1e6
        for v, name in ((923, 'sliced_plus_finish'), (924, 'one_workgroup_per_image')):
            lib.scl_debug_set_variant(v)
            with torch.no_grad():
                for _ in range(5):
                    nets.netvlad(x, wt, ct, True)
                torch.cuda.synchronize()
                best = 1e9
                for _ in range(3):
                    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                    e0.record()
                    for _ in range(30):
                        nets.netvlad(x, wt, ct, True)
                    e1.record()
                    torch.cuda.synchronize()
                    best = min(best, e0.elapsed_time(e1) / 30 * 1e3)
            rec[name] = {'us': round(best, 1), 'frac_of_hbm_bound': round(bound_us / best, 3)}
        lib.scl_debug_set_variant(0)
        rec['hbm_bound_us'] = round(bound_us, 1)
        print(json.dumps(rec))


if __name__ == '__main__':
    main()
