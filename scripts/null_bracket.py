#!/usr/bin/env python
"""Calibration of bench.py's bracket correction: event-bracketed duration of the library's empty
kernel (scl_prof_null through the scl_prof_* sink).  Run it plain for the event figure and under
`rocprofv3 --kernel-trace --stats` for the device time of scl_null_kernel; the difference is what
the HIP-event bracket adds to every kernel of the library (profiles/r04/null_kernel_bracket.txt)."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402
import torch  # noqa: E402

from soft_contrastive_learning_amd import _lib as L  # noqa: E402


def main():
    lib = L.load()
    x = torch.zeros(1 << 20, device='cuda:0')
    st = L.stream_of(x)
    for spaced in (False, True):
        for _ in range(20):
            lib.scl_prof_null(st)
        torch.cuda.synchronize()
        with L.KernelTimer(capacity=400) as kt:
            for _ in range(200):
                if spaced:
                    x.add_(1.0)            # a real kernel in between, as in a step
                lib.scl_prof_null(st)
            torch.cuda.synchronize()
        us = np.array([t for n, t in kt.records if n == 'scl_null_kernel']) * 1e3
        print('%s: n %d  median %.2f us  p10 %.2f  p90 %.2f' % (
            'between other kernels' if spaced else 'back to back', us.size, np.median(us),
            np.percentile(us, 10), np.percentile(us, 90)))


if __name__ == '__main__':
    main()
