#!/usr/bin/env python
"""Timing ablation of conv_first_kernel: scl_debug_set_variant(70000 + bits), bit 0 no image
loads, bit 1 no x0 stores, bit 2 no output stores (results meaningless)."""
import os, sys

os.environ.setdefault('SCL_DIAG', '1')   # the diagnostic build carries the variants (csrc/Makefile)
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from soft_contrastive_learning_amd import _lib
from soft_contrastive_learning_amd.model import nets

dev = torch.device('cuda:0')
lib = _lib.load()
img = torch.randint(0, 256, (24, 480, 640, 3), generator=torch.Generator().manual_seed(1)).float().to(dev)
avg = torch.tensor([123.68, 116.78, 103.94], device=dev)
w = (torch.randn(64, 3, 3, 3) * 0.1).to(dev)
bias = torch.randn(64).to(dev)
for var in (0, 70001, 70002, 70003, 70004, 70005, 70006, 70007, 0):
    lib.scl_debug_set_variant(var)
    for _ in range(2):
        nets._FirstConv.apply(img, avg, w, bias, torch.bfloat16, None)
    torch.cuda.synchronize()
    with _lib.KernelTimer(capacity=64) as kt:
        for _ in range(10):
            nets._FirstConv.apply(img, avg, w, bias, torch.bfloat16, None)
        torch.cuda.synchronize()
    print(var, {k: round(v[1] * 1e3, 1) for k, v in kt.summary().items()})
lib.scl_debug_set_variant(0)
