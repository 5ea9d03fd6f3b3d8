#!/usr/bin/env python
"""Probe: where does the split forward cost time in a training step?"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from soft_contrastive_learning_amd import parallel
from soft_contrastive_learning_amd.model import nets

dev = torch.device('cuda:0')
model = nets.VGG16NetVLAD(compute_dtype=torch.bfloat16, seed=1, fused_relu=True).to(dev)
img = torch.randint(0, 256, (24, 480, 640, 3), generator=torch.Generator().manual_seed(1)).float().to(dev)
g = torch.randn(24, 30, 40, 512, device=dev).bfloat16()
buckets = parallel.GradBuckets(list(model.parameters()))
nets.GRAD_SINK = buckets

def timeit(name, fn, n=10):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    print(name, round((time.perf_counter() - t0) / n * 1e3, 3), 'ms', flush=True)

def fwd_nograd():
    with torch.no_grad():
        model.features(img)

def fwd_grad():
    model.features(img)

def fwd_bwd():
    buckets.zero()
    model.features(img).backward(g)
    buckets.finish()

for split in (False, True, False, True):
    nets.USE_SPLIT_FWD = split
    for side in (True, False):
        nets.USE_SIDE_WRW = side
        timeit('split=%d side=%d fwd+bwd' % (split, side), fwd_bwd)
    nets.USE_SIDE_WRW = True
    timeit('split=%d fwd no_grad' % split, fwd_nograd)
    timeit('split=%d fwd grad   ' % split, fwd_grad)
