"""Where does the one-rank RCCL step lose time against the plain step?  configs[1] step with the
embedding all-gather and the bucket all-reduces switched on one at a time; per variant the device
time per step (events) and the HOST time per step() call (no synchronisation inside).

    python scripts/force_dist_ab.py [--steps 20]
"""
import argparse
import os
import sys
import time

import numpy as np
import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--steps', type=int, default=20)
    ap.add_argument('--side-wrw', type=int, default=1)
    args = ap.parse_args()
    from soft_contrastive_learning_amd import parallel
    from soft_contrastive_learning_amd.model import losses, nets
    from soft_contrastive_learning_amd.train.optim import TFAdam
    dev = torch.device('cuda:0')
    torch.cuda.set_device(dev)
    group = parallel.init_process_group(dev, backend='nccl', force_single=True)
    nets.USE_SIDE_WRW = bool(args.side_wrw)
    g = torch.Generator().manual_seed(42)
    images = torch.randint(0, 256, (24, 480, 640, 3), generator=g).float().to(dev)
    xy = np.random.default_rng(7).uniform(0.0, 200.0, size=(24, 2))
    dmat = torch.tensor(np.sqrt(((xy[:, None] - xy[None]) ** 2).sum(2)).astype(np.float32)[None], device=dev)
    model = nets.VGG16NetVLAD(compute_dtype=torch.bfloat16, seed=1234, fused_relu=True).to(dev)
    params = list(model.parameters())
    opt = TFAdam(params, lr=5e-6, fused=True)
    rows = []
    for name, gather, reduce_, on_side in (('plain', 0, 0, 1), ('gather only', 1, 0, 1),
                                           ('buckets only, from side stream', 0, 1, 1),
                                           ('buckets only, joined', 0, 1, 0),
                                           ('both, from side stream', 1, 1, 1), ('plain again', 0, 0, 1)):
        parallel.ON_SIDE_STREAM = bool(on_side)
        buckets = parallel.GradBuckets(params, group, force_collectives=bool(reduce_))
        nets.GRAD_SINK = buckets

        def step():
            buckets.zero()
            emb = model(images)
            if gather:
                loss = parallel.wms_loss_dp(dmat, emb, 0.8, 15.0)
            else:
                loss = losses.wms_loss(dmat, emb, d_alpha=0.8, d_beta=15.0)
            loss.backward()
            buckets.finish()
            opt.step()
        for _ in range(30):
            step()
        torch.cuda.synchronize()
        best = (1e9, 0)
        for _ in range(3):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            host = 0.0
            e0.record()
            for _ in range(args.steps):
                t0 = time.perf_counter()
                step()
                host += time.perf_counter() - t0
            e1.record()
            torch.cuda.synchronize()
            best = min(best, (e0.elapsed_time(e1) / args.steps, host / args.steps * 1e3))
        rows.append((name, best))
        nets.GRAD_SINK = None
        buckets.close()
    for name, (dev_ms, host_ms) in rows:
        print('%-34s device %.3f ms/step   host %.3f ms per step() call' % (name, dev_ms, host_ms))
    dist.destroy_process_group()


if __name__ == '__main__':
    main()
