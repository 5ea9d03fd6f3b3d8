"""Does the step gain when its MAIN chain (forward, backward-data, head, loss) runs on a HIGH-priority
stream and the weight-gradient kernels stay on their normal-priority side stream?  Same process,
alternating; configs[1] step.  (torch.cuda.Stream.priority_range() on this image: -1 = high, 0 = normal.)

    python scripts/stream_priority_ab.py [--steps 20]
"""
import argparse
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--steps', type=int, default=20)
    args = ap.parse_args()
    from soft_contrastive_learning_amd import parallel
    from soft_contrastive_learning_amd.model import losses, nets
    from soft_contrastive_learning_amd.train.optim import TFAdam
    dev = torch.device('cuda:0')
    torch.cuda.set_device(dev)
    g = torch.Generator().manual_seed(42)
    images = torch.randint(0, 256, (24, 480, 640, 3), generator=g).float().to(dev)
    xy = np.random.default_rng(7).uniform(0.0, 200.0, size=(24, 2))
    dmat = torch.tensor(np.sqrt(((xy[:, None] - xy[None]) ** 2).sum(2)).astype(np.float32)[None], device=dev)
    model = nets.VGG16NetVLAD(compute_dtype=torch.bfloat16, seed=1234, fused_relu=True).to(dev)
    params = list(model.parameters())
    opt = TFAdam(params, lr=5e-6, fused=True)
    buckets = parallel.GradBuckets(params, None)
    nets.GRAD_SINK = buckets
    nets.USE_SIDE_WRW = True
    nets.USE_FUSED_FIRST_WRW = True
    print('priority range', torch.cuda.Stream.priority_range())
    high = torch.cuda.Stream(device=dev, priority=-1)

    def step():
        buckets.zero()
        emb = model(images)
        loss = losses.wms_loss(dmat, emb, d_alpha=0.8, d_beta=15.0)
        loss.backward()
        buckets.finish()
        opt.step()

    def timed(stream):
        ctx = torch.cuda.stream(stream) if stream is not None else torch.cuda.stream(torch.cuda.current_stream())
        with ctx:
            for _ in range(10):
                step()
            torch.cuda.synchronize()
            best = 1e9
            for _ in range(3):
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for _ in range(args.steps):
                    step()
                e1.record()
                torch.cuda.synchronize()
                best = min(best, e0.elapsed_time(e1) / args.steps)
        return best
    for _ in range(20):
        step()
    torch.cuda.synchronize()
    for name, st in (('default stream (priority 0)', None), ('main chain on a priority -1 stream', high),
                     ('default stream again', None), ('priority -1 again', high)):
        print('%-40s %.3f ms/step' % (name, timed(st)))


if __name__ == '__main__':
    main()
