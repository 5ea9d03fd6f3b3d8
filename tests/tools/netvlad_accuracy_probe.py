#!/usr/bin/env python
"""What the two-plane operands of the fused NetVLAD kernels (VF_NPL = 2 in csrc/netvlad.hip: every
float32 operand as bf16 high + low, 2^-17 relative) cost in accuracy, at the bench shape: the
descriptors and all three gradients against the float64 autograd twin on the same bf16-rounded
feature map.  Gates (tests/test_gpu_config1.py): 1e-4, 2e-4, 2e-4, 2.5e-3."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from oracle import netvlad_np as NV, twin_torch as TT
from soft_contrastive_learning_amd.model import nets
from tests import util_data as U
dev = torch.device('cuda:0')
b, n = 24, 1200
x = U.feature_map(b, n, seed=2400); xb = torch.tensor(x).to(torch.bfloat16); xr = xb.float().numpy()
w, c = U.vlad_params(seed=8)
g = np.random.default_rng(5).standard_normal((b, 32768)).astype(np.float32)
xt = xb.to(dev).reshape(b, 30, 40, 512).requires_grad_(True)
wt = torch.tensor(w, device=dev).reshape(1, 1, 512, 64).requires_grad_(True)
ct = torch.tensor(c, device=dev).reshape(1, 1, 1, 512, 64).requires_grad_(True)
out = nets.netvlad(xt, wt, ct, True); out.backward(torch.tensor(g, device=dev))
got = out.detach().cpu().numpy()
x64 = torch.tensor(xr, dtype=torch.float64, requires_grad=True); w64 = torch.tensor(w, dtype=torch.float64, requires_grad=True); c64 = torch.tensor(c, dtype=torch.float64, requires_grad=True)
o64 = TT.netvlad(x64, w64, c64); o64.backward(torch.tensor(g, dtype=torch.float64))
mr = lambda a, bb: float(np.abs(a - bb).max() / np.abs(bb).max())
nr = lambda a, bb: float(np.linalg.norm(a - bb) / np.linalg.norm(bb))
print('emb maxrel vs f64 %.3g (gate 1e-4)' % mr(got, o64.detach().numpy()))
print('gw nrel %.3g gc nrel %.3g (gate 2e-4)' % (nr(wt.grad.cpu().numpy().reshape(512, 64), w64.grad.numpy()), nr(ct.grad.cpu().numpy().reshape(512, 64), c64.grad.numpy())))
print('gx nrel %.3g (gate 2.5e-3)' % nr(xt.grad.float().cpu().numpy().reshape(b, n, 512), x64.grad.numpy()))
