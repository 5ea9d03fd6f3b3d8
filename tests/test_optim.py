"""The reference's optimiser (train/train.py:865-878, A13 of SURVEY.md section 8): `TFAdam` must be
`tf.train.AdamOptimizer` — epsilon outside the bias correction — not torch's Adam; checked against
oracle/adam_np.py (float32 restatement of TF 1.10's ApplyAdam) and a pencil case."""
import copy
import math

import numpy as np
import pytest
import torch

from oracle import adam_np
from soft_contrastive_learning_amd.train.optim import TFAdam, make_optimizer


def _grads(rng, shapes, lo=-9.0, hi=-2.0):
    """Magnitudes log-uniform over 1e-9 .. 1e-2 (the range of the trainer's gradients at 5e-6:
    where epsilon = 1e-8 matters and where it does not), random signs."""
    return [(10.0 ** rng.uniform(lo, hi, s) * rng.choice([-1.0, 1.0], s)).astype(np.float32) for s in shapes]


def test_first_step_pencil_case():
    g = np.float32(1e-8 / math.sqrt(1e-3))                     # epsilon / sqrt(1 - beta2)
    lr = 1e-3
    var = [np.zeros(4, np.float32)]
    adam_np.tf_adam_step(var, [np.full(4, g, np.float32)], adam_np.TFAdamState([(4,)]), lr)
    assert np.allclose(var[0], -lr / 2, rtol=1e-5)
    for cls, want in ((TFAdam, 0.5), (torch.optim.Adam, 1.0 / (1.0 + math.sqrt(1e-3)))):
        p = torch.nn.Parameter(torch.zeros(4))
        opt = cls([p], lr=lr)
        p.grad = torch.full((4,), float(g))
        opt.step()
        assert np.allclose(p.detach().numpy(), -lr * want, rtol=1e-4), cls


@pytest.mark.parametrize("foreach", [False, True])
def test_tfadam_follows_the_tensorflow_update(foreach):
    rng = np.random.RandomState(3)
    shapes = [(64, 27), (512,), (3,)]
    lr = 5e-6
    init = [rng.randn(*s).astype(np.float32) * 0.05 for s in shapes]
    ref = [a.copy() for a in init]
    st = adam_np.TFAdamState(shapes)
    params = [torch.nn.Parameter(torch.from_numpy(a.copy())) for a in init]
    plain = [torch.nn.Parameter(torch.from_numpy(a.copy())) for a in init]
    opt = TFAdam(params, lr=lr, foreach=foreach)
    opt_plain = torch.optim.Adam(plain, lr=lr, foreach=foreach)
    for _ in range(60):
        gs = _grads(rng, shapes)
        adam_np.tf_adam_step(ref, gs, st, lr)
        for ps, o in ((params, opt), (plain, opt_plain)):
            for p, g in zip(ps, gs):
                p.grad = torch.from_numpy(g.copy())
            o.step()
    plain_off = []
    for a0, r, p, q in zip(init, ref, params, plain):
        moved = np.abs(r - a0).max()
        assert moved > 10 * lr                                                    # 60 steps of ~lr each
        assert np.abs(p.detach().numpy() - r).max() < 2e-4 * moved                # TFAdam == TF
        plain_off.append(np.abs(q.detach().numpy() - r).max() / moved)
    assert max(plain_off) > 2e-2, plain_off                                       # torch's Adam is not


def test_step_count_survives_a_state_dict_round_trip():
    rng = np.random.RandomState(5)
    shapes = [(16, 8)]
    a, b = (torch.nn.Parameter(torch.zeros(16, 8)) for _ in range(2))
    oa = TFAdam([a], lr=1e-3)
    gs = [_grads(rng, shapes)[0] for _ in range(8)]
    for g in gs[:5]:
        a.grad = torch.from_numpy(g.copy())
        oa.step()
    ob = TFAdam([b], lr=1e-3)
    with torch.no_grad():
        b.copy_(a)
    ob.load_state_dict(copy.deepcopy(oa.state_dict()))      # (torch shares the step tensor otherwise)
    for g in gs[5:]:
        for p, o in ((a, oa), (b, ob)):
            p.grad = torch.from_numpy(g.copy())
            o.step()
    assert ob._t == oa._t == 8
    assert torch.equal(a, b)


def test_momentum_is_torch_sgd():
    rng = np.random.RandomState(7)
    shapes = [(32, 9), (32,)]
    init = [rng.randn(*s).astype(np.float32) for s in shapes]
    ref = [a.copy() for a in init]
    acc = [np.zeros(s, np.float32) for s in shapes]
    params = [torch.nn.Parameter(torch.from_numpy(a.copy())) for a in init]
    opt = make_optimizer('momentum', params, lr=1e-2, momentum=0.9)
    assert isinstance(opt, torch.optim.SGD) and isinstance(make_optimizer('adam', params, 1e-3), TFAdam)
    for _ in range(20):
        gs = _grads(rng, shapes, -3, 0)
        adam_np.tf_momentum_step(ref, gs, acc, 1e-2)
        for p, g in zip(params, gs):
            p.grad = torch.from_numpy(g.copy())
        opt.step()
    for r, p in zip(ref, params):
        assert np.abs(p.detach().numpy() - r).max() < 1e-5 * np.abs(r).max()
