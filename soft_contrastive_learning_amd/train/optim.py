"""The reference's optimisers (train/train.py:865-878) on torch's fused kernels.

`tf.train.AdamOptimizer(learning_rate)` (train/train.py:870; tensorflow==1.10.0, README.md:6 —
kernel ApplyAdam, non-Nesterov) updates

    lr_t = lr * sqrt(1 - beta2^t) / (1 - beta1^t)
    m   += (1 - beta1) * (g - m)
    v   += (1 - beta2) * (g * g - v)
    var -= lr_t * m / (sqrt(v) + epsilon)                   beta1 0.9, beta2 0.999, epsilon 1e-8

with epsilon OUTSIDE the bias correction ("epsilon hat" of the Adam paper, section 2).
`torch.optim.Adam` divides by sqrt(v / (1 - beta2^t)) + eps.  The two are the same formula with

    eps_torch(t) = epsilon / sqrt(1 - beta2^t)

— 31.6 x epsilon at the first step, 3.3 x after 100, 1.16 x after 1000.  At the reference's learning
rate (5e-6) and batch (25 images) most weights of the lower layers see gradients of 1e-8 … 1e-6:
there the denominators differ by that factor, i.e. a fixed-eps torch Adam takes steps up to
several times LARGER than the reference's during the first hundreds of steps.  `TFAdam` feeds
torch's fused kernel the step-dependent eps, which makes it `tf.train.AdamOptimizer` to rounding
(tests/test_optim.py against oracle/adam_np.py).

`tf.train.MomentumOptimizer(lr, 0.9)` (train/train.py:868): accum = 0.9 accum + g; var -= lr accum
is `torch.optim.SGD(momentum=0.9)` as it stands (dampening 0, no Nesterov).
"""
import math

import torch


class TFAdam(torch.optim.Adam):
    def __init__(self, params, lr, betas=(0.9, 0.999), epsilon=1e-8, **kw):
        super().__init__(params, lr=lr, betas=betas, eps=epsilon, **kw)
        self.tf_epsilon = float(epsilon)
        self._t = None                    # steps taken; read from the state once (one device sync)

    def _steps_done(self):
        for g in self.param_groups:
            for p in g['params']:
                st = self.state.get(p)
                if st and 'step' in st:
                    return int(st['step'])
        return 0

    def eps_for_step(self, t, beta2):
        return self.tf_epsilon / math.sqrt(1.0 - beta2 ** t)

    @torch.no_grad()
    def step(self, closure=None):
        if self._t is None:
            self._t = self._steps_done()
        self._t += 1
        for g in self.param_groups:
            if g.get('capturable'):
                continue                  # a captured graph bakes eps in: plain torch Adam semantics
            g['eps'] = self.eps_for_step(self._t, g['betas'][1])
        return super().step(closure)

    def load_state_dict(self, state_dict):
        super().load_state_dict(state_dict)
        self._t = None


def make_optimizer(kind, params, lr, momentum=0.9):
    """train/train.py:865-870: 'momentum' -> MomentumOptimizer, anything else -> AdamOptimizer."""
    params = list(params)
    if kind == 'momentum':
        return torch.optim.SGD(params, lr=lr, momentum=momentum)
    return TFAdam(params, lr=lr, fused=bool(params) and params[0].is_cuda)
