"""CPU oracle of the reference's optimiser update.  TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).

`tf.train.AdamOptimizer(learning_rate)` — train/train.py:870, with TensorFlow's defaults
beta1 = 0.9, beta2 = 0.999, epsilon = 1e-8 — is third-party code absent from /root/reference:
TensorFlow 1.10.0 (README.md:6: "tested using").  Its published algorithm (python/training/adam.py docstring and
the ApplyAdam kernel, core/kernels/training_ops.cc, non-Nesterov branch), restated in float32:

    beta1_power, beta2_power : float32 variables, initialised to beta1, beta2, multiplied by
                               beta1, beta2 AFTER every update (adam.py `_finish`)
    alpha = lr * sqrt(1 - beta2_power) / (1 - beta1_power)
    m  += (g - m) * (1 - beta1)
    v  += (g * g - v) * (1 - beta2)
    var -= (m * alpha) / (sqrt(v) + epsilon)

PARITY UNPINNED: the reference holds no vector for it; pinned by the pencil case of
tests/test_optim.py — the first step from zero slots is var -= lr * g / (|g| + epsilon / sqrt(1 - beta2)),
so a gradient of epsilon / sqrt(0.001) = 3.1623e-7 moves the variable by exactly lr / 2 (a fixed-eps
torch Adam moves it by 0.969 lr).

`tf.train.MomentumOptimizer(lr, momentum)` (train/train.py:868; ApplyMomentum, no Nesterov):
    accum = accum * momentum + g;  var -= lr * accum
"""
import numpy as np

F = np.float32


class TFAdamState:
    def __init__(self, shapes, beta1=0.9, beta2=0.999):
        self.m = [np.zeros(s, F) for s in shapes]
        self.v = [np.zeros(s, F) for s in shapes]
        self.beta1_power = F(beta1)
        self.beta2_power = F(beta2)


def tf_adam_step(variables, grads, state, lr, beta1=0.9, beta2=0.999, epsilon=1e-8):
    """One `apply_gradients` over a list of float32 arrays, in place; returns the variables."""
    lr, b1, b2, eps = F(lr), F(beta1), F(beta2), F(epsilon)
    alpha = lr * np.sqrt(F(1) - state.beta2_power, dtype=F) / (F(1) - state.beta1_power)
    for var, g, m, v in zip(variables, grads, state.m, state.v):
        g = g.astype(F)
        m += (g - m) * (F(1) - b1)
        v += (g * g - v) * (F(1) - b2)
        var -= (m * alpha) / (np.sqrt(v, dtype=F) + eps)
    state.beta1_power = F(state.beta1_power * b1)
    state.beta2_power = F(state.beta2_power * b2)
    return variables


def tf_momentum_step(variables, grads, accums, lr, momentum=0.9):
    lr, mu = F(lr), F(momentum)
    for var, g, a in zip(variables, grads, accums):
        a *= mu
        a += g.astype(F)
        var -= lr * a
    return variables
