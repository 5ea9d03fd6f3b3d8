"""float32 NumPy restatement of the NetVLAD head.  TEST INFRASTRUCTURE ONLY
(see ``oracle/__init__.py``; parity unpinned by the reference).

Follows ``model/nets.py:66-67``: ``tf.nn.l2_normalize(x, axis=-1)`` over the 512
channels then ``netvlad_tf.layers.netVLAD(x, 64)``.  The latter is third-party and
absent from /root/reference; its published algorithm
(uzh-rpg/netvlad_tf_open, python/netvlad_tf/layers.py, unpinned) is restated here:

    s = conv1x1(x; assignment/kernel [1,1,D,K], no bias);  a = softmax_K(s)
    v = sum_{h,w} a[..., None, :] * (x[..., :, None] + cluster_centers[1,1,1,D,K])
    v = matconvnetNormalize(transpose(v, [0,2,1]), 1e-12)   # per cluster, over D
    v = matconvnetNormalize(flatten(transpose(v, [0,2,1])), 1e-12)   # index d*K+k
    matconvnetNormalize(x, e) = x / sqrt(sum(x^2, -1) + e)  (epsilon inside the sqrt)

Two equivalent forms are provided: ``netvlad_literal`` materialises the
[N,D,K] tensor exactly like the TF graph (this is also the "TF-style" CPU timing
variant of BASELINE.md §3), ``netvlad_fused`` uses two matmuls.
"""
import numpy as np

from .losses_np import F32, _f32, l2_normalize


def softmax_last(s):
    """tf.nn.softmax over the last axis (max-subtracted)."""
    s = _f32(s)
    m = np.max(s, axis=-1, keepdims=True)
    e = np.exp(s - m).astype(F32)
    return (e / np.sum(e, axis=-1, keepdims=True, dtype=F32)).astype(F32)


def matconvnet_normalize(x, epsilon=1e-12):
    x = _f32(x)
    return (x / np.sqrt(np.sum(x * x, axis=-1, keepdims=True, dtype=F32)
                        + F32(epsilon))).astype(F32)


def _post_norm(v):
    """v: [B,D,K] un-normalised VLAD -> [B, D*K]."""
    v = matconvnet_normalize(np.transpose(v, (0, 2, 1)))       # [B,K,D], over D
    v = np.transpose(v, (0, 2, 1)).reshape(v.shape[0], -1)     # [B,D*K], d-major
    return matconvnet_normalize(v)


def netvlad_literal(x, assign_w, centers, pre_l2=True):
    """x [B,N,D] (conv5_3 map, channels last, spatial dims flattened);
    assign_w [D,K] (= assignment/kernel[0,0]); centers [D,K] (= cluster_centers[0,0,0]).
    Returns [B, D*K].  Materialises [N,D,K] per image."""
    x = _f32(x)
    if pre_l2:
        x = l2_normalize(x, axis=-1)
    w, c = _f32(assign_w), _f32(centers)
    out = np.empty((x.shape[0], w.shape[0], w.shape[1]), dtype=F32)
    for b in range(x.shape[0]):
        a = softmax_last(x[b] @ w)                              # [N,K]
        v = a[:, None, :] * (x[b][:, :, None] + c[None, :, :])  # [N,D,K]
        out[b] = np.sum(v, axis=0, dtype=F32)
    return _post_norm(out)


def netvlad_fused(x, assign_w, centers, pre_l2=True, return_aux=False):
    """Same result as ``netvlad_literal`` by two matmuls per image:
    V = x^T a + C * colsum(a)."""
    x = _f32(x)
    if pre_l2:
        x = l2_normalize(x, axis=-1)
    w, c = _f32(assign_w), _f32(centers)
    a = softmax_last(x @ w)                                     # [B,N,K]
    v = np.einsum('bnd,bnk->bdk', x, a).astype(F32)
    v = v + c[None] * np.sum(a, axis=1, dtype=F32)[:, None, :]
    out = _post_norm(v)
    if return_aux:
        return out, a, v
    return out
