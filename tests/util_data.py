"""Seeded synthetic inputs shared by the parity tests, the golden generator, the smoke
test and bench.py.  NumPy Generators are stable across platforms, so
the same seed gives the same tensors here and on the GPU box."""
import numpy as np

F32 = np.float32


def embeddings(b, e, seed=99, rank=6, mix=0.9):
    """[B,E] rows with a spread of cosine similarities (pure Gaussian rows in 32768-d
    are all ~orthogonal and would leave the mining / exp terms trivial)."""
    rng = np.random.default_rng(seed)
    noise = rng.standard_normal((b, e)) / np.sqrt(e)
    u = rng.standard_normal((b, rank))
    v = rng.standard_normal((rank, e)) / np.sqrt(e)
    x = noise + mix * (u @ v) / np.sqrt(rank)
    x *= rng.uniform(0.5, 2.0, size=(b, 1))        # the losses must be scale-invariant
    return x.astype(F32)


def positions_distances(b, side=200.0, seed=7):
    """B points uniform in a side x side square -> pairwise Euclidean distances [B,B]
    (float64 -> float32 cast like train/train.py:275)."""
    rng = np.random.default_rng(seed)
    xy = rng.uniform(0.0, side, size=(b, 2))
    d = np.sqrt(((xy[:, None, :] - xy[None, :, :]) ** 2).sum(axis=2))
    return d.astype(F32)


def feature_map(b, n, d=512, seed=5, dtype=F32):
    rng = np.random.default_rng(seed)
    return rng.standard_normal((b, n, d)).astype(dtype)


def vlad_params(d=512, k=64, seed=1234, logit_scale=3.0):
    rng = np.random.default_rng(seed)
    w = (rng.standard_normal((d, k)) * logit_scale).astype(F32)
    c = (rng.standard_normal((d, k)) * 0.05).astype(F32)
    return w, c


def tuple_batch(t, p, n, e, seed=21, quad=False, scale=0.02):
    rng = np.random.default_rng(seed)
    s = 1 + p + n + (1 if quad else 0)
    base = rng.standard_normal((t, 1, e))
    out = (base + rng.standard_normal((t, s, e)) * 1.5) * scale
    return out.astype(F32)


def retrieval_sets(r, q, d, seed_r=11, seed_q=12):
    ref = np.random.default_rng(seed_r).standard_normal((r, d)).astype(F32)
    qry = np.random.default_rng(seed_q).standard_normal((q, d)).astype(F32)
    return ref, qry
