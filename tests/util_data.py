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


def pose_images(b, h, w, seed=42):
    """[B,H,W,3] float32 in 0..255 whose CONTENT differs from image to image (oriented
    sinusoidal textures of image-specific frequency and colour, a few flat rectangles, light
    noise) — uniform-noise images all have the same statistics and give a random-init VGG16
    nearly parallel descriptors (cosine > 0.99), which turns every loss gradient into a
    difference of almost equal terms.  Cheap stand-in for 'pose-correlated' camera frames."""
    rng = np.random.default_rng(seed)
    yy, xx = np.meshgrid(np.arange(h, dtype=F32) / h, np.arange(w, dtype=F32) / w, indexing='ij')
    out = np.empty((b, h, w, 3), F32)
    for i in range(b):
        img = np.full((h, w, 3), 127.5, F32) + rng.uniform(-60, 60, size=3).astype(F32)
        for _ in range(3):
            th = rng.uniform(0, np.pi)
            fr = rng.uniform(2.0, 60.0)
            ph = rng.uniform(0, 2 * np.pi)
            wave = np.sin(2 * np.pi * fr * (np.cos(th) * xx + np.sin(th) * yy) + ph).astype(F32)
            img += wave[:, :, None] * rng.uniform(-50, 50, size=3).astype(F32)
        for _ in range(4):
            y0, x0 = int(rng.integers(0, h - 8)), int(rng.integers(0, w - 8))
            y1, x1 = int(rng.integers(y0 + 4, h + 1)), int(rng.integers(x0 + 4, w + 1))
            img[y0:y1, x0:x1] = 0.5 * img[y0:y1, x0:x1] + rng.uniform(0, 128, size=3).astype(F32)
        img += rng.normal(0, 6.0, size=(h, w, 3)).astype(F32)
        out[i] = np.clip(np.rint(img), 0, 255)
    return out


VGG_CONVS = (('1_1', 3, 64), ('1_2', 64, 64), ('2_1', 64, 128), ('2_2', 128, 128), ('3_1', 128, 256),
             ('3_2', 256, 256), ('3_3', 256, 256), ('4_1', 256, 512), ('4_2', 512, 512), ('4_3', 512, 512),
             ('5_1', 512, 512), ('5_2', 512, 512), ('5_3', 512, 512))


def vgg_variables(seed=77, scope='vgg16_netvlad_pca'):
    """The backbone's variables by TensorFlow name, in the checkpoint layout (kernels [3,3,in,out],
    model/nets.py:12-63): He-scaled kernels, biases of a size that matters, a mean image."""
    rng = np.random.default_rng(seed)
    sd = {scope + '/average_rgb': np.array([123.68, 116.78, 103.94], F32) + rng.uniform(-2, 2, 3).astype(F32)}
    for name, cin, cout in VGG_CONVS:
        sd['%s/conv%s/kernel' % (scope, name)] = (
            rng.standard_normal((3, 3, cin, cout)) * np.sqrt(2.0 / (9 * cin))).astype(F32)
        sd['%s/conv%s/bias' % (scope, name)] = (rng.standard_normal(cout) * 0.1).astype(F32)
    return sd


def retrieval_dataset(seed=5, n_pca=640, n_ref=900, n_query=120, e=288, rank=24):
    """A small place-recognition data set for evaluation/top-n.py: a reference traverse (steps of
    0.2-3 m along a wandering path), queries near random reference poses, and descriptors that are
    a smooth function of the pose (random Fourier features through a rank-`rank` map) plus noise —
    so that whitening has a real spectrum to equalise and neighbours in feature space are mostly
    neighbours on the ground.  e = 288 with 640 PCA rows and d = 256 makes scikit-learn's 'auto'
    solver the exact one (randomized is picked only when d < 0.8 * min(shape))."""
    rng = np.random.default_rng(seed)

    def path(n, start):
        head = np.cumsum(rng.normal(0.0, 0.08, n))
        step = rng.uniform(0.2, 3.0, n)
        return start + np.cumsum(np.stack([step * np.cos(head), step * np.sin(head)], 1), axis=0)

    ref_xy = path(n_ref, np.array([620000.0, 5700000.0]))
    pca_xy = path(n_pca, ref_xy[n_ref // 3] + 40.0)
    query_xy = ref_xy[rng.integers(0, n_ref, n_query)] + rng.normal(0.0, 2.0, (n_query, 2))
    freq = rng.normal(0.0, 1.0 / 15.0, (2, 4 * rank))
    phase = rng.uniform(0.0, 2 * np.pi, 4 * rank)
    mix = rng.standard_normal((4 * rank, e)) * (np.arange(4 * rank)[:, None] % rank + 1.0) ** -0.7

    def feats(xy):
        f = np.cos((xy - ref_xy[0]) @ freq + phase) @ mix / np.sqrt(4 * rank)
        f = f + rng.standard_normal(f.shape) * 0.05
        return (f / np.linalg.norm(f, axis=1, keepdims=True)).astype(F32)

    return {'pca_f': feats(pca_xy), 'ref_f': feats(ref_xy), 'query_f': feats(query_xy),
            'ref_xy': ref_xy, 'query_xy': query_xy}


def sampler_dataset(seed=3, n=360, laps=2):
    """Poses for the tuple sampler (train/train.py:433-582): `laps` passes of a closed course with
    2-3 m between frames, lateral noise, heading from the direction of travel — every frame has
    positives within 15 m and 30 degrees of heading (other laps, neighbours), and frames on the far
    side of the course to draw negatives from.  -> xy [n,2] float64, yaw [n] float64."""
    rng = np.random.default_rng(seed)
    per = n // laps
    ang = np.concatenate([np.linspace(0.0, 2 * np.pi, per, endpoint=False) + rng.normal(0, 0.002, per)
                          for _ in range(laps)] + [np.zeros(n - per * laps)])
    rad = 2.5 * per / (2 * np.pi)
    xy = np.stack([rad * np.cos(ang) * 1.4, rad * np.sin(ang)], 1) + rng.normal(0.0, 0.6, (n, 2))
    yaw = np.arctan2(np.cos(ang), -1.4 * np.sin(ang)) + rng.normal(0.0, 0.05, n)
    return xy + np.array([620000.0, 5700000.0]), yaw


def localization_inputs(seed, num_q, k):
    """Inputs of the in-training localisation check (train/train.py:1181-1193): reference / query poses from
    the sampler course; the 'retrieved' references are the k geographically nearest of a NOISY copy of the
    query pose (60 % of the queries displaced by ~25 m), so that the curves are neither 0 nor 100 %.
    -> ref_xy, query_xy, nearest_latent_indices [Q,k], nearest_d_dist [Q,1], nearest_d_indices [Q,1]."""
    from sklearn.neighbors import KDTree
    rng = np.random.default_rng(seed)
    xy, _ = sampler_dataset()
    ref_xy = xy[::2]
    query_xy = xy[1::2][rng.permutation(len(xy[1::2]))[:num_q]] + rng.normal(0.0, 1.0, (num_q, 2))
    noisy = query_xy + rng.normal(0.0, 25.0, query_xy.shape) * (rng.random((num_q, 1)) < 0.6)
    _, nearest_latent = KDTree(ref_xy).query(noisy, k=k)
    nearest_d_dist, nearest_d_idx = KDTree(ref_xy).query(query_xy, k=1)
    return ref_xy, query_xy, nearest_latent, nearest_d_dist, nearest_d_idx
