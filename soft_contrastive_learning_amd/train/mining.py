"""Hard-negative mining cache on the GPU.

The reference caches the descriptors of ``mining_cache_size`` images plus the next anchors
every ``mining_step`` steps, builds ``KDTree(CACHED_FEATURES)`` (train/train.py:1014-1066)
and, per anchor, asks for the WHOLE cache sorted by descriptor distance
(``query(k=MINING_CACHE_SIZE, sort_results=True)``, :446-453); ``get_tuple`` then walks that
list for hard negatives (front) and hard positives (back) (:462-484).

A full ranking of every cached descriptor against every other is one pairwise
squared-distance matrix — the library's ``scl_pairwise_sqdist`` (exact-f32 MFMA Gram,
csrc/gram_loss.hip) — followed by a row sort.  Distances use r_i - 2 G_ij + r_j in float32,
which is what mining needs (an ordering of candidates), not the float64 exactness of the
evaluation-time retrieval in ``evaluation/retrieval.py``.
"""
import numpy as np
import torch

from ..model import losses


class MiningCache:
    """CACHED_FEATURES / CACHED_FEATURE_INDICES / CACHED_FEATURE_TREE of the reference."""

    def __init__(self):
        self.indices = None        # dataset indices of the cached rows (np.int64 [C])
        self.order = None          # [C, C] int64 on the device: row i = cache rows by distance
        self.sqdist = None

    def update(self, features, dataset_indices):
        """features [C,E] float32 on a HIP device, dataset_indices [C] (train/train.py:1032-1036)."""
        f = features.float().contiguous()
        d2 = losses._pairwise_squared_distances(f[None])[0]
        self.sqdist = d2
        # stable sort: ties keep cache order, the row itself (distance 0) comes first
        self.order = torch.argsort(d2, dim=1, stable=True)
        self.indices = np.asarray(dataset_indices, dtype=np.int64)

    def sorted_neighbours(self, dataset_index, k=None):
        """``true_sorted`` of get_tuple (train/train.py:446-453): dataset indices of the cache
        ordered by descriptor distance to the cached copy of ``dataset_index``; ``None`` when
        the image is not cached (the reference then has no hard candidates either)."""
        hits = np.where(self.indices == dataset_index)[0]
        if len(hits) == 0:
            return None
        row = self.order[int(hits[0])]
        if k is not None:
            row = row[:k]
        return self.indices[row.cpu().numpy()].tolist()
