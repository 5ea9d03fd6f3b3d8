"""util/meta.py of the reference, as far as the path's scripts use it (evaluation/top-n.py:9)."""
import numpy as np


def get_xy(meta):
    """[M,2] float positions from a list's easting / northing columns (train/train.py:1152-1153)."""
    return np.array([[e, n] for e, n in zip(meta['easting'], meta['northing'])], dtype=float)
