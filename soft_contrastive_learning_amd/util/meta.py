"""``learnlarge.util.meta`` as far as the path's scripts import it (evaluation/top-n.py:9)."""
import numpy as np


def get_xy(meta):
    """Positions of a list's rows as one [M, 2] float64 array: column 0 = easting, column 1 =
    northing (the CSV columns may still be strings).  Used like train/train.py:1152-1153."""
    xy = np.empty((len(meta['easting']), 2), dtype=np.float64)
    xy[:, 0] = np.asarray(meta['easting'], dtype=np.float64)
    xy[:, 1] = np.asarray(meta['northing'], dtype=np.float64)
    return xy
