"""File formats of the reference's scripts (util/io.py): images as RGB uint8 arrays, CSV lists as a
dict of string columns, pickles."""
import csv
import pickle

import numpy as np


def load_img(in_file):
    """util/io.py:16-20 — ``cv2.imread`` (8-bit, three channels, EXIF orientation applied) turned to
    RGB: here through PIL, which decodes PNG to the same bytes."""
    from PIL import Image, ImageOps
    with Image.open(str(in_file)) as im:
        im = ImageOps.exif_transpose(im)
        if im.mode in ('I;16', 'I;16B', 'I;16L', 'I'):
            # imread without IMREAD_ANYDEPTH hands 16-bit files back as 8-bit: the high byte (recalled,
            # not checked against OpenCV here: no cv2 in the image).  Through NumPy — Image.point with a
            # function is not defined for the 16-bit modes in every Pillow version.
            hi = (np.asarray(im).astype(np.int64) >> 8).clip(0, 255).astype(np.uint8)
            return np.repeat(hi[:, :, None], 3, axis=2)
        return np.asarray(im.convert('RGB'), dtype=np.uint8)


def save_img(img, out_file):
    from PIL import Image
    Image.fromarray(np.asarray(img, dtype=np.uint8)).save(str(out_file))


def load_csv(in_file, delimiter=',', has_header=True, keys=()):
    """util/io.py:46-83: a dict ``column -> list of strings``.  Without a header the columns are
    ``keys`` (if their count fits) or 0 .. n-1; a file that holds nothing but its first row returns
    the list of keys instead of a dict (the reference's own convention for one-line files)."""
    with open(in_file, newline='') as f:
        rows = list(csv.reader(f, delimiter=delimiter))
    if not rows:
        return {}
    if has_header:
        names, body = list(rows[0]), rows[1:]
    else:
        names = list(keys) if len(keys) == len(rows[0]) else list(range(len(rows[0])))
        body = rows
    if not body:
        return names
    return {name: [r[i] for r in body] for i, name in enumerate(names)}


def save_csv(columns, out_file, delimiter=','):
    """util/io.py:86-106: header + one line per row; scalar values make a one-row file."""
    names = list(columns)
    listy = bool(names) and isinstance(columns[names[0]], (list, tuple, np.ndarray))
    rows = zip(*(columns[n] for n in names)) if listy else [[columns[n] for n in names]]
    with open(out_file, 'w', newline='') as f:
        f.write(delimiter.join(str(n) for n in names) + '\n')
        for row in rows:
            f.write(delimiter.join('{}'.format(v) for v in row) + '\n')


def save_pickle(data, out_file):
    with open(out_file, 'wb') as f:
        pickle.dump(data, f)


def load_pickle(in_file):
    with open(in_file, 'rb') as f:
        return pickle.load(f)
