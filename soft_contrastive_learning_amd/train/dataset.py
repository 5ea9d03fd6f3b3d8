"""Pose-tagged image sets for the trainer: the per-epoch CSV lists of the reference
(``<set>_<epoch:03d>.csv`` with the columns date, folder, t, easting, northing, yaw —
util/io.py:46-83 ``load_csv``, ``get_xy`` train/train.py:1152-1153) and a synthetic stand-in
with the same interface for machines without the RobotCar data.

A set exposes what the training loop touches: ``xy`` [M,2] metres, ``yaw`` [M] radians,
``len``, and ``load_images(indices) -> float32 [n,H,W,3]`` (0..255 RGB, the tensors
``load_images`` hands to the GPU thread, train/train.py:423-430).
"""
import csv
import os

import numpy as np


def load_frame(path, vlad_cores=64, max_side=240, standard=(180, 240)):
    """One frame as the network sees it (train/train.py:423-430) — a module-level function so that a
    process pool can run it."""
    from ..util import cv, io
    img = io.load_img(path)
    if vlad_cores > 0:
        return cv.resize_img(img, max_side)
    return cv.standard_size(img, h=standard[0], w=standard[1])


def _worker_hello(delay):
    """First task of every loader process: stay busy long enough for the pool to start them all."""
    import time
    time.sleep(delay)
    return os.getpid()


class LoaderPool:
    """Worker PROCESSES for CsvImageSet(pool=...): PNG decoding scales with cores only across
    processes.  Every worker is started (spawned, i.e. exec'd) INSIDE the constructor — the executor
    would otherwise create them lazily on the first ``map``, after the trainer has initialised the
    GPU, and an exec from a process that holds the device is what this pool of machines forbids.
    The executor never starts a process later (it only adds one when it has fewer than
    ``max_workers``, and a dead worker breaks it instead of being replaced); once broken, ``map``
    returns None for good and the sets use their thread pools."""

    def __init__(self, processes):
        import multiprocessing
        from concurrent.futures import ProcessPoolExecutor
        self.processes = int(processes)
        self._ex = ProcessPoolExecutor(max_workers=self.processes,
                                       mp_context=multiprocessing.get_context('spawn'))
        # a submit() adds a process while none is idle: `processes` tasks that keep their worker busy
        # bring up all of them (repeat if an early worker finished and took a second task)
        pids = set()
        for _ in range(8):
            futs = [self._ex.submit(_worker_hello, 0.5) for _ in range(self.processes)]
            pids.update(f.result() for f in futs)
            if len(self._ex._processes) >= self.processes:
                break
        if len(self._ex._processes) != self.processes:
            self._ex.shutdown(cancel_futures=True)
            raise RuntimeError('loader pool: %d of %d worker processes started'
                               % (len(self._ex._processes), self.processes))
        self.pids = sorted(p.pid for p in self._ex._processes.values())
        self.broken = False

    def map(self, fn, *iterables):
        """list(map(fn, ...)) on the workers, or None when the pool is (now) unusable."""
        from concurrent.futures.process import BrokenProcessPool
        if self.broken:
            return None
        try:
            return list(self._ex.map(fn, *iterables))
        except (BrokenProcessPool, RuntimeError):     # a worker died / the pool was shut down
            self.broken = True
            return None

    def shutdown(self):
        self.broken = True
        self._ex.shutdown(cancel_futures=True)


def make_loader_pool(processes):
    """LoaderPool(processes), all workers running on return.  Must be called before the process
    initialises the GPU; returns None when that is too late or ``processes`` < 1, and the sets fall
    back to their thread pools."""
    if processes < 1:
        return None
    try:
        import torch
        if torch.cuda.is_initialized():
            return None
    except ImportError:
        pass
    return LoaderPool(processes)


class CsvImageSet:
    """One reference CSV + an image root.  File of row i (``img_path``, train/train.py:124-128):
    ``<root>/<date>_stereo_centre_<folder:02d>/<t>.png``.  Images (``load_images``,
    train/train.py:423-430): with the NetVLAD head the longer side is brought to ``max_side`` = 240
    (``resize_img``), without it the image is scaled to cover ``standard`` = (180, 240) and
    centre-cropped (``standard_size``) — both with OpenCV's un-filtered bilinear resampling,
    restated in util/cv.py."""

    def __init__(self, csv_file, img_root, vlad_cores=64, max_side=240, standard=(180, 240), ext='.png',
                 need_yaw=True, loader_threads=None, pool=None):
        with open(csv_file) as f:
            rows = list(csv.DictReader(f))
        need = ('date', 'folder', 't', 'easting', 'northing') + (('yaw',) if need_yaw else ())
        if not rows or any(k not in rows[0] for k in need):
            raise ValueError('%s must have the columns %s' % (csv_file, ', '.join(need)))
        self.meta = {k: [r[k] for r in rows] for k in rows[0]}
        self.xy = np.array([[float(e), float(n)] for e, n in
                            zip(self.meta['easting'], self.meta['northing'])], dtype=float)
        # (the localisation reference lists, train/train.py:1158-1167, are read for t / easting /
        # northing only)
        self.yaw = (np.array(self.meta['yaw'], dtype=float) if 'yaw' in self.meta
                    else np.zeros(len(rows), dtype=float))
        self.img_root, self.ext = img_root, ext
        self.vlad_cores, self.max_side, self.standard = vlad_cores, max_side, tuple(standard)
        # decoding a 1280 x 960 PNG takes ~15 ms of one core, a 25-image batch 0.4 s — thirty train
        # steps of this backend: the frames of a batch are decoded and resized by a pool of threads
        # (PIL's decoders and numpy release the interpreter lock); None = min(16, cores)
        self.loader_threads = (min(16, os.cpu_count() or 1) if loader_threads is None
                               else max(int(loader_threads), 1))
        self._pool = None
        self._procs = pool               # make_loader_pool(...): worker processes, shared by the sets

    def __len__(self):
        return len(self.yaw)

    def path(self, i):
        return os.path.join(self.img_root,
                            '{}_stereo_centre_{:02d}'.format(self.meta['date'][i], int(self.meta['folder'][i])),
                            '{}{}'.format(self.meta['t'][i], self.ext))

    def load_raw(self, i):
        """The frame as it lies on disk (the example pictures of the localisation check)."""
        from ..util import io
        return io.load_img(self.path(int(i)))

    def load_image(self, i):
        return load_frame(self.path(int(i)), self.vlad_cores, self.max_side, self.standard)

    def load_images(self, indices):
        """float32 [n,H,W,3], 0..255 RGB (all images of a batch must come out the same size, as
        in the reference, whose feed would fail otherwise)."""
        indices = [int(i) for i in indices]
        frames = None
        if self._procs is not None and len(indices) > 1:
            n = len(indices)
            frames = self._procs.map(load_frame, [self.path(i) for i in indices], [self.vlad_cores] * n,
                                     [self.max_side] * n, [self.standard] * n)     # None: pool broken
        if frames is not None:
            pass
        elif self.loader_threads > 1 and len(indices) > 1:
            if self._pool is None:
                from concurrent.futures import ThreadPoolExecutor
                self._pool = ThreadPoolExecutor(max_workers=self.loader_threads)
            frames = list(self._pool.map(self.load_image, indices))
        else:
            frames = [self.load_image(i) for i in indices]
        return np.stack(frames).astype(np.float32)


class SyntheticImageSet:
    """A closed loop driven ``laps`` times: poses every ``spacing`` metres with heading along
    the track, one deterministic image per pose.  An image is a smooth function of the pose
    (low-frequency pattern indexed by position + per-image noise), so nearby poses look alike
    and the descriptors carry place information — enough for mining and the localisation
    evaluation to have something to find."""

    def __init__(self, num, height=64, width=80, spacing=2.0, laps=2, seed=0, distractor=0.0):
        """``distractor`` in [0, 1): share of an image's contrast taken by a per-IMAGE pattern that
        says nothing about the place (random orientation, frequency and phase — "illumination"),
        with the per-pixel noise raised in step: 0 (the default) leaves a set that untrained
        descriptors already localise perfectly; 0.7 one on which training has something to learn
        (scripts/train_dtype_ab.py, tests/test_gpu_training.py)."""
        self.h, self.w, self.seed = height, width, seed
        self.distractor = float(distractor)
        per_lap = max(num // laps, 1)
        s = (np.arange(num) % per_lap) * spacing                 # arc length along the loop
        length = per_lap * spacing
        ang = 2.0 * np.pi * s / length
        radius = length / (2.0 * np.pi)
        rng = np.random.default_rng(seed)
        jitter = rng.normal(0.0, 0.3, size=(num, 2))             # laps do not coincide exactly
        self.xy = np.stack([radius * np.cos(ang), radius * np.sin(ang)], 1) + jitter
        self.yaw = (ang + np.pi / 2.0) % (2.0 * np.pi)
        self._s = s
        yy, xx = np.mgrid[0:height, 0:width]
        self._grid = (yy / float(height), xx / float(width))

    def __len__(self):
        return len(self.yaw)

    def load_raw(self, i):
        return self.load_images([int(i)])[0].clip(0, 255).astype(np.uint8)

    def load_images(self, indices):
        yy, xx = self._grid
        out = np.empty((len(indices), self.h, self.w, 3), dtype=np.float32)
        for k, i in enumerate(indices):
            s = self._s[int(i)]
            rng = np.random.default_rng(self.seed * 1000003 + int(i))
            a = 1.0 - self.distractor
            base = [127.5 + 100.0 * a * np.sin(0.05 * s * (c + 1) + 6.0 * xx * (c + 1) + 3.0 * yy)
                    for c in range(3)]
            img = np.stack(base, -1)
            sigma = 8.0
            if self.distractor > 0.0:
                fx, fy = rng.uniform(-9.0, 9.0, 2)
                ph, tint = rng.uniform(0.0, 2.0 * np.pi), rng.uniform(0.5, 1.0, 3)
                img = img + (100.0 * self.distractor * np.sin(fx * xx + fy * yy + ph))[..., None] * tint
                sigma = 8.0 + 24.0 * self.distractor
            out[k] = np.clip(img + rng.normal(0.0, sigma, (self.h, self.w, 3)), 0, 255)
        return out
