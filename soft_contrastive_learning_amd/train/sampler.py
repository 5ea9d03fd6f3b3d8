"""Tuple sampler and threaded input pipeline of the trainer.

``TupleSampler.get_tuple`` follows ``get_tuple`` of the reference (train/train.py:433-582):
per anchor, positives are drawn (with replacement, ``np.random.choice``) from the images
within ``max_pos_radius`` whose heading differs by less than 30 degrees (:455-457, :467),
negatives one at a time from everything outside ``min_neg_radius`` of the anchor and — with
``mutually_exclusive_negs`` — of every negative already taken (:471-498); quadruplet shapes
add one "other negative" far from all of them (:503-519).  With a mining cache the walk over
the anchor's cached neighbours supplies hard negatives from the front of the list and hard
positives from its back (:446-453, :459-466, :472-484).  The per-loss ``distances`` payload
is built exactly as :525-571 (squared / plain Euclidean, anchor / pairwise / all-pairs /
pos|neg layouts).

Deliberate differences, all on paths where the reference misbehaves:
  * an anchor without cached neighbours has no hard candidates (the reference would reuse
    the previous anchor's stale ``true_sorted`` or hit a NameError);
  * without ``mutually_exclusive_negs`` the drawn negative itself is excluded (the reference
    adds the stale loop variable ``ti``, :495);
  * an anchor without potential positives drops the batch (``np.random.choice`` on an empty
    list raises in the reference);
  * the candidate scan uses a boolean mask instead of a Python list comprehension over the
    whole dataset per negative (same ascending candidate order, so the same RandomState
    reproduces the same draws).
The weighted-ratio payloads ('wrd', 'swrd') belong to losses outside the hot path and are
refused.

``InputPipeline`` is the CPU_IN -> GPU_IN queue pair of train_cpu_thread (:226-260): a
worker thread samples tuples and loads their images while the device trains.
"""
import math
import queue
import threading

import numpy as np


class TupleSampler:
    def __init__(self, xy, yaw, positives_per_tuple=12, negatives_per_tuple=12,
                 max_pos_radius=15.0, min_neg_radius=15.0, hard_positives_per_tuple=6,
                 hard_negatives_per_tuple=6, mutually_exclusive_negs=True, distance_type='none',
                 cache=None, mining_cache_size=1000, rng=None):
        from sklearn.neighbors import KDTree
        self.xy = np.asarray(xy, dtype=float)
        self.yaw = np.asarray(yaw, dtype=float)
        self.tree = KDTree(self.xy)                               # ref_tree (:236)
        self.p, self.n = positives_per_tuple, negatives_per_tuple
        self.max_pos_radius, self.min_neg_radius = max_pos_radius, min_neg_radius
        self.hard_p, self.hard_n = hard_positives_per_tuple, hard_negatives_per_tuple
        self.exclusive = mutually_exclusive_negs
        if distance_type in ('wrd', 'swrd'):
            raise ValueError("distance type %r belongs to losses outside the hot path"
                             % (distance_type,))
        if distance_type not in ('none', 'anchor', 'pairwise', 'wms', 'logratio'):
            raise ValueError("unknown distance type %r" % (distance_type,))
        self.distance_type = distance_type
        self.cache = cache
        self.cache_k = mining_cache_size
        self.rng = rng if rng is not None else np.random.RandomState(42)   # np.random.seed(42)

    def _radius(self, index, r):
        return self.tree.query_radius(self.xy[index, :].reshape(1, -1), r=r)[0]

    def _one(self, index, tuple_shape, use_hard_negatives):
        if len(tuple_shape) not in (3, 4):
            return None                                            # 'Invalid tuple shape.'
        p_want = tuple_shape[1]
        n_want = tuple_shape[2]
        true_sorted = None
        if use_hard_negatives and self.cache is not None and self.cache.indices is not None:
            true_sorted = self.cache.sorted_neighbours(index, self.cache_k)
        true_sorted = true_sorted or []

        dirty = np.setdiff1d(self._radius(index, self.max_pos_radius), [index])
        potential = [p for p in dirty
                     if abs(self.yaw[index] - self.yaw[p]) % (2 * math.pi) < (math.pi / 6.0)]
        if not potential:
            return None
        hard_positives = []
        if use_hard_negatives and self.hard_p > 0:
            pot = set(potential)
            for ti in reversed(true_sorted):
                if ti in pot:
                    hard_positives.append(ti)
                    if len(hard_positives) >= self.hard_p:
                        break
        hard_positives = hard_positives[:p_want]
        positives = self.rng.choice(potential, p_want - len(hard_positives))
        if hard_positives:
            positives = np.concatenate((positives, hard_positives))
        positives = [int(i) for i in positives]

        excluded = np.zeros(len(self.yaw), dtype=bool)
        excluded[self._radius(index, self.min_neg_radius)] = True
        hard_negatives = []
        if use_hard_negatives:
            for ti in true_sorted:
                if len(hard_negatives) >= min(self.hard_n, n_want):
                    break
                if not excluded[ti]:
                    hard_negatives.append(ti)
                    if self.exclusive:
                        excluded[self._radius(ti, self.min_neg_radius)] = True
                    else:
                        excluded[ti] = True
        remaining = n_want - len(hard_negatives)
        negatives = []
        while len(negatives) < remaining:
            candidates = np.flatnonzero(~excluded)
            if len(candidates) == 0:
                return None                                        # 'Not enough negatives.'
            nxt = int(self.rng.choice(candidates))
            negatives.append(nxt)
            if self.exclusive:
                excluded[self._radius(nxt, self.min_neg_radius)] = True
            else:
                excluded[nxt] = True
        negatives = negatives + [int(i) for i in hard_negatives]

        if len(tuple_shape) == 3:
            tuple_indices = [index] + positives + negatives
        else:
            if not self.exclusive:                                 # :506-510
                for original in np.flatnonzero(excluded):
                    excluded[self._radius(original, self.min_neg_radius)] = True
            candidates = np.flatnonzero(~excluded)
            if len(candidates) == 0:
                return None
            tuple_indices = [index] + positives + negatives + [int(self.rng.choice(candidates))]
        return tuple_indices, positives, negatives

    def _distances(self, index, positives, negatives):
        from sklearn.metrics import pairwise_distances
        dt = self.distance_type
        if dt == 'none':
            return []
        pos_loc = np.array([self.xy[i, :] for i in [index] + positives], dtype=float)
        anchor = self.xy[index, :].reshape(1, -1)
        if dt == 'anchor':
            return np.squeeze(pairwise_distances(pos_loc[1:], anchor, metric='sqeuclidean'))
        if dt == 'pairwise':
            return pairwise_distances(pos_loc, pos_loc, metric='sqeuclidean')
        if dt == 'wms':
            neg_loc = np.array([self.xy[int(i), :] for i in negatives], dtype=float)
            every = np.concatenate((pos_loc, neg_loc), 0)
            return pairwise_distances(every, every, metric='euclidean')
        neg_loc = np.array([self.xy[int(i), :] for i in negatives], dtype=float)      # logratio
        pos_d = np.squeeze(pairwise_distances(pos_loc[1:], anchor, metric='sqeuclidean'))
        neg_d = np.squeeze(pairwise_distances(neg_loc, anchor, metric='sqeuclidean'))
        return np.concatenate((np.atleast_1d(pos_d), np.atleast_1d(neg_d)))

    def get_tuple(self, original_indices, tuple_shape, use_hard_negatives=False):
        """-> (distances per anchor, dataset indices of all T*S images in tuple-major order),
        or ([], []) when the batch has to be dropped."""
        distances, every = [], []
        for index in original_indices:
            index = int(index)
            got = self._one(index, tuple_shape, use_hard_negatives)
            if got is None:
                return [], []
            tuple_indices, positives, negatives = got
            if len(tuple_indices) != sum(tuple_shape):
                return [], []                                      # 'faulty tuple'
            distances.append(self._distances(index, positives, negatives))
            every.extend(tuple_indices)
        return distances, np.asarray(every, dtype=int)


class _WorkerError:
    """An exception raised in a pipeline worker, on its way to the consumer."""

    def __init__(self, exc):
        self.exc = exc


class InputPipeline:
    """TRAIN_CPU_IN_QUEUE -> train_cpu_thread -> TRAIN_GPU_IN_QUEUE (train/train.py:226-260).

    ``put(anchor_indices)`` enqueues one batch of anchors; the worker samples the tuples,
    calls ``load_images(dataset_indices) -> float32 [T*S,H,W,3]`` and makes
    ``(distances, images, dataset_indices)`` available through ``get()``.  Dropped batches
    produce nothing, like the reference's 'Faulty training batch'.  ``used_images`` collects
    every index that reached the device queue (USED_IMAGES, :253-254)."""

    def __init__(self, sampler, load_images, tuple_shape, use_hard_negatives=True, depth=2,
                 workers=1, emit_dropped=False):
        self.sampler, self.load_images, self.tuple_shape = sampler, load_images, tuple_shape
        # emit_dropped: a dropped batch yields a None item, so that every put() is answered by
        # exactly one get() (the single-threaded training loop counts on it)
        self.emit_dropped = emit_dropped
        self.use_hard = use_hard_negatives
        self.cpu_in = queue.Queue()
        self.gpu_in = queue.Queue(maxsize=depth)
        self.used_images = set()
        self.dropped = 0
        self._lock = threading.Lock()
        self._threads = [threading.Thread(target=self._work, daemon=True) for _ in range(workers)]
        for t in self._threads:
            t.start()

    def _work(self):
        while True:
            anchors = self.cpu_in.get()
            try:
                if anchors is None:
                    return
                with self._lock:                      # the sampler's RandomState is shared
                    distances, indices = self.sampler.get_tuple(anchors, self.tuple_shape,
                                                                self.use_hard)
                if len(indices) == len(anchors) * sum(self.tuple_shape):
                    images = self.load_images(indices)
                    self.gpu_in.put((distances, images, indices), block=True)
                    with self._lock:
                        self.used_images.update(int(i) for i in indices)
                else:
                    with self._lock:
                        self.dropped += 1
                    if self.emit_dropped:
                        self.gpu_in.put(None, block=True)
            except BaseException as exc:              # noqa: BLE001 (handed to the consumer)
                # a failing sampler / image loader (missing or corrupt file) must not leave the
                # consumer blocked in get() for ever: the exception travels through the queue in
                # the batch's place and get() re-raises it in the training thread
                self.gpu_in.put(_WorkerError(exc), block=True)
            finally:
                self.cpu_in.task_done()

    def put(self, anchor_indices):
        self.cpu_in.put(list(anchor_indices))

    def get(self, timeout=None):
        """Next batch (None for a dropped one with ``emit_dropped``).  Re-raises, in the calling
        thread, whatever the worker raised while preparing it; with no ``timeout`` the wait
        still wakes up once a second and fails if every worker thread has died."""
        while True:
            try:
                item = self.gpu_in.get(timeout=1.0 if timeout is None else timeout)
                break
            except queue.Empty:
                if timeout is not None:
                    raise
                if not any(t.is_alive() for t in self._threads):
                    raise RuntimeError('InputPipeline: all worker threads have exited and the '
                                       'queue is empty')
        self.gpu_in.task_done()
        if isinstance(item, _WorkerError):
            raise RuntimeError('InputPipeline worker failed: %r' % (item.exc,)) from item.exc
        return item

    def join(self):
        """TRAIN_CPU_IN_QUEUE.join() before mining / evaluation (:1015-1018)."""
        self.cpu_in.join()

    def close(self):
        for _ in self._threads:
            self.cpu_in.put(None)
        for t in self._threads:
            t.join(timeout=5)
