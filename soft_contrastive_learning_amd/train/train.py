"""Trainer counterpart of the reference's ``train/train.py`` for the hot path.

Keeps the reference's flag names and defaults (train/train.py:1226-1312), the loss
dispatch of ``build_model`` (:585-879), the tuple/batch layout (``[anchor, P positives,
N negatives(, other negative)]`` per tuple, :502-503, 561, 589-594, 654), the learning-rate
schedule (:118-121), the optimisers (:865-878) and the three checkpoint cadences
(:935-937, 1070-1102).  Two data routes:

* ``--synthetic_dataset M`` or ``--shuffled_root DIR`` (+ ``--img_root``): the reference's
  epoch loop (``train_one_epoch`` :987-1109) on a pose-tagged image set — anchors go through
  ``TupleSampler`` -> ``InputPipeline`` (a worker thread samples tuples and loads images while
  the device trains, :226-260); every ``mining_step`` anchors the descriptors of
  ``mining_cache_size`` images plus the next anchors are cached for hard-negative mining
  (``MiningCache``, :1014-1068); every ``eval_step`` anchors a rolling checkpoint is written,
  the loss on the other region is measured (``get_eval_loss`` :1112-1149) and localisation is
  evaluated on both regions with the exact HIP top-5 (``evaluate_localization`` :1156-1193);
  every ``save_step`` a part checkpoint (:1094-1102).
* default: ``SyntheticTuples`` — random images with the per-loss ``distances`` payload the GPU
  thread receives (:263-275); what bench.py's workload looks like.

The PCA threads and the eigenvalue / incremental-PCA losses stay out.

    python -m soft_contrastive_learning_amd.train.train --loss wms --vlad_cores 64 \
        --reduction none --tuples_per_batch 1 --steps 20
    python -m soft_contrastive_learning_amd.train.train --loss wms --synthetic_dataset 400 \
        --height 64 --width 80 --positives_per_tuple 4 --negatives_per_tuple 4
    python -m torch.distributed.run --nproc-per-node 8 -m soft_contrastive_learning_amd.train.train ...
"""
import argparse
import json
import os
import time

import numpy as np
import torch
import torch.distributed as dist

from .. import checkpoint, parallel, pointnetvlad_cls
from ..model import losses, nets
from .optim import make_optimizer


def make_parser():
    p = argparse.ArgumentParser()
    # output / restore (train/train.py:1233-1238)
    p.add_argument('--checkpoint', default='')
    p.add_argument('--resume', action='store_true',
                   help='also restore the optimizer slots and the global step from --checkpoint '
                        '(the reference only warm-starts the weights, train/train.py:882-905)')
    p.add_argument('--out_root', default='./scl_logs')
    p.add_argument('--out_folder', default='')
    p.add_argument('--max_to_keep', type=int, default=1)
    # tuple size (:1241-1249)
    p.add_argument('--positives_per_tuple', type=int, default=12)
    p.add_argument('--negatives_per_tuple', type=int, default=12)
    p.add_argument('--hard_negatives_per_tuple', type=int, default=6)
    p.add_argument('--hard_positives_per_tuple', type=int, default=6)
    p.add_argument('--mutually_exclusive_negs', type=bool, default=True)
    # loss (:1252-1266)
    p.add_argument('--loss', default='wms')
    p.add_argument('--margin_1', type=float, default=0.1)
    p.add_argument('--margin_2', type=float, default=0.2)
    p.add_argument('--alpha', type=float, default=0.8)
    p.add_argument('--beta', type=float, default=15)
    p.add_argument('--wfunction', default='exp', help='exp, lin, tanh')
    p.add_argument('--sumfunction', default='ms', help='ms, plain')
    p.add_argument('--msmining', type=bool, default=False)   # type=bool as in the reference
    p.add_argument('--lam', type=float, default=0.5, help='Scaling factor between loss components.')
    p.add_argument('--max_pos_radius', type=float, default=15)
    p.add_argument('--min_neg_radius', type=float, default=15)
    # training (:1269-1281)
    p.add_argument('--tuples_per_batch', type=int, default=1)
    p.add_argument('--max_epoch', type=int, default=5)
    p.add_argument('--base_lr', type=float, default=5e-6)
    p.add_argument('--minimal_lr', type=float, default=5e-12)
    p.add_argument('--lr_down_factor', type=float, default=0.5)
    p.add_argument('--lr_down_frequency', type=float, default=1)
    p.add_argument('--momentum', type=float, default=0.9)
    p.add_argument('--optimizer', default='adam', help='adam, momentum')
    # head (:1284-1289); only the NetVLAD / no-reduction path is on the hot path
    p.add_argument('--reduction', default='none')
    p.add_argument('--vlad_cores', default=64, type=int)
    # hard negative mining (:1288-1293)
    p.add_argument('--mining_step', type=int, default=250)
    p.add_argument('--mining_cache_size', type=int, default=1000)
    # cadence, validation loss and localisation testing (:1296-1300)
    p.add_argument('--eval_step', type=int, default=100)
    p.add_argument('--save_step', type=int, default=500)
    p.add_argument('--num_eval_queries', type=int, default=50)
    p.add_argument('--eval_ref_r', default=5, type=int)
    # data set names (:1303-1307) and locations (:1226-1231)
    p.add_argument('--local_ref_set', default='train_ref')
    p.add_argument('--local_query_set', default='train_query')
    p.add_argument('--other_ref_set', default='test_ref')
    p.add_argument('--other_query_set', default='test_query')
    p.add_argument('--train_ref_r', default=1, type=int)
    p.add_argument('--img_root', default='')
    p.add_argument('--anchor_root', default='',
                   help='directory of the anchor lists <local_ref_set>_<train_ref_r>_<epoch:03d>.csv '
                        '(column idx; train/train.py:1007-1009); empty: a seeded permutation of '
                        'every train_ref_r-th image')
    p.add_argument('--loc_ref_root', default='',
                   help='directory of the localisation reference lists <ref_set>_<eval_ref_r>.csv '
                        '(train/train.py:1158); empty: every eval_ref_r-th image of the set')
    p.add_argument('--shuffled_root', default='',
                   help='directory of the per-epoch lists <set>_<epoch:03d>.csv; empty: synthetic')
    p.add_argument('--loader_processes', type=int, default=0,
                   help='worker processes decoding the frames of --shuffled_root sets (0: threads only). '
                        'Created before the device is initialised; ignored when it already is')
    p.add_argument('--force_dist', type=int, default=0,
                   help='single process only: form a ONE-rank RCCL process group and take the data-parallel '
                        'routes through it (the collective path on a one-GPU box)')
    p.add_argument('--dist_timeout', type=float, default=300.0,
                   help='seconds: rendezvous and every collective')
    p.add_argument('--tensorboard', type=int, default=1,
                   help='TensorBoard event files <out>/local and <out>/other with the reference\'s tags '
                        '(train/train.py:304, 380-397, 929-932, 1139-1147)')
    p.add_argument('--save_examples', type=int, default=-1,
                   help='curve PDFs and example pictures of every localisation check (train/train.py:368-420): '
                        '1 / 0; -1 = with --shuffled_root only')
    p.add_argument('--synthetic_dataset', type=int, default=0,
                   help='M > 0: train on a synthetic pose-tagged set of M images through the '
                        'sampler / pipeline / mining / evaluation route')
    p.add_argument('--synthetic_distractor', type=float, default=0.0,
                   help='synthetic set: share of the image contrast that is place-independent')
    # synthetic stand-in for the dataset pipeline
    p.add_argument('--steps', type=int, default=10, help='steps per epoch')
    p.add_argument('--height', type=int, default=180)
    p.add_argument('--width', type=int, default=240)
    p.add_argument('--dtype', default='f32', choices=['f32', 'bf16'])
    p.add_argument('--seed', type=int, default=42)
    return p


def distance_type(loss):
    """train/train.py:1378-1391."""
    if 'pairwise' in loss:
        return 'pairwise'
    if 'distance' in loss:
        return 'anchor'
    if 'swrd' in loss:
        return 'swrd'
    if 'wrd' in loss:
        return 'wrd'
    if 'wms' in loss:
        return 'wms'
    if 'logratio' in loss:
        return 'logratio'
    return 'none'


def tuple_shape_for(loss, positives, negatives):
    """train/train.py:589-594: quadruplet losses turn the last negative into 'other'."""
    if 'quadruplet' in loss:
        return [1, positives, negatives - 1, 1]
    return [1, positives, negatives]


def get_learning_rate(epoch, flags):
    """train/train.py:118-121."""
    lr = flags.base_lr * (flags.lr_down_factor ** (epoch // flags.lr_down_frequency))
    return max(lr, flags.minimal_lr)


SUPPORTED_LOSSES = ('triplet', 'lazy_triplet', 'evil_triplet', 'quadruplet', 'lazy_quadruplet',
                    'evil_quadruplet', 'ms_loss', 'wms', 'logratio',
                    'distance_triplet', 'distance_lazy_triplet', 'distance_quadruplet',
                    'distance_lazy_quadruplet', 'huber_distance_triplet',
                    'huber_distance_lazy_triplet', 'huber_distance_quadruplet',
                    'huber_distance_lazy_quadruplet')


def compute_loss(flags, tuple_shape, output, distances, local_rows=None, group=None):
    """The loss dispatch of build_model (train/train.py:700-855) on ``output`` [T*S, E]."""
    t = flags.tuples_per_batch
    s = sum(tuple_shape)
    loss = flags.loss
    if loss in ('ms_loss', 'wms') and group is not None:
        if loss == 'wms':
            return parallel.wms_loss_dp(distances, output, flags.alpha, flags.beta, group=group,
                                        wfunction=flags.wfunction, sumfunction=flags.sumfunction)
        return parallel.ms_loss_dp(distances, output, group=group, ms_mining=flags.msmining)
    if group is not None and loss in SUPPORTED_LOSSES:
        # per-tuple losses shard by tuple: the local mean, then the mean over ranks
        return parallel.tuple_loss_dp(compute_loss(flags, tuple_shape, output, distances), group)
    outs = torch.split(output.reshape(t, s, -1), tuple_shape, dim=1)       # :654
    if loss == 'triplet':
        return pointnetvlad_cls.triplet_loss(outs[0], outs[1], outs[2], flags.margin_1)
    if loss == 'lazy_triplet':
        return pointnetvlad_cls.lazy_triplet_loss(outs[0], outs[1], outs[2], flags.margin_1)
    if loss == 'evil_triplet':
        return losses.evil_triplet_loss(outs[0], outs[1], outs[2], flags.margin_1)
    if loss == 'quadruplet':
        return pointnetvlad_cls.quadruplet_loss(outs[0], outs[1], outs[2], outs[3],
                                                flags.margin_1, flags.margin_2)
    if loss == 'lazy_quadruplet':
        return pointnetvlad_cls.lazy_quadruplet_loss(outs[0], outs[1], outs[2], outs[3],
                                                     flags.margin_1, flags.margin_2)
    if loss == 'evil_quadruplet':
        return losses.evil_quadruplet_loss(outs[0], outs[1], outs[2], outs[3], flags.margin_1,
                                           flags.margin_2)
    if loss in SUPPORTED_LOSSES and 'distance' in loss:                        # :719-763
        d_max_squared = float(flags.max_pos_radius) ** 2                       # :695
        f_max_squared = 2.0                                                    # :696
        trip = 'lazy_triplet_loss' if 'lazy' in loss else 'triplet_loss'
        dist = 'huber_distance_loss' if 'huber' in loss else 'distance_loss'
        if 'quadruplet' in loss:
            return losses.distance_quadruplet_loss(outs[0], outs[1], outs[2], outs[3],
                                                   flags.margin_1, flags.margin_2, flags.lam,
                                                   distances, d_max_squared, f_max_squared, trip,
                                                   dist)
        return losses.distance_triplet_loss(outs[0], outs[1], outs[2], flags.margin_1, flags.lam,
                                            distances, d_max_squared, f_max_squared, trip, dist)
    if loss == 'ms_loss':
        return losses.ms_loss(distances, output, ms_mining=flags.msmining)     # :821-827
    if loss == 'wms':
        # only d_alpha, d_beta, wfunction, sumfunction are forwarded (:851-852)
        return losses.wms_loss(distances, output, d_alpha=flags.alpha, d_beta=flags.beta,
                               wfunction=flags.wfunction, sumfunction=flags.sumfunction)
    if loss == 'logratio':
        p = flags.positives_per_tuple
        pos_d, neg_d = torch.split(distances.reshape(t, -1, 1), [p, flags.negatives_per_tuple], 1)
        return losses.logratio_loss(outs[0], outs[1], outs[2], pos_d, neg_d)   # :854-855
    raise ValueError("loss %r is outside the hot path (supported: %s)" % (loss, SUPPORTED_LOSSES))


class SyntheticTuples:
    """Stand-in for get_tuple + load_images (train/train.py:423-582): per step one batch of
    T tuples with the reference's row order and per-loss ``distances`` tensor."""

    def __init__(self, flags, tuple_shape, device, rank=0, world=1):
        self.f, self.shape, self.dev = flags, tuple_shape, device
        self.rank, self.world = rank, world
        self.gen = torch.Generator().manual_seed(flags.seed + rank)
        self.rng = np.random.default_rng(7 + flags.seed)

    def batch(self):
        f = self.f
        t, s = f.tuples_per_batch, sum(self.shape)
        img = torch.randint(0, 256, (t * s, f.height, f.width, 3), generator=self.gen).float()
        dtype = distance_type(f.loss)
        gb = t * s * self.world
        xy = self.rng.uniform(0.0, 200.0, size=(gb, 2))         # same on every rank
        mine = slice(self.rank * t, (self.rank + 1) * t)        # per-tuple payloads: this rank's tuples
        if dtype == 'wms':
            # sklearn pairwise_distances(all, all, 'euclidean') (:557-563), rank-3 [T,S,S]
            d = np.sqrt(((xy[:, None] - xy[None]) ** 2).sum(2)).astype(np.float32)[None]
        elif dtype == 'anchor':
            # squared metres anchor -> positives inside max_pos_radius (:529-533), [T,P]
            d = self.rng.uniform(0.0, f.max_pos_radius ** 2,
                                 (t * self.world, f.positives_per_tuple)).astype(np.float32)[mine]
        elif dtype == 'logratio':
            p, n = f.positives_per_tuple, f.negatives_per_tuple      # squared metres (:569-571)
            d = np.concatenate([self.rng.uniform(1, 15 ** 2, (t * self.world, p)),
                                self.rng.uniform(15 ** 2, 200 ** 2, (t * self.world, n))],
                               1).astype(np.float32)[mine]
        elif f.loss == 'ms_loss':
            # labels built in build_model (:822-826), globally unique across ranks
            p = f.positives_per_tuple
            one = np.concatenate((np.zeros(1 + p), np.arange(f.negatives_per_tuple) + 1))
            d = np.concatenate([one + k * (f.negatives_per_tuple + 1)
                                for k in range(t * self.world)])
        else:
            d = None
        dist_t = None if d is None else torch.as_tensor(d).to(self.dev)
        return dist_t, img.to(self.dev)


def batch_distances(flags, distances, device, world=1, rank_offset=0):
    """The sampler's per-anchor payloads -> the tensor ``ops['distances']`` holds for the loss
    (train/train.py:665-691); ms_loss takes the labels of :822-826 instead."""
    t = flags.tuples_per_batch
    if flags.loss == 'ms_loss':
        one = np.concatenate((np.zeros(1 + flags.positives_per_tuple),
                              np.arange(flags.negatives_per_tuple) + 1))
        lab = np.concatenate([one + (k + rank_offset) * (flags.negatives_per_tuple + 1)
                              for k in range(t * world)])
        return torch.as_tensor(lab).to(device)
    if distance_type(flags.loss) == 'none':
        return None
    return torch.as_tensor(np.asarray(distances, dtype=np.float32)).to(device)


def cadence_due(step, every, stride, world):
    """Does the cadence `every` (in anchors) fire at anchor position `step`?  One rank: the
    reference's `step % every == 0` (train/train.py:1014, 1070, 1094); several ranks: a step covers
    `stride` = tuples_per_batch * world anchors and the cadence fires when a multiple lies inside it."""
    return step % every == 0 if world == 1 else step % every < stride


def next_cadence(step, every, stride, world, n_anchors):
    """First anchor position after `step` (in steps of `stride`) at which cadence_due fires again,
    capped at `n_anchors`: the end of the window a mining-cache refresh at `step` must cover."""
    nxt = step + stride
    while nxt < n_anchors and not cadence_due(nxt, every, stride, world):
        nxt += stride
    return min(nxt, n_anchors)


def loc_ref_set(flags, ref_set_name):
    """The thinned reference list the localisation check runs against:
    <loc_ref_root>/<ref_set>_<eval_ref_r>.csv (train/train.py:1158)."""
    from . import dataset
    return dataset.CsvImageSet(os.path.join(flags.loc_ref_root,
                                            '{}_{}.csv'.format(ref_set_name, flags.eval_ref_r)),
                               flags.img_root, vlad_cores=flags.vlad_cores, need_yaw=False)


def open_sets(flags, epoch):
    """(local_ref, local_query, other_ref, other_query) image sets of an epoch."""
    from . import dataset
    if flags.shuffled_root:
        def one(name):
            # (image size: the reference's fixed rule, train/train.py:423-430 — longer side 240 with
            # the NetVLAD head, 180 x 240 cover-and-crop without; --height / --width are the
            # synthetic set's)
            return dataset.CsvImageSet(os.path.join(flags.shuffled_root,
                                                    '%s_%03d.csv' % (name, epoch)),
                                       flags.img_root, vlad_cores=flags.vlad_cores,
                                       pool=getattr(flags, 'loader_pool', None))
        return (one(flags.local_ref_set), one(flags.local_query_set), one(flags.other_ref_set),
                one(flags.other_query_set))
    m = flags.synthetic_dataset
    dis = flags.synthetic_distractor

    def make(num, seed):
        return dataset.SyntheticImageSet(num, flags.height, flags.width, seed=seed, distractor=dis)
    local, other = make(m, flags.seed), make(max(m // 2, 8), flags.seed + 1)
    if dis > 0.0:
        # queries from ANOTHER traverse of the same track (own jitter, noise and distractors), like
        # the reference's query sets: with the references themselves as queries every nearest
        # descriptor is the query's own and the localisation check says 100 % whatever the weights
        return local, make(m, flags.seed + 100), other, make(max(m // 2, 8), flags.seed + 101)
    return local, local, other, other


def train_dataset_epoch(flags, epoch, state, log):
    """``train_one_epoch`` (train/train.py:987-1109) on this process's device.

    With more than one rank (``state['group']``; new work, the reference is single-GPU) a step takes
    ``tuples_per_batch`` anchors PER RANK: rank r trains on anchors [r t, (r + 1) t) of every
    block of world * t, with its own sampler stream (seeded by (epoch, rank), with and without the
    mining cache); a batch dropped on
    one rank ('Faulty training batch') is dropped on all of them (one MIN all-reduce of a flag per
    step), the per-tuple losses become the mean over all ranks' tuples (parallel.tuple_loss_dp),
    the pairwise losses take the gathered batch (for wms the ranks exchange the image indices and
    build the full distance matrix from the poses every rank holds), the mining cache is
    extracted in shards and all-gathered, and rank 0 alone logs and writes checkpoints; the
    evaluations run on every rank alike (no collective inside them)."""
    from . import evaluate, mining
    from .sampler import InputPipeline, TupleSampler
    model, opt, buckets, saver, dev = (state[k] for k in ('model', 'opt', 'buckets', 'saver', 'dev'))
    group = state.get('group')
    world = dist.get_world_size(group) if group is not None else 1
    rank = dist.get_rank(group) if group is not None else 0
    tuple_shape = state['tuple_shape']
    t, s_img = flags.tuples_per_batch, flags.tuples_per_batch * sum(state['tuple_shape'])
    local_ref, local_query, other_ref, other_query = open_sets(flags, epoch)
    dtype = distance_type(flags.loss)
    dtype = dtype if dtype in ('anchor', 'pairwise', 'wms', 'logratio') else 'none'
    cache = mining.MiningCache()

    def make_sampler(image_set, use_cache):
        return TupleSampler(image_set.xy, image_set.yaw, flags.positives_per_tuple,
                            flags.negatives_per_tuple, flags.max_pos_radius, flags.min_neg_radius,
                            flags.hard_positives_per_tuple, flags.hard_negatives_per_tuple,
                            flags.mutually_exclusive_negs, dtype, cache if use_cache else None,
                            flags.mining_cache_size,
                            np.random.RandomState(42 + epoch if world == 1 else [42 + epoch, rank]))
    sampler = make_sampler(local_ref, True)
    other_sampler = make_sampler(other_ref, False)
    pipe = InputPipeline(sampler, local_ref.load_images, tuple_shape, use_hard_negatives=True,
                         depth=2, emit_dropped=True)
    if flags.anchor_root:                                     # train/train.py:1007-1009
        from ..util import io
        anchors = np.array(io.load_csv(os.path.join(
            flags.anchor_root, '{}_{}_{:03d}.csv'.format(flags.local_ref_set, flags.train_ref_r, epoch)))['idx'],
            dtype=int)
        if len(anchors) and (anchors.min() < 0 or anchors.max() >= len(local_ref)):
            raise ValueError('anchor list of epoch %d points outside %s' % (epoch, flags.local_ref_set))
    else:
        anchors = np.random.RandomState(1000 + epoch).permutation(
            np.arange(0, len(local_ref), max(flags.train_ref_r, 1)))
    if flags.steps > 0:
        anchors = anchors[:flags.steps * t * world]           # --steps = batches per rank
    # whole batches only: compute_loss / batch_distances reshape with tuples_per_batch, a short
    # tail batch would raise at the end of the epoch, before the epoch checkpoint is written
    anchors = anchors[:(len(anchors) // (t * world)) * (t * world)]
    lr = get_learning_rate(epoch, flags)
    for g in opt.param_groups:
        g['lr'] = lr
    if rank == 0:
        log({'event': 'epoch', 'epoch': epoch, 'anchors': int(len(anchors)),
             'anchor_source': 'list' if flags.anchor_root else 'permutation',
             'first_anchors': [int(a) for a in anchors[:4]], 'learning_rate': lr})

    def loss_of(distances, images):
        """Single-process loss (the evaluation on the other region: every rank alike)."""
        out = nets.full_out(images)
        return compute_loss(flags, tuple_shape, out, batch_distances(flags, distances, dev))

    def train_loss(distances, images, indices):
        out = nets.full_out(images)
        if group is None:
            return compute_loss(flags, tuple_shape, out, batch_distances(flags, distances, dev))
        if flags.loss == 'wms':
            # the global batch's pairwise geographic distances (train/train.py:557-563) from the
            # poses of ALL ranks' images: the indices travel, every rank holds the poses
            idx_all = parallel.all_gather_ragged(
                torch.as_tensor(np.asarray(indices, dtype=np.int64), device=dev), group).cpu().numpy()
            xy = np.asarray(local_ref.xy, dtype=np.float64)[idx_all]
            dmat = np.sqrt(((xy[:, None] - xy[None]) ** 2).sum(2)).astype(np.float32)[None]
            payload = torch.as_tensor(dmat).to(dev)
        elif flags.loss == 'ms_loss':
            payload = batch_distances(flags, distances, dev, world=world)   # globally unique labels
        else:
            payload = batch_distances(flags, distances, dev)
        return compute_loss(flags, tuple_shape, out, payload, group=group)

    def train_on(item):
        have = 0 if item is None else 1
        if group is not None:                                   # a batch dropped anywhere is dropped everywhere
            flag = torch.tensor([have], device=dev)
            dist.all_reduce(flag, op=dist.ReduceOp.MIN, group=group)
            have = int(flag)
        if not have:
            return                                              # 'Faulty training batch'
        distances, images, indices = item
        buckets.zero()
        loss = train_loss(distances, torch.from_numpy(images).to(dev), indices)
        loss.backward()
        buckets.finish()
        opt.step()
        state['step'] += 1
        rec = {'step': state['step'], 'epoch': epoch, 'loss': float(loss.detach()),
               'learning_rate': lr}
        if rank == 0:
            print('Train batch loss: {}'.format(rec['loss']))      # :289
            log(rec)

    def drain(outstanding):
        while outstanding:
            train_on(pipe.get())
            outstanding -= 1
        return 0

    outstanding, mining_count = 0, 0
    try:
        stride = t * world

        def due(every):
            return cadence_due(step, every, stride, world)
        for step in range(0, len(anchors), stride):
            if due(flags.mining_step):                               # :1014-1068
                outstanding = drain(outstanding)
                mining_indices = np.arange(mining_count * flags.mining_cache_size,
                                           (mining_count + 1) * flags.mining_cache_size) % len(local_ref)
                # every anchor trained before the NEXT refresh: with several ranks the cadence fires
                # up to stride - 1 anchors late, so the window runs to the next firing, not to
                # step + mining_step (anchors outside the cache get no hard positives / negatives)
                window_end = next_cadence(step, flags.mining_step, stride, world, len(anchors))
                to_mine = anchors[step:window_end]
                mining_indices = np.concatenate([mining_indices, to_mine])
                if group is None:
                    feats = evaluate.extract_features(model, local_ref, mining_indices, s_img)
                else:                        # each rank embeds a contiguous share, rank order = list order
                    share = np.array_split(mining_indices, world)[rank]
                    feats = parallel.all_gather_ragged(
                        evaluate.extract_features(model, local_ref, share, s_img)
                        if len(share) else torch.zeros((0, 0), device=dev), group)   # (width: from the others)
                cache.update(feats, mining_indices)
                mining_count += 1
                if rank == 0:
                    log({'step': state['step'], 'event': 'mining_cache', 'images': int(len(mining_indices)),
                         'anchor_window': [int(step), int(window_end)]})
            if due(flags.eval_step):                                 # :1070-1092
                outstanding = drain(outstanding)
                if rank == 0:
                    saver.save_rolling(model, state['step'], opt)
                test_number = state['step'] // flags.eval_step
                nq = (flags.num_eval_queries // t) * t
                test_idx = np.arange(test_number * nq, (test_number + 1) * nq) % len(other_ref)
                with torch.no_grad():
                    ev, used = evaluate.eval_loss(loss_of, other_sampler, other_ref, test_idx, t,
                                                  tuple_shape, dev)
                rec = {'step': state['step'], 'event': 'eval', 'other_region_loss': ev,
                       'eval_batches': used}
                for mode, rset, qset, rname in (('other', other_ref, other_query, flags.other_ref_set),
                                                ('local', local_ref, local_query, flags.local_ref_set)):
                    if flags.loc_ref_root and flags.shuffled_root:     # train/train.py:1158-1167
                        rset = loc_ref_set(flags, rname)
                        refs = np.arange(len(rset))
                    else:
                        refs = np.arange(0, len(rset), max(flags.eval_ref_r, 1))
                    q = np.arange(test_number * flags.num_eval_queries,
                                  (test_number + 1) * flags.num_eval_queries) % len(qset)
                    want = (flags.save_examples if flags.save_examples >= 0 else bool(flags.shuffled_root)) \
                        and rank == 0
                    out_name = '{:02d}_checkpoint-{}'.format(epoch, state['step'])   # :1079-1080
                    metrics, nearest = evaluate.evaluate_localization(
                        model, rset, refs, qset, q, s_img,
                        plots=(saver.out_dir, mode, out_name) if want else None)    # :368-396
                    rec[mode] = metrics
                    if want:                                         # :400-420
                        evaluate.save_example_pictures(saver.out_dir, mode, out_name, qset, q, rset, refs,
                                                       nearest)
                if rank == 0:
                    print('Other region loss: {}'.format(ev))        # :1144
                    log(rec)
            if due(flags.save_step):                                 # :1094-1102
                outstanding = drain(outstanding)
                if rank == 0:
                    saver.save_part(model, state['step'], opt)
            pipe.put(anchors[step + rank * t:step + (rank + 1) * t])
            outstanding += 1
            if outstanding > 1:                                      # one batch stays in flight
                train_on(pipe.get())
                outstanding -= 1
        drain(outstanding)
    except Exception:
        # a failure on ONE rank (a worker error re-raised by pipe.get(), a bad batch) would leave
        # the others blocked in the per-step MIN all-reduce: take the process group down with it
        # (no destroy_process_group() first: with RCCL it can block behind this rank's outstanding
        # collectives while the peers wait in theirs — a hang is not an exception)
        if group is not None and os.environ.get('SCL_TRAIN_ABORT_ON_RANK_FAILURE', '1') != '0':
            try:
                pipe.close()
                if getattr(flags, 'loader_pool', None) is not None:
                    flags.loader_pool.shutdown()
                for b in (state.get('boards') or {}).values():
                    b.close()
            except Exception:
                pass
            parallel.abort_rank(1)        # peers see the closed connection instead of waiting
        raise
    finally:
        pipe.close()
    if rank == 0:
        saver.save_epoch(model, epoch, state['step'], opt)           # :984


def main(argv=None):
    flags = make_parser().parse_args(argv)
    if flags.vlad_cores not in (0, 64) or flags.reduction != 'none':
        raise SystemExit('only --vlad_cores 64 | 0 with --reduction none is on the hot path')
    world = int(os.environ.get('WORLD_SIZE', '1'))
    rank = int(os.environ.get('RANK', '0'))
    local_rank = int(os.environ.get('LOCAL_RANK', '0'))
    # TEST ONLY (SCL_TRAIN_ONE_GPU_GLOO=1): every rank on cuda:0 with gloo carrying the collectives —
    # the data-parallel routes on a one-GPU box (tests/test_gpu_dist.py)
    one_gpu = os.environ.get('SCL_TRAIN_ONE_GPU_GLOO') == '1'
    if one_gpu:
        local_rank = 0
    # worker processes for the image files: spawned BEFORE anything touches the device
    flags.loader_pool = None
    if flags.shuffled_root and flags.loader_processes > 0:
        from . import dataset
        flags.loader_pool = dataset.make_loader_pool(flags.loader_processes)
    torch.cuda.set_device(local_rank)
    dev = torch.device('cuda', local_rank)
    if flags.force_dist and world != 1:
        raise SystemExit('--force_dist is the one-rank form of the collective path')
    group = parallel.init_process_group(dev, backend='gloo' if one_gpu else 'nccl',
                                        timeout_s=flags.dist_timeout, force_single=bool(flags.force_dist))

    np.random.seed(42)                                    # train/train.py:1463-1464
    tuple_shape = tuple_shape_for(flags.loss, flags.positives_per_tuple,
                                  flags.negatives_per_tuple)
    cdt = torch.bfloat16 if flags.dtype == 'bf16' else torch.float32
    model = nets.set_default_model(nets.VGG16NetVLAD(compute_dtype=cdt,
                                                     vlad_cores=flags.vlad_cores).to(dev))
    params = nets.trainable_parameters(model)       # train/train.py:606-611: the head decides
    buckets = parallel.GradBuckets(params, group, force_collectives=bool(flags.force_dist))
    nets.GRAD_SINK = buckets       # conv weight / bias gradients go straight into the flat buffer
    # train/train.py:865-870: MomentumOptimizer / AdamOptimizer — TF's Adam (epsilon outside the bias
    # correction), not torch's: train/optim.py
    opt = make_optimizer(flags.optimizer, params, flags.base_lr, flags.momentum)
    # restore_weights (:882-905) + the slot variables a tf.train.Saver checkpoint carries
    step = 0
    if flags.checkpoint:
        got = checkpoint.load(model, flags.checkpoint, optimizer=opt if flags.resume else None)
        step = got if flags.resume else 0
    out_dir = os.path.join(flags.out_root, flags.out_folder or flags.loss)
    saver = checkpoint.Saver(out_dir, flags.max_to_keep)
    data = SyntheticTuples(flags, tuple_shape, dev, rank, world)
    log = open(os.path.join(out_dir, 'train_log.txt'), 'a') if rank == 0 and (
        os.makedirs(out_dir, exist_ok=True) or True) else None

    # writers['local'] / writers['other'] (train/train.py:929-932): the training loss and learning
    # rate of every step and the training region's localisation numbers / the other region's loss and
    # localisation numbers, as TensorBoard event files (tf_events.py)
    boards = None
    if rank == 0 and flags.tensorboard:
        from .. import tf_events
        boards = {m: tf_events.SummaryWriter(os.path.join(out_dir, m)) for m in ('local', 'other')}

    def to_boards(rec):
        if boards is None:
            return
        if 'loss' in rec and 'event' not in rec:                                  # :304
            boards['local'].add_scalars({'loss': rec['loss'], 'learning_rate': rec['learning_rate']},
                                        rec['step'])
        elif rec.get('event') == 'eval':
            if rec.get('other_region_loss') is not None:                           # :1145-1147
                boards['other'].add_scalars({'loss': rec['other_region_loss']}, rec['step'])
            for mode in ('other', 'local'):                                        # :380-397
                vals = {k: v for k, v in (rec.get(mode) or {}).items() if isinstance(v, (int, float))}
                if vals:
                    boards[mode].add_scalars(vals, rec['step'])
            for b in boards.values():
                b.flush()

    if flags.synthetic_dataset > 0 or flags.shuffled_root:
        state = dict(model=model, opt=opt, buckets=buckets, saver=saver, dev=dev,
                     tuple_shape=tuple_shape, step=step, group=group, boards=boards)

        def write(rec):
            if log is not None:
                log.write(json.dumps(rec) + '\n')
                log.flush()
            to_boards(rec)
        try:
            for epoch in range(flags.max_epoch):
                train_dataset_epoch(flags, epoch, state, write)
        finally:
            nets.GRAD_SINK = None        # also after an exception: the sink outlives nothing
            for b in (boards or {}).values():
                b.close()
            if flags.loader_pool is not None:
                flags.loader_pool.shutdown()
        return state

    for epoch in range(flags.max_epoch):
        lr = get_learning_rate(epoch, flags)
        for g in opt.param_groups:
            g['lr'] = lr
        for _ in range(flags.steps):
            t0 = time.time()
            distances, img = data.batch()
            buckets.zero()
            output = nets.full_out(img)                           # ops['output'] (:606-629)
            loss = compute_loss(flags, tuple_shape, output, distances, group=group)
            loss.backward()
            buckets.finish()
            opt.step()
            step += 1
            if rank == 0:
                rec = {'step': step, 'epoch': epoch, 'loss': float(loss.detach()),
                       'learning_rate': lr, 'sec': round(time.time() - t0, 4)}
                print('Train batch loss: {}'.format(rec['loss']))   # :289
                log.write(json.dumps(rec) + '\n')
                log.flush()
                to_boards(rec)
                if step % flags.eval_step == 0:
                    saver.save_rolling(model, step, opt)              # :1079
                if step % flags.save_step == 0:
                    saver.save_part(model, step, opt)                 # :1102
        if rank == 0:
            saver.save_epoch(model, epoch, step, opt)                 # :984
    nets.GRAD_SINK = None
    for b in (boards or {}).values():
        b.close()
    if group is not None:
        dist.destroy_process_group()


if __name__ == '__main__':
    main()
