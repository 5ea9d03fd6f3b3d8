"""Data-parallel step for the soft-contrastive path (new work: the reference is
single-process, single-GPU — SURVEY.md F2 / §8e).

One process per GPU, ``torch.distributed`` (backend "nccl" = RCCL over xGMI on the GPU
box, "gloo" in the CPU tests).  The batch shards by image:

  1. each rank embeds its ``b`` images -> [b, 32768] unit-norm descriptors;
  2. ONE all-gather makes the full [B, 32768] matrix on every rank (3.1 MB per rank at
     b=24), so every rank forms the same B x B pairwise matrix and the same loss;
  3. every rank asks the loss backward for ITS OWN rows only (``_rows``): because
     d loss / d E_i needs both row and column terms of the pair matrix, replicating the
     B x B work is cheaper than a second exchange, and the all-gather's backward is then
     a plain slice — no reduce-scatter;
  4. parameter gradients are SUMMED over ranks (each rank back-propagated the same loss
     through its own images only), in buckets launched as soon as their gradients are
     final so the collective overlaps the rest of the VGG backward.
"""
import datetime
import os
import sys

import torch
import torch.distributed as dist


def init_process_group(dev=None, backend='nccl', timeout_s=None, force_single=False):
    """Join the job's process group (RANK / WORLD_SIZE / MASTER_* from the launcher's environment)
    and return it, or None for a single process that was not asked to form one.

    * ``backend`` "nccl" IS RCCL on this platform; the communicator is bound to ``dev`` at
      creation (``device_id``), so the first collective does not have to guess the device.
    * ``timeout_s`` (default ``SCL_DIST_TIMEOUT_S`` or 300): rendezvous AND every collective.  A
      rank that never arrives makes the others fail after that long (the watchdog aborts the
      communicator and the process exits non-zero) instead of sitting in a kernel until the
      launcher's own limit.
    * ``force_single``: with WORLD_SIZE 1 still create a one-rank group, so that the whole
      collective path (communicator creation, the collectives' own stream and events, the
      asynchronous work handles of GradBuckets) runs on a one-GPU box."""
    world = int(os.environ.get('WORLD_SIZE', '1'))
    if world == 1 and not force_single:
        return None
    if timeout_s is None:
        timeout_s = float(os.environ.get('SCL_DIST_TIMEOUT_S', '300'))
    os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
    if world == 1:
        os.environ.setdefault('MASTER_PORT', str(_free_port()))
        os.environ.setdefault('RANK', '0')
        os.environ.setdefault('WORLD_SIZE', '1')
    kw = dict(timeout=datetime.timedelta(seconds=timeout_s))
    if backend == 'nccl':
        kw['device_id'] = dev
    dist.init_process_group(backend, **kw)
    return dist.group.WORLD


def _free_port():
    import socket
    with socket.socket() as sk:
        sk.bind(('127.0.0.1', 0))
        return sk.getsockname()[1]


def abort_rank(code=1):
    """A rank failed: print the traceback of the exception being handled and leave at once.  No
    destroy_process_group() — with RCCL that can block behind this rank's own outstanding
    collectives while the peers wait in theirs, which is the hang this exists to avoid; the peers
    see the closed connection (or their collective timeout) and fail too."""
    import traceback
    traceback.print_exc()
    sys.stderr.flush()
    sys.stdout.flush()
    os._exit(code)


# launch a bucket's all-reduce from the weight-gradient stream (GradBuckets._launch); 0: join that
# stream into the compute stream first, as rounds 3-5 did
ON_SIDE_STREAM = os.environ.get('SCL_BUCKETS_ON_SIDE', '1') != '0'

# bench.py sets this to {'allgather': [], 'finish': []} for its diagnostic steps: pairs of device
# events on the compute stream around the embedding all-gather and around the waits of
# GradBuckets.finish() (how long the step sat behind the last gradient all-reduce).  None: no
# events are recorded.
COMM_LOG = None


def _comm_events(kind):
    log = COMM_LOG
    if log is None or not torch.cuda.is_available():
        return None
    pair = (torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True))
    log[kind].append(pair)
    pair[0].record()
    return pair


class _AllGatherRows(torch.autograd.Function):
    """[b,E] per rank -> [world*b,E] on every rank; backward = own rows of the gradient."""

    @staticmethod
    def forward(ctx, local, group):
        world = dist.get_world_size(group)
        ctx.rank = dist.get_rank(group)
        ctx.b = local.shape[0]
        local = local.contiguous()
        full = torch.empty((world * local.shape[0],) + tuple(local.shape[1:]),
                           dtype=local.dtype, device=local.device)
        ev = _comm_events('allgather') if local.is_cuda else None
        dist.all_gather_into_tensor(full, local, group=group)
        if ev is not None:
            ev[1].record()
        return full

    @staticmethod
    def backward(ctx, grad_full):
        return grad_full[ctx.rank * ctx.b:(ctx.rank + 1) * ctx.b].contiguous(), None


def all_gather_rows(local, group=None):
    return _AllGatherRows.apply(local, group)


def local_rows(local_batch, group=None):
    """(row_begin, row_count) of this rank's images inside the gathered batch."""
    return dist.get_rank(group) * local_batch, local_batch


def wms_loss_dp(distances, local_embeddings, d_alpha, d_beta, group=None, **kw):
    """``model.losses.wms_loss`` on the global batch: ``distances`` is the replicated
    [1,B,B] / [B,B] matrix of ALL images, ``local_embeddings`` this rank's [b,E]."""
    from .model import losses
    full = all_gather_rows(local_embeddings, group)
    rows = local_rows(local_embeddings.shape[0], group)
    return losses.wms_loss(distances, full, d_alpha, d_beta, _rows=rows, **kw)


def ms_loss_dp(labels, local_embeddings, group=None, **kw):
    """``model.losses.ms_loss`` on the global batch; ``labels`` must be globally unique
    across ranks."""
    from .model import losses
    full = all_gather_rows(local_embeddings, group)
    rows = local_rows(local_embeddings.shape[0], group)
    return losses.ms_loss(labels, full, _rows=rows, **kw)


class _MeanOverRanks(torch.autograd.Function):
    """Scalar mean over this rank's tuples -> mean over the tuples of ALL ranks (every rank holds
    the same number of tuples); backward: d global / d local = 1 / world on every rank."""

    @staticmethod
    def forward(ctx, local, group):
        ctx.world = dist.get_world_size(group)
        total = local.detach().clone()
        dist.all_reduce(total, op=dist.ReduceOp.SUM, group=group)
        return total / ctx.world

    @staticmethod
    def backward(ctx, grad):
        return grad / ctx.world, None


def tuple_loss_dp(local_loss, group=None):
    """The per-tuple losses (triplet / quadruplet families, log-ratio, the distance-term losses:
    a mean over tuples of terms that each involve one tuple only) shard BY TUPLE (SURVEY section
    8e): every rank evaluates the loss of its own tuples with the HIP kernels, this makes the scalar
    the mean over all ranks' tuples — the single-process loss on the concatenated batch — and
    scales the local gradient by 1 / world; ``GradBuckets`` then SUMS the parameter gradients.
    No embedding travels: the only collectives are this scalar and the gradient buckets."""
    return _MeanOverRanks.apply(local_loss, group)


def all_gather_ragged(local, group=None):
    """Rows of every rank, concatenated in rank order, for row counts that may differ by rank
    (mining-cache descriptors, image indices): padded to the longest, gathered, trimmed.  A rank
    with NO rows need not know the row width (a [0, anything] tensor will do): widths travel with
    the counts and an empty rank adopts the others'."""
    world = dist.get_world_size(group)
    width = 1
    for d in local.shape[1:]:
        width *= int(d)
    n = torch.tensor([local.shape[0], width if local.shape[0] else 0], dtype=torch.int64,
                     device=local.device)
    counts = [torch.zeros_like(n) for _ in range(world)]
    dist.all_gather(counts, n, group=group)
    widths = {int(c[1]) for c in counts if int(c[0])}
    counts = [int(c[0]) for c in counts]
    if len(widths) > 1:
        raise ValueError('all_gather_ragged: ranks disagree on the row width: %s' % sorted(widths))
    if not widths:                                  # nobody has rows
        return local[:0]
    if local.shape[0] == 0 and width != next(iter(widths)):
        if local.dim() != 2:
            raise ValueError('all_gather_ragged: an empty rank of rank-%d rows must match the row '
                             'shape of the others' % local.dim())
        local = local.new_zeros((0, next(iter(widths))))
    longest = max(counts)
    pad = torch.zeros((longest,) + tuple(local.shape[1:]), dtype=local.dtype, device=local.device)
    pad[:local.shape[0]] = local
    parts = [torch.empty_like(pad) for _ in range(world)]
    dist.all_gather(parts, pad, group=group)
    return torch.cat([p[:c] for p, c in zip(parts, counts)], 0)


def topn_l2_sharded(ref_shard, query, n, shard_offset, group=None, score='f32', local_fn=None,
                    force_exchange=False):
    """Retrieval with the reference set sharded over ranks: every rank scans
    its own rows ``[shard_offset, shard_offset + len(ref_shard))`` for all (replicated)
    queries, the [Q, n] (distance, index) candidates are all-gathered (Q * n * 16 bytes per
    rank) and merged by (distance, index) — the same lists on every rank as the single-device
    call on the concatenated reference set.  ``local_fn`` replaces the HIP kernel in the
    CPU tests of the exchange; ``force_exchange`` runs the all-gather + merge in a one-rank group."""
    from .evaluation import retrieval
    if local_fn is None:
        d, i = retrieval.topn_l2(ref_shard, query, n, idx_offset=shard_offset, score=score)
    else:
        d, i = local_fn(ref_shard, query, n, shard_offset)
    world = dist.get_world_size(group) if dist.is_available() and dist.is_initialized() else 1
    if world == 1 and not (force_exchange and dist.is_initialized()):
        return d, i
    ds = [torch.empty_like(d) for _ in range(world)]
    idx = [torch.empty_like(i) for _ in range(world)]
    dist.all_gather(ds, d.contiguous(), group=group)
    dist.all_gather(idx, i.contiguous(), group=group)
    return retrieval.merge_topn(ds, idx, n)


class GradBuckets:
    """Flat gradient storage + bucketed asynchronous all-reduce (SUM).

    Every parameter's ``.grad`` is a view into one flat buffer, so a bucket is a
    contiguous slice and needs no packing.  Buckets follow reverse registration order
    (≈ the order gradients become final in backward); a bucket's collective is launched
    from the post-accumulate hook of its last parameter.  14.78 M f32 parameters = 59 MB:
    the default 16 MB buckets give 4 collectives, each large enough to run at link rate
    on the point-to-point xGMI fabric and small enough to hide under the conv backward.
    """

    def __init__(self, params, group=None, bucket_bytes=16 << 20, force_collectives=False):
        """``force_collectives``: launch the bucket all-reduces in a ONE-rank group too (a sum over
        one rank: the gradients come back unchanged) — the asynchronous path on a one-GPU box."""
        self.params = [p for p in params if p.requires_grad]
        self.group = group
        total = sum(p.numel() for p in self.params)
        ref = self.params[0]
        self.flat = torch.zeros(total, dtype=ref.dtype, device=ref.device)
        self.buckets = []          # (begin, end) element ranges
        self._bucket_of = {}
        self._pending = []
        self._handles = []
        order = list(reversed(self.params))
        off = 0
        begin, count = 0, 0
        members = []
        for p in order:
            n = p.numel()
            p.grad = self.flat[off:off + n].view_as(p)
            members.append(p)
            off += n
            count += n
            if count * self.flat.element_size() >= bucket_bytes:
                self._close(begin, off, members)
                begin, count, members = off, 0, []
        if members:
            self._close(begin, off, members)
        self.enabled = dist.is_available() and dist.is_initialized() and \
            (dist.get_world_size(group) > 1 or force_collectives)
        self._hook_handles = [p.register_post_accumulate_grad_hook(self._hook) for p in self.params]
        self._remaining = [len(m) for m in self._members]
        self._reported = {}
        # direct gradient sink (nets.GRAD_SINK): a backward kernel may write a parameter's
        # gradient straight into its slice of the flat buffer — no temporary, no cast, no
        # `grad += new` pass — and report it with done(); the bucket logic is the same as for
        # gradients that arrive through autograd
        self._by_ptr = {p.data_ptr(): p for p in self.params}
        self._streams = []         # side streams that write into the flat buffer (note_stream)
        self._keep = []            # what those streams still read: released once they are joined

    def _close(self, begin, end, members):
        if not hasattr(self, '_members'):
            self._members = []
        idx = len(self.buckets)
        self.buckets.append((begin, end))
        self._members.append(list(members))
        for p in members:
            self._bucket_of[id(p)] = idx

    def _hook(self, p):
        """Post-accumulate hook (the autograd path)."""
        self._report(p, False)

    def _report(self, p, via_sink):
        # The engine runs the post-accumulate hooks of a parameter even when its Function
        # returned None for it (the sink path, already counted by done()): that follow-up call
        # is ignored.  Anything else arriving twice before zero() is a second backward pass —
        # the gradient sink OVERWRITES conv gradients and no further all-reduce would fire, so
        # refuse instead of silently dropping the first pass / letting the ranks diverge.
        seen = self._reported.get(id(p))
        if seen is not None:
            if seen == 'sink' and not via_sink:
                self._reported[id(p)] = 'sink+hook'
                return
            raise RuntimeError('GradBuckets: a parameter gradient was reported twice before '
                               'zero(); call zero() before every backward pass (gradient '
                               'accumulation needs nets.GRAD_SINK = None)')
        self._reported[id(p)] = 'sink' if via_sink else 'hook'
        if not self.enabled:
            return
        idx = self._bucket_of[id(p)]
        self._remaining[idx] -= 1
        if self._remaining[idx] == 0:
            b, e = self.buckets[idx]
            self._handles.append(self._launch(self.flat[b:e]))

    def _launch(self, part):
        """Start the all-reduce of one bucket.  The collective orders itself after the stream that
        is current when it is called.  With weight gradients on a side stream that stream is made
        current for the call (after it has been ordered behind the compute stream's position,
        which every weight-gradient launch does anyway): the collective then waits for exactly the
        kernels that write the bucket, and the COMPUTE stream never waits for the side stream at a
        bucket boundary — it only meets the collectives again in finish().  Joining the side stream
        into the compute stream here instead (rounds 3-5) stalled the backward-data chain behind the
        weight-gradient kernels four times per step."""
        if not self._streams or not ON_SIDE_STREAM:
            self._join_streams()
            return dist.all_reduce(part, op=dist.ReduceOp.SUM, group=self.group, async_op=True)
        side = self._streams[0]
        side.wait_stream(torch.cuda.current_stream(self.flat.device))
        for other in self._streams[1:]:
            side.wait_stream(other)
        with torch.cuda.stream(side):
            return dist.all_reduce(part, op=dist.ReduceOp.SUM, group=self.group, async_op=True)

    def note_stream(self, stream, *inputs):
        """A kernel on `stream` (not the current one) writes a gradient into the flat buffer:
        the stream is joined before a collective reads the buffer and in finish().  `inputs`
        (tensors that kernel reads) are kept alive until finish() has joined the stream, so that
        the caching allocator cannot hand their memory to the current stream early."""
        if all(s is not stream for s in self._streams):
            self._streams.append(stream)
        self._keep.extend(inputs)

    def _join_streams(self):
        for s in self._streams:
            torch.cuda.current_stream(self.flat.device).wait_stream(s)

    def view(self, p_like):
        """The flat-buffer slice of the parameter that owns p_like's storage, shaped like it
        (None if p_like is not one of the parameters, e.g. a cast copy)."""
        p = self._by_ptr.get(p_like.data_ptr())
        if p is None or p.grad is None or p.shape != p_like.shape or p.dtype != self.flat.dtype:
            return None
        return p.grad

    def done(self, p_like):
        """The gradient of that parameter is final in the flat buffer (written by a kernel
        enqueued on the current stream): what the post-accumulate hook would have done."""
        self._report(self._by_ptr[p_like.data_ptr()], True)

    def zero(self):
        """Zero all gradients in one memset and re-arm the buckets."""
        # If finish() was skipped (an exception, a dropped step after backward) two things may
        # still be touching the flat buffer: all-reduces launched from the abandoned backward pass
        # (their late result would land in the next step's gradients) and side-stream kernels
        # writing weight gradients.  Wait for both before the memset.
        for h in self._handles:
            h.wait()
        self._handles = []
        self._join_streams()
        self._keep = []
        self.flat.zero_()
        self._remaining = [len(m) for m in self._members]
        self._reported = {}

    def close(self):
        """Detach from the parameters (hooks removed, .grad views dropped): a second GradBuckets
        over the same parameters must not find this one still reporting."""
        self.zero()
        for h in self._hook_handles:
            h.remove()
        self._hook_handles = []
        for p in self.params:
            p.grad = None

    def finish(self):
        """Wait for the collectives launched during backward (call before optimizer.step)."""
        ev = _comm_events('finish') if (self._handles and self.flat.is_cuda) else None
        for h in self._handles:
            h.wait()
        if ev is not None:
            ev[1].record()
        self._handles = []
        self._join_streams()
        self._keep = []
