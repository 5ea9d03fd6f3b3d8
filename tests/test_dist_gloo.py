"""World-size-2 checks of the data-parallel layer on CPU (gloo).

The HIP loss cannot run here, so a differentiable torch stand-in plays the role of the
pairwise loss; what is under test is the communication contract of
``soft_contrastive_learning_amd.parallel``:
  * the autograd-aware all-gather (backward = own rows, no reduce-scatter),
  * "own rows only" loss gradients + SUM all-reduce == the single-process gradient of the
    global-batch loss,
  * bucketed flat-gradient all-reduce launched from post-accumulate hooks.
"""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

WORLD = 2
B_LOCAL, F_IN, E = 3, 5, 7


def _free_port():
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    port = s.getsockname()[1]
    s.close()
    return port


def _pair_loss(full):
    """Stand-in for the B x B pairwise loss: couples every row with every other row."""
    g = full @ full.T
    return (torch.tanh(g) * torch.arange(1, g.numel() + 1, dtype=g.dtype).reshape(g.shape)).mean()


def _make(seed=0):
    gen = torch.Generator().manual_seed(seed)
    x = torch.randn(WORLD * B_LOCAL, F_IN, generator=gen, dtype=torch.float64)
    w1 = torch.randn(F_IN, 6, generator=gen, dtype=torch.float64)
    w2 = torch.randn(6, E, generator=gen, dtype=torch.float64)
    return x, w1, w2


def _reference():
    x, w1, w2 = _make()
    w1.requires_grad_(True)
    w2.requires_grad_(True)
    loss = _pair_loss(torch.tanh(x @ w1) @ w2)
    loss.backward()
    return float(loss), w1.grad.clone(), w2.grad.clone()


class _DirectMatmul(torch.autograd.Function):
    """y = h @ w whose weight gradient goes straight into the gradient sink (what the HIP
    weight-gradient kernels do with nets.GRAD_SINK): autograd gets None for it."""

    @staticmethod
    def forward(ctx, h, w, sink):
        ctx.save_for_backward(h, w)
        ctx.sink = sink
        return h @ w

    @staticmethod
    def backward(ctx, gy):
        h, w = ctx.saved_tensors
        view = ctx.sink.view(w)
        assert view is not None and view.shape == w.shape
        view.copy_(h.t() @ gy)
        ctx.sink.done(w)
        return gy @ w.t(), None, None


def _worker(rank, port, out, bucket_bytes=8):
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    dist.init_process_group('gloo', rank=rank, world_size=WORLD)
    try:
        from soft_contrastive_learning_amd import parallel
        x, w1, w2 = _make()
        xl = x[rank * B_LOCAL:(rank + 1) * B_LOCAL]
        p1 = torch.nn.Parameter(w1.clone())
        p2 = torch.nn.Parameter(w2.clone())
        # tiny buckets: every parameter gets its own collective
        # tiny buckets: every parameter gets its own collective; one large bucket: its
        # all-reduce must wait for BOTH gradients, also when one arrives through the sink
        # (the engine still runs that parameter's post-accumulate hook afterwards)
        holder = {}
        if bucket_bytes != 8:
            # runs before GradBuckets' own hook: p1's gradient is the last to arrive, so the
            # shared bucket's collective must not have been launched yet
            p1.register_post_accumulate_grad_hook(
                lambda p: holder.__setitem__('early', len(holder['b']._handles)))
        buckets = parallel.GradBuckets([p1, p2], bucket_bytes=bucket_bytes)
        holder['b'] = buckets
        assert len(buckets.buckets) == (2 if bucket_bytes == 8 else 1)
        for it in range(3):                      # second pass checks zero()/re-arming, the
            buckets.zero()                       # third the direct gradient sink
            if it < 2:
                local = torch.tanh(xl @ p1) @ p2
            else:
                assert buckets.view(p2.detach().clone()) is None      # not a parameter's storage
                local = _DirectMatmul.apply(torch.tanh(xl @ p1), p2, buckets)
            full = parallel.all_gather_rows(local)
            assert full.shape == (WORLD * B_LOCAL, E)
            begin, count = parallel.local_rows(B_LOCAL)
            assert (begin, count) == (rank * B_LOCAL, B_LOCAL)

            def own_rows_only(g, begin=begin, count=count):
                m = torch.zeros_like(g)
                m[begin:begin + count] = g[begin:begin + count]
                return m
            full.register_hook(own_rows_only)    # what `_rows` does inside the HIP backward
            loss = _pair_loss(full)
            loss.backward()
            assert holder.get('early', 0) == 0, 'all-reduce launched before the bucket was complete'
            buckets.finish()
        out[rank] = (float(loss), p1.grad.clone(), p2.grad.clone())
    finally:
        dist.destroy_process_group()


@pytest.mark.timeout(300)
@pytest.mark.parametrize('bucket_bytes', [8, 1 << 20])
def test_dp_gradients_equal_single_process(bucket_bytes):
    want_loss, g1, g2 = _reference()
    mgr = mp.Manager()
    out = mgr.dict()
    mp.spawn(_worker, args=(_free_port(), out, bucket_bytes), nprocs=WORLD, join=True)
    assert sorted(out.keys()) == [0, 1]
    for rank in range(WORLD):
        loss, q1, q2 = out[rank]
        assert abs(loss - want_loss) < 1e-12           # same loss on every rank
        torch.testing.assert_close(q1, g1, rtol=1e-10, atol=1e-12)
        torch.testing.assert_close(q2, g2, rtol=1e-10, atol=1e-12)


def test_grad_buckets_layout_without_process_group():
    from soft_contrastive_learning_amd import parallel
    ps = [torch.nn.Parameter(torch.zeros(4, 3)), torch.nn.Parameter(torch.zeros(5)),
          torch.nn.Parameter(torch.zeros(2, 2))]
    gb = parallel.GradBuckets(ps, bucket_bytes=4 * 9)
    assert gb.flat.numel() == 21 and not gb.enabled
    # reverse registration order: [2x2 | 5] fill the first bucket (9 floats), [4x3] the next
    assert gb.buckets == [(0, 9), (9, 21)]
    (ps[0].sum() * 2 + ps[1].sum() * 3 + ps[2].sum() * 5).backward()
    assert gb.flat.tolist() == [5.0] * 4 + [3.0] * 5 + [2.0] * 12
    assert ps[0].grad.data_ptr() == gb.flat[9:].data_ptr()          # grads are views
    gb.zero()
    assert float(gb.flat.abs().sum()) == 0.0


def test_grad_buckets_refuse_a_second_backward_before_zero():
    """The gradient sink overwrites conv gradients and the bucket counters would go negative
    (no all-reduce on the second pass): a second report before zero() must fail loudly."""
    from soft_contrastive_learning_amd import parallel
    p = torch.nn.Parameter(torch.ones(3))
    gb = parallel.GradBuckets([p])
    (p.sum() * 2).backward()
    with pytest.raises(RuntimeError, match='reported twice'):
        (p.sum() * 2).backward()
    gb.zero()
    (p.sum() * 3).backward()                       # re-armed
    assert gb.flat.tolist() == [3.0] * 3


def _bruteforce_local(ref, query, n, offset):
    d2 = ((query[:, None, :].double() - ref[None, :, :].double()) ** 2).sum(-1)
    idx = torch.arange(ref.shape[0])[None].expand_as(d2)
    o = torch.argsort(d2, dim=1, stable=True)[:, :n]
    return torch.gather(d2, 1, o).sqrt(), torch.gather(idx, 1, o) + offset


def _retrieval_worker(rank, port, out):
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    dist.init_process_group('gloo', rank=rank, world_size=WORLD)
    try:
        from soft_contrastive_learning_amd import parallel
        gen = torch.Generator().manual_seed(5)
        ref = torch.randn(90, 16, generator=gen)
        ref[70] = ref[3]                         # an exact tie across the two shards
        qry = torch.cat([torch.randn(11, 16, generator=gen), ref[3:4]])
        per = 45
        d, i = parallel.topn_l2_sharded(ref[rank * per:(rank + 1) * per], qry, 7, rank * per,
                                        local_fn=_bruteforce_local)
        out[rank] = (d, i)
    finally:
        dist.destroy_process_group()


def test_sharded_retrieval_equals_single_process():
    """Reference set split over two ranks, queries replicated (SURVEY.md §8e retrieval row)."""
    port = _free_port()
    with mp.Manager() as m:
        out = m.dict()
        mp.spawn(_retrieval_worker, args=(port, out), nprocs=WORLD, join=True)
        res = dict(out)
    gen = torch.Generator().manual_seed(5)
    ref = torch.randn(90, 16, generator=gen)
    ref[70] = ref[3]
    qry = torch.cat([torch.randn(11, 16, generator=gen), ref[3:4]])
    want_d, want_i = _bruteforce_local(ref, qry, 7, 0)
    # ties: (distance, index) order, i.e. index 3 before its duplicate 70
    assert want_i[-1, :2].tolist() == [3, 70]
    for rank in range(WORLD):
        d, i = res[rank]
        assert torch.equal(i, want_i)
        assert torch.allclose(d, want_d, rtol=0, atol=0)


# ---- per-tuple losses shard by tuple ------------------------------------------
def _tuple_batch(seed=3):
    """Two tuples (one per rank) of [anchor, 3 positives, 3 negatives, other] in E = 16."""
    gen = torch.Generator().manual_seed(seed)
    x = torch.randn(WORLD, 8, 6, generator=gen, dtype=torch.float64)
    w = torch.randn(6, 16, generator=gen, dtype=torch.float64)
    return x, w


def _tuple_losses(out, kind):
    """The float64 autograd twins of the per-tuple losses on out [T, 8, E] (oracle = test infrastructure)."""
    from oracle import twin_torch as TT
    q, pos, neg, oth = out[:, :1], out[:, 1:4], out[:, 4:7], out[:, 7:]
    if kind == 'triplet':
        return TT.triplet_loss(q, pos, neg, 0.5)
    if kind == 'lazy_quadruplet':
        return TT.lazy_quadruplet_loss(q, pos, neg, oth, 0.5, 0.2)
    sp = torch.tensor([[1.0], [4.0], [9.0]], dtype=torch.float64)[None]
    sn = torch.tensor([[400.0], [900.0], [2500.0]], dtype=torch.float64)[None]
    # the log-ratio loss takes one tuple per call (the reference's broadcasting): the
    # mean over tuples of the per-tuple values is what a batch of T = 1 calls gives
    return torch.stack([TT.logratio_loss(q[k:k + 1], pos[k:k + 1], neg[k:k + 1], sp, sn)
                        for k in range(out.shape[0])]).mean()


def _tuple_worker(rank, port, out):
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    dist.init_process_group('gloo', rank=rank, world_size=WORLD)
    try:
        from soft_contrastive_learning_amd import parallel
        x, w = _tuple_batch()
        res = {}
        for kind in ('triplet', 'lazy_quadruplet', 'logratio'):
            p = torch.nn.Parameter(w.clone())
            buckets = parallel.GradBuckets([p])
            buckets.zero()
            local = _tuple_losses(torch.tanh(x[rank:rank + 1] @ p), kind)
            loss = parallel.tuple_loss_dp(local)
            loss.backward()
            buckets.finish()
            res[kind] = (float(loss), p.grad.clone())
        out[rank] = res
    finally:
        dist.destroy_process_group()


@pytest.mark.timeout(300)
def test_tuple_losses_shard_by_tuple():
    """parallel.tuple_loss_dp: the loss of a rank's own tuples -> the mean over all ranks' tuples,
    local gradients scaled by 1 / world and SUMMED by GradBuckets == the single-process loss and
    gradient on the concatenated batch, for triplet, lazy_quadruplet and logratio."""
    x, w = _tuple_batch()
    mgr = mp.Manager()
    out = mgr.dict()
    mp.spawn(_tuple_worker, args=(_free_port(), out), nprocs=WORLD, join=True)
    for kind in ('triplet', 'lazy_quadruplet', 'logratio'):
        p = w.clone().requires_grad_(True)
        want = _tuple_losses(torch.tanh(x @ p), kind)
        want.backward()
        for rank in range(WORLD):
            loss, grad = out[rank][kind]
            assert abs(loss - float(want)) < 1e-12, (kind, rank)
            torch.testing.assert_close(grad, p.grad, rtol=1e-10, atol=1e-12)


def _ragged_worker(rank, port, out):
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    dist.init_process_group('gloo', rank=rank, world_size=WORLD)
    try:
        from soft_contrastive_learning_amd import parallel
        rows = torch.arange(10, dtype=torch.float32).reshape(5, 2)
        share = rows[:3] if rank == 0 else rows[3:]
        out[rank] = parallel.all_gather_ragged(share)
        # a rank with no rows that does not know the row width (the mining share of a short list)
        out[10 + rank] = parallel.all_gather_ragged(rows if rank == 1 else torch.zeros((0, 0)))
        out[20 + rank] = parallel.all_gather_ragged(torch.zeros((0, 7))).shape
    finally:
        dist.destroy_process_group()


@pytest.mark.timeout(300)
def test_all_gather_ragged_concatenates_in_rank_order():
    mgr = mp.Manager()
    out = mgr.dict()
    mp.spawn(_ragged_worker, args=(_free_port(), out), nprocs=WORLD, join=True)
    for rank in range(WORLD):
        assert torch.equal(out[rank], torch.arange(10, dtype=torch.float32).reshape(5, 2))
        assert torch.equal(out[10 + rank], torch.arange(10, dtype=torch.float32).reshape(5, 2))
        assert tuple(out[20 + rank]) == (0, 7)


# ---- a missing or silent rank fails the job after the timeout instead of hanging it ---------------
_TIMEOUT_RANK = r"""
import os, sys, time
sys.path.insert(0, %r)
import torch
import torch.distributed as dist
from soft_contrastive_learning_amd import parallel
mode = sys.argv[1]
group = parallel.init_process_group(backend='gloo', timeout_s=4.0)
if mode == 'silent':              # joined, then never calls the collective
    time.sleep(120)
    sys.exit(0)
try:
    t = torch.ones(4)
    dist.all_reduce(t, group=group)
    print('all_reduce returned', flush=True)
except Exception:
    parallel.abort_rank(3)
"""


def _rank_proc(mode, rank, world, port):
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, RANK=str(rank), WORLD_SIZE=str(world), MASTER_ADDR='127.0.0.1',
               MASTER_PORT=str(port))
    return subprocess.Popen([sys.executable, '-c', _TIMEOUT_RANK % root, mode], env=env,
                            stdout=subprocess.PIPE, stderr=subprocess.STDOUT)


@pytest.mark.timeout(300)
def test_rendezvous_times_out_when_a_rank_never_arrives():
    """parallel.init_process_group(timeout_s=4) with WORLD_SIZE 2 and only rank 0 present: the
    process fails (non-zero) within the timeout's order of magnitude — it does not wait for the
    launcher's limit."""
    import time
    t0 = time.time()
    pr = _rank_proc('wait', 0, 2, _free_port())
    try:
        out, _ = pr.communicate(timeout=120)
    finally:
        if pr.poll() is None:
            pr.kill()
    assert pr.returncode not in (0, None), out.decode(errors='replace')[-800:]
    assert time.time() - t0 < 90


@pytest.mark.timeout(300)
def test_collective_times_out_and_the_rank_leaves_nonzero():
    """Rank 1 joins the group and then never calls the collective: rank 0's all-reduce raises after
    the group's timeout and parallel.abort_rank() ends the process with the given code, traceback
    printed, no destroy_process_group()."""
    import time
    port = _free_port()
    silent = _rank_proc('silent', 1, 2, port)
    active = _rank_proc('reduce', 0, 2, port)
    t0 = time.time()
    try:
        out, _ = active.communicate(timeout=150)
    finally:
        for pr in (active, silent):
            if pr.poll() is None:
                pr.kill()
        silent.communicate()
    text = out.decode(errors='replace')
    assert active.returncode == 3, text[-1500:]
    assert 'Traceback' in text and 'all_reduce returned' not in text
    assert time.time() - t0 < 120
