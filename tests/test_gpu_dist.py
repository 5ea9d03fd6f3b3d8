"""Two ranks on ONE GPU (gloo carries the collective; both ranks compute on cuda:0): the only
way to run the data-parallel gradient path with real asynchronous HIP kernels on a one-GPU box.

What is under test (parallel.GradBuckets + nets.GRAD_SINK + nets.USE_SIDE_WRW):
the weight-gradient kernels write into the flat gradient buffer from a second stream while the
bucket logic launches the all-reduce of a bucket as soon as its last parameter is reported —
the collective must see finished gradients.  The all-reduced buffer of each rank has to equal,
bit for bit, the sum of the two ranks' single-process gradients (a + b of two floats has one
result), with small buckets (many collectives in flight during backward) and with one bucket.
"""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

pytestmark = pytest.mark.gpu


def _free_port():
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    port = s.getsockname()[1]
    s.close()
    return port


def _images(rank):
    g = torch.Generator().manual_seed(200 + rank)
    return torch.randint(0, 256, (1, 480, 640, 3), generator=g).float()


def _backbone_params(model):
    """Only features() runs here: the NetVLAD parameters get no gradient and would keep their
    bucket from ever completing."""
    return [p for n, p in model.named_parameters()
            if not n.startswith(('assignment', 'cluster'))]


def _local_grads(rank, dev):
    """Flat gradient buffer of one rank's batch, no process group."""
    from soft_contrastive_learning_amd import parallel
    from soft_contrastive_learning_amd.model import nets
    model = nets.VGG16NetVLAD(compute_dtype=torch.bfloat16, seed=21, fused_relu=True).to(dev)
    buckets = parallel.GradBuckets(_backbone_params(model))
    buckets.enabled = False                # no collective: this rank's own gradients only
    g = torch.randn(1, 30, 40, 512, generator=torch.Generator().manual_seed(300 + rank)).to(dev).bfloat16()
    nets.GRAD_SINK = buckets
    try:
        buckets.zero()
        model.features(_images(rank).to(dev)).backward(g)
        buckets.finish()
    finally:
        nets.GRAD_SINK = None
    torch.cuda.synchronize()
    return buckets.flat.clone()


def _worker(rank, world, port, bucket_bytes, out):
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port))
    import faulthandler
    faulthandler.dump_traceback_later(180, exit=False)      # a hung rank says where
    dist.init_process_group('gloo', rank=rank, world_size=world)
    try:
        from soft_contrastive_learning_amd import parallel
        from soft_contrastive_learning_amd.model import nets
        dev = torch.device('cuda:0')
        want = _local_grads(0, dev) + _local_grads(1, dev)
        model = nets.VGG16NetVLAD(compute_dtype=torch.bfloat16, seed=21, fused_relu=True).to(dev)
        buckets = parallel.GradBuckets(_backbone_params(model), bucket_bytes=bucket_bytes)
        assert buckets.enabled and nets.USE_SIDE_WRW
        g = torch.randn(1, 30, 40, 512, generator=torch.Generator().manual_seed(300 + rank)).to(dev).bfloat16()
        img = _images(rank).to(dev)
        nets.GRAD_SINK = buckets
        try:
            for _ in range(3):                      # stale events, buffer reuse across steps
                buckets.zero()
                model.features(img).backward(g)
                buckets.finish()
                torch.cuda.synchronize()
                same = bool(torch.equal(buckets.flat, want))
                worst = float((buckets.flat - want).abs().max())
                if not same:
                    break
        finally:
            nets.GRAD_SINK = None
        out.put((rank, same, worst, len(buckets.buckets), len(buckets._streams)))
    finally:
        faulthandler.cancel_dump_traceback_later()
        dist.destroy_process_group()


def _run_ranks(procs, limit=240):
    """Start the rank processes and wait; a rank that hangs or fails is KILLED before the assert
    (a live non-daemon child would otherwise keep pytest itself from exiting)."""
    for p in procs:
        p.start()
    try:
        for p in procs:
            p.join(timeout=limit)
        codes = [p.exitcode for p in procs]
    finally:
        for p in procs:
            if p.is_alive():
                p.kill()
                p.join(timeout=10)
    assert codes == [0] * len(procs), 'rank exit codes %s (None = hung, killed)' % (codes,)


@pytest.mark.parametrize('bucket_bytes', [1 << 20, 1 << 30])
def test_two_ranks_on_one_gpu_all_reduce_finished_gradients(bucket_bytes):
    if not torch.cuda.is_available():
        pytest.skip('needs a HIP device')
    ctx = mp.get_context('spawn')
    out = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, bucket_bytes, out)) for r in range(2)]
    _run_ranks(procs)
    res = sorted(out.get(timeout=10) for _ in range(2))
    for rank, same, worst, nb, ns in res:
        assert ns == 1, 'the weight gradients did not run on the second stream'
        assert nb >= (8 if bucket_bytes == 1 << 20 else 1)
        assert same, 'rank %d: all-reduced gradients differ from the sum of the ranks by %g' % (rank, worst)


def _loss_worker(rank, world, port, out):
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port))
    dist.init_process_group('gloo', rank=rank, world_size=world)
    try:
        import numpy as np
        from soft_contrastive_learning_amd import parallel
        from soft_contrastive_learning_amd.evaluation import retrieval
        from soft_contrastive_learning_amd.model import losses
        dev = torch.device('cuda:0')
        b, e = (12 if world == 2 else 24), 32768      # world 8: configs[3], global batch 192
        g = torch.Generator().manual_seed(400)
        emb_all = torch.randn(world * b, e, generator=g)
        emb_all = (emb_all / emb_all.norm(dim=1, keepdim=True)).to(dev)
        xy = torch.rand(world * b, 2, generator=g) * 60.0
        dmat = (xy[:, None] - xy[None]).norm(dim=2)[None].to(dev)
        # single-process loss and gradient on the whole batch
        ref = emb_all.clone().requires_grad_(True)
        loss_ref = losses.wms_loss(dmat, ref, d_alpha=0.8, d_beta=15.0)
        loss_ref.backward()
        # this rank's rows through the data-parallel wrapper (all-gather + own-rows backward)
        mine = emb_all[rank * b:(rank + 1) * b].clone().requires_grad_(True)
        loss = parallel.wms_loss_dp(dmat, mine, 0.8, 15.0)
        loss.backward()
        torch.cuda.synchronize()
        want = ref.grad[rank * b:(rank + 1) * b]
        gerr = float((mine.grad - want).abs().max() / want.abs().max())
        lerr = abs(float(loss) - float(loss_ref)) / abs(float(loss_ref))
        # sharded retrieval: each rank scans half of the references, lists merged over ranks
        refs = torch.randn(4096, 256, generator=g).to(dev)
        qry = torch.randn(64, 256, generator=g).to(dev)
        half = refs.shape[0] // world
        d_sh, i_sh = parallel.topn_l2_sharded(refs[rank * half:(rank + 1) * half], qry, 25, rank * half)
        d_all, i_all = retrieval.topn_l2(refs, qry, 25)
        out.put((rank, lerr, gerr, bool(torch.equal(i_sh.cpu(), i_all.cpu())),
                 float((d_sh.cpu() - d_all.cpu()).abs().max())))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize('world', [2, 8])
def test_ranks_on_one_gpu_loss_and_sharded_retrieval(world):
    """parallel.wms_loss_dp (autograd all-gather, the HIP Gram loss on the gathered batch, backward
    for the rank's own rows) against the single-process loss on the same descriptors: equal loss
    (<= 1e-6 relative: same kernels, same inputs) and equal gradients for the own rows — with 2
    ranks of 12 and with 8 ranks of 24 (BASELINE.json configs[3]: global batch 192); and
    parallel.topn_l2_sharded against retrieval.topn_l2 on the unsharded references: the same index
    lists."""
    if not torch.cuda.is_available():
        pytest.skip('needs a HIP device')
    ctx = mp.get_context('spawn')
    out = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_loss_worker, args=(r, world, port, out)) for r in range(world)]
    _run_ranks(procs)
    for rank, lerr, gerr, same_idx, derr in sorted(out.get(timeout=10) for _ in range(world)):
        assert lerr <= 1e-6, (rank, lerr)
        assert gerr <= 1e-6, (rank, gerr)
        assert same_idx and derr <= 1e-9, (rank, derr)


def test_bench_two_ranks_on_one_gpu():
    """`python bench.py --gpus 2` end to end on a one-GPU box (SCL_BENCH_ONE_GPU_GLOO=1: both
    ranks on cuda:0, gloo for the collectives): self-launch, the data-parallel step (all-gather of
    the embeddings, loss on the global batch of 48, bucketed all-reduce against the second
    stream), max-over-ranks timing, one JSON line from rank 0.  The timing itself means nothing."""
    import json
    import subprocess
    import sys
    if not torch.cuda.is_available():
        pytest.skip('needs a HIP device')
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, SCL_BENCH_ONE_GPU_GLOO='1')
    for k in ('RANK', 'LOCAL_RANK', 'WORLD_SIZE', 'MASTER_ADDR', 'MASTER_PORT'):
        env.pop(k, None)
    r = subprocess.run([sys.executable, os.path.join(root, 'bench.py'), '--gpus', '2', '--steps', '2',
                        '--warmup', '1', '--no-cpu-baseline'], env=env, capture_output=True,
                       text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith('{')]
    assert len(lines) == 1
    d = json.loads(lines[0])
    assert d['n_gpus'] == 2 and d['config']['global_batch'] == 48 and d['scaling'] == 'weak'
    assert d['value'] > 0 and 0.0 < d['config']['loss'] < 100.0
    # the line explains its own communication
    c = d['comm']
    assert c['world_seen'] == 2 and c['backend'] == 'gloo'
    assert c['allgather_us_median'] > 0 and c['finish_wait_us_median'] is not None
    assert c['allgather_bytes_per_rank'] == 24 * 32768 * 4 and c['allreduce_buckets'] >= 3
    assert set(c['ms_per_step_by_reserved_cus']) == {'0', '8'}
    assert c['reserved_cus_warmup_choice']['chosen'] in (0, 8)
    assert c['reserved_cus_in_timed_region'] == c['reserved_cus_warmup_choice']['chosen']
    assert all(v > 0 for v in c['ms_per_step_by_reserved_cus'].values())
    assert d['switches'].get('SCL_BENCH_ONE_GPU_GLOO') == '1'
    assert 'retrieval' not in d and 'cpu_baseline' not in d          # N = 1 objects


def test_bench_eight_ranks_on_one_gpu_reduced_images():
    """`python bench.py --gpus 8` — the driver's last scaling point, configs[3]'s world size — kept
    warm on a one-GPU box: eight self-launched ranks on cuda:0 over gloo, 4 images of 128 x 160 per
    rank (global batch 32: the B <= 32 one-launch loss on the gathered batch, own-rows backward,
    bucketed all-reduce over 8 ranks), so that the first real 8-GPU run cannot fail on plumbing."""
    import json
    import subprocess
    import sys
    if not torch.cuda.is_available():
        pytest.skip('needs a HIP device')
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, SCL_BENCH_ONE_GPU_GLOO='1')
    for k in ('RANK', 'LOCAL_RANK', 'WORLD_SIZE', 'MASTER_ADDR', 'MASTER_PORT'):
        env.pop(k, None)
    r = subprocess.run([sys.executable, os.path.join(root, 'bench.py'), '--gpus', '8', '--steps', '2',
                        '--warmup', '1', '--batch', '4', '--height', '128', '--width', '160',
                        '--no-cpu-baseline'], env=env, capture_output=True, text=True, timeout=1500)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith('{')]
    assert len(lines) == 1
    d = json.loads(lines[0])
    assert d['n_gpus'] == 8 and d['config']['global_batch'] == 32 and d['scaling'] == 'weak'
    assert d['value'] > 0 and 0.0 < d['config']['loss'] < 100.0
    c = d['comm']
    assert c['world_seen'] == 8 and c['backend'] == 'gloo'
    assert c['allgather_bytes_per_rank'] == 4 * 32768 * 4 and c['allreduce_buckets'] >= 3


def test_bench_retrieval_workload_two_ranks_on_one_gpu():
    """`python bench.py --workload retrieval --gpus 2` (configs[4], reference set sharded over
    the ranks) end to end on a one-GPU box, and the same workload at --gpus 1: same index lists
    (checksum), queries/s reported, exchange time separated from the local scan."""
    import json
    import subprocess
    import sys
    if not torch.cuda.is_available():
        pytest.skip('needs a HIP device')
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, SCL_BENCH_ONE_GPU_GLOO='1')
    for k in ('RANK', 'LOCAL_RANK', 'WORLD_SIZE', 'MASTER_ADDR', 'MASTER_PORT'):
        env.pop(k, None)
    recs = {}
    for gpus in (1, 2):
        r = subprocess.run([sys.executable, os.path.join(root, 'bench.py'), '--workload', 'retrieval',
                            '--gpus', str(gpus), '--steps', '2', '--warmup', '1', '--refs', '20000',
                            '--queries', '1000', '--n1-ref', '1000.0'], env=env, capture_output=True,
                           text=True, timeout=900)
        assert r.returncode == 0, r.stderr[-2000:]
        lines = [l for l in r.stdout.splitlines() if l.startswith('{')]
        assert len(lines) == 1
        recs[gpus] = json.loads(lines[0])
    for gpus, d in recs.items():
        assert d['n_gpus'] == gpus and d['unit'] == 'queries/sec' and d['scaling'] == 'strong'
        assert d['value'] > 0 and d['world_seen'] == gpus and 'scaling_efficiency' in d
        assert d['config']['refs'] == 20000 and d['config']['parallelism'] == 'ref-shard%d' % gpus
    assert recs[1]['checksum_idx'] == recs[2]['checksum_idx']


# ---- per-tuple losses and the dataset route with several ranks ---------------
def _tuple_dp_worker(rank, world, port, out):
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port))
    dist.init_process_group('gloo', rank=rank, world_size=world)
    try:
        import argparse
        from soft_contrastive_learning_amd.train import train as T
        dev = torch.device('cuda:0')
        g = torch.Generator().manual_seed(77)
        res = []
        for loss_name in ('triplet', 'lazy_quadruplet', 'logratio', 'huber_distance_triplet'):
            flags = T.make_parser().parse_args(['--loss', loss_name, '--positives_per_tuple', '3',
                                                '--negatives_per_tuple', '3', '--tuples_per_batch', '1'])
            shape = T.tuple_shape_for(loss_name, 3, 3)
            s = sum(shape)
            emb_all = torch.randn(world * s, 4096, generator=g)
            emb_all = (emb_all / emb_all.norm(dim=1, keepdim=True)).to(dev)
            if T.distance_type(loss_name) == 'logratio':
                pay = torch.rand(world, 6, generator=g) * 50.0 + 1.0
            elif T.distance_type(loss_name) == 'anchor':
                pay = torch.rand(world, 3, generator=g) * 200.0
            else:
                pay = None
            # single process on the concatenated batch (T = world tuples); the log-ratio loss takes
            # one tuple per call, so its single-process value is the mean of the calls
            ref = emb_all.clone().requires_grad_(True)
            if loss_name == 'logratio':
                parts = [T.compute_loss(flags, shape, ref[k * s:(k + 1) * s], pay[k:k + 1].to(dev))
                         for k in range(world)]
                want = torch.stack(parts).mean()
            else:
                fl2 = argparse.Namespace(**dict(vars(flags), tuples_per_batch=world))
                want = T.compute_loss(fl2, shape, ref, None if pay is None else pay.to(dev))
            want.backward()
            mine = emb_all[rank * s:(rank + 1) * s].clone().requires_grad_(True)
            got = T.compute_loss(flags, shape, mine, None if pay is None else pay[rank:rank + 1].to(dev),
                                 group=dist.group.WORLD)
            got.backward()
            torch.cuda.synchronize()
            wg = ref.grad[rank * s:(rank + 1) * s]
            res.append((loss_name, abs(float(got) - float(want)) / max(abs(float(want)), 1e-30),
                        float((mine.grad - wg).abs().max() / wg.abs().max().clamp_min(1e-30))))
        out.put((rank, res))
    finally:
        dist.destroy_process_group()


def test_tuple_losses_two_ranks_on_one_gpu():
    """train.compute_loss with a process group for the per-tuple losses (HIP kernels on the rank's own
    tuple, parallel.tuple_loss_dp for the mean over ranks) against the single-process loss on the
    concatenated batch: same value, and own-tuple gradients equal to the single-process rows."""
    if not torch.cuda.is_available():
        pytest.skip('needs a HIP device')
    ctx = mp.get_context('spawn')
    out = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_tuple_dp_worker, args=(r, 2, port, out)) for r in range(2)]
    _run_ranks(procs)
    for rank, res in sorted(out.get(timeout=10) for _ in range(2)):
        for name, lerr, gerr in res:
            assert lerr <= 1e-6, (rank, name, lerr)
            assert gerr <= 1e-5, (rank, name, gerr)


@pytest.mark.parametrize('loss_name', ['wms', 'lazy_quadruplet'])
def test_dataset_route_two_ranks_on_one_gpu(loss_name, tmp_path):
    """`train.py --synthetic_dataset` with two ranks (both on cuda:0, gloo): sampler -> pipeline ->
    step with the gathered batch (wms: image indices exchanged, distances from the shared poses) or the
    tuple-sharded loss, sharded mining-cache refresh, evaluation on every rank, rank 0 writing the log
    and the checkpoints — runs to the end and logs finite losses."""
    if not torch.cuda.is_available():
        pytest.skip('needs a HIP device')
    import json
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    port = _free_port()
    procs = []
    for r in range(2):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE='2', MASTER_ADDR='127.0.0.1',
                   MASTER_PORT=str(port), SCL_TRAIN_ONE_GPU_GLOO='1', PYTHONPATH=root)
        procs.append(subprocess.Popen(
            [sys.executable, '-m', 'soft_contrastive_learning_amd.train.train', '--loss', loss_name,
             '--synthetic_dataset', '120', '--height', '64', '--width', '80', '--positives_per_tuple', '3',
             '--negatives_per_tuple', '3', '--hard_negatives_per_tuple', '1', '--hard_positives_per_tuple', '1',
             '--steps', '6', '--max_epoch', '1', '--mining_step', '4', '--mining_cache_size', '16',
             '--eval_step', '4', '--save_step', '4', '--num_eval_queries', '4', '--dtype', 'bf16',
             '--out_root', str(tmp_path)], env=env, cwd=root, stdout=subprocess.PIPE, stderr=subprocess.STDOUT))
    outs = []
    for pr in procs:
        try:
            o, _ = pr.communicate(timeout=400)
        except subprocess.TimeoutExpired:
            pr.kill()
            o, _ = pr.communicate()
        outs.append(o.decode(errors='replace'))
    assert all(pr.returncode == 0 for pr in procs), outs
    recs = [json.loads(line) for line in open(os.path.join(str(tmp_path), loss_name, 'train_log.txt'))]
    losses = [r['loss'] for r in recs if 'loss' in r]
    assert len(losses) >= 4 and all(np.isfinite(losses)), recs
    assert any(r.get('event') == 'mining_cache' for r in recs)
    assert any(r.get('event') == 'eval' for r in recs)
