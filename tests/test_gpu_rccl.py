"""The RCCL ("nccl") path on the ONE GPU a test box has: a one-rank process group.

Everything multi-rank in tests/test_gpu_dist.py rides on gloo (several ranks sharing cuda:0), so
until round 6 no test, smoke or bench run had ever created an RCCL communicator.  A one-rank
group does: ``init_process_group('nccl', device_id=...)``, communicator creation, the
collectives' own stream and its events against the compute stream, the asynchronous work handles
``GradBuckets`` waits on.  The sums over one rank change nothing, so every result must be
BIT-equal to the step without a process group (new work — the reference is single-process,
SURVEY.md §8e; call sites bench.py `--force-dist`, train/train.py `--force_dist`).
"""
import json
import os
import socket
import subprocess
import sys

import pytest
import torch
import torch.multiprocessing as mp

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    port = s.getsockname()[1]
    s.close()
    return port


def _one_rank_env(port):
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), RANK='0', WORLD_SIZE='1',
                      LOCAL_RANK='0')
    os.environ.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')


def _collectives_worker(port, out):
    import faulthandler
    faulthandler.dump_traceback_later(240, exit=True)
    _one_rank_env(port)
    import torch.distributed as dist
    from soft_contrastive_learning_amd import parallel
    from soft_contrastive_learning_amd.evaluation import retrieval
    from soft_contrastive_learning_amd.model import losses
    dev = torch.device('cuda:0')
    torch.cuda.set_device(dev)
    group = parallel.init_process_group(dev, backend='nccl', timeout_s=120, force_single=True)
    res = {'backend': dist.get_backend(), 'world': dist.get_world_size(group)}
    g = torch.Generator().manual_seed(500)
    b, e = 24, 32768
    emb = torch.randn(b, e, generator=g)
    emb = (emb / emb.norm(dim=1, keepdim=True)).to(dev)
    # all_gather_rows forward + backward
    x = emb.clone().requires_grad_(True)
    full = parallel.all_gather_rows(x, group)
    w = torch.randn(b, e, generator=g).to(dev)
    (full * w).sum().backward()
    res['gather_fwd_equal'] = bool(torch.equal(full.detach(), emb))
    res['gather_bwd_equal'] = bool(torch.equal(x.grad, w))
    # the loss on the gathered batch, own-rows backward, against the plain call
    xy = torch.rand(b, 2, generator=g) * 60.0
    dmat = (xy[:, None] - xy[None]).norm(dim=2)[None].to(dev)
    ref = emb.clone().requires_grad_(True)
    loss_ref = losses.wms_loss(dmat, ref, d_alpha=0.8, d_beta=15.0)
    loss_ref.backward()
    mine = emb.clone().requires_grad_(True)
    loss = parallel.wms_loss_dp(dmat, mine, 0.8, 15.0, group=group)
    loss.backward()
    res['wms_loss_equal'] = float(loss) == float(loss_ref)
    res['wms_grad_equal'] = bool(torch.equal(mine.grad, ref.grad))
    labels = torch.arange(b, device=dev) // 3
    ref = emb.clone().requires_grad_(True)
    l0 = losses.ms_loss(labels, ref)
    l0.backward()
    mine = emb.clone().requires_grad_(True)
    l1 = parallel.ms_loss_dp(labels, mine, group=group)
    l1.backward()
    res['ms_equal'] = float(l0) == float(l1) and bool(torch.equal(mine.grad, ref.grad))
    # scalar mean over ranks (the per-tuple losses), ragged gather
    t = torch.tensor(1.25, device=dev, requires_grad=True)
    m = parallel.tuple_loss_dp(t * 2.0, group)
    m.backward()
    res['tuple_mean'] = (float(m), float(t.grad))
    rows = torch.arange(10, dtype=torch.float32, device=dev).reshape(5, 2)
    res['ragged_equal'] = bool(torch.equal(parallel.all_gather_ragged(rows, group), rows))
    # sharded retrieval with the exchange forced through the one-rank group
    refs = torch.randn(4096, 256, generator=g).to(dev)
    qry = torch.randn(64, 256, generator=g).to(dev)
    d_sh, i_sh = parallel.topn_l2_sharded(refs, qry, 25, 0, group=group, force_exchange=True)
    d_all, i_all = retrieval.topn_l2(refs, qry, 25)
    res['topn_idx_equal'] = bool(torch.equal(i_sh.cpu(), i_all.cpu()))
    res['topn_dist_equal'] = bool(torch.equal(d_sh.cpu(), d_all.cpu()))
    dist.barrier()
    torch.cuda.synchronize()
    out.put(res)
    dist.destroy_process_group()
    faulthandler.cancel_dump_traceback_later()


def _run(target, args, limit=420):
    ctx = mp.get_context('spawn')
    out = ctx.Queue()
    p = ctx.Process(target=target, args=args + (out,))
    p.start()
    try:
        p.join(timeout=limit)
        code = p.exitcode
    finally:
        if p.is_alive():
            p.kill()
            p.join(timeout=10)
    assert code == 0, 'worker exit code %s (None = hung, killed)' % (code,)
    return out.get(timeout=10)


def test_rccl_one_rank_collectives_match_the_plain_calls():
    if not torch.cuda.is_available():
        pytest.skip('needs a HIP device')
    res = _run(_collectives_worker, (_free_port(),))
    assert res['backend'] == 'nccl' and res['world'] == 1
    assert res['gather_fwd_equal'] and res['gather_bwd_equal']
    assert res['wms_loss_equal'] and res['wms_grad_equal'] and res['ms_equal']
    assert res['tuple_mean'] == (2.5, 2.0)
    assert res['ragged_equal'] and res['topn_idx_equal'] and res['topn_dist_equal']


def _step_worker(port, bucket_bytes, out):
    """bf16 train steps with the bucket all-reduces FORCED ON in a one-rank RCCL group (gradient
    sink, weight gradients on the second stream) against the same steps with no collective.

    (i) backbone only, fixed upstream gradient — a deterministic computation (the two-rank gloo
    test relies on the same): the flat gradient buffer must be BIT-equal over three steps (buffer
    reuse, stale events, the all-reduce of step k against the memset of step k + 1).
    (ii) the full step (NetVLAD + wms loss through wms_loss_dp, Adam): the head's backward
    accumulates with atomics, so two plain runs already differ in the last bits and the bf16
    backbone amplifies that; forced-vs-plain must differ no more than plain-vs-plain does."""
    import faulthandler
    faulthandler.dump_traceback_later(300, exit=True)
    _one_rank_env(port)
    import torch.distributed as dist
    from soft_contrastive_learning_amd import parallel
    from soft_contrastive_learning_amd.model import losses, nets
    from soft_contrastive_learning_amd.train.optim import TFAdam
    dev = torch.device('cuda:0')
    torch.cuda.set_device(dev)
    g = torch.Generator().manual_seed(600)
    b = 6
    images = torch.randint(0, 256, (b, 240, 320, 3), generator=g).float().to(dev)
    xy = torch.rand(b, 2, generator=g) * 60.0
    dmat = (xy[:, None] - xy[None]).norm(dim=2)[None].to(dev)
    # (i) runs at the shape of the two-rank gloo test: one 640 x 480 image, every layer on the
    # hand-written kernels whose accumulation order is fixed
    image1 = torch.randint(0, 256, (1, 480, 640, 3), generator=g).float().to(dev)
    up = torch.randn(1, 30, 40, 512, generator=g).to(dev).bfloat16()

    def backbone(group, force):
        model = nets.VGG16NetVLAD(compute_dtype=torch.bfloat16, seed=21, fused_relu=True).to(dev)
        params = [p for n, p in model.named_parameters() if not n.startswith(('assignment', 'cluster'))]
        buckets = parallel.GradBuckets(params, group, bucket_bytes=bucket_bytes, force_collectives=force)
        assert buckets.enabled == force
        nets.GRAD_SINK = buckets
        flats = []
        try:
            for _ in range(3):
                buckets.zero()
                model.features(image1).backward(up)
                launched = len(buckets._handles)
                buckets.finish()
                torch.cuda.synchronize()
                flats.append(buckets.flat.clone())
        finally:
            nets.GRAD_SINK = None
        return flats, launched, len(buckets.buckets), len(buckets._streams)

    def full(group, force):
        model = nets.VGG16NetVLAD(compute_dtype=torch.bfloat16, seed=21, fused_relu=True).to(dev)
        params = list(model.parameters())
        buckets = parallel.GradBuckets(params, group, bucket_bytes=bucket_bytes, force_collectives=force)
        opt = TFAdam(params, lr=1e-6, fused=True)
        nets.GRAD_SINK = buckets
        flats, vals = [], []
        try:
            for _ in range(3):
                buckets.zero()
                emb = model(images)
                if force:
                    loss = parallel.wms_loss_dp(dmat, emb, 0.8, 15.0, group=group)
                else:
                    loss = losses.wms_loss(dmat, emb, d_alpha=0.8, d_beta=15.0)
                loss.backward()
                launched = len(buckets._handles)
                buckets.finish()
                torch.cuda.synchronize()
                flats.append(buckets.flat.clone())
                vals.append(float(loss.detach()))
                opt.step()
        finally:
            nets.GRAD_SINK = None
        torch.cuda.synchronize()
        return flats, vals, launched, len(buckets.buckets)

    def rel(a, c):
        return float((a - c).norm() / a.norm().clamp_min(1e-30))

    plain_b, again_b = backbone(None, False), backbone(None, False)
    plain_f, again_f = full(None, False), full(None, False)
    group = parallel.init_process_group(dev, backend='nccl', timeout_s=120, force_single=True)
    forced_b, forced_f = backbone(group, True), full(group, True)
    dist.barrier()
    out.put({'backend': dist.get_backend(),
             'backbone_flat_equal': [bool(torch.equal(a, c)) for a, c in zip(plain_b[0], forced_b[0])],
             'backbone_plain_repeatable': [bool(torch.equal(a, c)) for a, c in zip(plain_b[0], again_b[0])],
             'backbone_rel_plain_forced': [rel(a, c) for a, c in zip(plain_b[0], forced_b[0])],
             'backbone_flat_nonzero': float(forced_b[0][0].abs().sum()) > 0,
             'backbone_launched': (plain_b[1], forced_b[1]), 'backbone_buckets': forced_b[2],
             'side_streams': forced_b[3],
             'full_launched': (plain_f[2], forced_f[2]), 'full_buckets': forced_f[3],
             'full_loss': (plain_f[1], again_f[1], forced_f[1]),
             'full_rel_plain_again': [rel(a, c) for a, c in zip(plain_f[0], again_f[0])],
             'full_rel_plain_forced': [rel(a, c) for a, c in zip(plain_f[0], forced_f[0])]})
    dist.destroy_process_group()
    faulthandler.cancel_dump_traceback_later()


@pytest.mark.parametrize('bucket_bytes', [1 << 20, 16 << 20])
def test_rccl_one_rank_train_step_with_forced_bucket_all_reduce(bucket_bytes):
    if not torch.cuda.is_available():
        pytest.skip('needs a HIP device')
    res = _run(_step_worker, (_free_port(), bucket_bytes))
    print(res)
    assert res['backend'] == 'nccl'
    few = 8 if bucket_bytes == 1 << 20 else 3
    assert res['backbone_launched'][0] == 0 and res['backbone_launched'][1] == res['backbone_buckets'] >= few, res
    assert res['full_launched'][0] == 0 and res['full_launched'][1] == res['full_buckets'] >= few, res
    assert res['side_streams'] == 1, 'the weight gradients did not run on the second stream'
    assert all(res['backbone_plain_repeatable']), ('the backbone pass itself is not repeatable', res)
    assert res['backbone_flat_nonzero'] and all(res['backbone_flat_equal']), res
    p, a, f = res['full_loss']
    for k in range(3):
        assert abs(f[k] - p[k]) <= 3.0 * abs(a[k] - p[k]) + 1e-5 * abs(p[k]), res
        assert res['full_rel_plain_forced'][k] <= 3.0 * res['full_rel_plain_again'][k] + 1e-6, res


def _bench(extra, timeout=1500):
    env = dict(os.environ)
    for k in ('RANK', 'LOCAL_RANK', 'WORLD_SIZE', 'MASTER_ADDR', 'MASTER_PORT', 'SCL_BENCH_ONE_GPU_GLOO'):
        env.pop(k, None)
    env.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')
    r = subprocess.run([sys.executable, os.path.join(ROOT, 'bench.py'), '--gpus', '1', '--no-cpu-baseline',
                        '--no-retrieval', '--no-batch-sweep', '--no-telemetry'] + extra, env=env,
                       capture_output=True, text=True, timeout=timeout)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith('{')]
    assert len(lines) == 1
    return json.loads(lines[0])


def test_bench_force_dist_line_over_rccl():
    """`bench.py --gpus 1 --force-dist`: the driver's data-parallel step at configs[1] through a
    one-rank RCCL group — same JSON line plus `comm` (backend nccl, world 1), the same loss as the
    plain line, and a step time close to it (the collectives of one rank are copies; 2 % is what
    profiles/r06 records, 10 % what this asserts on a shared box)."""
    if not torch.cuda.is_available():
        pytest.skip('needs a HIP device')
    plain = _bench(['--steps', '10', '--warmup', '3'])
    forced = _bench(['--steps', '10', '--warmup', '3', '--force-dist'])
    c = forced['comm']
    assert c['backend'] == 'nccl' and c['world_seen'] == 1
    assert c['allgather_us_median'] > 0 and c['finish_wait_us_median'] is not None
    assert c['allgather_bytes_per_rank'] == 24 * 32768 * 4 and c['allreduce_buckets'] >= 3
    assert forced['n_gpus'] == 1 and forced['config']['global_batch'] == 24
    assert forced['switches'].get('--force-dist') is True
    assert 'comm' not in plain
    # (the loss of the LAST step: the forced run's warm-up makes more Adam updates — it also times
    # the 0 / 8 reserved-CU settings — so the two values are close, not equal)
    assert abs(forced['config']['loss'] - plain['config']['loss']) <= 0.05 * abs(plain['config']['loss'])
    print('ms_per_step plain %.3f forced %.3f comm %s' % (plain['ms_per_step'], forced['ms_per_step'], c))
    assert forced['ms_per_step'] <= 1.10 * plain['ms_per_step'], (forced['ms_per_step'], plain['ms_per_step'])


def test_bench_retrieval_force_dist_over_rccl():
    """`bench.py --workload retrieval --gpus 1 --force-dist`: candidate all-gather + merge through the
    one-rank RCCL group; same index lists (checksum) as the plain call."""
    if not torch.cuda.is_available():
        pytest.skip('needs a HIP device')
    common = ['--workload', 'retrieval', '--steps', '2', '--warmup', '1', '--refs', '20000', '--queries', '1000']
    plain = _bench(common, timeout=900)
    forced = _bench(common + ['--force-dist'], timeout=900)
    assert forced['backend'] == 'nccl' and forced['world_seen'] == 1
    assert forced['checksum_idx'] == plain['checksum_idx']


def test_trainer_force_dist_over_rccl(tmp_path):
    """`train.py --force_dist 1` on the dataset route (sampler -> pipeline -> gathered-batch loss,
    sharded mining-cache refresh, evaluation) through a one-rank RCCL group: runs to the end with
    finite losses, exits 0."""
    if not torch.cuda.is_available():
        pytest.skip('needs a HIP device')
    env = dict(os.environ, PYTHONPATH=ROOT)
    for k in ('RANK', 'LOCAL_RANK', 'WORLD_SIZE', 'MASTER_ADDR', 'MASTER_PORT', 'SCL_TRAIN_ONE_GPU_GLOO'):
        env.pop(k, None)
    r = subprocess.run(
        [sys.executable, '-m', 'soft_contrastive_learning_amd.train.train', '--loss', 'wms',
         '--synthetic_dataset', '120', '--height', '64', '--width', '80', '--positives_per_tuple', '3',
         '--negatives_per_tuple', '3', '--hard_negatives_per_tuple', '1', '--hard_positives_per_tuple', '1',
         '--steps', '6', '--max_epoch', '1', '--mining_step', '4', '--mining_cache_size', '16',
         '--eval_step', '4', '--save_step', '4', '--num_eval_queries', '4', '--dtype', 'bf16',
         '--force_dist', '1', '--tensorboard', '0', '--out_root', str(tmp_path)],
        env=env, cwd=ROOT, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, (r.stdout[-2000:], r.stderr[-3000:])
    import numpy as np
    recs = [json.loads(line) for line in open(os.path.join(str(tmp_path), 'wms', 'train_log.txt'))]
    vals = [x['loss'] for x in recs if 'loss' in x]
    assert len(vals) >= 4 and all(np.isfinite(vals)), recs
    assert any(x.get('event') == 'mining_cache' for x in recs)
