"""The reference's calling pattern: ONE session driven concurrently from three Python threads —
training, evaluation loss and localisation (train/train.py:967-975; bodies :143-161, :193-223,
:263-309; SURVEY.md §8b "Threading").  Here: one model object, three threads, each on its own
HIP stream.  The C entry points are stateless; what this test pins down is the Python mirror's
module-level state (packed weight images, half-batch streams, the gradient sink).

* the training thread's losses and final weights must equal a single-thread run BIT FOR BIT
  (20 steps of forward + backward + fused Adam, gradient sink and second stream on, while the
  other two threads run forwards of the same model whose weights it is updating);
* the evaluation threads' results on a frozen snapshot model, computed while all of that is
  going on, must equal a serial run bit for bit; on the live model (weights moving under them —
  the reference has the same race by design) they must merely be finite unit-norm descriptors.
"""
import threading

import numpy as np
import pytest
import torch

from tests import util_data as U

pytestmark = pytest.mark.gpu

B, H, W = 4, 480, 640           # every layer on its own kernel (conv5_x maps are 30 x 40)
STEPS = 20


class _Set:
    """The slice of train/dataset.py's image-set interface that extract_features touches."""

    def __init__(self, images, xy):
        self.images, self.xy = images, xy

    def __len__(self):
        return len(self.images)

    def load_images(self, indices):
        return self.images[np.asarray(indices, dtype=int)]


@pytest.fixture(scope="module")
def dev():
    assert torch.cuda.is_available()
    return torch.device("cuda:0")


def _train(model, img, dist, steps, record):
    from soft_contrastive_learning_amd import parallel
    from soft_contrastive_learning_amd.model import losses, nets
    params = list(model.parameters())
    buckets = parallel.GradBuckets(params)
    opt = torch.optim.Adam(params, lr=1e-4, fused=True)
    nets.GRAD_SINK = buckets
    try:
        for _ in range(steps):
            buckets.zero()
            loss = losses.wms_loss(dist, model(img), d_alpha=0.8, d_beta=15.0)
            loss.backward()
            buckets.finish()
            opt.step()
            record.append(loss.detach().clone())
    finally:
        nets.GRAD_SINK = None
    torch.cuda.current_stream().synchronize()


def _eval_loss(model, img, dist):
    from soft_contrastive_learning_amd.model import losses
    with torch.no_grad():
        emb = model(img)
        return emb, losses.wms_loss(dist, emb, d_alpha=0.8, d_beta=15.0)


def _localise(model, ref_set, qry_set):
    from soft_contrastive_learning_amd.evaluation import retrieval
    from soft_contrastive_learning_amd.train import evaluate
    ref_f = evaluate.extract_features(model, ref_set, np.arange(len(ref_set)), B)
    qry_f = evaluate.extract_features(model, qry_set, np.arange(len(qry_set)), B)
    d, i = retrieval.topn_l2(ref_f, qry_f, 5)
    return ref_f, d, i


def test_three_threads_on_one_model(dev):
    from soft_contrastive_learning_amd.model import nets
    assert nets.USE_SIDE_WRW and nets.USE_PREPACK
    imgs = U.pose_images(B + 8 + 4, H, W, seed=3)
    img_train = torch.tensor(imgs[:B], device=dev)
    img_eval = torch.tensor(imgs[B:2 * B], device=dev)
    xy = np.random.default_rng(1).uniform(0, 100, size=(12, 2))
    ref_set, qry_set = _Set(imgs[B:B + 8], xy[:8]), _Set(imgs[B + 8:], xy[8:])
    dist = torch.tensor(U.positions_distances(B, side=60.0)[None], device=dev)

    def fresh(seed):
        return nets.VGG16NetVLAD(compute_dtype=torch.bfloat16, seed=seed).to(dev)

    # ---- serial reference
    serial, losses_serial = fresh(3), []
    _train(serial, img_train, dist, STEPS, losses_serial)
    snap = fresh(9)
    emb_s, loss_s = _eval_loss(snap, img_eval, dist)
    ref_s, d_s, i_s = _localise(snap, ref_set, qry_set)
    torch.cuda.synchronize()

    # ---- the same, concurrently on one live model
    live, losses_live = fresh(3), []
    done, errors = threading.Event(), []
    counts = {'eval': 0, 'loc': 0}

    def guarded(fn):
        def run():
            try:
                with torch.cuda.stream(torch.cuda.Stream(device=dev)):
                    fn()
                    torch.cuda.current_stream().synchronize()
            except BaseException as exc:              # noqa: BLE001 (re-raised in the test thread)
                errors.append(exc)
                done.set()
        return run

    def train_thread():
        _train(live, img_train, dist, STEPS, losses_live)
        done.set()

    def eval_thread():
        while not (done.is_set() and counts['eval'] >= STEPS):
            e_live, l_live = _eval_loss(live, img_eval, dist)          # weights moving under it
            e_snap, l_snap = _eval_loss(snap, img_eval, dist)
            assert torch.isfinite(e_live).all() and torch.isfinite(l_live)
            assert float((e_live.float().norm(dim=1) - 1).abs().max()) < 1e-3
            assert torch.equal(e_snap, emb_s) and torch.equal(l_snap, loss_s)
            counts['eval'] += 1

    def loc_thread():
        while not (done.is_set() and counts['loc'] >= 3):
            f_live, _, i_live = _localise(live, ref_set, qry_set)
            assert torch.isfinite(f_live).all() and int(i_live.min()) >= 0
            f_snap, d_snap, i_snap = _localise(snap, ref_set, qry_set)
            assert torch.equal(f_snap, ref_s) and torch.equal(i_snap, i_s) and torch.equal(d_snap, d_s)
            counts['loc'] += 1

    threads = [threading.Thread(target=guarded(f)) for f in (train_thread, eval_thread, loc_thread)]
    for t in threads:
        t.start()
    for t in threads:
        t.join(timeout=600)
    assert not any(t.is_alive() for t in threads), 'a thread hung'
    if errors:
        raise errors[0]
    torch.cuda.synchronize()
    assert counts['eval'] >= STEPS and counts['loc'] >= 3
    assert len(losses_live) == STEPS
    for k, (a, b) in enumerate(zip(losses_live, losses_serial)):
        assert torch.equal(a, b), ('loss of step %d differs from the single-thread run' % k,
                                   float(a), float(b))
    for (n, p), (_, q) in zip(live.named_parameters(), serial.named_parameters()):
        assert torch.equal(p, q), 'parameter %s differs from the single-thread run' % n
