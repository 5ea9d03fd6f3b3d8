"""Does the bf16 path TRAIN like the float32 path?  Every other GPU test is single-step; the
single-step bf16-vs-f32 gradient agreement in the lower layers is 0.7-0.95 cosine
(tests/test_gpu_config1.py), which says nothing about where the optimisation goes.  Here the
trainer's dataset route (sampler -> pipeline -> mining cache -> step -> evaluation: the reference
loop train/train.py:987-1109) runs 200 steps of the soft-contrastive loss twice, same seed, at the
REFERENCE'S OWN training shape — 25 images of 240 x 180 per step (train/train.py:423-428), where
since round 5 every bf16 convolution is an own kernel — once `--dtype bf16`, once `--dtype f32`
(library convolutions), on the synthetic pose-tagged set with a place-independent distractor in
every image and queries from another traverse (an untrained net localises 30 % of them).

Stated bands (measured: scripts/train_dtype_ab.py, profiles/r05/train_bf16_vs_f32_240x180.json —
with the trainer's TensorFlow-form Adam (train/optim.py): window means within 0.08 %, loss drop
0.0196 / 0.0176 over 200 steps, other-region %<25m@Top1 32.5 -> 70 / 60):
  * means of the training loss over windows of 25 steps: bf16 within 1 % of f32, every window;
  * the loss falls in both runs, by amounts within a factor of two of each other;
  * the localisation metric of train/evaluate.py on the OTHER region (%<25m@Top1) improves in both
    runs and ends within 20 points (8 of 40 queries) of each other.
"""
import importlib.util
import os

import pytest
import torch

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_bf16_trains_like_float32_at_the_reference_shape():
    assert torch.cuda.is_available()
    spec = importlib.util.spec_from_file_location('train_dtype_ab',
                                                  os.path.join(ROOT, 'scripts', 'train_dtype_ab.py'))
    ab = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(ab)
    out = ab.main(['--height', '180', '--width', '240', '--positives', '12', '--negatives', '12',
                   '--steps', '200', '--steps_per_epoch', '100', '--images', '360', '--eval_step', '50',
                   '--mining_step', '50', '--lr', '1e-4', '--distractor', '0.7'])
    c = out['comparison']
    assert out['runs']['bf16']['steps'] == out['runs']['f32']['steps'] >= 180
    assert c['max_relative_gap_of_window_means'] < 0.01, c
    db, df = c['loss_drop_bf16'], c['loss_drop_f32']
    assert db > 0.008 and df > 0.008, c
    assert 0.5 < db / df < 2.0, c
    ev = c['evaluations_bf16_vs_f32']
    assert len(ev) >= 3
    key = 'other %<25m@Top1'
    first, last = ev[0][key], ev[-1][key]
    assert first[0] == first[1]                              # same weights, same descriptors' ranking
    # (evaluations at anchors 0, 50, 100, 150 of each epoch; the last one of the run)
    assert last[0] >= first[0] + 15 and last[1] >= first[1] + 15, ev
    assert abs(last[0] - last[1]) <= 20, ev
