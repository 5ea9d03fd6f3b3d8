"""Checkpoint files with the reference's variable layout and file stems.

The reference uses ``tf.train.Saver`` bundles: ``checkpoint-<global_step>`` (rolling,
``max_to_keep``), ``epoch-checkpoint-<epoch>`` and ``part-checkpoint-<step>`` (keep all)
(train/train.py:935-937, 984, 1079, 1102); restore maps every trainable variable whose
name contains ``vgg16_netvlad_pca`` (train/train.py:882-905; inference restores every
variable except ``Variable:0``, evaluation/inference.py:122-144).

Here a checkpoint is ``<stem>.npz`` holding the same variable NAMES and SHAPES
(``vgg16_netvlad_pca/conv1_1/kernel`` [3,3,3,64] HWIO, ``.../assignment/kernel``
[1,1,512,64], ``.../cluster_centers`` [1,1,1,512,64], ``.../average_rgb`` [3]) plus
``global_step``.  Reading TF1 bundle binaries (.index/.data) without TensorFlow is the
first "next" row of SURVEY.md §8(f) and is not part of this round.
"""
import glob
import os
import re

import numpy as np
import torch


def save(model, path_stem, global_step=0, extra=None):
    """Write ``<path_stem>.npz``; returns the file name."""
    sd = {k: v.detach().cpu().numpy() for k, v in model.state_dict_tf().items()}
    sd['global_step'] = np.asarray(int(global_step), dtype=np.int64)
    for k, v in (extra or {}).items():
        sd[k] = np.asarray(v)
    os.makedirs(os.path.dirname(os.path.abspath(path_stem)) or '.', exist_ok=True)
    fname = path_stem + '.npz'
    tmp = fname + '.tmp.npz'
    np.savez(tmp[:-4], **sd)
    os.replace(tmp, fname)
    return fname


def load(model, path, strict=True):
    """Restore by variable name (scope filter like restore_weights); returns global_step."""
    fname = path if path.endswith('.npz') else path + '.npz'
    with np.load(fname) as z:
        sd = {k: z[k] for k in z.files}
    step = int(sd.pop('global_step', 0))
    model.load_state_dict_tf({k: torch.from_numpy(np.asarray(v)) for k, v in sd.items()},
                             strict=strict)
    return step


class Saver:
    """The trainer's three savers (train/train.py:935-937)."""

    def __init__(self, out_dir, max_to_keep=1):
        self.out_dir = out_dir
        self.max_to_keep = max_to_keep

    def _prune(self, prefix):
        files = glob.glob(os.path.join(self.out_dir, prefix + '-*.npz'))

        def step_of(f):
            m = re.search(r'-(\d+)\.npz$', f)
            return int(m.group(1)) if m else -1
        files.sort(key=step_of)
        for f in files[:-self.max_to_keep] if self.max_to_keep > 0 else []:
            os.remove(f)

    def save_rolling(self, model, global_step):          # saver.save(..., 'checkpoint', step)
        f = save(model, os.path.join(self.out_dir, 'checkpoint-%d' % global_step), global_step)
        self._prune('checkpoint')
        return f

    def save_epoch(self, model, epoch, global_step):     # epoch_saver: keeps all
        return save(model, os.path.join(self.out_dir, 'epoch-checkpoint-%d' % epoch), global_step)

    def save_part(self, model, global_step):             # part_saver: keeps all
        return save(model, os.path.join(self.out_dir, 'part-checkpoint-%d' % global_step),
                    global_step)
