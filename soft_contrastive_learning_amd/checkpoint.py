"""Checkpoint files with the reference's variable layout and file stems.

The reference uses ``tf.train.Saver`` bundles: ``checkpoint-<global_step>`` (rolling,
``max_to_keep``), ``epoch-checkpoint-<epoch>`` and ``part-checkpoint-<step>`` (keep all)
(train/train.py:935-937, 984, 1079, 1102); restore maps every trainable variable whose
name contains ``vgg16_netvlad_pca`` (train/train.py:882-905; inference restores every
variable except ``Variable:0``, evaluation/inference.py:122-144).

Two on-disk formats carry the same variable NAMES and SHAPES
(``vgg16_netvlad_pca/conv1_1/kernel`` [3,3,3,64] HWIO, ``.../assignment/kernel``
[1,1,512,64], ``.../cluster_centers`` [1,1,1,512,64], ``.../average_rgb`` [3]):

* ``fmt='tf'`` (default): a TensorFlow bundle ``<stem>.index`` + ``<stem>.data-00000-of-00001``
  (tf_bundle.py) plus the directory's ``checkpoint`` state file — what the reference reads
  and writes.  The global step is the int32 variable ``Variable`` (the reference's unnamed
  ``tf.Variable(0)``, cf. the ``Variable:0`` filter in evaluation/inference.py:126); Adam
  slots use TF's names (``<var>/Adam``, ``<var>/Adam_1``, ``beta1_power``, ``beta2_power``).
* ``fmt='npz'``: one ``<stem>.npz`` with the same keys (``global_step`` instead of
  ``Variable``), handy for tests and quick inspection.
"""
import os

import numpy as np
import torch

from . import tf_bundle

GLOBAL_STEP_VAR = 'Variable'


def _param_table(model):
    """(tf name, parameter, is_conv_kernel) for every trainable variable."""
    from .model.nets import SCOPE
    rows = [(SCOPE + '/average_rgb', model.average_rgb, False)]
    for name in model.conv_names:
        rows.append(('%s/conv%s/kernel' % (SCOPE, name), getattr(model, 'conv%s_kernel' % name),
                     True))
        rows.append(('%s/conv%s/bias' % (SCOPE, name), getattr(model, 'conv%s_bias' % name), False))
    if getattr(model, 'vlad_cores', 64) == 64:     # the vgg16() graph has no head variables
        rows.append((SCOPE + '/assignment/kernel', model.assignment_kernel, False))
        rows.append((SCOPE + '/cluster_centers', model.cluster_centers, False))
    return rows


def _to_tf(t, is_conv):
    t = t.detach().float().cpu()
    return (t.permute(2, 3, 1, 0) if is_conv else t).contiguous().numpy()      # OIHW -> HWIO


def _from_tf(a, is_conv, like):
    t = torch.as_tensor(np.asarray(a), dtype=torch.float32)
    t = t.permute(3, 2, 0, 1) if is_conv else t
    if tuple(t.shape) != tuple(like.shape):
        raise ValueError('slot shape %s != %s' % (tuple(t.shape), tuple(like.shape)))
    return t.to(like.device, like.dtype).contiguous()


def optimizer_slots(model, optimizer):
    """torch.optim.Adam state under TF's slot-variable names (AdamOptimizer creates
    ``<var>/Adam`` = m, ``<var>/Adam_1`` = v and the two ``beta*_power`` accumulators, which
    hold beta^(t+1) after t updates); SGD momentum uses ``<var>/Momentum``."""
    out = {}
    steps = 0
    beta1 = beta2 = None
    for name, p, is_conv in _param_table(model):
        st = optimizer.state.get(p, {})
        if 'exp_avg' in st:
            out[name + '/Adam'] = _to_tf(st['exp_avg'], is_conv)
            out[name + '/Adam_1'] = _to_tf(st['exp_avg_sq'], is_conv)
            steps = max(steps, int(st['step']))
        elif 'momentum_buffer' in st and st['momentum_buffer'] is not None:
            out[name + '/Momentum'] = _to_tf(st['momentum_buffer'], is_conv)
    for g in optimizer.param_groups:
        if 'betas' in g:
            beta1, beta2 = g['betas']
    if beta1 is not None:
        out['beta1_power'] = np.asarray(beta1 ** (steps + 1), dtype=np.float32)
        out['beta2_power'] = np.asarray(beta2 ** (steps + 1), dtype=np.float32)
    return out


def restore_optimizer(model, optimizer, variables):
    """Inverse of optimizer_slots for whatever slot variables the checkpoint holds; returns
    the number of parameters whose slots were restored."""
    betas = None
    for g in optimizer.param_groups:
        if 'betas' in g:
            betas = g['betas']
    steps = 0
    if betas is not None and 'beta1_power' in variables and 0.0 < betas[0] < 1.0:
        b1p = float(np.asarray(variables['beta1_power']))
        if b1p > 0.0:
            steps = max(int(round(np.log(b1p) / np.log(betas[0]))) - 1, 0)
    done = 0
    for name, p, is_conv in _param_table(model):
        if name + '/Adam' in variables and name + '/Adam_1' in variables and betas is not None:
            st = optimizer.state[p]
            st['exp_avg'] = _from_tf(variables[name + '/Adam'], is_conv, p)
            st['exp_avg_sq'] = _from_tf(variables[name + '/Adam_1'], is_conv, p)
            # like torch's _init_group: a fused / capturable Adam reads the step on the device
            # (torch._fused_adam_ hands the kernel its data pointer), the plain one on the host
            g = next((g for g in optimizer.param_groups if any(q is p for q in g['params'])), {})
            on_dev = bool(g.get('fused') or g.get('capturable'))
            st['step'] = torch.tensor(float(steps), dtype=torch.float32,
                                      device=p.device if on_dev else 'cpu')
            done += 1
        elif name + '/Momentum' in variables and betas is None:
            optimizer.state[p]['momentum_buffer'] = _from_tf(variables[name + '/Momentum'],
                                                             is_conv, p)
            done += 1
    return done


def variables_of(model, global_step=0, optimizer=None, extra=None):
    sd = {k: v.detach().float().cpu().numpy() for k, v in model.state_dict_tf().items()}
    if optimizer is not None:
        sd.update(optimizer_slots(model, optimizer))
    for k, v in (extra or {}).items():
        sd[k] = np.asarray(v)
    # ops['step'] = tf.Variable(0) (train/train.py:655) is an int32 variable
    sd[GLOBAL_STEP_VAR] = np.asarray(int(global_step), dtype=np.int32)
    return sd


RESERVE_CUS_VAR = 'scl/reserve_cus'


def run_switches():
    """What a bit-for-bit reproduction of the run needs besides the variables: the CUs the
    persistent convolution grids leave free (include/scl_hip.h, scl_set_reserve_cus: the weight
    and bias gradients differ in rounding between values).  An int32 variable next to
    ``global_step``; restoring ignores it (only float variables under the scope are taken).  Empty
    where the HIP library is not in use."""
    if not torch.cuda.is_available():
        return {}
    from . import _lib
    return {RESERVE_CUS_VAR: np.asarray(int(_lib.load().scl_get_reserve_cus()), dtype=np.int32)}


def save(model, path_stem, global_step=0, extra=None, fmt='tf', optimizer=None):
    """Write one checkpoint; returns its prefix (tf) or file name (npz)."""
    sd = variables_of(model, global_step, optimizer, extra)
    os.makedirs(os.path.dirname(os.path.abspath(path_stem)) or '.', exist_ok=True)
    if fmt == 'tf':
        tf_bundle.write(path_stem, sd)
        return path_stem
    if fmt != 'npz':
        raise ValueError("checkpoint format must be 'tf' or 'npz', got %r" % (fmt,))
    sd['global_step'] = sd.pop(GLOBAL_STEP_VAR)
    fname = path_stem + '.npz'
    tmp = fname + '.tmp.npz'
    np.savez(tmp[:-4], **sd)
    os.replace(tmp, fname)
    return fname


def read_variables(path):
    """All numeric variables of a checkpoint (either format) as NumPy arrays."""
    if path.endswith('.index'):
        path = path[:-len('.index')]
    if tf_bundle.exists(path):
        return tf_bundle.read(path)
    fname = path if path.endswith('.npz') else path + '.npz'
    if not os.path.isfile(fname):
        raise FileNotFoundError('no checkpoint at %s (.index or .npz)' % path)
    with np.load(fname) as z:
        sd = {k: z[k] for k in z.files}
    if 'global_step' in sd:
        sd[GLOBAL_STEP_VAR] = sd.pop('global_step')
    return sd


def load(model, path, strict=True, optimizer=None):
    """Restore by variable name (scope filter like restore_weights); returns global_step.
    ``path`` is what the reference's --checkpoint flag takes: the bundle prefix (or an .npz)."""
    sd = read_variables(path)
    step = int(np.asarray(sd.get(GLOBAL_STEP_VAR, 0)))
    model.load_state_dict_tf({k: torch.from_numpy(np.ascontiguousarray(v))
                              for k, v in sd.items() if v.dtype.kind == 'f'}, strict=strict)
    if optimizer is not None:
        restore_optimizer(model, optimizer, sd)
    return step


class Saver:
    """The trainer's three savers (train/train.py:935-937)."""

    def __init__(self, out_dir, max_to_keep=1, fmt='tf'):
        self.out_dir = out_dir
        self.max_to_keep = max_to_keep
        self.fmt = fmt
        self._kept = {}

    def _remove(self, stem):
        if self.fmt == 'tf':
            tf_bundle.remove(stem)
        else:
            os.remove(stem + '.npz')

    def _save(self, model, prefix, number, global_step, keep, optimizer):
        stem = os.path.join(self.out_dir, '%s-%d' % (prefix, number))
        out = save(model, stem, global_step, fmt=self.fmt, optimizer=optimizer, extra=run_switches())
        # like tf.train.Saver's _last_checkpoints: only what THIS saver wrote, in save order —
        # files an earlier run left in the directory are never touched
        stems = self._kept.setdefault(prefix, [])
        if stem in stems:
            stems.remove(stem)
        stems.append(stem)
        if keep > 0:
            while len(stems) > keep:
                self._remove(stems.pop(0))
        if self.fmt == 'tf':
            tf_bundle.write_state(self.out_dir, os.path.basename(stem),
                                  [os.path.basename(s) for s in stems])
        return out

    def save_rolling(self, model, global_step, optimizer=None):   # saver.save(.., 'checkpoint')
        return self._save(model, 'checkpoint', global_step, global_step, self.max_to_keep,
                          optimizer)

    def save_epoch(self, model, epoch, global_step, optimizer=None):   # epoch_saver: keeps all
        return self._save(model, 'epoch-checkpoint', epoch, global_step, 0, optimizer)

    def save_part(self, model, global_step, optimizer=None):           # part_saver: keeps all
        return self._save(model, 'part-checkpoint', global_step, global_step, 0, optimizer)
