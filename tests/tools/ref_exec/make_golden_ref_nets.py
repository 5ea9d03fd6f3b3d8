#!/usr/bin/env python
"""Executes the reference's own ``model/nets.py`` (where it lies under /root/reference — nothing of
it is copied) on tests/tools/ref_exec/tf_shim.py and freezes what ``vgg16`` returns, and what
``vgg16Netvlad`` hands to the NetVLAD layer, in tests/golden/golden_ref_nets_v1.json.
BUILD CONTAINER ONLY: needs /root/reference.

    python tests/tools/ref_exec/make_golden_ref_nets.py

``netvlad_tf.layers`` (third party, absent: SURVEY.md section 8c) is imported by the reference at
module level.  A RECORDING stand-in takes its place: ``netVLAD(x, k)`` stores its arguments and
returns ``x`` — so the fixture holds the head's INPUT and its cluster count as the reference's
call site passes them (model/nets.py:66-67), and says nothing about the head itself.

Variables and images come from tests/util_data.py (seeded; the tests rebuild them).
See tf_shim.py for what a stand-in library does and does not pin.
"""
import base64
import importlib.util
import json
import os
import sys
import types

import numpy as np

sys.dont_write_bytecode = True          # nothing may be written under /root/reference (no __pycache__)
HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(HERE)))
sys.path.insert(0, ROOT)
sys.path.insert(0, HERE)

import tf_shim  # noqa: E402
from tests import util_data as U  # noqa: E402

REF = '/root/reference/model/nets.py'
OUT = os.path.join(ROOT, 'tests', 'golden', 'golden_ref_nets_v1.json')
F32 = np.float32
HEAD_CALLS = []


def _recording_netvlad(x, k, *args, **kwargs):
    HEAD_CALLS.append({'x': np.asarray(x), 'k': int(k), 'extra_args': len(args) + len(kwargs)})
    return x


def load_reference():
    sys.modules['tensorflow'] = tf_shim
    pkg = types.ModuleType('netvlad_tf')
    lay = types.ModuleType('netvlad_tf.layers')
    lay.netVLAD = _recording_netvlad
    pkg.layers = lay
    sys.modules['netvlad_tf'] = pkg
    sys.modules['netvlad_tf.layers'] = lay
    spec = importlib.util.spec_from_file_location('reference_model_nets', REF)
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def b64(a):
    return base64.b64encode(np.ascontiguousarray(a, dtype='<f4').tobytes()).decode('ascii')


def images(c):
    img = U.pose_images(c['b'], c['h'], c['w'], seed=c['seed'])
    if c['channels'] == 1:
        img = img.mean(axis=3, keepdims=True).astype(F32)
    return img


def main():
    R = load_reference()
    cases = []
    for name, fn, b, h, w, ch in (('vgg16_rgb_2x40x56', 'vgg16', 2, 40, 56, 3),     # 40 -> 20,10,5,2: floor pooling
                                  ('vgg16_grey_1x32x32', 'vgg16', 1, 32, 32, 1),
                                  ('vgg16Netvlad_rgb_1x32x48_head_input', 'vgg16Netvlad', 1, 32, 48, 3),
                                  ('vgg16Netvlad_grey_2x48x32_head_input', 'vgg16Netvlad', 2, 48, 32, 1)):
        c = {'name': name, 'fn': fn, 'b': b, 'h': h, 'w': w, 'channels': ch, 'seed': 42, 'var_seed': 77}
        tf_shim.VARIABLES = U.vgg_variables(c['var_seed'])
        del tf_shim.CREATED[:]
        del HEAD_CALLS[:]
        out = np.asarray(getattr(R, fn)(tf_shim._t(images(c))))
        c['variables_created'] = list(tf_shim.CREATED)
        if fn == 'vgg16Netvlad':
            assert len(HEAD_CALLS) == 1 and HEAD_CALLS[0]['extra_args'] == 0
            c['head_clusters'] = HEAD_CALLS[0]['k']
            out = HEAD_CALLS[0]['x']
        c['shape'] = list(out.shape)
        c['out_f32_b64'] = b64(out)
        cases.append(c)
    meta = {
        'made_by': 'tests/tools/ref_exec/make_golden_ref_nets.py',
        'what': 'outputs of /root/reference/model/nets.py executed on tests/tools/ref_exec/tf_shim.py '
                '(NumPy stand-ins: float64 accumulation, float32 between layers); the NetVLAD layer '
                'is a recording stand-in; variables and images from tests/util_data.py',
        'numpy': np.__version__,
        'shim_ops_called': dict(sorted(tf_shim.CALLS.items())),
    }
    with open(OUT, 'w') as f:
        json.dump({'meta': meta, 'cases': cases}, f, indent=1)
    print('wrote %s: %d cases; shim ops used: %s' % (OUT, len(cases), ', '.join(sorted(tf_shim.CALLS))))


if __name__ == '__main__':
    main()
