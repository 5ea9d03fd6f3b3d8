"""Loads the reference's own ``train/train.py`` (where it lies under /root/reference) as a module in
the build container: TensorFlow is tests/tools/ref_exec/tf_shim.py, and the names the file imports
but the image / the reference lack are supplied — none on an executed path except the recalled
pointnetvlad losses (see make_golden_ref_trainer.py, which documents each).  TEST TOOLING, BUILD
CONTAINER ONLY."""
import importlib.util
import os
import sys
import types

import numpy as np

sys.dont_write_bytecode = True          # nothing may be written under /root/reference (no __pycache__)
HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(HERE)))
for _p in (ROOT, HERE):
    if _p not in sys.path:
        sys.path.insert(0, _p)

import tf_shim  # noqa: E402
from oracle import losses_np as O  # noqa: E402

REF_ROOT = '/root/reference'
TRAIN = os.path.join(REF_ROOT, 'train', 'train.py')
F32 = np.float32


def install_names():
    sys.modules['tensorflow'] = tf_shim
    pkg = types.ModuleType('learnlarge')
    pkg.__path__ = [REF_ROOT]
    sys.modules['learnlarge'] = pkg
    sys.modules['cv2'] = types.ModuleType('cv2')
    nv, nvl = types.ModuleType('netvlad_tf'), types.ModuleType('netvlad_tf.layers')
    nv.layers = nvl                                    # imported by model/nets.py; the embedder is not run
    sys.modules['netvlad_tf'], sys.modules['netvlad_tf.layers'] = nv, nvl
    pn = types.ModuleType('pointnetvlad_cls')
    for name in ('triplet_loss', 'lazy_triplet_loss', 'quadruplet_loss', 'lazy_quadruplet_loss'):
        fn = getattr(O, name)
        setattr(pn, name, (lambda f: lambda *a: tf_shim._t(np.asarray(
            f(*[np.asarray(x) for x in a]), dtype=F32)))(fn))
    pn.best_pos_distance = lambda q, p: tf_shim._t(O.best_pos_distance(np.asarray(q), np.asarray(p)))
    outer = types.ModuleType('pointnetvlad')
    outer.pointnetvlad_cls = pn
    sys.modules['pointnetvlad'] = outer
    sys.modules['pointnetvlad.pointnetvlad_cls'] = pn
    sys.modules['pointnetvlad_cls'] = pn
    import learnlarge.util.helper as helper            # the reference's own module
    helper.debugging = lambda: False
    helper.location = lambda: 'here'
    helper.srv_root = helper.fs_root
    skl = types.ModuleType('learnlarge.model.incremental_skl')
    skl.skl_init = skl.single_skl_increment = skl.multiple_skl_increments = None
    sys.modules['learnlarge.model.incremental_skl'] = skl
    mac = types.ModuleType('learnlarge.model.mac')
    mac.spp = None
    sys.modules['learnlarge.model.mac'] = mac


def load_trainer():
    spec = importlib.util.spec_from_file_location('reference_train_train', TRAIN)
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod
