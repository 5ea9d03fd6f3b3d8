#!/usr/bin/env python
"""RUNS the reference's own ``evaluation/top-n.py`` (as ``__main__``, where it lies under
/root/reference — nothing of it is copied) on a small synthetic data set and freezes the pickle it
writes in tests/golden/golden_ref_topn_v1.json.  BUILD CONTAINER ONLY: needs /root/reference.

    python tests/tools/ref_exec/make_golden_ref_topn.py

Unlike the loss / network fixtures this one runs on the REAL libraries the script computes with —
NumPy, scikit-learn's PCA, KDTree and pairwise_distances (1.7.2 here; the reference names no
version) — and on the reference's own util/io.py, util/meta.py and util/helper.py.  What is stood
in for, none of it arithmetic:
  * the package name ``learnlarge`` is pointed at /root/reference (the reference imports itself
    under that name, train/train.py:15-19);
  * ``cv2`` (absent; util/io.py imports it for image files, which this script never touches):
    an empty module;
  * ``learnlarge.util.experiments.get_checkpoints`` (the module is missing from the reference,
    SURVEY.md F4; evaluation/top-n.py:25 uses it to choose its L / D sweeps): returns [] — so the
    script takes its own ``else`` branch, L = [0.0], D = [256] (:38-39);
  * ``learnlarge.util.helper.srv_root`` (missing too; used for two flag DEFAULTS that the command
    line overrides): returns the scratch directory.
Descriptors and poses come from tests/util_data.retrieval_dataset (the tests rebuild them); the
shapes make scikit-learn's 'auto' PCA solver the exact one, so the run is deterministic.
"""
import base64
import hashlib
import json
import os
import runpy
import sys
import tempfile
import types

import numpy as np

sys.dont_write_bytecode = True          # nothing may be written under /root/reference (no __pycache__)
HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(HERE)))
sys.path.insert(0, ROOT)

from tests import util_data as U  # noqa: E402

REF_ROOT = '/root/reference'
SCRIPT = os.path.join(REF_ROOT, 'evaluation', 'top-n.py')
OUT = os.path.join(ROOT, 'tests', 'golden', 'golden_ref_topn_v1.json')


def b64(a, dtype):
    return base64.b64encode(np.ascontiguousarray(a, dtype=dtype).tobytes()).decode('ascii')


def install_names(scratch):
    pkg = types.ModuleType('learnlarge')
    pkg.__path__ = [REF_ROOT]
    sys.modules['learnlarge'] = pkg
    sys.modules['cv2'] = types.ModuleType('cv2')
    exp = types.ModuleType('learnlarge.util.experiments')
    exp.get_checkpoints = lambda name: []
    sys.modules['learnlarge.util.experiments'] = exp
    import learnlarge.util.helper as helper            # the reference's own module
    helper.srv_root = lambda: scratch


def write_inputs(ds, scratch):
    """With the reference's own util/io.py: CSV lists (easting / northing) and descriptor pickles as
    evaluation/inference.py writes them (a list of one array per image, :172-192)."""
    import learnlarge.util.io as rio
    paths = {}
    for name, xy in (('ref', ds['ref_xy']), ('query', ds['query_xy'])):
        paths[name + '_csv'] = os.path.join(scratch, 'set_%s.csv' % name)
        rio.save_csv({'easting': [repr(float(v)) for v in xy[:, 0]],
                      'northing': [repr(float(v)) for v in xy[:, 1]]}, paths[name + '_csv'])
    for name in ('pca', 'ref', 'query'):
        paths[name + '_lv_pickle'] = os.path.join(scratch, 'set_%s.v1.pickle' % name)
        rio.save_pickle([row for row in ds[name + '_f']], paths[name + '_lv_pickle'])
    return paths


def main():
    args = {'seed': 5, 'n_pca': 640, 'n_ref': 900, 'n_query': 120, 'e': 288}
    ds = U.retrieval_dataset(**args)
    n = 25
    with tempfile.TemporaryDirectory() as scratch:
        install_names(scratch)
        paths = write_inputs(ds, scratch)
        out_root = os.path.join(scratch, 'top_n')
        argv = ['top-n.py', '--N', str(n), '--out_root', out_root, '--log_dir', os.path.join(scratch, 'logs')]
        for k, v in paths.items():
            argv += ['--' + k, v]
        old = sys.argv
        sys.argv = argv
        try:
            runpy.run_path(SCRIPT, run_name='__main__')
        finally:
            sys.argv = old
        written = [os.path.relpath(os.path.join(d, f), out_root) for d, _, fs in os.walk(out_root) for f in fs]
        assert written == ['l0.0_dim256/set_queryv1.pickle'], written     # :43-45: the dots leave the name
        import learnlarge.util.io as rio
        top_i, top_g, top_f, gt_i, gt_g, ref_idx = rio.load_pickle(os.path.join(out_root, written[0]))
    types = [type(v).__name__ for v in (top_i, top_g, top_f, gt_i, gt_g, ref_idx)]
    top_i = np.asarray(top_i)
    case = dict(args)
    case.update({
        'name': 'top_n_l0.0_dim256', 'N': n, 'd': 256, 'l': 0.0, 'written': written,
        'types': types,
        'ref_idx_len': len(ref_idx), 'ref_idx_head': [int(v) for v in ref_idx[:4]],
        'ref_idx_sha1': hashlib.sha1(np.asarray(ref_idx, '<i8').tobytes()).hexdigest(),
        'top_i_shape': list(top_i.shape), 'top_i_i32_b64': b64(top_i, '<i4'),
        'top_f_dists_f64_b64': b64(np.asarray(top_f), '<f8'),
        'top_g_dists_f64_b64': b64(np.asarray(top_g), '<f8'),
        'gt_i_i32_b64': b64(np.asarray(gt_i), '<i4'), 'gt_g_dist_f64_b64': b64(np.asarray(gt_g), '<f8'),
    })
    import sklearn
    meta = {'made_by': 'tests/tools/ref_exec/make_golden_ref_topn.py',
            'what': 'the pickle /root/reference/evaluation/top-n.py wrote when run as __main__ on '
                    'tests/util_data.retrieval_dataset (real NumPy / scikit-learn; see the generator for '
                    'the four names stood in for)',
            'numpy': np.__version__, 'sklearn': sklearn.__version__}
    with open(OUT, 'w') as f:
        json.dump({'meta': meta, 'cases': [case]}, f, indent=1)
    print('wrote %s; ref_idx starts %s (len %d); first query hits %s' % (OUT, ref_idx[:4], len(ref_idx), top_i[0, :6]))


if __name__ == '__main__':
    main()
