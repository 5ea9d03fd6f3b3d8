#!/usr/bin/env python
"""Runs ``evaluate_localization_thread`` of the reference's own ``train/train.py`` (:360-420, where it
lies under /root/reference — nothing of it is copied) and freezes the summary values it adds and the
plot files it writes in tests/golden/golden_ref_localization_v1.json.  BUILD CONTAINER ONLY.

    python tests/tools/ref_exec/make_golden_ref_localization.py

The numbers come from real NumPy and scikit-learn (``sklearn.metrics.auc``) and real matplotlib (Agg)
for the three PDFs; TensorFlow's stand-in supplies ``tf.Summary`` as a recording list and is otherwise
involved only in importing the file (ref_trainer.py).  The example pictures at the end of the function go
through ``load_img`` / ``put_text`` / ``merge_images`` / ``save_img`` (OpenCV, absent): those four names
in the trainer's namespace are pointed at array stand-ins — what is recorded of that part is only which
files it writes.
"""
import json
import os
import sys
import tempfile

import numpy as np

sys.dont_write_bytecode = True
HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
from ref_trainer import ROOT, install_names, load_trainer  # noqa: E402
from tests import util_data as U  # noqa: E402

OUT = os.path.join(ROOT, 'tests', 'golden', 'golden_ref_localization_v1.json')


class Writer:
    def __init__(self):
        self.calls = []

    def add_summary(self, summary, step):
        self.calls.append((int(step), [(t, float(v)) for t, v in summary.value]))


def main():
    install_names()
    T = load_trainer()
    T.LOG = open(os.devnull, 'w')
    cases = []
    for name, seed, num_q, k, mode, out_name, step in (('q50_k5', 3, 50, 5, 'local', '0_checkpoint-100', 100),
                                                       ('q32_k3', 4, 32, 3, 'other', '2_checkpoint-7', 7)):
        ref_xy, query_xy, nearest_latent, nearest_d_dist, nearest_d_idx = U.localization_inputs(seed, num_q, k)
        with tempfile.TemporaryDirectory() as tmp:
            out_dir = os.path.join(tmp, 'runs', 'wms_run')
            os.makedirs(out_dir)
            T.NUM_EVAL_QUERIES, T.OUT_DIR, T.IMG_ROOT = num_q, out_dir, tmp
            T.load_img = lambda path: np.zeros((6, 8, 3), np.uint8)
            T.put_text = lambda text, image, scale=1, color=(0, 255, 0): image
            T.merge_images = lambda a, b: np.concatenate([a, b], axis=1)
            saved = []
            T.save_img = lambda img, path: saved.append((os.path.relpath(path, out_dir), list(np.asarray(img).shape)))
            info = [('d', '1', str(i)) for i in range(max(len(ref_xy), num_q))]
            w = Writer()
            np.random.seed(seed)
            T.evaluate_localization_thread(step, mode, nearest_d_dist, nearest_d_idx, nearest_latent, out_name,
                                           info[:num_q], query_xy, info[:len(ref_xy)], ref_xy, w)
            files = sorted(os.path.relpath(os.path.join(d, f), out_dir) for d, _, fs in os.walk(out_dir) for f in fs)
            dirs = sorted(os.path.relpath(os.path.join(d, x), out_dir) for d, xs, _ in os.walk(out_dir) for x in xs)
        assert len(w.calls) == 1
        cases.append({'name': name, 'seed': seed, 'num_q': num_q, 'k': k, 'mode': mode, 'out_name': out_name,
                      'step': step, 'summary_step': w.calls[0][0], 'summary': w.calls[0][1], 'files_written': files,
                      'dirs_made': dirs, 'pictures_saved': len(saved), 'picture_shape': saved[0][1]})
    meta = {'made_by': 'tests/tools/ref_exec/make_golden_ref_localization.py',
            'what': 'summary values and files of evaluate_localization_thread() of /root/reference/train/train.py '
                    '(real NumPy / scikit-learn / matplotlib; image functions replaced by array stand-ins)',
            'numpy': np.__version__}
    with open(OUT, 'w') as f:
        json.dump({'meta': meta, 'cases': cases}, f, indent=1)
    for c in cases:
        print(c['name'], c['summary'], c['files_written'], c['dirs_made'], c['pictures_saved'])


if __name__ == '__main__':
    main()
