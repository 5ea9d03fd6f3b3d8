"""tests/golden/golden_ref_topn_v1.json: the pickle the reference's own evaluation/top-n.py WROTE
when it was run as ``__main__`` in the build container on tests/util_data.retrieval_dataset — on
real NumPy and scikit-learn (PCA, KDTree, pairwise_distances) and the reference's own util/io.py,
util/meta.py, util/helper.py (tests/tools/ref_exec/make_golden_ref_topn.py lists the four names it
had to supply; none computes anything).  This is an output of the reference itself for SURVEY.md
section 8 rows A14 and (f) 2: whitening, the thinning loop with its doubled index 0 at l = 0, the
top-N lists, ground truth, index translation, the pickle's element types and its file name.

CPU part: the retrieval oracle (oracle/topn_np.py) and the host pieces of the package (CSV reader,
get_xy, thinning, output path) against it.
GPU part: the package's evaluation/top_n.py script, files in -> pickle out, against it element
by element.  Lists must be IDENTICAL except inside a run of reference distances closer than 1e-5
relative (float32 whitened descriptors: such neighbours may legitimately swap); feature distances
within 1e-4 relative (BASELINE.json north_star).
"""
import base64
import hashlib
import json
import os

import numpy as np
import pytest

from oracle import topn_np as O
from soft_contrastive_learning_amd.evaluation import top_n
from soft_contrastive_learning_amd.util import io as sio
from soft_contrastive_learning_amd.util.meta import get_xy
from tests import util_data as U

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden', 'golden_ref_topn_v1.json')
DOC = json.load(open(GOLDEN))
C = DOC['cases'][0]


def _arr(key, dtype, shape=None):
    a = np.frombuffer(base64.b64decode(C[key]), dtype=dtype)
    return a.reshape(shape) if shape is not None else a


def _want():
    shp = C['top_i_shape']
    return {'top_i': _arr('top_i_i32_b64', '<i4', shp), 'top_f': _arr('top_f_dists_f64_b64', '<f8', shp),
            'top_g': _arr('top_g_dists_f64_b64', '<f8', shp), 'gt_i': _arr('gt_i_i32_b64', '<i4'),
            'gt_g': _arr('gt_g_dist_f64_b64', '<f8')}


def _dataset():
    return U.retrieval_dataset(seed=C['seed'], n_pca=C['n_pca'], n_ref=C['n_ref'], n_query=C['n_query'], e=C['e'])


def _lists_agree(got_i, want_i, want_f, gap=1e-5):
    """Equal, or different only where the reference's own neighbouring distances are within `gap`."""
    bad = 0
    for q in range(want_i.shape[0]):
        for k in np.nonzero(got_i[q] != want_i[q])[0]:
            near = [abs(want_f[q, j] - want_f[q, k]) <= gap * want_f[q, k]
                    for j in (k - 1, k + 1) if 0 <= j < want_i.shape[1]]
            if k == want_i.shape[1] - 1:
                near.append(True)            # the last place may go to the (unseen) 26th neighbour
            bad += not any(near)
    return bad


def test_fixture_is_a_run_of_the_reference_script():
    assert DOC['meta']['made_by'] == 'tests/tools/ref_exec/make_golden_ref_topn.py'
    assert C['written'] == ['l0.0_dim256/set_queryv1.pickle']
    assert C['types'] == ['list', 'list', 'ndarray', 'list', 'ndarray', 'list']      # evaluation/top-n.py:104-119
    assert C['ref_idx_head'] == [0, 0, 1, 2] and C['ref_idx_len'] == C['n_ref'] + 1  # evaluation/top-n.py:91-94


def test_host_pieces_give_what_the_reference_run_used(tmp_path):
    ds = _dataset()
    p = str(tmp_path / 'ref.csv')
    sio.save_csv({'easting': [repr(float(v)) for v in ds['ref_xy'][:, 0]],
                  'northing': [repr(float(v)) for v in ds['ref_xy'][:, 1]]}, p)
    xy = get_xy(sio.load_csv(p))
    assert xy.dtype == np.float64 and np.array_equal(xy, ds['ref_xy'])
    ref_idx = top_n.thin_reference(xy, C['l'])
    assert hashlib.sha1(np.asarray(ref_idx, '<i8').tobytes()).hexdigest() == C['ref_idx_sha1']
    out = top_n.out_pickle_path('top_n', '/somewhere/set_query.v1.pickle', C['l'], C['d'])
    assert os.path.relpath(out, 'top_n') == C['written'][0]


def test_oracle_retrieval_gives_the_reference_runs_lists():
    """scikit-learn's PCA (the reference's own call) + the float64 brute force of oracle/topn_np.py on
    the thinned set, translated back: the lists, distances and ground truth of the reference run."""
    from sklearn.decomposition import PCA
    from sklearn.metrics import pairwise_distances
    ds, want = _dataset(), _want()
    pca = PCA(whiten=True, n_components=C['d']).fit(ds['pca_f'])
    ref_idx = top_n.thin_reference(ds['ref_xy'], C['l'])
    ref_f = pca.transform(ds['ref_f'])[ref_idx]
    qry_f = pca.transform(ds['query_f'])
    dist, idx = O.topn_bruteforce(ref_f, qry_f, C['N'])
    got_i = np.asarray(ref_idx)[idx]
    assert _lists_agree(got_i, want['top_i'], want['top_f']) == 0
    assert np.abs(dist - want['top_f']).max() <= 1e-6 * want['top_f'].max()
    xy = pairwise_distances(ds['query_xy'], ds['ref_xy'], metric='euclidean')
    assert np.array_equal(np.asarray(ref_idx)[np.argmin(xy[:, ref_idx], axis=1)], want['gt_i'])
    assert np.array_equal(O.recall_at_threshold(want['top_g'], [5.0, 25.0], n=1),
                          top_n.recall_at(want['top_g'], [5.0, 25.0], n=1))


@pytest.mark.gpu
@pytest.mark.parametrize('backend', ['device', 'sklearn'])
def test_the_packages_script_writes_the_reference_runs_pickle(tmp_path, backend):
    ds, want = _dataset(), _want()
    paths = {}
    for name, xy in (('ref', ds['ref_xy']), ('query', ds['query_xy'])):
        paths[name + '_csv'] = str(tmp_path / ('set_%s.csv' % name))
        sio.save_csv({'easting': [repr(float(v)) for v in xy[:, 0]],
                      'northing': [repr(float(v)) for v in xy[:, 1]]}, paths[name + '_csv'])
    for name in ('pca', 'ref', 'query'):
        paths[name + '_lv_pickle'] = str(tmp_path / ('set_%s.v1.pickle' % name))
        sio.save_pickle([row for row in ds[name + '_f']], paths[name + '_lv_pickle'])
    argv = ['--N', str(C['N']), '--out_root', str(tmp_path / 'top_n'), '--pca_backend', backend]
    for k, v in paths.items():
        argv += ['--' + k, v]
    written = top_n.main(argv)
    assert [os.path.relpath(w, str(tmp_path / 'top_n')) for w in written] == C['written']
    top_i, top_g, top_f, gt_i, gt_g, ref_idx = sio.load_pickle(written[0])
    assert [type(v).__name__ for v in (top_i, top_g, top_f, gt_i, gt_g, ref_idx)] == C['types']
    assert hashlib.sha1(np.asarray(ref_idx, '<i8').tobytes()).hexdigest() == C['ref_idx_sha1']
    top_i, top_g, top_f = np.asarray(top_i), np.asarray(top_g), np.asarray(top_f, dtype=np.float64)
    assert top_i.shape == tuple(C['top_i_shape'])
    assert _lists_agree(top_i, want['top_i'], want['top_f']) == 0
    assert (top_i == want['top_i']).mean() >= 0.995
    assert np.abs(top_f - want['top_f']).max() <= 1e-4 * want['top_f'].max()
    same = top_i == want['top_i']
    assert np.array_equal(top_g[same], want['top_g'][same])            # the same float64 table lookups
    assert np.array_equal(np.asarray(gt_i), want['gt_i'])
    assert np.array_equal(np.asarray(gt_g), want['gt_g'])
