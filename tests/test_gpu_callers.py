"""GPU checks of the caller counterparts: one trainer step per hot-path loss, the
inference product and the top-n harness against the reference's own scikit-learn calls."""
import pickle

import numpy as np
import pytest
import torch

from oracle import topn_np as TN
from tests import util_data as U

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def dev():
    assert torch.cuda.is_available()
    return torch.device("cuda:0")


@pytest.mark.parametrize("loss", ["wms", "ms_loss", "triplet", "lazy_triplet", "evil_triplet",
                                  "quadruplet", "lazy_quadruplet", "evil_quadruplet", "logratio"])
def test_trainer_step_runs_and_updates_every_variable(dev, loss):
    from soft_contrastive_learning_amd.model import nets
    from soft_contrastive_learning_amd.train import train as T
    flags = T.make_parser().parse_args(['--loss', loss, '--height', '64', '--width', '80',
                                        '--positives_per_tuple', '3', '--negatives_per_tuple', '3',
                                        '--margin_1', '0.5', '--margin_2', '0.5'])
    shape = T.tuple_shape_for(loss, 3, 3)
    model = nets.set_default_model(nets.VGG16NetVLAD().to(dev))
    before = {k: v.clone() for k, v in model.state_dict_tf().items()}
    opt = torch.optim.Adam(model.parameters(), lr=1e-4)
    distances, img = T.SyntheticTuples(flags, shape, dev).batch()
    out = nets.vgg16Netvlad(img)
    assert out.shape == (sum(shape), 32768)
    lv = T.compute_loss(flags, shape, out, distances)
    lv.backward()
    assert torch.isfinite(lv)
    opt.step()
    changed = [k for k, v in model.state_dict_tf().items() if not torch.equal(v, before[k])]
    # all trainable variables are updated, average_rgb and the VLAD variables included (A13)
    assert len(changed) == len(before), sorted(set(before) - set(changed))


def test_inference_product_is_the_reference_pickle(dev, tmp_path):
    from soft_contrastive_learning_amd.evaluation import inference
    from soft_contrastive_learning_amd.model import nets
    model = nets.VGG16NetVLAD().to(dev)
    loader = inference.synthetic_loader(64, 80)
    feats = inference.extract_features(model, loader, 5, images_per_pass=4)
    assert len(feats) == 5 and all(f.shape == (32768,) and f.dtype == np.float32 for f in feats)
    # order: feature i belongs to image i, whatever the batching
    again = inference.extract_features(model, loader, 5, images_per_pass=2)
    for a, b in zip(feats, again):
        np.testing.assert_allclose(a, b, rtol=2e-3, atol=2e-5)
    out = tmp_path / 'set_name.pickle'
    inference.save_pickle(feats, str(out))
    back = pickle.load(open(out, 'rb'))
    assert isinstance(back, list) and len(back) == 5


def test_top_n_harness_matches_the_reference_pipeline(dev):
    from sklearn.decomposition import PCA
    from sklearn.metrics import pairwise_distances
    from soft_contrastive_learning_amd.evaluation import top_n
    rng = np.random.default_rng(5)
    pca_f = rng.standard_normal((400, 96)).astype(np.float32)
    ref_f = rng.standard_normal((300, 96)).astype(np.float32)
    qry_f = rng.standard_normal((40, 96)).astype(np.float32)
    ref_xy = np.cumsum(rng.uniform(0.2, 1.5, size=(300, 2)), axis=0)
    qry_xy = rng.uniform(0, 250, size=(40, 2))
    got = top_n.get_top_n(pca_f, ref_f, qry_f, ref_xy, qry_xy, n=25, d=64, l=1.0)
    top_i, top_g, top_f, gt_i, gt_g, ref_idx = got
    # the reference's pipeline with its own calls (evaluation/top-n.py:74-117)
    pca = PCA(whiten=True, n_components=64).fit(pca_f)
    r, q = pca.transform(ref_f), pca.transform(qry_f)
    want_idx = top_n.thin_reference(ref_xy, 1.0)
    assert ref_idx == want_idx
    # the kernel consumes float32 features: pin the KDTree on the same float32 values
    wd, wi = TN.topn_kdtree(r[want_idx].astype(np.float32), q.astype(np.float32), 25)
    np.testing.assert_array_equal(np.asarray(top_i), np.asarray(want_idx)[wi])
    np.testing.assert_allclose(top_f, wd, rtol=1e-12)
    xy = pairwise_distances(qry_xy, ref_xy)[:, want_idx]
    np.testing.assert_allclose(np.asarray(top_g), np.take_along_axis(xy, wi, axis=1))
    np.testing.assert_array_equal(gt_i, np.asarray(want_idx)[xy.argmin(axis=1)])
    np.testing.assert_allclose(gt_g, xy.min(axis=1))
    assert top_n.get_top_n(pca_f, ref_f[:10], qry_f, ref_xy[:10], qry_xy, n=25, d=64) is None
