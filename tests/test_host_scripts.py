"""CPU-side checks of the caller counterparts (trainer glue, inference padding, top-n
harness helpers, checkpoint files)."""
import glob
import os
from types import SimpleNamespace

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

from soft_contrastive_learning_amd import checkpoint
from soft_contrastive_learning_amd.evaluation import inference, top_n
from soft_contrastive_learning_amd.model import nets
from soft_contrastive_learning_amd.train import train as T


def test_distance_type_follows_the_reference_chain():
    # train/train.py:1378-1391 (order matters: 'swrd' before 'wrd')
    assert T.distance_type('wms') == 'wms'
    assert T.distance_type('logratio') == 'logratio'
    assert T.distance_type('triplet') == 'none'
    assert T.distance_type('huber_distance_triplet') == 'anchor'
    assert T.distance_type('pairwise_distance_neg_eigenvalue') == 'pairwise'
    assert T.distance_type('swrd') == 'swrd' and T.distance_type('prodwrd') == 'wrd'


def test_tuple_shape_and_lr_schedule():
    assert T.tuple_shape_for('wms', 12, 12) == [1, 12, 12]
    assert T.tuple_shape_for('lazy_quadruplet', 12, 12) == [1, 12, 11, 1]   # :589-592
    f = SimpleNamespace(base_lr=5e-6, lr_down_factor=0.5, lr_down_frequency=1, minimal_lr=5e-12)
    assert T.get_learning_rate(0, f) == 5e-6
    assert T.get_learning_rate(2, f) == 5e-6 * 0.25
    assert T.get_learning_rate(100, f) == 5e-12                               # floor (:120)
    f.lr_down_frequency = 2
    assert T.get_learning_rate(3, f) == 5e-6 * 0.5


def test_parser_keeps_reference_flag_names_and_defaults():
    flags = T.make_parser().parse_args([])
    assert (flags.positives_per_tuple, flags.negatives_per_tuple) == (12, 12)
    assert (flags.margin_1, flags.margin_2) == (0.1, 0.2)
    assert (flags.alpha, flags.beta, flags.wfunction, flags.sumfunction) == (0.8, 15, 'exp', 'ms')
    assert (flags.base_lr, flags.minimal_lr, flags.lr_down_factor) == (5e-6, 5e-12, 0.5)
    assert flags.optimizer == 'adam' and flags.momentum == 0.9
    assert (flags.eval_step, flags.save_step, flags.max_to_keep) == (100, 500, 1)
    # type=bool flags are truthy for any non-empty string, as in the reference (:1249,1263)
    assert T.make_parser().parse_args(['--msmining', 'False']).msmining is True


def test_synthetic_tuples_layout_cpu():
    flags = T.make_parser().parse_args(['--loss', 'wms', '--height', '32', '--width', '48',
                                        '--positives_per_tuple', '2', '--negatives_per_tuple', '3'])
    shape = T.tuple_shape_for(flags.loss, 2, 3)
    d, img = T.SyntheticTuples(flags, shape, torch.device('cpu')).batch()
    assert img.shape == (6, 32, 48, 3) and d.shape == (1, 6, 6)
    assert torch.equal(d[0], d[0].T) and float(d[0].diagonal().abs().max()) == 0.0
    flags.loss = 'ms_loss'
    lab, _ = T.SyntheticTuples(flags, shape, torch.device('cpu')).batch()
    assert lab.tolist() == [0, 0, 0, 1, 2, 3]


def test_inference_padding_matches_the_reference_quirk():
    # evaluation/inference.py:172-175: a full extra pass when num is already a multiple
    assert inference.pad_indices(5, 4).tolist() == [0, 1, 2, 3, 4, 0, 0, 0]
    assert inference.pad_indices(8, 4).tolist() == list(range(8)) + [0, 0, 0, 0]


def test_thin_reference_and_recall():
    xy = np.array([[0.0, 0], [0.5, 0], [1.2, 0], [1.3, 0], [3.0, 0]])
    assert top_n.thin_reference(xy, 0.0) == [0, 0, 1, 2, 3, 4]      # literal: index 0 twice
    assert top_n.thin_reference(xy, 1.0) == [0, 2, 4]
    g = np.array([[3.0, 1.0], [9.0, 20.0]])
    assert top_n.recall_at(g, [2.0, 10.0], n=1).tolist() == [0.0, 1.0]
    assert top_n.recall_at(g, [2.0, 10.0], n=2).tolist() == [0.5, 1.0]


def test_pca_whitening_on_the_host_matches_sklearn():
    # the device PCA is torch plumbing, so its algebra can be checked without a GPU
    from sklearn.decomposition import PCA
    from soft_contrastive_learning_amd.evaluation.pca import PCAWhitening
    rng = np.random.default_rng(3)
    x = (rng.standard_normal((90, 30)) * np.linspace(2.0, 0.3, 30)).astype(np.float32) \
        @ rng.standard_normal((30, 300)).astype(np.float32) + 1.5
    for d, data in ((12, x), (10, x[:, :40])):                # n < E and n > E code paths
        want = PCA(whiten=True, n_components=d, svd_solver='full').fit(data)
        pca = PCAWhitening(d, device='cpu', chunk=64).fit(data)
        np.testing.assert_allclose(pca.explained_variance_.numpy(), want.explained_variance_,
                                   rtol=1e-4)
        got, ref = pca.transform(data[:17]).numpy(), want.transform(data[:17])
        sign = np.sign(np.sum(got * ref, axis=0))
        np.testing.assert_allclose(got * sign, ref, atol=2e-3)
        # every component's largest-magnitude entry is positive (svd_flip, v-based)
        c = pca.components_.numpy()
        assert (c[np.arange(d), np.abs(c).argmax(axis=1)] > 0).all()
    with pytest.raises(ValueError):
        PCAWhitening(91, device='cpu').fit(x)
    with pytest.raises(RuntimeError):
        PCAWhitening(4, device='cpu').transform(x)


def test_checkpoint_roundtrip_and_saver_cadence(tmp_path):
    a, b = nets.VGG16NetVLAD(seed=3), nets.VGG16NetVLAD(seed=4)
    f = checkpoint.save(a, str(tmp_path / 'checkpoint-7'), global_step=7, fmt='npz')
    assert f.endswith('checkpoint-7.npz')
    with np.load(f) as z:
        assert z['vgg16_netvlad_pca/conv2_1/kernel'].shape == (3, 3, 64, 128)
        assert int(z['global_step']) == 7
    assert checkpoint.load(b, str(tmp_path / 'checkpoint-7')) == 7
    for k, v in b.state_dict_tf().items():
        assert torch.equal(v, a.state_dict_tf()[k])
    for fmt, ext in (('npz', '.npz'), ('tf', '.index')):
        run = tmp_path / ('run_' + fmt)
        s = checkpoint.Saver(str(run), max_to_keep=1, fmt=fmt)
        s.save_rolling(a, 100)
        s.save_rolling(a, 200)
        s.save_epoch(a, 0, 200)
        s.save_part(a, 500)
        names = sorted(os.path.basename(p) for p in glob.glob(str(run / ('*' + ext))))
        assert names == ['checkpoint-200' + ext, 'epoch-checkpoint-0' + ext,
                         'part-checkpoint-500' + ext]
    assert not glob.glob(str(tmp_path / 'run_tf' / 'checkpoint-100*'))


def test_compute_loss_rejects_losses_outside_the_hot_path():
    flags = T.make_parser().parse_args(['--loss', 'residual_det'])
    with pytest.raises(ValueError):
        T.compute_loss(flags, [1, 12, 12], torch.zeros(25, 8), None)


def test_bench_launches_its_own_ranks(tmp_path):
    """`python bench.py --gpus 2` with no outer launcher must start two rank processes itself
    and print ONE JSON line with n_gpus 2 (gloo + a stub step here: no GPU in this container);
    a failing rank must make the parent exit non-zero."""
    import json
    import subprocess
    import sys
    env = {k: v for k, v in os.environ.items()
           if k not in ('WORLD_SIZE', 'RANK', 'LOCAL_RANK', 'MASTER_PORT')}
    bench = os.path.join(ROOT, 'bench.py')
    out = subprocess.run([sys.executable, bench, '--gpus', '2', '--steps', '3', '--warmup', '1',
                          '--stub-cpu'], env=env, capture_output=True, text=True, timeout=240)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [l for l in out.stdout.splitlines() if l.startswith('{')]
    assert len(lines) == 1, out.stdout
    rec = json.loads(lines[0])
    assert rec['n_gpus'] == 2 and rec['steps'] == 3 and rec['data'] == 'stub'
    # without --stub-cpu the ranks need a HIP device: both fail here, the parent reports it
    bad = subprocess.run([sys.executable, bench, '--gpus', '2', '--steps', '1', '--warmup', '0'],
                         env=env, capture_output=True, text=True, timeout=240)
    assert bad.returncode != 0 and 'ranks failed' in bad.stderr


def test_profiling_sink_names_are_the_kernel_names():
    """Every SCL_LAUNCH site reports its kernel under the kernel's OWN name (template arguments may
    follow), so the rows of bench.py's `kernels` / `roofline*` objects can be looked up in a
    rocprofv3 trace as they stand; and bench.py prices every NetVLAD / loss kernel the library
    can launch."""
    import glob
    import re
    names = set()
    for path in sorted(glob.glob(os.path.join(ROOT, 'soft_contrastive_learning_amd', 'csrc', '*.hip'))):
        src = open(path).read()
        for m in re.finditer(r'SCL_LAUNCH\(', src):
            if '#define' in src[max(0, m.start() - 20):m.start()]:
                continue                                     # the macro's own definition
            depth, i = 0, m.end()
            while not (src[i] == ',' and depth == 0):        # the label expression has no commas
                depth += (src[i] == '(') - (src[i] == ')')
                i += 1
            labels = re.findall(r'"([^"]+)"', src[m.end():i])
            kernel = re.match(r'[\s\\]*\(?\s*(\w+)', src[i + 1:]).group(1)
            assert labels, (path, src[m.end():i])
            for lab in labels:
                assert re.sub(r'<.*$', '', lab) == kernel, (os.path.basename(path), lab, kernel)
                names.add(lab)
    import bench
    models = bench.kernel_models(24, 1200, 24, 2)
    head = sorted(n for n in names if re.match(
        r'(vlad_|gram|finish_|bwd_d|rowtile16|aggregate_kernel|dx16|wgrad_finish)', n))
    assert len(head) >= 20, head
    missing = [n for n in head if n not in models]
    assert not missing, missing
