"""tests/golden/golden_ref_nets_v1.json: what the reference's own model/nets.py returns when its
text is EXECUTED (on NumPy stand-ins for the eight TensorFlow calls it makes —
tests/tools/ref_exec/; build container only, the JSON is what travels): ``vgg16`` outputs and the
tensor ``vgg16Netvlad`` hands to ``layers.netVLAD`` together with its cluster count.

What this fixes from the reference's statements rather than from a reading of them: the variable
names and kernel layout ([3,3,in,out], cross-correlation), the mean image subtracted AFTER the
grey -> RGB replication, 'same' padding on every convolution, 'valid' 2x2 pooling (odd maps lose
their last row / column), no ReLU after conv5_3, the L2 normalisation over the channel axis before
the head, and K = 64.

Tolerances: the fixture is float64 accumulation rounded to float32 between layers; the float32
composition (CPU, and the HIP path with float32 maps) must agree to 2e-5 of the largest entry,
the bf16 training path to 3e-2 (13 layers of bf16 maps; stated in DESIGN.md section 7).
"""
import base64
import json
import os

import numpy as np
import pytest
import torch

from soft_contrastive_learning_amd.model import nets
from tests import util_data as U

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden', 'golden_ref_nets_v1.json')
DOC = json.load(open(GOLDEN))
CASES = DOC['cases']
F32 = np.float32


def _expected(c):
    return np.frombuffer(base64.b64decode(c['out_f32_b64']), dtype='<f4').reshape(c['shape'])


def _images(c):
    img = U.pose_images(c['b'], c['h'], c['w'], seed=c['seed'])
    if c['channels'] == 1:
        img = img.mean(axis=3, keepdims=True).astype(F32)
    return img


def _model(c, dtype=torch.float32, vlad_cores=0):
    m = nets.VGG16NetVLAD(compute_dtype=dtype, vlad_cores=vlad_cores)
    m.load_state_dict_tf({k: torch.from_numpy(v) for k, v in U.vgg_variables(c['var_seed']).items()},
                         strict=vlad_cores == 0)
    return m


def test_fixture_file_is_the_generators():
    assert DOC['meta']['made_by'] == 'tests/tools/ref_exec/make_golden_ref_nets.py'
    assert len(CASES) == 4
    used = DOC['meta']['shim_ops_called']
    for op in ('layers.conv2d', 'layers.max_pooling2d', 'nn.l2_normalize', 'nn.relu', 'get_variable',
               'nn.conv2d', 'variable_scope'):
        assert used.get(op, 0) > 0, op


@pytest.mark.parametrize('c', CASES, ids=[c['name'] for c in CASES])
def test_variables_the_reference_creates_are_the_checkpoint_names_of_the_model(c):
    own = _model(c).state_dict_tf()
    assert sorted(c['variables_created']) == sorted(own)          # the vgg16() graph: no head variables
    if c['fn'] == 'vgg16Netvlad':
        assert c['head_clusters'] == 64                           # model/nets.py:67
        head = nets.VGG16NetVLAD(vlad_cores=64).state_dict_tf()
        assert set(head) - set(own) == {'vgg16_netvlad_pca/assignment/kernel', 'vgg16_netvlad_pca/cluster_centers'}


@pytest.mark.parametrize('c', CASES, ids=[c['name'] for c in CASES])
def test_float32_composition_gives_the_executed_references_map(c):
    want = _expected(c)
    assert want.shape == (c['b'], c['h'] // 16, c['w'] // 16, 512)
    with torch.no_grad():
        got = _model(c).forward_vgg16(torch.from_numpy(_images(c))).numpy()
    assert got.shape == want.shape
    assert np.abs(got - want).max() <= 2e-5 * np.abs(want).max()
    assert (want < 0).any()                                       # conv5_3 has no ReLU (nets.py:62)


def test_the_two_graphs_share_everything_up_to_the_head():
    """vgg16Netvlad's head input is vgg16's output for the same images and variables
    (model/nets.py:12-66 and 77-130 are the same statements)."""
    c = dict(CASES[2])
    with torch.no_grad():
        got = _model(c).forward_vgg16(torch.from_numpy(_images(c))).numpy()
    assert np.abs(got - _expected(c)).max() <= 2e-5 * np.abs(got).max()


# ------------------------------------------------------------------ GPU: the HIP path
@pytest.mark.gpu
@pytest.mark.parametrize('c', CASES, ids=[c['name'] for c in CASES])
@pytest.mark.parametrize('dtype,tol', [(torch.float32, 2e-5), (torch.bfloat16, 3e-2)], ids=['f32', 'bf16'])
def test_hip_backbone_gives_the_executed_references_map(c, dtype, tol):
    dev = torch.device('cuda:0')
    want = _expected(c)
    m = _model(c, dtype).to(dev)
    with torch.no_grad():
        got = m.forward_vgg16(torch.from_numpy(_images(c)).to(dev)).float().cpu().numpy()
    assert got.shape == want.shape
    assert np.abs(got - want).max() <= tol * np.abs(want).max()


@pytest.mark.gpu
def test_hip_head_on_the_references_head_input_equals_the_whole_model():
    """The NetVLAD kernels fed the tensor the reference hands its head (already normalised:
    pre_l2 is idempotent) give what the whole model gives from the images."""
    dev = torch.device('cuda:0')
    c = CASES[2]
    m = _model(c, torch.float32, vlad_cores=64).to(dev)
    with torch.no_grad():
        whole = m(torch.from_numpy(_images(c)).to(dev))
        x = torch.from_numpy(_expected(c).copy()).to(dev)
        head = nets.netvlad(x, m.assignment_kernel, m.cluster_centers, True)
    assert whole.shape == (c['b'], 32768)
    assert (whole - head).abs().max().item() <= 1e-4 * whole.abs().max().item()
