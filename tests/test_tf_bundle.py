"""TensorFlow checkpoint bundle reader / writer (soft_contrastive_learning_amd/tf_bundle.py).

No TensorFlow and no released checkpoint exists in the build container, so the format is
pinned by: published CRC-32C vectors (RFC 3720 B.4), a table assembled byte by byte in this
file (independent of write_table), a hand-made snappy stream, and round trips.
"""
import os
import struct

import numpy as np
import pytest
import torch

from soft_contrastive_learning_amd import checkpoint, tf_bundle as TB
from soft_contrastive_learning_amd.model import nets


def test_crc32c_known_answers():
    assert TB.crc32c(b'123456789') == 0xe3069283
    assert TB.crc32c(bytes(32)) == 0x8a9136aa
    assert TB.crc32c(b'\xff' * 32) == 0x62a8ab43
    assert TB.crc32c(bytes(range(32))) == 0x46dd794e
    assert TB.crc32c(bytes(range(31, -1, -1))) == 0x113fdb5c
    assert TB.crc32c(b'') == 0
    # continuation and unaligned starts
    blob = np.random.default_rng(0).integers(0, 256, 4099, dtype=np.uint8).tobytes()
    for cut in (1, 7, 8, 1000, 4098):
        assert TB.crc32c(blob[cut:], TB.crc32c(blob[:cut])) == TB.crc32c(blob)
    # bit-at-a-time definition
    def slow(data):
        c = 0xffffffff
        for b in data:
            c ^= b
            for _ in range(8):
                c = (c >> 1) ^ 0x82F63B78 if c & 1 else c >> 1
        return c ^ 0xffffffff
    assert TB.crc32c(blob[:257]) == slow(blob[:257])


def test_crc_mask_is_a_rotation_plus_delta():
    c = TB.crc32c(b'foo')
    assert TB.mask_crc(c) != c and TB.mask_crc(TB.mask_crc(c)) != c
    assert TB.unmask_crc(TB.mask_crc(c)) == c
    assert TB.unmask_crc(TB.unmask_crc(TB.mask_crc(TB.mask_crc(c)))) == c
    assert TB.mask_crc(0) == 0xa282ead8
    assert TB.mask_crc(0x00008000) == (1 + 0xa282ead8) & 0xffffffff


def test_varints():
    assert TB.put_varint(0) == b'\x00' and TB.put_varint(127) == b'\x7f'
    assert TB.put_varint(300) == b'\xac\x02'
    for n in (0, 1, 127, 128, 16383, 16384, 2 ** 32, 2 ** 63 - 1):
        assert TB.get_varint(TB.put_varint(n) + b'\x55', 0) == (n, len(TB.put_varint(n)))
    with pytest.raises(TB.BundleError):
        TB.get_varint(b'\x80\x80', 0)


def _hand_table(tensor_bytes):
    """[header, 'w' -> float32[2]] assembled from the format description, byte by byte."""
    header = bytes([0x08, 0x01, 0x1a, 0x02, 0x08, 0x01])       # num_shards=1, version{producer=1}
    crc = TB.mask_crc(TB.crc32c(tensor_bytes))
    entry = bytes([0x08, 0x01,                                 # dtype = DT_FLOAT
                   0x12, 0x04, 0x12, 0x02, 0x08, 0x02,         # shape { dim { size: 2 } }
                   0x28, 0x08,                                 # size = 8 (offset 0 omitted)
                   0x35]) + struct.pack('<I', crc)             # fixed32 crc32c
    block = bytes([0, 0, len(header)]) + header                # shared, unshared, value_len, key ''
    block += bytes([0, 1, len(entry)]) + b'w' + entry
    block += struct.pack('<II', 0, 1)                          # restart[0] = 0; one restart

    def with_trailer(b):
        return b + b'\x00' + struct.pack('<I', TB.mask_crc(TB.crc32c(b + b'\x00')))
    out = with_trailer(block)
    meta = struct.pack('<II', 0, 1)
    meta_off = len(out)
    out += with_trailer(meta)
    index = bytes([0, 1, 2]) + b'w' + bytes([0, len(block)]) + struct.pack('<II', 0, 1)
    index_off = len(out)
    out += with_trailer(index)
    footer = bytes([meta_off, len(meta), index_off, len(index)])
    footer += bytes(40 - len(footer)) + bytes([0x57, 0xfb, 0x80, 0x8b, 0x24, 0x75, 0x47, 0xdb])
    return out + footer


def test_reader_on_a_hand_assembled_bundle(tmp_path):
    w = np.array([1.5, -2.25], dtype='<f4')
    prefix = str(tmp_path / 'model')
    with open(prefix + '.index', 'wb') as f:
        f.write(_hand_table(w.tobytes()))
    with open(prefix + '.data-00000-of-00001', 'wb') as f:
        f.write(w.tobytes())
    header, entries = TB.list_variables(prefix)
    assert header == {'num_shards': 1, 'endianness': 0, 'producer': 1}
    assert entries['w']['shape'] == [2] and entries['w']['dtype'] == 1 and entries['w']['size'] == 8
    got = TB.read(prefix)
    assert list(got) == ['w'] and got['w'].dtype == np.float32
    assert got['w'].tolist() == [1.5, -2.25]
    # the writer produces the same index bytes for the same content
    TB.write(str(tmp_path / 'again'), {'w': w})
    with open(str(tmp_path / 'again.index'), 'rb') as f:
        assert f.read() == _hand_table(w.tobytes())


def test_snappy_blocks():
    # 12 bytes: literal 'abc' then a copy of 9 from offset 3 (overlapping)
    stream = bytes([12, 0x08]) + b'abc' + bytes([(5 << 2) | 1, 3])
    assert TB.snappy_uncompress(stream) == b'abc' * 4
    # 2-byte-offset copy and a long literal (length byte follows the tag)
    lit = bytes(range(70))
    stream = bytes([74, 60 << 2, 69]) + lit + bytes([(3 << 2) | 2, 70, 0])
    assert TB.snappy_uncompress(stream) == lit + lit[:4]
    with pytest.raises(TB.BundleError):
        TB.snappy_uncompress(bytes([5, 0x08]) + b'abc')


def test_round_trip_many_variables_and_dtypes(tmp_path, monkeypatch):
    monkeypatch.setattr(TB, 'BLOCK_SIZE', 4096)       # many index blocks without megabytes of names
    rng = np.random.default_rng(1)
    tensors = {}
    for i in range(3000):                     # restarts, shared prefixes, block boundaries
        tensors['vgg16_netvlad_pca/some/long/scope/name_%05d/kernel' % i] = \
            rng.standard_normal(3).astype(np.float32)
    tensors['Variable'] = np.asarray(123456789012, dtype=np.int64)
    tensors['a/scalar'] = np.asarray(2.5, dtype=np.float64)
    tensors['a/flags'] = np.array([True, False, True])
    tensors['a/empty'] = np.zeros((0, 4), dtype=np.float32)
    tensors['z/big'] = rng.standard_normal((3, 3, 64, 128)).astype(np.float32)
    prefix = str(tmp_path / 'ckpt-5')
    TB.write(prefix, tensors)
    assert os.path.getsize(prefix + '.index') > 16 * TB.BLOCK_SIZE
    header, entries = TB.list_variables(prefix)
    assert set(entries) == set(tensors) and header['num_shards'] == 1
    got = TB.read(prefix)
    for k, v in tensors.items():
        assert got[k].dtype == v.dtype and got[k].shape == v.shape
        assert np.array_equal(got[k], v)
    only = TB.read(prefix, names=lambda n: n.startswith('a/'))
    assert sorted(only) == ['a/empty', 'a/flags', 'a/scalar']
    with pytest.raises(KeyError):
        TB.read(prefix, names=['nope'])


def test_corruption_is_detected(tmp_path):
    prefix = str(tmp_path / 'c')
    TB.write(prefix, {'w': np.arange(64, dtype=np.float32)})
    data = prefix + '.data-00000-of-00001'
    raw = bytearray(open(data, 'rb').read())
    raw[17] ^= 0x40
    open(data, 'wb').write(bytes(raw))
    with pytest.raises(TB.BundleError, match='tensor checksum'):
        TB.read(prefix)
    assert TB.read(prefix, verify=False)['w'].shape == (64,)
    idx = bytearray(open(prefix + '.index', 'rb').read())
    idx[5] ^= 0x01
    open(prefix + '.index', 'wb').write(bytes(idx))
    with pytest.raises(TB.BundleError, match='block checksum'):
        TB.read(prefix)
    open(prefix + '.index', 'wb').write(b'not a table')
    with pytest.raises(TB.BundleError):
        TB.read(prefix)


def test_model_checkpoint_in_tf_format_with_adam_slots(tmp_path):
    a, b = nets.VGG16NetVLAD(seed=5), nets.VGG16NetVLAD(seed=6)
    oa = torch.optim.Adam(a.parameters(), lr=1e-3)
    ob = torch.optim.Adam(b.parameters(), lr=1e-3)
    g = torch.Generator().manual_seed(0)
    for _ in range(3):
        for p in a.parameters():
            p.grad = torch.randn(p.shape, generator=g) * 1e-2
        oa.step()
    stem = str(tmp_path / 'run' / 'checkpoint-3')
    assert checkpoint.save(a, stem, global_step=3, optimizer=oa) == stem
    header, entries = TB.list_variables(stem)
    assert entries['vgg16_netvlad_pca/conv1_1/kernel']['shape'] == [3, 3, 3, 64]       # HWIO
    assert entries['vgg16_netvlad_pca/conv1_1/kernel/Adam_1']['shape'] == [3, 3, 3, 64]
    assert entries['vgg16_netvlad_pca/assignment/kernel']['shape'] == [1, 1, 512, 64]
    assert entries['vgg16_netvlad_pca/cluster_centers']['shape'] == [1, 1, 1, 512, 64]
    # ops['step'] = tf.Variable(0) is int32 (train/train.py:655): DT_INT32 = 3
    assert entries['Variable']['dtype'] == 3 and entries['Variable']['shape'] == []
    v = TB.read(stem, names=['beta1_power', 'beta2_power'])
    assert np.isclose(v['beta1_power'], 0.9 ** 4) and np.isclose(v['beta2_power'], 0.999 ** 4)
    assert checkpoint.load(b, stem, optimizer=ob) == 3
    for k, t in a.state_dict_tf().items():
        assert torch.equal(t, b.state_dict_tf()[k])
    for pa, pb in zip(a.parameters(), b.parameters()):
        assert torch.equal(oa.state[pa]['exp_avg'], ob.state[pb]['exp_avg'])
        assert torch.equal(oa.state[pa]['exp_avg_sq'], ob.state[pb]['exp_avg_sq'])
        assert int(ob.state[pb]['step']) == 3
    # the next update is identical on both sides
    for pa, pb in zip(a.parameters(), b.parameters()):
        pa.grad = torch.randn(pa.shape, generator=g) * 1e-2
        pb.grad = pa.grad.clone()
    oa.step()
    ob.step()
    for pa, pb in zip(a.parameters(), b.parameters()):
        assert torch.equal(pa, pb)


def test_checkpoint_state_file(tmp_path):
    d = str(tmp_path)
    assert TB.latest_checkpoint(d) is None
    TB.write(os.path.join(d, 'checkpoint-10'), {'w': np.zeros(2, dtype=np.float32)})
    TB.write_state(d, 'checkpoint-10', ['checkpoint-5', 'checkpoint-10'])
    text = open(os.path.join(d, 'checkpoint')).read()
    assert text.splitlines()[0] == 'model_checkpoint_path: "checkpoint-10"'
    assert 'all_model_checkpoint_paths: "checkpoint-5"' in text
    assert TB.latest_checkpoint(d) == os.path.join(d, 'checkpoint-10')
    TB.remove(os.path.join(d, 'checkpoint-10'))
    assert TB.latest_checkpoint(d) is None


def test_saver_rotates_only_what_it_wrote_in_save_order(tmp_path):
    """tf.train.Saver keeps its own _last_checkpoints in save order: a higher-numbered
    checkpoint left by an earlier run must neither be deleted nor cause the fresh one to be."""
    d = str(tmp_path)
    m = nets.VGG16NetVLAD(seed=1)
    TB.write(os.path.join(d, 'checkpoint-900'), {'w': np.zeros(2, dtype=np.float32)})   # earlier run
    sv = checkpoint.Saver(d, max_to_keep=1)
    sv.save_rolling(m, 100)
    assert TB.exists(os.path.join(d, 'checkpoint-100')) and TB.exists(os.path.join(d, 'checkpoint-900'))
    assert TB.latest_checkpoint(d) == os.path.join(d, 'checkpoint-100')
    sv.save_rolling(m, 200)
    assert not TB.exists(os.path.join(d, 'checkpoint-100'))
    assert TB.exists(os.path.join(d, 'checkpoint-200')) and TB.exists(os.path.join(d, 'checkpoint-900'))
    assert TB.latest_checkpoint(d) == os.path.join(d, 'checkpoint-200')
    sv.save_epoch(m, 0, 200)
    sv.save_epoch(m, 1, 400)                       # epoch saver keeps everything
    assert TB.exists(os.path.join(d, 'epoch-checkpoint-0')) and TB.exists(os.path.join(d, 'epoch-checkpoint-1'))
