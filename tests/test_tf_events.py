"""TensorBoard event files (soft_contrastive_learning_amd/tf_events.py): the record framing and the
Event / Summary messages the reference's tf.summary.FileWriter produces (train/train.py:304,
380-397, 929-932, 1139-1147), checked on a hand-assembled record and by round trip."""
import os
import struct

import pytest

from soft_contrastive_learning_amd import tf_bundle as B
from soft_contrastive_learning_amd import tf_events as E


def test_hand_assembled_scalar_event():
    # Event{wall_time = 2.0, step = 3, summary{value{tag = "loss", simple_value = 0.5}}}
    value = bytes([0x0a, 4]) + b'loss' + bytes([0x15]) + struct.pack('<f', 0.5)     # 1: tag, 2: fixed32
    summary = bytes([0x0a, len(value)]) + value                                       # 1: Value
    event = bytes([0x09]) + struct.pack('<d', 2.0) + bytes([0x10, 3]) + bytes([0x2a, len(summary)]) + summary
    assert E.encode_event(2.0, 3, None, {'loss': 0.5}) == event
    rec = E._record(event)
    assert rec[:8] == struct.pack('<Q', len(event)) and len(rec) == len(event) + 16
    assert B.unmask_crc(struct.unpack('<I', rec[8:12])[0]) == B.crc32c(rec[:8])
    assert B.unmask_crc(struct.unpack('<I', rec[-4:])[0]) == B.crc32c(event)
    # the version record every event file starts with
    first = E.encode_event(1.5, 0, 'brain.Event:2')
    assert first == bytes([0x09]) + struct.pack('<d', 1.5) + bytes([0x1a, 13]) + b'brain.Event:2'


def test_writer_round_trip_and_corruption(tmp_path):
    w = E.SummaryWriter(str(tmp_path / 'local'), now=1700000000.25)
    assert os.path.basename(w.path).startswith('events.out.tfevents.1700000000.')
    w.add_scalars({'loss': 2.4375, 'learning_rate': 5e-6}, 1, now=1700000001.0)
    w.add_scalars({'50m-auc@Top1': 1234.5, '%<50m@Top1': 87.5}, 100, now=1700000002.0)
    w.close()
    ev = E.read_events(w.path)
    assert [e[1] for e in ev] == [0, 1, 100] and ev[0][2] == 'brain.Event:2' and ev[0][3] == {}
    assert ev[1][0] == 1700000001.0 and ev[1][3]['loss'] == 2.4375
    assert abs(ev[1][3]['learning_rate'] - 5e-6) < 1e-12                      # float32 on disk
    assert ev[2][3] == {'50m-auc@Top1': 1234.5, '%<50m@Top1': 87.5}
    raw = bytearray(open(w.path, 'rb').read())
    raw[-6] ^= 1
    bad = tmp_path / 'bad'
    bad.write_bytes(bytes(raw))
    with pytest.raises(B.BundleError):
        E.read_events(str(bad))
    assert len(E.read_events(str(bad), verify=False)) == 3
    bad.write_bytes(bytes(raw[:-3]))
    with pytest.raises(B.BundleError):
        E.read_events(str(bad))
