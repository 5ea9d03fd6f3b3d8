"""TensorBoard event files written and read without TensorFlow.

The reference's trainer reports through two ``tf.summary.FileWriter``s — ``OUT_DIR/local`` (the
training loss and learning rate of every step, the localisation numbers of the training region)
and ``OUT_DIR/other`` (the other region's loss and localisation numbers) — train/train.py:304,
380-397, 929-932, 1139-1147.  A user of the reference follows a run in TensorBoard, so the trainer
here writes the same files with the same tags.

Format (third-party: TensorFlow 1.10, README.md:6; tensorflow/core/lib/io/record_writer.cc,
core/util/event.proto, core/framework/summary.proto), restated:
  file   ``events.out.tfevents.<unix seconds>.<hostname>``: a sequence of records
  record uint64 length | uint32 masked CRC-32C of those 8 bytes | payload | uint32 masked CRC-32C
         of the payload   (little endian; the mask of tf_bundle.mask_crc)
  Event  1: double wall_time, 2: int64 step, 3: string file_version (first record: "brain.Event:2"),
         5: Summary;   Summary  1: repeated Value;   Value  1: string tag, 2: float simple_value
PARITY UNPINNED (no TensorFlow / TensorBoard here): pinned by the record framing's own checksums
(RFC 3720 vectors behind tf_bundle.crc32c), a hand-assembled record in tests/test_tf_events.py and
the round trip.
"""
import os
import socket
import struct
import time

from . import tf_bundle as B


def _record(payload):
    head = struct.pack('<Q', len(payload))
    return (head + struct.pack('<I', B.mask_crc(B.crc32c(head))) + payload +
            struct.pack('<I', B.mask_crc(B.crc32c(payload))))


def _field(num, wire):
    return B.put_varint((num << 3) | wire)


def encode_event(wall_time, step=0, file_version=None, scalars=None):
    out = _field(1, 1) + struct.pack('<d', float(wall_time))
    if step:
        out += _field(2, 0) + B.put_varint(int(step))
    if file_version is not None:
        v = file_version.encode()
        out += _field(3, 2) + B.put_varint(len(v)) + v
    if scalars:
        summary = b''
        for tag, value in scalars.items():
            t = str(tag).encode()
            val = _field(1, 2) + B.put_varint(len(t)) + t + _field(2, 5) + struct.pack('<f', float(value))
            summary += _field(1, 2) + B.put_varint(len(val)) + val
        out += _field(5, 2) + B.put_varint(len(summary)) + summary
    return out


class SummaryWriter:
    """``tf.summary.FileWriter(logdir)`` for scalar summaries: ``add_scalars({'loss': 2.4}, step)``
    is ``writer.add_summary(summary, step)`` with one ``summary.value.add(tag=..., simple_value=...)``
    per entry."""

    def __init__(self, logdir, now=None):
        os.makedirs(logdir, exist_ok=True)
        now = time.time() if now is None else now
        self.path = os.path.join(logdir, 'events.out.tfevents.%010d.%s' % (int(now), socket.gethostname()))
        self._f = open(self.path, 'ab')
        self._f.write(_record(encode_event(now, file_version='brain.Event:2')))
        self._f.flush()

    def add_scalars(self, scalars, step, now=None):
        self._f.write(_record(encode_event(time.time() if now is None else now, step, None, scalars)))

    def flush(self):
        self._f.flush()

    def close(self):
        if not self._f.closed:
            self._f.close()


def read_events(path, verify=True):
    """[(wall_time, step, file_version or None, {tag: simple_value})] of an event file."""
    data = open(path, 'rb').read()
    out, pos = [], 0
    while pos < len(data):
        if pos + 12 > len(data):
            raise B.BundleError('truncated record header')
        head = data[pos:pos + 8]
        (n,), (hcrc,) = struct.unpack('<Q', head), struct.unpack('<I', data[pos + 8:pos + 12])
        payload = data[pos + 12:pos + 12 + n]
        if len(payload) != n or pos + 16 + n > len(data):
            raise B.BundleError('truncated record')
        (pcrc,) = struct.unpack('<I', data[pos + 12 + n:pos + 16 + n])
        if verify and (B.unmask_crc(hcrc) != B.crc32c(head) or B.unmask_crc(pcrc) != B.crc32c(payload)):
            raise B.BundleError('record checksum mismatch at byte %d' % pos)
        pos += 16 + n
        wall, step, version, scalars = 0.0, 0, None, {}
        for field, wt, val in B._pb_fields(payload):
            if field == 1 and wt == 1:
                (wall,) = struct.unpack('<d', struct.pack('<Q', val))
            elif field == 2 and wt == 0:
                step = B._signed(val)
            elif field == 3 and wt == 2:
                version = bytes(val).decode()
            elif field == 5 and wt == 2:
                for f2, w2, v2 in B._pb_fields(val):
                    if f2 == 1 and w2 == 2:
                        tag, num = None, None
                        for f3, w3, v3 in B._pb_fields(v2):
                            if f3 == 1 and w3 == 2:
                                tag = bytes(v3).decode()
                            elif f3 == 2 and w3 == 5:
                                (num,) = struct.unpack('<f', struct.pack('<I', v3))
                        if tag is not None and num is not None:
                            scalars[tag] = num
        out.append((wall, step, version, scalars))
    return out
