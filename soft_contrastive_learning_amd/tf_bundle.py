"""TensorFlow checkpoint bundles (``<prefix>.index`` + ``<prefix>.data-00000-of-00001``)
read and written without TensorFlow.

The reference restores released models with ``tf.train.Saver(...).restore(sess, CHECKPOINT)``
(train/train.py:882-905, evaluation/inference.py:122-144) and writes its own with
``saver.save(sess, os.path.join(OUT_DIR, 'checkpoint'), global_step=...)``
(train/train.py:984, 1079, 1102).  Both go through TensorFlow's TensorBundle V2 format
(tensorflow 1.10, core/util/tensor_bundle/tensor_bundle.cc — not part of /root/reference;
restated here from its published on-disk format):

* ``.index`` is a LevelDB-style sorted string table (core/lib/io/table_builder.cc,
  format.cc): prefix-compressed key/value blocks with restart points, each followed by a
  5-byte trailer (compression type, masked CRC-32C), an index block, an empty metaindex
  block and a 48-byte footer ending in the magic ``0xdb4775248b80fb57``.
  Key ``""`` holds a ``BundleHeaderProto`` (num_shards, endianness, version), every other
  key is a variable name holding a ``BundleEntryProto`` (dtype, shape, shard_id, offset,
  size, masked crc32c of the tensor bytes).
* ``.data-SSSSS-of-NNNNN`` holds the raw little-endian tensor bytes at those offsets.
* the directory's ``checkpoint`` file is a text ``CheckpointState`` proto.

PARITY UNPINNED: no TensorFlow and no released checkpoint exists in the build container,
so this module is checked by round trips, by hand-assembled tables (tests/test_tf_bundle.py)
and by the CRC-32C / varint known answers only.
"""
import os
import re
import struct

import numpy as np

from . import _lib

TABLE_MAGIC = 0xdb4775248b80fb57
FOOTER_LEN = 48
BLOCK_TRAILER = 5
BLOCK_SIZE = 262144            # table::Options::block_size in TF
RESTART_INTERVAL = 16
MASK_DELTA = 0xa282ead8

# tensorflow/core/framework/types.proto
DT_OF_NUMPY = {
    np.dtype('float32'): 1, np.dtype('float64'): 2, np.dtype('int32'): 3, np.dtype('uint8'): 4,
    np.dtype('int16'): 5, np.dtype('int8'): 6, np.dtype('int64'): 9, np.dtype('bool'): 10,
    np.dtype('uint16'): 17, np.dtype('float16'): 19, np.dtype('uint32'): 22,
    np.dtype('uint64'): 23,
}
NUMPY_OF_DT = {v: k for k, v in DT_OF_NUMPY.items()}
DT_STRING, DT_BFLOAT16 = 7, 14


class BundleError(ValueError):
    pass


# ----------------------------------------------------------------------------- checksums
def crc32c(data, crc=0):
    """CRC-32C of a bytes-like object (host routine of libscl_hip.so)."""
    mv = memoryview(data).cast('B')
    if len(mv) == 0:
        return crc & 0xffffffff
    buf = np.frombuffer(mv, dtype=np.uint8)
    return int(_lib.load().scl_crc32c(crc, buf.ctypes.data, buf.size)) & 0xffffffff


def mask_crc(crc):
    """crc32c::Mask — stored CRCs are rotated and offset so that CRCs of CRCs stay sound."""
    return ((((crc >> 15) | (crc << 17)) & 0xffffffff) + MASK_DELTA) & 0xffffffff


def unmask_crc(masked):
    rot = (masked - MASK_DELTA) & 0xffffffff
    return ((rot >> 17) | (rot << 15)) & 0xffffffff


# ------------------------------------------------------------------------------- varints
def put_varint(n):
    if n < 0:
        n += 1 << 64
    out = bytearray()
    while n >= 0x80:
        out.append((n & 0x7f) | 0x80)
        n >>= 7
    out.append(n)
    return bytes(out)


def get_varint(buf, pos):
    shift = result = 0
    while True:
        if pos >= len(buf):
            raise BundleError('truncated varint')
        b = buf[pos]
        pos += 1
        result |= (b & 0x7f) << shift
        if b < 0x80:
            return result, pos
        shift += 7
        if shift > 63:
            raise BundleError('varint too long')


# ------------------------------------------------------------------- protobuf (minimal)
def _pb_fields(buf):
    """Yield (field number, wire type, value) of one serialized message."""
    pos = 0
    while pos < len(buf):
        tag, pos = get_varint(buf, pos)
        field, wt = tag >> 3, tag & 7
        if wt == 0:
            val, pos = get_varint(buf, pos)
        elif wt == 1:
            val = struct.unpack_from('<Q', buf, pos)[0]
            pos += 8
        elif wt == 2:
            n, pos = get_varint(buf, pos)
            val = bytes(buf[pos:pos + n])
            if len(val) != n:
                raise BundleError('truncated length-delimited field')
            pos += n
        elif wt == 5:
            val = struct.unpack_from('<I', buf, pos)[0]
            pos += 4
        else:
            raise BundleError('unsupported protobuf wire type %d' % wt)
        yield field, wt, val


def _signed(v):
    return v - (1 << 64) if v >= (1 << 63) else v


def _pb_varint_field(field, value):
    return put_varint(field << 3) + put_varint(value)


def _pb_bytes_field(field, payload):
    return put_varint((field << 3) | 2) + put_varint(len(payload)) + payload


def encode_header(num_shards=1):
    # BundleHeaderProto{num_shards=1; endianness=LITTLE(0, default: omitted); version{producer=1}}
    return _pb_varint_field(1, num_shards) + _pb_bytes_field(3, _pb_varint_field(1, 1))


def decode_header(buf):
    h = {'num_shards': 0, 'endianness': 0, 'producer': 0}
    for f, _, v in _pb_fields(buf):
        if f == 1:
            h['num_shards'] = v
        elif f == 2:
            h['endianness'] = v
        elif f == 3:
            for f2, _, v2 in _pb_fields(v):
                if f2 == 1:
                    h['producer'] = v2
    return h


def encode_entry(dtype, shape, shard_id, offset, size, masked_crc):
    dims = b''.join(_pb_bytes_field(2, _pb_varint_field(1, int(d))) for d in shape)
    out = _pb_varint_field(1, dtype) + _pb_bytes_field(2, dims)
    if shard_id:
        out += _pb_varint_field(3, shard_id)
    if offset:
        out += _pb_varint_field(4, offset)
    if size:
        out += _pb_varint_field(5, size)
    if masked_crc:
        out += put_varint((6 << 3) | 5) + struct.pack('<I', masked_crc)
    return out


def decode_entry(buf):
    e = {'dtype': 0, 'shape': [], 'shard_id': 0, 'offset': 0, 'size': 0, 'crc32c': 0,
         'sliced': False}
    for f, _, v in _pb_fields(buf):
        if f == 1:
            e['dtype'] = v
        elif f == 2:
            for f2, _, v2 in _pb_fields(v):
                if f2 == 2:          # Dim
                    size = 0
                    for f3, _, v3 in _pb_fields(v2):
                        if f3 == 1:
                            size = _signed(v3)
                    e['shape'].append(size)
                elif f2 == 3 and v2:
                    raise BundleError('tensor of unknown rank in checkpoint')
        elif f == 3:
            e['shard_id'] = v
        elif f == 4:
            e['offset'] = _signed(v)
        elif f == 5:
            e['size'] = _signed(v)
        elif f == 6:
            e['crc32c'] = v
        elif f == 7:
            e['sliced'] = True
    return e


# ---------------------------------------------------------------------------- snappy (read)
def snappy_uncompress(buf):
    """Raw snappy block format (index blocks may be compressed with it)."""
    n, pos = get_varint(buf, 0)
    out = bytearray()
    while pos < len(buf):
        tag = buf[pos]
        pos += 1
        kind = tag & 3
        if kind == 0:
            ln = tag >> 2
            if ln >= 60:
                nb = ln - 59
                ln = int.from_bytes(buf[pos:pos + nb], 'little')
                pos += nb
            ln += 1
            out += buf[pos:pos + ln]
            pos += ln
            continue
        if kind == 1:
            ln = ((tag >> 2) & 7) + 4
            off = ((tag >> 5) << 8) | buf[pos]
            pos += 1
        elif kind == 2:
            ln = (tag >> 2) + 1
            off = int.from_bytes(buf[pos:pos + 2], 'little')
            pos += 2
        else:
            ln = (tag >> 2) + 1
            off = int.from_bytes(buf[pos:pos + 4], 'little')
            pos += 4
        if off == 0 or off > len(out):
            raise BundleError('corrupt snappy block')
        for _ in range(ln):          # copies may overlap their own output
            out.append(out[-off])
    if len(out) != n:
        raise BundleError('snappy length mismatch')
    return bytes(out)


# --------------------------------------------------------------------------- table blocks
class _BlockBuilder:
    def __init__(self):
        self.buf = bytearray()
        self.restarts = [0]
        self.counter = 0
        self.last_key = b''

    def add(self, key, value):
        shared = 0
        if self.counter < RESTART_INTERVAL:
            lim = min(len(key), len(self.last_key))
            while shared < lim and key[shared] == self.last_key[shared]:
                shared += 1
        else:
            self.restarts.append(len(self.buf))
            self.counter = 0
        self.buf += put_varint(shared) + put_varint(len(key) - shared) + put_varint(len(value))
        self.buf += key[shared:] + value
        self.last_key = key
        self.counter += 1

    def size_estimate(self):
        return len(self.buf) + 4 * len(self.restarts) + 4

    def empty(self):
        return not self.buf

    def finish(self):
        return bytes(self.buf) + b''.join(struct.pack('<I', r) for r in self.restarts) + \
            struct.pack('<I', len(self.restarts))


def _block_entries(block):
    """Iterate (key, value) over one uncompressed block."""
    if len(block) < 4:
        raise BundleError('block too small')
    nrestart = struct.unpack_from('<I', block, len(block) - 4)[0]
    limit = len(block) - 4 - 4 * nrestart
    if limit < 0:
        raise BundleError('bad restart array')
    pos, key = 0, b''
    while pos < limit:
        shared, pos = get_varint(block, pos)
        unshared, pos = get_varint(block, pos)
        vlen, pos = get_varint(block, pos)
        if shared > len(key) or pos + unshared + vlen > limit:
            raise BundleError('corrupt block entry')
        key = key[:shared] + bytes(block[pos:pos + unshared])
        pos += unshared
        yield key, bytes(block[pos:pos + vlen])
        pos += vlen


def _handle(offset, size):
    return put_varint(offset) + put_varint(size)


def _read_block(data, offset, size, verify):
    end = offset + size + BLOCK_TRAILER
    if end > len(data):
        raise BundleError('block handle past the end of the index file')
    body, ctype = data[offset:offset + size], data[offset + size]
    if verify:
        stored = unmask_crc(struct.unpack_from('<I', data, offset + size + 1)[0])
        if stored != crc32c(data[offset:offset + size + 1]):
            raise BundleError('index block checksum mismatch')
    if ctype == 0:
        return body
    if ctype == 1:
        return snappy_uncompress(body)
    raise BundleError('unknown block compression %d' % ctype)


def read_table(data, verify=True):
    """All (key, value) pairs of a sorted string table, in key order."""
    if len(data) < FOOTER_LEN:
        raise BundleError('index file shorter than a table footer')
    footer = data[-FOOTER_LEN:]
    if struct.unpack_from('<Q', footer, FOOTER_LEN - 8)[0] != TABLE_MAGIC:
        raise BundleError('not a TensorFlow checkpoint index (bad table magic)')
    _, pos = get_varint(footer, 0)           # metaindex handle (unused)
    _, pos = get_varint(footer, pos)
    ioff, pos = get_varint(footer, pos)
    isize, pos = get_varint(footer, pos)
    out = []
    for _, handle in _block_entries(_read_block(data, ioff, isize, verify)):
        boff, p = get_varint(handle, 0)
        bsize, _ = get_varint(handle, p)
        out.extend(_block_entries(_read_block(data, boff, bsize, verify)))
    return out


def write_table(items):
    """Serialize sorted (key, value) pairs the way TableBuilder does (no compression)."""
    out = bytearray()
    index = _BlockBuilder()
    block = _BlockBuilder()

    def emit(builder):
        body = builder.finish()
        off = len(out)
        out.extend(body)
        out.append(0)                                        # kNoCompression
        out.extend(struct.pack('<I', mask_crc(crc32c(body + b'\x00'))))
        return off, len(body)

    prev = None
    for key, value in items:
        if prev is not None and key <= prev:
            raise BundleError('table keys must be strictly increasing')
        block.add(key, value)
        prev = key
        if block.size_estimate() >= BLOCK_SIZE:
            off, size = emit(block)
            index.add(block.last_key, _handle(off, size))
            block = _BlockBuilder()
    if not block.empty():
        off, size = emit(block)
        index.add(block.last_key, _handle(off, size))
    moff, msize = emit(_BlockBuilder())                      # empty metaindex block
    ioff, isize = emit(index)
    footer = _handle(moff, msize) + _handle(ioff, isize)
    footer += bytes(FOOTER_LEN - 8 - len(footer)) + struct.pack('<Q', TABLE_MAGIC)
    out.extend(footer)
    return bytes(out)


# -------------------------------------------------------------------------------- bundles
def _data_name(prefix, shard, num_shards):
    return '%s.data-%05d-of-%05d' % (prefix, shard, num_shards)


def list_variables(prefix, verify=True):
    """name -> entry dict (dtype, shape, shard_id, offset, size, crc32c) plus the header."""
    with open(prefix + '.index', 'rb') as f:
        data = f.read()
    header, entries = None, {}
    for key, value in read_table(data, verify):
        if key == b'':
            header = decode_header(value)
        else:
            entries[key.decode('utf-8')] = decode_entry(value)
    if header is None:
        raise BundleError('checkpoint index has no bundle header')
    if header['endianness'] != 0:
        raise BundleError('big-endian checkpoints are not supported')
    return header, entries


def read(prefix, names=None, verify=True):
    """Load variables of a checkpoint bundle as NumPy arrays.

    names: iterable of variable names or a predicate ``name -> bool``; None loads every
    numeric variable (string tensors and partitioned variables are skipped / refused).
    bfloat16 tensors come back as uint16 bit patterns."""
    header, entries = list_variables(prefix, verify)
    if names is None:
        want = [n for n, e in entries.items() if e['dtype'] != DT_STRING]
    elif callable(names):
        want = [n for n in entries if names(n)]
    else:
        want = list(names)
    shards, out = {}, {}
    try:
        for name in want:
            if name not in entries:
                raise KeyError('variable %r not in checkpoint %s' % (name, prefix))
            e = entries[name]
            if e['sliced']:
                raise BundleError('partitioned variable %r is not supported' % name)
            if e['dtype'] == DT_BFLOAT16:
                dt = np.dtype('uint16')
            elif e['dtype'] in NUMPY_OF_DT:
                dt = NUMPY_OF_DT[e['dtype']]
            else:
                raise BundleError('variable %r has unsupported dtype enum %d' % (name, e['dtype']))
            count = int(np.prod(e['shape'], dtype=np.int64)) if e['shape'] else 1
            if count * dt.itemsize != e['size']:
                raise BundleError('variable %r: %d bytes stored for shape %s'
                                  % (name, e['size'], e['shape']))
            sid = e['shard_id']
            if sid not in shards:
                shards[sid] = open(_data_name(prefix, sid, header['num_shards']), 'rb')
            f = shards[sid]
            f.seek(e['offset'])
            raw = f.read(e['size'])
            if len(raw) != e['size']:
                raise BundleError('variable %r: data shard truncated' % name)
            if verify and unmask_crc(e['crc32c']) != crc32c(raw):
                raise BundleError('variable %r: tensor checksum mismatch' % name)
            out[name] = np.frombuffer(raw, dtype=dt).reshape(e['shape']).copy()
    finally:
        for f in shards.values():
            f.close()
    return out


def write(prefix, tensors):
    """Write ``{name: array}`` as a single-shard bundle (what tf.train.Saver produces for an
    unpartitioned graph); returns the two file names."""
    names = sorted(tensors, key=lambda s: s.encode('utf-8'))
    os.makedirs(os.path.dirname(os.path.abspath(prefix)) or '.', exist_ok=True)
    data_name = _data_name(prefix, 0, 1)
    items = [(b'', encode_header(1))]
    offset = 0
    with open(data_name + '.tmp', 'wb') as f:
        for name in names:
            if not name:
                raise BundleError('empty variable name')
            a = np.asarray(tensors[name])
            if a.ndim and not a.flags.c_contiguous:
                a = np.ascontiguousarray(a)
            if a.dtype not in DT_OF_NUMPY:
                raise BundleError('variable %r: dtype %s cannot be stored' % (name, a.dtype))
            a = a.astype(a.dtype.newbyteorder('<'), copy=False)
            raw = a.tobytes()
            f.write(raw)
            items.append((name.encode('utf-8'),
                          encode_entry(DT_OF_NUMPY[np.dtype(a.dtype.name)], a.shape, 0, offset,
                                       len(raw), mask_crc(crc32c(raw)))))
            offset += len(raw)
    with open(prefix + '.index.tmp', 'wb') as f:
        f.write(write_table(items))
    os.replace(data_name + '.tmp', data_name)
    os.replace(prefix + '.index.tmp', prefix + '.index')
    return prefix + '.index', data_name


def exists(prefix):
    return os.path.isfile(prefix + '.index')


def remove(prefix):
    for f in [prefix + '.index', prefix + '.meta'] + \
            [os.path.join(os.path.dirname(prefix) or '.', n)
             for n in os.listdir(os.path.dirname(prefix) or '.')
             if n.startswith(os.path.basename(prefix) + '.data-')]:
        if os.path.isfile(f):
            os.remove(f)


# ---------------------------------------------------------------- CheckpointState text file
def write_state(directory, latest, all_paths, filename='checkpoint'):
    """The ``checkpoint`` file tf.train.Saver keeps next to its bundles (relative paths)."""
    lines = ['model_checkpoint_path: "%s"' % latest]
    lines += ['all_model_checkpoint_paths: "%s"' % p for p in all_paths]
    tmp = os.path.join(directory, filename + '.tmp')
    with open(tmp, 'w') as f:
        f.write('\n'.join(lines) + '\n')
    os.replace(tmp, os.path.join(directory, filename))


def latest_checkpoint(directory, filename='checkpoint'):
    """tf.train.latest_checkpoint: prefix named by the state file, or None."""
    path = os.path.join(directory, filename)
    if not os.path.isfile(path):
        return None
    with open(path) as f:
        m = re.search(r'^model_checkpoint_path:\s*"((?:[^"\\]|\\.)*)"', f.read(), re.M)
    if not m:
        return None
    p = m.group(1)
    p = p if os.path.isabs(p) else os.path.join(directory, p)
    return p if exists(p) else None
