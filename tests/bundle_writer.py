"""Test-only writer of TensorFlow tensor bundles (``variables.index`` + one data shard) and of the
TrackableObjectGraph inside them, written from the published format description independently
of the reader in pclsegmentation_amd/savedmodel.py (LevelDB table: prefix-compressed blocks with
restart points every 16 entries, block trailer = type byte + masked CRC32C, index block, footer).
It exists because no TensorFlow is available to produce real fixtures; it is not product code.
"""
import os
import struct

import numpy as np

from pclsegmentation_amd.savedmodel import crc32c, mask_crc

MAGIC = 0xdb4775248b80fb57


def varint(n):
  out = bytearray()
  while True:
    b = n & 0x7F
    n >>= 7
    if n:
      out.append(b | 0x80)
    else:
      out.append(b)
      return bytes(out)


def field(num, wt, payload):
  key = varint((num << 3) | wt)
  if wt == 0:
    return key + varint(payload)
  if wt == 2:
    return key + varint(len(payload)) + payload
  if wt == 5:
    return key + struct.pack("<I", payload)
  raise ValueError(wt)


def snappy_literal_only(data):
  """A valid snappy stream made of literals (enough to exercise the reader's snappy path)."""
  out = bytearray(varint(len(data)))
  pos = 0
  while pos < len(data):
    chunk = data[pos:pos + 60]
    out.append((len(chunk) - 1) << 2)
    out += chunk
    pos += len(chunk)
  return bytes(out)


class TableWriter:
  def __init__(self, block_size=512, restart_interval=16, snappy=False):
    self.block_size, self.restart_interval, self.snappy = block_size, restart_interval, snappy
    self.file = bytearray()
    self.index = []          # (last key, offset, size)
    self._reset()

  def _reset(self):
    self.buf, self.restarts, self.count, self.last = bytearray(), [0], 0, b""

  def add(self, key, value):
    shared = 0
    if self.count % self.restart_interval == 0 and self.count:
      self.restarts.append(len(self.buf))
    elif self.count:
      while shared < min(len(key), len(self.last)) and key[shared] == self.last[shared]:
        shared += 1
    self.buf += varint(shared) + varint(len(key) - shared) + varint(len(value)) + key[shared:] + value
    self.last = key
    self.count += 1
    if len(self.buf) >= self.block_size:
      self._flush()

  def _emit(self, body):
    ctype = 0
    if self.snappy:
      body, ctype = snappy_literal_only(body), 1
    off = len(self.file)
    self.file += body + bytes([ctype])
    self.file += struct.pack("<I", mask_crc(crc32c(bytes(self.file[off:]))))
    return off, len(body)

  def _flush(self):
    if not self.count:
      return
    body = bytes(self.buf) + b"".join(struct.pack("<I", r) for r in self.restarts) + struct.pack("<I", len(self.restarts))
    off, size = self._emit(body)
    self.index.append((self.last, off, size))
    self._reset()

  def finish(self):
    self._flush()
    empty = struct.pack("<I", 0) + struct.pack("<I", 1)
    moff, msize = self._emit(empty)                       # metaindex: no entries
    blocks, self.index = self.index, []
    self._reset()
    self.block_size = 1 << 30
    for key, off, size in blocks:
      self.add(key, varint(off) + varint(size))
    body = bytes(self.buf) + b"".join(struct.pack("<I", r) for r in self.restarts) + struct.pack("<I", len(self.restarts))
    ioff, isize = self._emit(body)
    footer = varint(moff) + varint(msize) + varint(ioff) + varint(isize)
    footer += b"\0" * (40 - len(footer)) + struct.pack("<Q", MAGIC)
    return bytes(self.file) + footer


def shape_proto(shape):
  return b"".join(field(2, 2, field(1, 0, int(d))) for d in shape)


def object_graph(paths_to_keys, extra_children=()):
  """TrackableObjectGraph for variables addressed by attribute path -> checkpoint key.  Every
  directory level becomes a node; ``extra_children`` adds (parent path, alias name, target path)
  edges such as Keras' ``layer_with_weights-N``."""
  nodes = {"": 0}
  order = [""]
  for p in paths_to_keys:
    parts = p.split("/")
    for i in range(1, len(parts) + 1):
      q = "/".join(parts[:i])
      if q not in nodes:
        nodes[q] = len(order)
        order.append(q)
  children = {q: [] for q in order}
  for q in order[1:]:
    parent = q.rsplit("/", 1)[0] if "/" in q else ""
    children[parent].append((q.rsplit("/", 1)[-1], nodes[q]))
  for parent, alias, target in extra_children:
    children[parent].insert(0, (alias, nodes[target]))
  out = b""
  for q in order:
    body = b"".join(field(1, 2, field(1, 0, nid) + field(2, 2, name.encode())) for name, nid in children[q])
    if q in paths_to_keys:
      body += field(2, 2, field(1, 2, b"VARIABLE_VALUE") + field(2, 2, q.encode()) +
                    field(3, 2, paths_to_keys[q].encode()))
    out += field(1, 2, body)
  return out


def write_bundle(prefix, tensors, graph_bytes=None, block_size=512, snappy=False):
  """tensors: {checkpoint key: ndarray}."""
  os.makedirs(os.path.dirname(prefix), exist_ok=True)
  data = bytearray()
  entries = {}
  for key in sorted(tensors):
    a = np.ascontiguousarray(tensors[key])
    dt = {np.dtype("float32"): 1, np.dtype("float64"): 2, np.dtype("int32"): 3, np.dtype("int64"): 9}[a.dtype]
    raw = a.tobytes()
    entries[key] = (field(1, 0, dt) + field(2, 2, shape_proto(a.shape)) + field(4, 0, len(data)) +
                    field(5, 0, len(raw)) + field(6, 5, mask_crc(crc32c(raw))))
    data += raw
  if graph_bytes is not None:
    lens = varint(len(graph_bytes))
    raw = lens + struct.pack("<I", mask_crc(crc32c(lens))) + graph_bytes
    entries["_CHECKPOINTABLE_OBJECT_GRAPH"] = (field(1, 0, 7) + field(2, 2, b"") + field(4, 0, len(data)) +
                                               field(5, 0, len(raw)) + field(6, 5, mask_crc(crc32c(raw))))
    data += raw
  tw = TableWriter(block_size=block_size, snappy=snappy)
  tw.add(b"", field(1, 0, 1) + field(3, 2, field(1, 0, 1)))     # header: num_shards = 1, version.producer = 1
  for key in sorted(entries, key=lambda k: k.encode()):
    tw.add(key.encode(), entries[key])
  with open(prefix + ".index", "wb") as f:
    f.write(tw.finish())
  with open(prefix + ".data-00000-of-00001", "wb") as f:
    f.write(bytes(data))
