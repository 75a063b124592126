"""TensorFlow-free reader for the weights of a reference SavedModel directory.

The reference stores a trained model with ``model.save(dir)`` (train.py:60) and restores it with
``tf.keras.models.load_model(dir)`` (inference.py:39, eval.py:40).  The tensors live in
``<dir>/variables/variables.index`` + ``variables.data-00000-of-00001`` — TensorFlow's *tensor
bundle*.  This module reads that pair with NumPy only and binds the tensors to the engine's
Keras-attribute-path names (``fire2/squeeze/kernel`` ...).

Format, as published in the TensorFlow sources (r2.9):
  * ``.index`` is a LevelDB-format table (core/lib/io/format.cc, table.cc, block.cc): data blocks
    of prefix-compressed (key, value) entries with a restart array, an index block, a 48-byte
    footer ending in the magic 0xdb4775248b80fb57; every block is followed by a 1-byte
    compression type (0 none, 1 snappy) and a masked CRC32C.
  * key "" -> BundleHeaderProto, every other key -> BundleEntryProto {dtype, shape, shard_id,
    offset, size, crc32c} (core/protobuf/tensor_bundle.proto); the bytes are at ``offset`` of data
    shard ``shard_id`` (core/util/tensor_bundle/tensor_bundle.cc).
  * key "_CHECKPOINTABLE_OBJECT_GRAPH" holds a serialized TrackableObjectGraph
    (core/protobuf/trackable_object_graph.proto): the object tree of the saved model.  Weights are
    resolved by walking it along the SAME attribute names the reference's classes use
    (root -> "fire2" -> "squeeze" -> "kernel"), so the result does not depend on which of its
    aliases ("layer_with_weights-N/..." or "fire2/squeeze/...") TensorFlow chose as checkpoint key.

PARITY UNPINNED: there is no TensorFlow in the build image, so no TensorFlow-written file was
available to check this reader against; the tests exercise it on bundles produced by an
independent writer of the same published format (tests/bundle_writer.py).  Where TensorFlow is at
hand, ``tools/export_tf_weights.py`` is the conservative route.
"""
import os
import struct

import numpy as np

TABLE_MAGIC = 0xdb4775248b80fb57
OBJECT_GRAPH_KEY = "_CHECKPOINTABLE_OBJECT_GRAPH"
# tensorflow/core/framework/types.proto
_DTYPES = {1: np.dtype("<f4"), 2: np.dtype("<f8"), 3: np.dtype("<i4"), 4: np.dtype("u1"),
           5: np.dtype("<i2"), 6: np.dtype("i1"), 9: np.dtype("<i8"), 10: np.dtype("?"),
           19: np.dtype("<f2")}
DT_STRING = 7


class BundleError(ValueError):
  pass


# ------------------------------------------------------------------ primitives
def _varint(buf, pos):
  result = shift = 0
  while True:
    if pos >= len(buf):
      raise BundleError("truncated varint")
    b = buf[pos]
    pos += 1
    result |= (b & 0x7F) << shift
    if not b & 0x80:
      return result, pos
    shift += 7
    if shift > 63:
      raise BundleError("varint too long")


_CRC_TABLE = None


def crc32c(data, crc=0):
  """CRC-32C (Castagnoli), bytewise table; used to verify blocks and, optionally, tensors."""
  global _CRC_TABLE
  if _CRC_TABLE is None:
    tab = []
    for i in range(256):
      c = i
      for _ in range(8):
        c = (c >> 1) ^ 0x82F63B78 if c & 1 else c >> 1
      tab.append(c)
    _CRC_TABLE = tab
  tab = _CRC_TABLE
  crc ^= 0xFFFFFFFF
  for b in bytes(data):
    crc = tab[(crc ^ b) & 0xFF] ^ (crc >> 8)
  return crc ^ 0xFFFFFFFF


def mask_crc(crc):
  return (((crc >> 15) | (crc << 17)) + 0xA282EAD8) & 0xFFFFFFFF


def snappy_decompress(src):
  """Raw snappy block format (the index is normally written uncompressed; kept for robustness)."""
  n, pos = _varint(src, 0)
  out = bytearray()
  while pos < len(src):
    tag = src[pos]
    pos += 1
    kind = tag & 3
    if kind == 0:
      ln = tag >> 2
      if ln >= 60:
        nb = ln - 59
        ln = int.from_bytes(src[pos:pos + nb], "little")
        pos += nb
      ln += 1
      out += src[pos:pos + ln]
      pos += ln
      continue
    if kind == 1:
      ln = ((tag >> 2) & 7) + 4
      off = ((tag >> 5) << 8) | src[pos]
      pos += 1
    elif kind == 2:
      ln = (tag >> 2) + 1
      off = int.from_bytes(src[pos:pos + 2], "little")
      pos += 2
    else:
      ln = (tag >> 2) + 1
      off = int.from_bytes(src[pos:pos + 4], "little")
      pos += 4
    if off == 0 or off > len(out):
      raise BundleError("bad snappy copy offset")
    for _ in range(ln):           # may overlap its own output
      out.append(out[-off])
  if len(out) != n:
    raise BundleError("snappy length mismatch")
  return bytes(out)


def parse_proto(buf):
  """Wire-level protobuf decode -> {field: [values]}; length-delimited values stay bytes."""
  out = {}
  pos = 0
  while pos < len(buf):
    key, pos = _varint(buf, pos)
    field, wt = key >> 3, key & 7
    if wt == 0:
      v, pos = _varint(buf, pos)
    elif wt == 1:
      v = struct.unpack_from("<Q", buf, pos)[0]
      pos += 8
    elif wt == 2:
      ln, pos = _varint(buf, pos)
      if pos + ln > len(buf):
        raise BundleError("truncated protobuf field")
      v = bytes(buf[pos:pos + ln])
      pos += ln
    elif wt == 5:
      v = struct.unpack_from("<I", buf, pos)[0]
      pos += 4
    else:
      raise BundleError("unsupported protobuf wire type %d" % wt)
    out.setdefault(field, []).append(v)
  return out


# ------------------------------------------------------------------ LevelDB-format table
def _read_block(data, offset, size, verify):
  end = offset + size
  if end + 5 > len(data):
    raise BundleError("block handle points past the end of the file")
  body, ctype = data[offset:end], data[end]
  if verify:
    want = struct.unpack_from("<I", data, end + 1)[0]
    if mask_crc(crc32c(data[offset:end + 1])) != want:
      raise BundleError("block checksum mismatch at offset %d" % offset)
  if ctype == 1:
    body = snappy_decompress(body)
  elif ctype != 0:
    raise BundleError("unknown block compression type %d" % ctype)
  return body


def _block_entries(block):
  if len(block) < 4:
    raise BundleError("block too small")
  nrestarts = struct.unpack_from("<I", block, len(block) - 4)[0]
  limit = len(block) - 4 - 4 * nrestarts
  if limit < 0:
    raise BundleError("bad restart array")
  pos, key = 0, b""
  while pos < limit:
    shared, pos = _varint(block, pos)
    non_shared, pos = _varint(block, pos)
    vlen, pos = _varint(block, pos)
    if shared > len(key) or pos + non_shared + vlen > limit:
      raise BundleError("corrupt block entry")
    key = key[:shared] + bytes(block[pos:pos + non_shared])
    pos += non_shared
    yield key, bytes(block[pos:pos + vlen])
    pos += vlen


def read_table(path, verify=True):
  """All (key, value) pairs of a LevelDB-format table file, in key order."""
  with open(path, "rb") as f:
    data = f.read()
  if len(data) < 48:
    raise BundleError("%s: too short for a table footer" % path)
  footer = data[-48:]
  if struct.unpack_from("<Q", footer, 40)[0] != TABLE_MAGIC:
    raise BundleError("%s: not a TensorFlow table file (bad magic)" % path)
  pos = 0
  _, pos = _varint(footer, pos)          # metaindex handle (unused)
  _, pos = _varint(footer, pos)
  ioff, pos = _varint(footer, pos)
  isize, pos = _varint(footer, pos)
  out = []
  for _, handle in _block_entries(_read_block(data, ioff, isize, verify)):
    boff, p = _varint(handle, 0)
    bsize, p = _varint(handle, p)
    out.extend(_block_entries(_read_block(data, boff, bsize, verify)))
  return out


# ------------------------------------------------------------------ tensor bundle
class BundleReader:
  """``BundleReader(prefix)`` with prefix = ".../variables/variables"."""

  def __init__(self, prefix, verify=True):
    self.prefix = prefix
    self.entries = {}
    header = None
    for key, value in read_table(prefix + ".index", verify):
      if key == b"":
        header = parse_proto(value)
      else:
        self.entries[key.decode("utf-8")] = parse_proto(value)
    if header is None:
      raise BundleError("%s.index has no bundle header" % prefix)
    self.num_shards = header.get(1, [1])[0]
    if header.get(2, [0])[0] != 0:
      raise BundleError("big-endian bundles are not supported")
    self._shards = {}

  def keys(self):
    return sorted(self.entries)

  def _shard(self, sid):
    if sid not in self._shards:
      path = "%s.data-%05d-of-%05d" % (self.prefix, sid, self.num_shards)
      self._shards[sid] = np.memmap(path, dtype=np.uint8, mode="r")
    return self._shards[sid]

  def _raw(self, key, verify_crc):
    if key not in self.entries:
      raise KeyError(key)
    e = self.entries[key]
    if 7 in e:
      raise BundleError("'%s' is stored in slices (partitioned variable): not supported" % key)
    off, size = e.get(4, [0])[0], e.get(5, [0])[0]
    shard = self._shard(e.get(3, [0])[0])
    if off + size > shard.size:
      raise BundleError("'%s' points past the end of its data shard" % key)
    raw = bytes(shard[off:off + size])
    if verify_crc and 6 in e and mask_crc(crc32c(raw)) != e[6][0]:
      raise BundleError("'%s': tensor checksum mismatch" % key)
    shape = tuple(parse_proto(d).get(1, [0])[0] for d in parse_proto(e[2][0]).get(2, [])) if 2 in e else ()
    return e.get(1, [0])[0], shape, raw

  def get_tensor(self, key, verify_crc=False):
    dtype, shape, raw = self._raw(key, verify_crc)
    if dtype not in _DTYPES:
      raise BundleError("'%s': unsupported dtype enum %d" % (key, dtype))
    a = np.frombuffer(raw, dtype=_DTYPES[dtype])
    if a.size != int(np.prod(shape, dtype=np.int64)):
      raise BundleError("'%s': %d bytes do not fill shape %s" % (key, len(raw), shape))
    return a.reshape(shape)

  def get_strings(self, key):
    """DT_STRING tensor: [varint64 length]*N, 4-byte masked crc of the lengths, then the bytes."""
    dtype, shape, raw = self._raw(key, False)
    if dtype != DT_STRING:
      raise BundleError("'%s' is not a string tensor" % key)
    n = int(np.prod(shape, dtype=np.int64)) if shape else 1
    pos, lens = 0, []
    for _ in range(n):
      ln, pos = _varint(raw, pos)
      lens.append(ln)
    pos += 4
    out = []
    for ln in lens:
      out.append(raw[pos:pos + ln])
      pos += ln
    return out


class ObjectGraph:
  """The saved object tree: node 0 is the model; children are named by attribute."""

  def __init__(self, serialized):
    self.nodes = []
    for raw in parse_proto(serialized).get(1, []):
      node = parse_proto(raw)
      children = {}
      for c in node.get(1, []):
        ref = parse_proto(c)
        children[ref.get(2, [b""])[0].decode("utf-8")] = ref.get(1, [0])[0]
      attrs = {}
      for t in node.get(2, []):
        st = parse_proto(t)
        attrs[st.get(1, [b""])[0].decode("utf-8")] = st.get(3, [b""])[0].decode("utf-8")
      self.nodes.append((children, attrs))
    if not self.nodes:
      raise BundleError("empty object graph")

  def checkpoint_key(self, path):
    """'fire2/squeeze/kernel' -> the bundle key of that variable's value."""
    node = 0
    for name in path.split("/"):
      children = self.nodes[node][0]
      if name not in children:
        raise KeyError("object graph has no '%s' (while resolving %s)" % (name, path))
      node = children[name]
    attrs = self.nodes[node][1]
    if "VARIABLE_VALUE" not in attrs:
      raise KeyError("'%s' is not a variable in the object graph" % path)
    return attrs["VARIABLE_VALUE"]


def is_savedmodel_dir(path):
  return os.path.isdir(path) and os.path.isfile(os.path.join(path, "variables", "variables.index"))


def load_savedmodel_weights(path, spec, verify_crc=False):
  """{Keras path: float32 array} for every tensor of ``spec`` (nets/spec.py) from the SavedModel
  directory ``path``; raises ValueError naming the first tensor that is missing or mis-shaped."""
  reader = BundleReader(os.path.join(path, "variables", "variables"))
  graph = None
  if OBJECT_GRAPH_KEY in reader.entries:
    graph = ObjectGraph(reader.get_strings(OBJECT_GRAPH_KEY)[0])
  out = {}
  for w in spec:
    key = None
    if graph is not None:
      try:
        key = graph.checkpoint_key(w.path)
      except KeyError:
        key = None
    if key is None or key not in reader.entries:
      key = w.path + "/.ATTRIBUTES/VARIABLE_VALUE"      # the attribute-path alias, spelled out
    if key not in reader.entries:
      raise ValueError("SavedModel %s has no variable for '%s'" % (path, w.path))
    a = reader.get_tensor(key, verify_crc)
    if tuple(a.shape) != tuple(w.shape):
      raise ValueError("'%s' has shape %s in the SavedModel, expected %s" % (w.path, a.shape, tuple(w.shape)))
    out[w.path] = np.ascontiguousarray(a, dtype=np.float32)
  return out
