"""A self-contained TFLite flatbuffer reader / writer (no `flatbuffers`, no LiteRT wheel).

The reference reads and writes models through `ai_edge_litert.tools.flatbuffer_utils`
(`read_model`, `read_model_from_bytearray`, `write_model`, `convert_object_to_bytearray`;
ref: utils/tfl_flatbuffer_utils.py:115-139, model_modifier.py:290-391). That wheel is a
third-party dependency; this module restates the *wire format* it speaks:

* `read_model(buf)` parses a `.tflite` into the object tree the reference manipulates
  (`ModelT`, `SubGraphT`, `TensorT`, `QuantizationParametersT`, `BufferT`, `OperatorT`, ... with
  the flatbuffers object-API attribute names). Constant buffers are **zero-copy** `uint8` views
  of the input bytes (ref quantizer.py:180-185), including buffers stored outside the
  flatbuffer through `Buffer.offset/size`.
* `write_model(model)` serializes the tree again, `serialize_with_external_buffers` lays
  buffers >= 1 KiB out behind the flatbuffer at 16-byte aligned offsets exactly like
  model_modifier.py:290-377.

Only the tables the quantizer touches are described field by field. Operator option tables are
carried as *opaque* scalar-only tables (raw vtable + inline bytes). Soundness does not rest on
that assumption: every parse accounts for **every byte** of the flatbuffer (tables, vtables,
vectors, strings, padding); a table with an offset field this module does not know about would
leave its children unaccounted for and the model is rejected instead of being silently damaged.
"""
from __future__ import annotations

import struct
from typing import Any, Callable, Optional

import numpy as np

FILE_IDENTIFIER = b"TFL3"

# --------------------------------------------------------------------------------------
# Schema description. kind: scalar code | "str" | ("vec", scalar) | ("vec_tab", Table) |
# (int32 vectors - shapes, tensor / op indices - are read as Python lists, like the reference's
# code expects (`if op.inputs:`); every other vector is a zero-copy NumPy view) |
# ("vec_str",) | ("tab", Table) | ("union", type_field, {code: Table}) | "dead" (deprecated slot)
# --------------------------------------------------------------------------------------
_SCALARS = {
    "bool": ("<?", 1), "i8": ("<b", 1), "u8": ("<B", 1), "i16": ("<h", 2), "u16": ("<H", 2),
    "i32": ("<i", 4), "u32": ("<I", 4), "i64": ("<q", 8), "u64": ("<Q", 8), "f32": ("<f", 4),
    "f64": ("<d", 8),
}
_NP = {"bool": np.bool_, "i8": np.int8, "u8": np.uint8, "i16": np.int16, "u16": np.uint16,
       "i32": np.int32, "u32": np.uint32, "i64": np.int64, "u64": np.uint64, "f32": np.float32,
       "f64": np.float64}

# Union member tables of QuantizationDetails / SparseIndexVector / the option unions that hold
# offsets. Everything else in BuiltinOptions / BuiltinOptions2 is carried opaquely.
_BUILTIN_OPTIONS = {3: "ConcatEmbeddingsOptions", 8: "FullyConnectedOptions", 17: "ReshapeOptions", 21: "MulOptions",
                    30: "SqueezeOptions", 101: "BatchMatMulOptions", 111: "VarHandleOptions",
                    115: "BucketizeOptions"}
_BUILTIN_OPTIONS2 = {21: "StableHLOCompositeOptions"}

SCHEMA: dict[str, list[tuple]] = {
    "Model": [("version", "u32", 0), ("operatorCodes", ("vec_tab", "OperatorCode")),
              ("subgraphs", ("vec_tab", "SubGraph")), ("description", "str"),
              ("buffers", ("vec_tab", "Buffer")), ("metadataBuffer", ("vec", "i32")),
              ("metadata", ("vec_tab", "Metadata")), ("signatureDefs", ("vec_tab", "SignatureDef"))],
    "OperatorCode": [("deprecatedBuiltinCode", "i8", 0), ("customCode", "str"), ("version", "i32", 1),
                     ("builtinCode", "i32", 0)],
    "SubGraph": [("tensors", ("vec_tab", "Tensor")), ("inputs", ("vec", "i32")),
                 ("outputs", ("vec", "i32")), ("operators", ("vec_tab", "Operator")), ("name", "str"),
                 ("debugMetadataIndex", "i32", -1)],
    "Tensor": [("shape", ("vec", "i32")), ("type", "i8", 0), ("buffer", "u32", 0), ("name", "str"),
               ("quantization", ("tab", "QuantizationParameters")), ("isVariable", "bool", False),
               ("sparsity", ("tab", "SparsityParameters")), ("shapeSignature", ("vec", "i32")),
               ("hasRank", "bool", False), ("variantTensors", ("vec_tab", "VariantSubType"))],
    "QuantizationParameters": [("min", ("vec", "f32")), ("max", ("vec", "f32")), ("scale", ("vec", "f32")),
                               ("zeroPoint", ("vec", "i64")), ("detailsType", "u8", 0),
                               ("details", ("union", "detailsType",
                                            {1: "CustomQuantization", 2: "BlockwiseQuantization"})),
                               ("quantizedDimension", "i32", 0)],
    "CustomQuantization": [("custom", ("vec", "u8"))],
    "BlockwiseQuantization": [("scales", "i32", 0), ("zeroPoints", "i32", 0), ("blockSize", "i32", 0)],
    "SparsityParameters": [("traversalOrder", ("vec", "i32")), ("blockMap", ("vec", "i32")),
                           ("dimMetadata", ("vec_tab", "DimensionMetadata"))],
    "DimensionMetadata": [("format", "i8", 0), ("denseSize", "i32", 0), ("arraySegmentsType", "u8", 0),
                          ("arraySegments", ("union", "arraySegmentsType",
                                             {1: "Int32Vector", 2: "Uint16Vector", 3: "Uint8Vector"})),
                          ("arrayIndicesType", "u8", 0),
                          ("arrayIndices", ("union", "arrayIndicesType",
                                            {1: "Int32Vector", 2: "Uint16Vector", 3: "Uint8Vector"}))],
    "Int32Vector": [("values", ("vec", "i32", "np"))],
    "Uint16Vector": [("values", ("vec", "u16"))],
    "Uint8Vector": [("values", ("vec", "u8"))],
    "VariantSubType": [("shape", ("vec", "i32")), ("type", "i8", 0), ("hasRank", "bool", False)],
    "Operator": [("opcodeIndex", "u32", 0), ("inputs", ("vec", "i32")), ("outputs", ("vec", "i32")),
                 ("builtinOptionsType", "u8", 0),
                 ("builtinOptions", ("union", "builtinOptionsType", _BUILTIN_OPTIONS)),
                 ("customOptions", ("vec", "u8")), ("customOptionsFormat", "i8", 0),
                 ("mutatingVariableInputs", ("vec", "bool")), ("intermediates", ("vec", "i32")),
                 ("largeCustomOptionsOffset", "u64", 0), ("largeCustomOptionsSize", "u64", 0),
                 ("builtinOptions2Type", "u8", 0),
                 ("builtinOptions2", ("union", "builtinOptions2Type", _BUILTIN_OPTIONS2)),
                 ("debugMetadataIndex", "i32", -1)],
    "Buffer": [("data", ("vec", "u8")), ("offset", "u64", 0), ("size", "u64", 0)],
    "Metadata": [("name", "str"), ("buffer", "u32", 0)],
    "SignatureDef": [("inputs", ("vec_tab", "TensorMap")), ("outputs", ("vec_tab", "TensorMap")),
                     ("signatureKey", "str"), ("deprecatedTag", "dead"), ("subgraphIndex", "u32", 0)],
    "TensorMap": [("name", "str"), ("tensorIndex", "u32", 0)],
    # The hot path's own op (typed so its fields are addressable by name).
    "FullyConnectedOptions": [("fusedActivationFunction", "i8", 0), ("weightsFormat", "i8", 0),
                              ("keepNumDims", "bool", False), ("asymmetricQuantizeInputs", "bool", False),
                              ("quantizedBiasType", "i8", 0)],
    "MulOptions": [("fusedActivationFunction", "i8", 0)],     # written by INSERT_MULTIPLY
    "BatchMatMulOptions": [("adjX", "bool", False), ("adjY", "bool", False),
                           ("asymmetricQuantizeInputs", "bool", False)],
    # Option tables that hold offsets.
    "ConcatEmbeddingsOptions": [("numChannels", "i32", 0), ("numColumnsPerChannel", ("vec", "i32")),
                                ("embeddingDimPerChannel", ("vec", "i32"))],
    "ReshapeOptions": [("newShape", ("vec", "i32"))],
    "SqueezeOptions": [("squeezeDims", ("vec", "i32"))],
    "VarHandleOptions": [("container", "str"), ("sharedName", "str")],
    "BucketizeOptions": [("boundaries", ("vec", "f32"))],
    "StableHLOCompositeOptions": [("name", "str"), ("decompositionSubgraphIndex", "i32", 0),
                                  ("compositeAttributes", ("vec", "u8")),
                                  ("compositeAttributesFormat", "i8", 0), ("version", "i32", 0)],
}

# Alignment the Buffer.data vector is written with (schema: force_align 16).
_VEC_ALIGN = {("Buffer", "data"): 16}


class FlatbufferError(ValueError):
  """The bytes are not a flatbuffer this module can carry without loss."""


# --------------------------------------------------------------------------------------
# Object API
# --------------------------------------------------------------------------------------
class TableT:
  """Attribute container with the flatbuffers object-API field names of one table."""
  _table: str = ""

  def __init__(self, **kw):
    for spec in SCHEMA[self._table]:
      name, kind = spec[0], spec[1]
      if kind == "dead":
        continue
      setattr(self, name, spec[2] if isinstance(kind, str) and kind in _SCALARS else None)
    for k, v in kw.items():
      setattr(self, k, v)

  def __repr__(self):
    body = ", ".join(f"{s[0]}={getattr(self, s[0], None)!r}" for s in SCHEMA[self._table] if s[1] != "dead")
    return f"{type(self).__name__}({body})"


def _make_classes() -> dict[str, type]:
  out = {}
  for table in SCHEMA:
    out[table] = type(table + "T", (TableT,), {"_table": table})
  return out


CLASSES = _make_classes()
ModelT = CLASSES["Model"]
SubGraphT = CLASSES["SubGraph"]
TensorT = CLASSES["Tensor"]
BufferT = CLASSES["Buffer"]
OperatorT = CLASSES["Operator"]
OperatorCodeT = CLASSES["OperatorCode"]
QuantizationParametersT = CLASSES["QuantizationParameters"]
BlockwiseQuantizationT = CLASSES["BlockwiseQuantization"]
CustomQuantizationT = CLASSES["CustomQuantization"]
MetadataT = CLASSES["Metadata"]
SignatureDefT = CLASSES["SignatureDef"]
TensorMapT = CLASSES["TensorMap"]
StableHLOCompositeOptionsT = CLASSES["StableHLOCompositeOptions"]
ReshapeOptionsT = CLASSES["ReshapeOptions"]
FullyConnectedOptionsT = CLASSES["FullyConnectedOptions"]
BatchMatMulOptionsT = CLASSES["BatchMatMulOptions"]
MulOptionsT = CLASSES["MulOptions"]
# union type code of each typed option table (Operator.builtinOptionsType / builtinOptions2Type)
BUILTIN_OPTIONS_CODE = {v: k for k, v in _BUILTIN_OPTIONS.items()}
BUILTIN_OPTIONS2_CODE = {v: k for k, v in _BUILTIN_OPTIONS2.items()}


class OpaqueTableT:
  """A scalar-only table whose field meanings are not described here.

  `vtable_fields[i]` is field i's byte offset inside `inline` (0 = absent); `inline` is the
  table's inline bytes *including* the leading 4-byte vtable reference (rewritten on output);
  `start_mod8` keeps the original 8-byte phase so 64-bit scalars stay aligned.
  """
  __slots__ = ("vtable_fields", "inline", "start_mod8")

  def __init__(self, vtable_fields=(), inline=b"\0\0\0\0", start_mod8=0):
    self.vtable_fields = tuple(vtable_fields)
    self.inline = bytes(inline)
    self.start_mod8 = start_mod8

  def scalar(self, field_id: int, code: str, default=0):
    """Decode scalar field `field_id` (schema order) as `code` ('i32', 'bool', ...)."""
    if field_id >= len(self.vtable_fields) or self.vtable_fields[field_id] == 0:
      return default
    fmt, _ = _SCALARS[code]
    return struct.unpack_from(fmt, self.inline, self.vtable_fields[field_id])[0]

  def __eq__(self, other):
    return (isinstance(other, OpaqueTableT) and self.vtable_fields == other.vtable_fields
            and self.inline[4:] == other.inline[4:])

  def __repr__(self):
    return f"OpaqueTableT(fields={len(self.vtable_fields)}, inline={len(self.inline) - 4}B)"


# --------------------------------------------------------------------------------------
# Reader
# --------------------------------------------------------------------------------------
class _Reader:
  def __init__(self, buf):
    self.mv = memoryview(buf).cast("B") if not isinstance(buf, memoryview) else buf.cast("B")
    self.u8 = np.frombuffer(self.mv, dtype=np.uint8)
    self.n = len(self.mv)
    self.spans: list[tuple[int, int]] = []   # byte ranges claimed by parsed objects

  # -- primitive access --------------------------------------------------------------
  def _check(self, pos: int, size: int, what: str):
    if pos < 0 or size < 0 or pos + size > self.n:
      raise FlatbufferError(f"{what}: [{pos}, {pos + size}) is outside the {self.n}-byte buffer")

  def u16(self, pos): self._check(pos, 2, "u16"); return struct.unpack_from("<H", self.mv, pos)[0]
  def u32(self, pos): self._check(pos, 4, "u32"); return struct.unpack_from("<I", self.mv, pos)[0]
  def i32(self, pos): self._check(pos, 4, "i32"); return struct.unpack_from("<i", self.mv, pos)[0]

  def mark(self, pos: int, size: int):
    if size:
      self.spans.append((pos, pos + size))

  def indirect(self, pos: int) -> int:
    off = self.u32(pos)
    if off == 0:
      raise FlatbufferError(f"null offset at {pos}")
    return pos + off

  # -- composite objects -------------------------------------------------------------
  def table_header(self, pos: int):
    """Returns (vtable field offsets, inline size) and accounts for vtable + inline bytes."""
    vt = pos - self.i32(pos)
    vsize, tsize = self.u16(vt), self.u16(vt + 2)
    if vsize < 4 or vsize % 2 or tsize < 4:
      raise FlatbufferError(f"bad vtable at {vt} (table at {pos})")
    self._check(vt, vsize, "vtable")
    self._check(pos, tsize, "table")
    fields = struct.unpack_from(f"<{(vsize - 4) // 2}H", self.mv, vt + 4)
    for f in fields:
      if f and not 4 <= f < tsize:
        raise FlatbufferError(f"field offset {f} outside table of {tsize} bytes at {pos}")
    self.mark(vt, vsize)
    self.mark(pos, tsize)
    return fields, tsize

  def string(self, pos: int) -> bytes:
    n = self.u32(pos)
    self._check(pos + 4, n + 1, "string")
    self.mark(pos, 4 + n + 1)
    return bytes(self.mv[pos + 4:pos + 4 + n])

  def vector(self, pos: int, code: str) -> np.ndarray:
    n = self.u32(pos)
    dt = np.dtype(_NP[code])
    nbytes = n * dt.itemsize
    self._check(pos + 4, nbytes, "vector")
    self.mark(pos, 4 + nbytes)
    if (pos + 4) % dt.itemsize:
      raise FlatbufferError(f"misaligned {code} vector at {pos}")
    return self.u8[pos + 4:pos + 4 + nbytes].view(dt)

  def opaque(self, pos: int) -> OpaqueTableT:
    fields, tsize = self.table_header(pos)
    return OpaqueTableT(fields, self.mv[pos:pos + tsize], pos % 8)

  def table(self, pos: int, name: str) -> TableT:
    spec = SCHEMA[name]
    fields, tsize = self.table_header(pos)
    if len(fields) > len(spec):
      extra = [i for i in range(len(spec), len(fields)) if fields[i]]
      if extra:
        raise FlatbufferError(f"table {name} at {pos} has unknown fields {extra}: the file uses a"
                              " newer schema than this module describes")
    obj = CLASSES[name]()
    for fid, fs in enumerate(spec):
      fname, kind = fs[0], fs[1]
      off = fields[fid] if fid < len(fields) else 0
      if kind == "dead":
        if off:
          raise FlatbufferError(f"{name}.{fname} (deprecated) is present; refusing to drop it")
        continue
      if not off:
        continue
      at = pos + off
      if isinstance(kind, str):
        if kind == "str":
          value = self.string(self.indirect(at))
        else:
          fmt, size = _SCALARS[kind]
          self._check(at, size, f"{name}.{fname}")
          value = struct.unpack_from(fmt, self.mv, at)[0]
          if value == fs[2]:          # stored although equal to the default: keep it stored
            obj.__dict__.setdefault("_explicit", set()).add(fname)
      elif kind[0] == "vec":
        value = self.vector(self.indirect(at), kind[1])
        if kind[1] == "i32" and len(kind) == 2:
          value = value.tolist()      # structural index / shape vectors are plain lists
      elif kind[0] == "vec_tab":
        vpos = self.indirect(at)
        n = self.u32(vpos)
        self._check(vpos + 4, 4 * n, "offset vector")
        self.mark(vpos, 4 + 4 * n)
        value = [self.table(self.indirect(vpos + 4 + 4 * i), kind[1]) for i in range(n)]
      elif kind[0] == "tab":
        value = self.table(self.indirect(at), kind[1])
      elif kind[0] == "union":
        code = getattr(obj, kind[1])
        if code == 0:
          raise FlatbufferError(f"{name}.{fname} present with union type NONE")
        target = self.indirect(at)
        member = kind[2].get(code)
        value = self.table(target, member) if member else self.opaque(target)
      else:
        raise AssertionError(kind)
      setattr(obj, fname, value)
    return obj

  # -- whole-buffer accounting --------------------------------------------------------
  def verify_coverage(self, start: int, end: int):
    """Every byte in [start, end) must belong to a parsed object or be alignment padding."""
    spans = sorted(self.spans)
    cur = start
    for a, b in spans:
      if a > cur:
        gap = self.u8[cur:a]
        if a - cur >= 16 or gap.any():
          raise FlatbufferError(
              f"{a - cur} unaccounted bytes at [{cur}, {a}): the model holds data this module's"
              " schema does not describe (refusing to rewrite it)")
      cur = max(cur, b)
    if cur < end:
      gap = self.u8[cur:end]
      if end - cur >= 16 or gap.any():
        raise FlatbufferError(f"{end - cur} unaccounted bytes at the end of the flatbuffer")


def read_model(buf, verify: bool = True) -> TableT:
  """Parses `.tflite` bytes into a ModelT tree; constant data are zero-copy views of `buf`.

  Buffers stored behind the flatbuffer (`Buffer.offset/size`, the >2 GiB / external layout the
  reference's serializer emits, model_modifier.py:316-377) are resolved to `data` views and
  their offset/size reset, which is what the LiteRT reader hands the reference.
  """
  r = _Reader(buf)
  if r.n < 8:
    raise FlatbufferError("buffer too small for a flatbuffer")
  root = r.u32(0)
  ident = bytes(r.mv[4:8])
  has_ident = ident == FILE_IDENTIFIER
  r.mark(0, 8 if has_ident else 4)
  model = r.table(root, "Model")
  # External buffers: data live after the flatbuffer proper.
  for b in model.buffers or []:
    if b.offset > 1 and b.size > 0:
      r._check(b.offset, b.size, "external buffer")
      b.data = r.u8[b.offset:b.offset + b.size]
      r.mark(b.offset, b.size)
      b.offset = 0
      b.size = 0
  if verify:
    r.verify_coverage(0, r.n)
  return model


def read_model_from_file(path) -> TableT:
  """mmap the file read-only and parse it (ref tfl_flatbuffer_utils.py:141-152)."""
  import mmap
  with open(path, "rb") as f:
    size = f.seek(0, 2)
    if size == 0:
      raise FlatbufferError(f"{path} is empty")
    mm = mmap.mmap(f.fileno(), 0, access=mmap.ACCESS_READ)
  return read_model(mm)


# --------------------------------------------------------------------------------------
# Writer: a back-to-front builder (children are emitted before their parents, so every
# uoffset points forward as the format requires).
# --------------------------------------------------------------------------------------
class Builder:
  def __init__(self):
    self.chunks: list[Any] = []       # in reverse file order
    self.size = 0                      # bytes emitted so far == distance from the end
    self.minalign = 1
    self.vtables: dict[bytes, int] = {}
    self.patch_points: dict[Any, int] = {}   # key -> offset-from-end of a field to patch later
    self.late_vectors: Optional[list] = None   # a list: vectors whose VALUES arrive later are accepted -> (offset-from-end of
                                               # the elements, the vector); None: every vector is read when it is packed

  def offset(self) -> int:
    return self.size

  def _put(self, b):
    n = len(b)
    if n:
      self.chunks.append(b)
      self.size += n

  def prep(self, align: int, additional: int):
    """Pad so that after `additional` more bytes the write position is `align`-aligned."""
    if align > self.minalign:
      self.minalign = align
    pad = (-(self.size + additional)) % align
    if pad:
      self._put(bytes(pad))

  def scalar(self, code: str, value):
    fmt, size = _SCALARS[code]
    self.prep(size, 0)
    self._put(struct.pack(fmt, value))

  def uoffset(self, target: int):
    self.prep(4, 0)
    self._put(struct.pack("<I", self.size + 4 - target))

  def string(self, s) -> int:
    if isinstance(s, str):
      s = s.encode("utf-8")
    s = bytes(s)
    self.prep(4, len(s) + 1)
    self._put(b"\0")
    self._put(s)
    self._put(struct.pack("<I", len(s)))
    return self.size

  def vector_bytes(self, payload, count: int, elem_align: int, align: int = 0) -> int:
    """`payload` = the elements' little-endian bytes (bytes / memoryview / ndarray)."""
    nbytes = len(payload) if not isinstance(payload, np.ndarray) else payload.nbytes
    self.prep(4, nbytes)
    self.prep(max(elem_align, align), nbytes)
    if isinstance(payload, np.ndarray):
      payload = payload.tobytes() if nbytes < 4096 else memoryview(
          np.ascontiguousarray(payload).reshape(-1).view(np.uint8))
    self._put(payload)
    self._put(struct.pack("<I", count))
    return self.size

  def offset_vector(self, targets: list[int]) -> int:
    self.prep(4, 4 * len(targets))
    for t in reversed(targets):
      self.uoffset(t)
    self._put(struct.pack("<I", len(targets)))
    return self.size

  def table(self, slots: list[Optional[tuple]]) -> int:
    """slots[i] = None | (code, value[, patch_key]) scalar | ("off", target). Fields are laid out
    in schema order back to front (first field at the highest address); trailing absent fields
    are trimmed from the vtable; identical vtables are shared."""
    floor = self.size
    placed = [0] * len(slots)
    for i, s in enumerate(slots):
      if s is None:
        continue
      if s[0] == "off":
        self.uoffset(s[1])
      else:
        self.scalar(s[0], s[1])
      placed[i] = self.size
      if len(s) > 2:
        self.patch_points[s[2]] = self.size
    self.prep(4, 0)
    self._put(b"\0\0\0\0")               # vtable reference, fixed up below
    table_off = self.size
    while placed and placed[-1] == 0:
      placed.pop()
    return self._finish_table(table_off, [table_off - p if p else 0 for p in placed],
                              table_off - floor)

  def _finish_table(self, table_off: int, field_offsets: list[int], inline_size: int) -> int:
    vt = struct.pack(f"<HH{len(field_offsets)}H", 4 + 2 * len(field_offsets), inline_size, *field_offsets)
    known = self.vtables.get(vt)
    if known is None:
      self.prep(2, 0)
      self._put(vt)
      known = self.size
      self.vtables[vt] = known
    # soffset = table position - vtable position = vtable_from_end - table_from_end
    self._patch_from_end(table_off, struct.pack("<i", known - table_off))
    return table_off

  def opaque_table(self, t: OpaqueTableT) -> int:
    """Re-emit a scalar-only table byte for byte, at the 8-byte phase it was read at (the
    finished buffer is a multiple of 8 long, so phase-from-the-end fixes the absolute phase)."""
    inline = t.inline
    self.minalign = max(self.minalign, 8)
    pad = ((-t.start_mod8) - (self.size + len(inline))) % 8
    if pad:
      self._put(bytes(pad))
    self._put(b"\0\0\0\0" + bytes(inline[4:]))
    return self._finish_table(self.size, list(t.vtable_fields), len(inline))

  def _patch_from_end(self, off_from_end: int, data: bytes):
    """Overwrite bytes that start `off_from_end` bytes before the end of what is built so far."""
    # walk chunks from the newest backwards until the one that holds the position
    pos = self.size
    for idx in range(len(self.chunks) - 1, -1, -1):
      c = self.chunks[idx]
      start = pos                        # this chunk occupies (pos - len, pos] from the end
      pos -= len(c)
      if pos < off_from_end <= start:
        inner = start - off_from_end     # index inside the chunk (chunk is in file order)
        if inner + len(data) > len(c):
          raise AssertionError("patch crosses a chunk boundary")
        b = bytearray(c)
        b[inner:inner + len(data)] = data
        self.chunks[idx] = bytes(b)
        return
    raise AssertionError("patch position not found")

  def finish(self, root: int, identifier: Optional[bytes] = FILE_IDENTIFIER) -> bytearray:
    extra = 4 + (4 if identifier else 0)
    self.prep(self.minalign, extra)
    if identifier:
      self._put(identifier)
    self.uoffset(root)
    out = bytearray(self.size)
    pos = 0
    for c in reversed(self.chunks):
      n = len(c)
      out[pos:pos + n] = c
      pos += n
    return out


def _pack(b: Builder, obj, name: str) -> int:
  spec = SCHEMA[name]
  # 1) children first, in schema order (the order the generated object API packs them in)
  child: dict[str, int] = {}
  for fs in spec:
    fname, kind = fs[0], fs[1]
    if kind == "dead":
      continue
    v = getattr(obj, fname, None)
    if v is None or (isinstance(kind, str) and kind in _SCALARS):
      continue
    if kind == "str":
      child[fname] = b.string(v)
    elif kind[0] == "vec" and b.late_vectors is not None and getattr(v, "late_values", False) and v.dtype == _NP[kind[1]]:
      # values still on their way from the accelerator (per-channel scales): count and size are known, the
      # elements are reserved and filled in when they are there (serialize_with_external_buffers)
      child[fname] = b.vector_bytes(bytes(v.nbytes), v.size, v.dtype.itemsize, _VEC_ALIGN.get((name, fname), 0))
      b.late_vectors.append((child[fname] - 4, v))       # (the count sits in front of the elements)
    elif kind[0] == "vec":
      arr = np.ascontiguousarray(np.asarray(v, dtype=_NP[kind[1]]) if not (
          isinstance(v, np.ndarray) and v.dtype == _NP[kind[1]]) else v).ravel()
      child[fname] = b.vector_bytes(arr, arr.size, arr.dtype.itemsize, _VEC_ALIGN.get((name, fname), 0))
    elif kind[0] == "vec_tab":
      offs = [_pack(b, e, kind[1]) for e in v]
      child[fname] = b.offset_vector(offs)
    elif kind[0] == "tab":
      child[fname] = _pack(b, v, kind[1])
    elif kind[0] == "union":
      child[fname] = b.opaque_table(v) if isinstance(v, OpaqueTableT) else _pack(b, v, v._table)
  # 2) the table itself
  slots: list[Optional[tuple]] = []
  for fs in spec:
    fname, kind = fs[0], fs[1]
    if kind == "dead":
      slots.append(None)
    elif isinstance(kind, str) and kind in _SCALARS:
      v = getattr(obj, fname, fs[2])
      v = fs[2] if v is None else v
      force = name == "Buffer" and fname in ("offset", "size") and getattr(obj, "_external", False)
      if v == fs[2] and not force and fname not in getattr(obj, "_explicit", ()):
        slots.append(None)
      elif force:
        slots.append((kind, int(v), (id(obj), fname)))
      else:
        slots.append((kind, bool(v) if kind == "bool" else (float(v) if kind[0] == "f" else int(v))))
    else:
      slots.append(("off", child[fname]) if fname in child else None)
  return b.table(slots)


def write_model(model, identifier: Optional[bytes] = FILE_IDENTIFIER) -> bytearray:
  """Serialize a ModelT tree to `.tflite` bytes (all buffers inline)."""
  b = Builder()
  root = _pack(b, model, "Model")
  return b.finish(root, identifier)


def _round_up_16(n: int) -> int:
  return (n + 15) & ~15


def serialize_with_external_buffers(model, min_size_bytes: int = 1024,
                                    sink: Optional[Callable[[int], Any]] = None,
                                    before_values: Optional[Callable[[], None]] = None,
                                    on_layout: Optional[Callable[[], None]] = None):
  """Flatbuffer + buffers >= `min_size_bytes` laid out behind it, each at a 16-byte aligned
  offset recorded in `Buffer.offset/size` (the reference's large-model layout,
  model_modifier.py:48-77, 290-377).

  `sink(total_bytes)` may return a writable buffer (e.g. an mmap of the output file) to build
  into; otherwise a bytearray is returned. Buffer payloads may be NumPy arrays of any dtype.

  The layout needs sizes only. Vectors whose values are still being computed (objects with `late_values`, a dtype
  and a size: per-channel scales in HBM) get their place in the flatbuffer and are filled in LAST, after every
  payload has been handed its place (`copy_into`: a device-resident payload starts its own way into the file there,
  behind its own producer) and after `before_values()` -- the caller's one wait for the accelerator (it may also raise to
  have the model built again: model_modifier.serialize_model).
  """
  ext: dict[int, Any] = {}      # buffer id -> memoryview of its bytes, or a device-resident payload
  sizes: dict[int, int] = {}
  packed = 0
  for i, buf in enumerate(model.buffers or []):
    d = buf.data
    if d is None:
      continue
    if hasattr(d, "copy_into"):   # lives on an accelerator: copies itself into the output mapping
      if d.nbytes >= min_size_bytes:
        ext[i], sizes[i] = d, d.nbytes
        packed = _round_up_16(packed + d.nbytes)
      else:
        buf.data = np.ravel(np.asarray(d)).view(np.uint8)
      continue
    arr = d if isinstance(d, np.ndarray) else np.frombuffer(bytes(d), dtype=np.uint8)
    if arr.nbytes >= min_size_bytes:
      ext[i] = memoryview(np.ascontiguousarray(arr).reshape(-1).view(np.uint8))
      sizes[i] = arr.nbytes
      packed = _round_up_16(packed + arr.nbytes)
  saved = {}
  for i in ext:
    buf = model.buffers[i]
    saved[i] = (buf.data, buf.offset, buf.size)
    buf.data, buf.offset, buf.size, buf._external = None, 1, 1, True
  try:
    b = Builder()
    b.late_vectors = []
    root = _pack(b, model, "Model")
    patch = dict(b.patch_points)
    fb = b.finish(root)
    total_fb = len(fb)
    start = _round_up_16(total_fb)
    if on_layout is not None:
      on_layout()
    out = sink(start + packed) if sink is not None else None
    if out is None:
      out = bytearray(start + packed)
    out[:total_fb] = fb
    out[total_fb:start] = bytes(start - total_fb)     # (the sink may hand out an older file's pages: every gap is written)
    cursor = start
    for i, view in ext.items():
      buf = model.buffers[i]
      n = sizes[i]
      for fname, value in (("offset", cursor), ("size", n)):
        pos = total_fb - patch[(id(buf), fname)]
        out[pos:pos + 8] = struct.pack("<Q", value)
      if hasattr(view, "copy_into"):
        view.copy_into(np.frombuffer(out, dtype=np.uint8, count=n, offset=cursor))
      else:
        out[cursor:cursor + n] = view
      out[cursor + n:min(_round_up_16(cursor + n), len(out))] = bytes(min(_round_up_16(cursor + n), len(out)) - cursor - n)
      cursor = _round_up_16(cursor + n)
    if before_values is not None:
      before_values()
    if b.late_vectors:
      for from_end, vec in b.late_vectors:
        pos = total_fb - from_end
        data = np.ascontiguousarray(np.asarray(vec)).reshape(-1)
        if data.nbytes != vec.nbytes:
          raise ValueError(f"a late vector changed its size: {data.nbytes} bytes for {vec.nbytes} reserved")
        out[pos:pos + data.nbytes] = data.view(np.uint8).tobytes()
  finally:
    for i, (d, o, s) in saved.items():
      buf = model.buffers[i]
      buf.data, buf.offset, buf.size = d, o, s
      del buf._external
  return out
