"""The FlexBuffer a custom op's options travel in: maps of scalars and vectors of scalars.

ref: transformations/insert_hadamard_rotation.py:24-33 -- the reference builds
  {"hadamard_size": int, "random_binary_vector": [..]} with the third-party
`flatbuffers.flexbuffers.Builder()` (package `flatbuffers`, not vendored with the reference and not
installed here: **parity of the bytes is unpinned**; what is pinned is the format -- `decode` below reads
it back by the published layout, and any FlexBuffers reader does). `encode_map` follows that builder's
steps for the pieces the reference uses (default options: keys shared, minimum width 8 bits):
  * a key is written where it is first needed (NUL-terminated) and pushed as an offset;
  * an Int / Float is pushed inline at the smallest width that holds it (floats: 32 bits when the value
    survives a round trip through float32, else 64);
  * a vector made from elements is UNTYPED: [length][elements ...][one packed type byte per element],
    every slot `byte_width` wide, the width chosen so that length, elements and relative offsets fit;
  * a map sorts its pairs by key bytes, writes the typed vector of key offsets, then the values vector
    prefixed by [offset of the keys vector][byte width of the keys vector];
  * the root: the value (or relative offset), its packed type, the root's byte width.
Packed type = (type << 2) | log2(byte width of what the slot refers to / holds).
"""
from __future__ import annotations

import struct
from typing import Any, Mapping, Sequence

# value types (flexbuffers.h / flexbuffers.py)
_INT, _FLOAT, _KEY, _MAP, _VECTOR, _VECTOR_KEY = 1, 3, 4, 9, 10, 14
_FMT_I = {1: "<b", 2: "<h", 4: "<i", 8: "<q"}
_FMT_U = {1: "<B", 2: "<H", 4: "<I", 8: "<Q"}
_FMT_F = {4: "<f", 8: "<d"}


def _width_u(v: int) -> int:
  """log2 of the bytes an unsigned value needs."""
  for k, bits in enumerate((8, 16, 32, 64)):
    if v < (1 << bits):
      return k
  raise ValueError("value does not fit 64 bits")


def _width_i(v: int) -> int:
  for k, bits in enumerate((8, 16, 32, 64)):
    if -(1 << (bits - 1)) <= v < (1 << (bits - 1)):
      return k
  raise ValueError("value does not fit 64 bits")


def _width_f(v: float) -> int:
  return 2 if struct.unpack("<f", struct.pack("<f", v))[0] == v else 3


class _Value:
  __slots__ = ("value", "type", "min_width")

  def __init__(self, value, type_, min_width):
    self.value, self.type, self.min_width = value, type_, min_width

  @property
  def inline(self) -> bool:
    return self.type in (_INT, _FLOAT)

  def elem_width(self, buf_size: int, elem_index: int = 0) -> int:
    """Width needed to store this value in a vector slot written at `buf_size` (+ padding)."""
    if self.inline:
      return self.min_width
    for k in range(4):
      byte_width = 1 << k
      offset_loc = buf_size + (-buf_size) % byte_width + elem_index * byte_width
      if byte_width == 1 << _width_u(offset_loc - self.value):
        return k
    raise ValueError("relative offset does not fit")

  def stored_packed_type(self, parent_width: int = 0) -> int:
    width = max(self.min_width, parent_width) if self.inline else self.min_width
    return (self.type << 2) | width


class _Builder:
  def __init__(self):
    self.buf = bytearray()
    self.keys: dict[bytes, int] = {}

  def _align(self, width: int) -> int:
    byte_width = 1 << width
    self.buf.extend(b"\x00" * ((-len(self.buf)) % byte_width))
    return byte_width

  def _write_any(self, v: _Value, byte_width: int) -> None:
    if v.type == _INT:
      self.buf += struct.pack(_FMT_I[byte_width], v.value)
    elif v.type == _FLOAT:
      self.buf += struct.pack(_FMT_F[byte_width], v.value)
    else:
      self.buf += struct.pack(_FMT_U[byte_width], len(self.buf) - v.value)

  def key(self, name: str) -> _Value:
    raw = name.encode("utf-8")
    loc = self.keys.get(raw)
    if loc is None:
      loc = self.keys[raw] = len(self.buf)
      self.buf += raw + b"\x00"
    return _Value(loc, _KEY, 0)

  @staticmethod
  def scalar(x) -> _Value:
    if isinstance(x, bool):
      raise TypeError("bool options are not used by the reference's custom ops")
    if isinstance(x, int):
      return _Value(int(x), _INT, _width_i(int(x)))
    if isinstance(x, float):
      return _Value(float(x), _FLOAT, _width_f(float(x)))
    raise TypeError(f"cannot encode {type(x).__name__}")

  def vector(self, elements: Sequence[_Value], typed: bool, keys: _Value | None = None) -> _Value:
    length = len(elements)
    width = max(0, _width_u(length))
    prefix = 1
    if keys is not None:
      width = max(width, keys.elem_width(len(self.buf)))
      prefix += 2
    for i, e in enumerate(elements):
      width = max(width, e.elem_width(len(self.buf), prefix + i))
    byte_width = self._align(width)
    if keys is not None:
      self.buf += struct.pack(_FMT_U[byte_width], len(self.buf) - keys.value)
      self.buf += struct.pack(_FMT_U[byte_width], 1 << keys.min_width)
    self.buf += struct.pack(_FMT_U[byte_width], length)
    loc = len(self.buf)
    for e in elements:
      self._write_any(e, byte_width)
    if not typed:
      for e in elements:
        self.buf.append(e.stored_packed_type(width))
    return _Value(loc, _MAP if keys is not None else (_VECTOR_KEY if typed else _VECTOR), width)

  def finish(self, root: _Value) -> bytes:
    byte_width = self._align(root.elem_width(len(self.buf)))
    self._write_any(root, byte_width)
    self.buf.append(root.stored_packed_type())
    self.buf.append(byte_width)
    return bytes(self.buf)


def encode_map(options: Mapping[str, Any]) -> bytes:
  """{"name": int | float | [int | float, ...]} -> FlexBuffer bytes; pairs in the caller's order (the
  order decides where keys and vectors land in the buffer, as with the reference's builder calls)."""
  b = _Builder()
  pairs = []
  for name, x in options.items():
    k = b.key(name)                   # (the key is written before the value it names)
    if isinstance(x, (list, tuple)):
      v = b.vector([b.scalar(e) for e in x], typed=False)
    else:
      v = b.scalar(x)
    pairs.append((name.encode("utf-8"), k, v))
  pairs.sort(key=lambda p: p[0])
  keys = b.vector([p[1] for p in pairs], typed=True)
  return b.finish(b.vector([p[2] for p in pairs], typed=False, keys=keys))


# ---- reader (tests, and anybody who wants to look inside a custom op's options) ----
def _read_u(buf: bytes, at: int, byte_width: int) -> int:
  return struct.unpack_from(_FMT_U[byte_width], buf, at)[0]


def _read_value(buf: bytes, at: int, parent_byte_width: int, packed_type: int):
  type_, byte_width = packed_type >> 2, 1 << (packed_type & 3)
  if type_ == _INT:
    return struct.unpack_from(_FMT_I[parent_byte_width], buf, at)[0]
  if type_ == _FLOAT:
    return struct.unpack_from(_FMT_F[parent_byte_width], buf, at)[0]
  target = at - _read_u(buf, at, parent_byte_width)
  if type_ == _KEY:
    end = buf.index(b"\x00", target)
    return buf[target:end].decode("utf-8")
  if type_ == _VECTOR:
    n = _read_u(buf, target - byte_width, byte_width)
    types = buf[target + n * byte_width:target + n * byte_width + n]
    return [_read_value(buf, target + i * byte_width, byte_width, types[i]) for i in range(n)]
  if type_ == _MAP:
    n = _read_u(buf, target - byte_width, byte_width)
    keys_at = target - 3 * byte_width
    keys_target = keys_at - _read_u(buf, keys_at, byte_width)
    keys_width = _read_u(buf, target - 2 * byte_width, byte_width)
    names = [_read_value(buf, keys_target + i * keys_width, keys_width, _KEY << 2) for i in range(n)]
    types = buf[target + n * byte_width:target + n * byte_width + n]
    return {names[i]: _read_value(buf, target + i * byte_width, byte_width, types[i]) for i in range(n)}
  raise ValueError(f"FlexBuffer type {type_} is not one this reader knows")


def decode(buf: bytes):
  """FlexBuffer bytes -> Python value (maps, untyped vectors, ints, floats, keys)."""
  if len(buf) < 3:
    raise ValueError("not a FlexBuffer")
  root_width = buf[-1]
  return _read_value(buf, len(buf) - 2 - root_width, root_width, buf[-2])
