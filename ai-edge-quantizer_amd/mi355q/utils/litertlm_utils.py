"""`.litertlm` container pass-through (ref: utils/litertlm_utils.py:69-283, aeq.py:61-181).

A LiteRT-LM file is: 8 magic bytes `LITERTLM`, three little-endian u32 version numbers, 4 zero
bytes, the u64 end offset of the header flatbuffer (file offset 24), the header flatbuffer at
offset 32 (system metadata + one record per section: key/value items, begin/end offsets, data
type), then the sections at 16 KiB-aligned offsets. The reference reads and re-packs the header
through the third-party `litert_lm_builder` wheel; here the header is parsed with this build's
own flatbuffer reader (full byte accounting, so an unknown layout is refused) and re-serialized
by **patching the section offsets in place** in a verbatim copy of the header - every other
byte of it (system metadata, item order, unknown value types) is carried unchanged. (The
reference additionally re-stamps system metadata through the wheel; that step is third-party
behaviour and is not reproduced.)
"""
from __future__ import annotations

import mmap
import os
import pathlib
import struct
from typing import Any, Mapping, Optional, Union

from .. import runtime
from . import tfl_flatbuffer_utils
from . import tflite_flatbuffer as fb

Path = Union[str, pathlib.Path]

HEADER_MAGIC_BYTES = b"LITERTLM"
HEADER_END_LOCATION_BYTE_OFFSET = 24
HEADER_BEGIN_BYTE_OFFSET = 32
BLOCK_SIZE = 16 * 1024


class AnySectionDataType:
  NONE = 0
  GenericBinaryData = 1
  Deprecated = 2
  TFLiteModel = 3


_STRING_VALUE = 9     # VData union member holding a string

# Header schema (checked field by field against the reference's .litertlm test fixture: the
# parse accounts for every byte of the header).
fb.SCHEMA.update({
    "LiteRTLMMetaData": [("systemMetadata", ("tab", "SystemMetadata")),
                         ("sectionMetadata", ("tab", "SectionMetadata"))],
    "SystemMetadata": [("entries", ("vec_tab", "KeyValuePair"))],
    "SectionMetadata": [("objects", ("vec_tab", "SectionObject"))],
    "SectionObject": [("items", ("vec_tab", "KeyValuePair")), ("beginOffset", "u64", 0),
                      ("endOffset", "u64", 0), ("dataType", "u8", 0)],
    "KeyValuePair": [("key", "str"), ("valueType", "u8", 0),
                     ("value", ("union", "valueType", {_STRING_VALUE: "StringValue"}))],
    "StringValue": [("value", "str")],
})
for _t in ("LiteRTLMMetaData", "SystemMetadata", "SectionMetadata", "SectionObject", "KeyValuePair",
           "StringValue"):
  fb.CLASSES[_t] = type(_t + "T", (fb.TableT,), {"_table": _t})
SectionObjectT = fb.CLASSES["SectionObject"]


def _text(v: Any) -> Any:
  return v.decode("utf-8") if isinstance(v, (bytes, bytearray)) else v


def _scalar_of(value: Any) -> Any:
  """Python value of a VData member: the string, or field 0 of a one-scalar table."""
  if isinstance(value, fb.TableT):
    return _text(value.value)
  return value


_KEPT: list = []      # the mapping of the last output file built in place (see _set_aside)


def _set_aside(holder: dict) -> None:
  """The file is complete and closed; what is left of its mapping is address space -- and a gigabyte of populated
  page-table entries, whose munmap() is 45-50 ms during which the process's other threads wait for the address-space lock
  (measured: on a thread of its own the caller's return still took them). The mapping is kept instead and taken down
  where that costs nothing: by the helper thread of the next output file (prepare_output), or by the process's exit."""
  mapping = holder.pop("mapping", None)
  if mapping is not None:
    _KEPT.append(mapping)


def _drop_kept() -> None:
  while _KEPT:
    try:
      _KEPT.pop().close()
    except (BufferError, ValueError):      # (a view of it is still alive somewhere: unmapped when that goes)
      pass


def _populate(mapping, length: int) -> None:
  """Page-table entries for the first `length` bytes of a shared mapping whose pages exist (Linux 5.14+
  MADV_POPULATE_WRITE; anything else: the first touch of each page maps it, as always)."""
  import ctypes
  import numpy as np
  try:
    libc = ctypes.CDLL(None, use_errno=True)
    address = np.frombuffer(mapping, dtype=np.uint8).ctypes.data
    libc.madvise(ctypes.c_void_p(address), ctypes.c_size_t(length), ctypes.c_int(23))     # MADV_POPULATE_WRITE
  except (OSError, AttributeError, ValueError):
    pass


class LiteRTLMFile:
  """Sections of a `.litertlm` file (memory mapped, nothing is copied until serialize)."""

  def __init__(self, path: Path):
    self._path = path
    self._buf = tfl_flatbuffer_utils.get_model_content(path)
    if bytes(self._buf[:8]) != HEADER_MAGIC_BYTES:
      raise ValueError(f"{path} is not a LiteRT-LM file (bad magic)")
    self.version = struct.unpack_from("<III", self._buf, 8)
    self._header_end = struct.unpack_from("<Q", self._buf, HEADER_END_LOCATION_BYTE_OFFSET)[0]
    if not HEADER_BEGIN_BYTE_OFFSET < self._header_end <= len(self._buf):
      raise ValueError(f"{path}: header end offset {self._header_end} is out of range")
    header = self._buf[HEADER_BEGIN_BYTE_OFFSET:self._header_end]
    reader = fb._Reader(header)  # pylint: disable=protected-access
    reader.mark(0, 4)
    self._offset_fields: list[tuple[int, int]] = []   # (begin, end) field positions per section
    self._metadata = reader.table(reader.u32(0), "LiteRTLMMetaData")
    reader.verify_coverage(0, len(header))
    self._sections = list((self._metadata.sectionMetadata and self._metadata.sectionMetadata.objects) or [])
    self._locate_offset_fields(reader)
    for s in self._sections:
      if not self._header_end <= s.beginOffset <= s.endOffset <= len(self._buf):
        raise ValueError(f"{path}: section [{s.beginOffset}, {s.endOffset}) is out of range")

  def _locate_offset_fields(self, reader) -> None:
    """Positions (inside the header) of every section's begin/end offset scalars, for patching."""
    root = reader.u32(0)
    fields, _ = reader.table_header(root)
    sm = root + fields[1]
    sm = sm + reader.u32(sm)
    sm_fields, _ = reader.table_header(sm)
    vec = sm + sm_fields[0]
    vec = vec + reader.u32(vec)
    for i in range(reader.u32(vec)):
      at = vec + 4 + 4 * i
      obj = at + reader.u32(at)
      f, _ = reader.table_header(obj)
      begin = obj + f[1] if len(f) > 1 and f[1] else -1
      end = obj + f[2] if len(f) > 2 and f[2] else -1
      self._offset_fields.append((begin, end))

  @property
  def sections(self) -> list[Any]:
    return self._sections

  def get_system_metadata(self) -> dict[str, Any]:
    sm = self._metadata.systemMetadata
    return {_text(e.key): _scalar_of(e.value) for e in (sm.entries if sm and sm.entries else [])}

  def get_section_metadata(self, section_id: int) -> dict[str, Any]:
    return {_text(i.key): _scalar_of(i.value) for i in (self._sections[section_id].items or [])}

  def get_model_type(self, section_id: int) -> Optional[str]:
    v = self.get_section_metadata(section_id).get("model_type")
    return v if isinstance(v, str) else None

  def get_section_buffer(self, section_id: int) -> memoryview:
    s = self._sections[section_id]
    return self._buf[s.beginOffset:s.endOffset]

  def read_model(self, section_id: int) -> Optional[Any]:
    if self._sections[section_id].dataType != AnySectionDataType.TFLiteModel:
      return None
    return tfl_flatbuffer_utils.read_model(self.get_section_buffer(section_id))

  def _layout(self, new_lengths: Mapping[int, int]):
    """(header bytes with patched offsets, section offsets, section lengths, total) when the
    sections in `new_lengths` get those lengths (ref :176-283: order kept, BLOCK_SIZE alignment)."""
    offsets = [min(s.beginOffset for s in self._sections)]
    lengths = []
    for sid, s in enumerate(self._sections):
      n = new_lengths.get(sid) or s.endOffset - s.beginOffset
      lengths.append(n)
      offsets.append((offsets[-1] + n + BLOCK_SIZE - 1) & ~(BLOCK_SIZE - 1))
    header = bytearray(self._buf[:self._header_end])
    for sid, (begin_at, end_at) in enumerate(self._offset_fields):
      if begin_at < 0 or end_at < 0:
        raise ValueError("section record without stored offsets cannot be re-addressed in place")
      struct.pack_into("<Q", header, HEADER_BEGIN_BYTE_OFFSET + begin_at, offsets[sid])
      struct.pack_into("<Q", header, HEADER_BEGIN_BYTE_OFFSET + end_at, offsets[sid] + lengths[sid])
    return header, offsets, lengths, offsets[-2] + lengths[-1]

  def prepare_output(self, path: Path, section_id: int, expected_section_bytes: int) -> None:
    """Creates the output file NOW -- before anything is quantized -- and has its pages allocated on a helper thread
    (posix_fallocate of the container's expected size: header, the other sections, `expected_section_bytes` for the
    one that will change). A gigabyte of fresh page-cache pages is 65 ms of allocation on a 256-core host and makes
    pwrite() 8 instead of 14 GB/s (tools/tmpfs_write_probe.py); done here it sits underneath calibration instead of
    behind the last kernel, where a C5 mixed call spent 0.15-0.33 of its 0.6 s waiting for the file
    (profiles/r05_c5_mixed_tail.txt). The expectation may be wrong either way: open_with_section sets the length it
    finds out, pages past it are given back, pages short of it are allocated by the writes as before."""
    import threading
    if not self._sections or expected_section_bytes <= 0:
      return
    self.discard_prepared()
    _, _, _, total = self._layout({section_id: int(expected_section_bytes)})
    fd = os.open(path, os.O_RDWR | os.O_CREAT | os.O_TRUNC, 0o644)
    state = {"path": path, "fd": fd, "bytes": total, "error": None}

    def allocate():
      try:
        _drop_kept()                 # (the previous output's mapping, if this process made one)
        os.posix_fallocate(fd, 0, total)
        # ... and mapped in: the page-table entries of 1 GB are 30 ms of minor faults for the threads that copy into the
        # mapping otherwise (16 against 30 GB/s, tools/tmpfs_write_probe.py). madvise(MADV_POPULATE_WRITE) through ctypes,
        # which lets go of the interpreter lock -- mmap.mmap(..., MAP_POPULATE) holds it for the 60 ms this takes, and the
        # thread that feeds the GPU stood still for them
        state["mapping"] = mmap.mmap(fd, total)
        _populate(state["mapping"], total)
      except (OSError, ValueError) as e:   # (a file system without room or without fallocate: the writes allocate, as before)
        state["error"] = e
    state["thread"] = threading.Thread(target=allocate, name="mi355q-output-pages", daemon=True)
    state["thread"].start()
    self._prepared = state

  def discard_prepared(self, remove: bool = True) -> None:
    """Drops a file prepare_output made and nobody built into (a call that failed before its writer ran)."""
    state = getattr(self, "_prepared", None)
    if state is None:
      return
    self._prepared = None
    state["thread"].join()
    if state.get("mapping") is not None:
      state["mapping"].close()
    os.close(state["fd"])
    if remove:
      try:
        os.remove(state["path"])
      except OSError:
        pass

  def open_with_section(self, path: Path, section_id: int, section_bytes: int):
    """Creates the output file for a container whose section `section_id` will be `section_bytes`
    long, writes the header and every other section, and returns (mapping, writable view of the
    section's place, total size): the new section is then built in place (a quantized model is
    serialized once, into the file, instead of into memory and from there into the file)."""
    if not self._sections:
      raise ValueError("LiteRT-LM file has no sections")
    self.close_built_in_place()      # (a writer that builds its model a second time asks again)
    header, offsets, lengths, total = self._layout({section_id: section_bytes})
    prepared = getattr(self, "_prepared", None)
    out, pages_exist = None, 0
    if prepared is not None and os.path.abspath(prepared["path"]) == os.path.abspath(path):
      self._prepared = None
      prepared["thread"].join()
      fd = prepared["fd"]            # (its pages are there; the length becomes what the layout says)
      early = prepared.get("mapping")
      if early is not None and prepared["error"] is None:
        pages_exist = min(prepared["bytes"], total)
        if prepared["bytes"] >= total:
          out = early                # longer than the file from here on: nothing past `total` is ever touched
        else:
          early.close()              # (expected too little: a mapping of the right length, its last pages fresh)
    else:
      self.discard_prepared()
      fd = os.open(path, os.O_RDWR | os.O_CREAT | os.O_TRUNC, 0o644)
    os.ftruncate(fd, total)
    if out is None:
      out = mmap.mmap(fd, total)
    # device-resident buffers of the model reach the file through the io ring (pinned staging, then pwrite() -- or a copy
    # into this mapping where its pages exist); the caller finishes the writes and closes `fd` (close_built_in_place)
    runtime.register_output_mapping(out, fd, pages_exist=pages_exist)
    self._built_in_place = (out, fd)
    out[:len(header)] = header
    for sid in range(len(self._sections)):
      if sid != section_id:
        out[offsets[sid]:offsets[sid] + lengths[sid]] = memoryview(self.get_section_buffer(sid)).cast("B")
    return out, memoryview(out)[offsets[section_id]:offsets[section_id] + section_bytes], total

  def close_built_in_place(self) -> None:
    """Waits for the io ring's writes into the file open_with_section made and closes its descriptor."""
    opened = getattr(self, "_built_in_place", None)
    if opened is None:
      return
    self._built_in_place = None
    mapping, fd = opened
    try:
      runtime.finish_downloads()
    finally:
      runtime.forget_output_mapping(mapping)
      os.close(fd)

  def serialize(self, path: Path, section_data_overrides: Mapping[int, Any]) -> int:
    """Writes the file again with some sections replaced; returns the number of bytes written.
    Sections keep their order and start at BLOCK_SIZE-aligned offsets (ref :176-283)."""
    if not self._sections:
      raise ValueError("LiteRT-LM file has no sections")
    new_lengths = {sid: len(d) for sid, d in section_data_overrides.items() if d is not None and len(d)}
    header, offsets, lengths, total = self._layout(new_lengths)
    with open(path, "w+b") as f:
      f.truncate(total)
      out = mmap.mmap(f.fileno(), total)
    out[:len(header)] = header
    for sid in range(len(self._sections)):
      data = section_data_overrides.get(sid)
      if data is None or not len(data):
        data = self.get_section_buffer(sid)
      out[offsets[sid]:offsets[sid] + lengths[sid]] = memoryview(data).cast("B")
    out.flush()
    out.close()
    return total


def _pick(table: Optional[Mapping], sid: int, model_type: Optional[str]) -> Any:
  """Entry of a per-model table: by section index, by model type, else "default"."""
  if table is None:
    return None
  for key in (sid, model_type, "default"):
    if key is not None and key in table:
      return table[key]
  return None


def _recipes(recipe: Any) -> Mapping:
  return recipe if isinstance(recipe, Mapping) else {"default": recipe}


def _tflite_sections(src: "LiteRTLMFile", recipes: Mapping):
  """(section index, model type, recipe) of every TFLite section that has a recipe."""
  for sid, section in enumerate(src.sections):
    if section.dataType != AnySectionDataType.TFLiteModel:
      continue
    model_type = src.get_model_type(sid)
    if model_type is None:
      continue
    model_recipe = _pick(recipes, sid, model_type)
    if model_recipe is not None:
      yield sid, model_type, model_recipe


def calibrate_litertlm(litertlm_path: Path, recipe: Any, calibration_data: Mapping,
                       previous_calibration_results: Optional[Mapping] = None,
                       tensor_provider: Optional[Any] = None, group: Any = None) -> dict[int, dict]:
  """Model QSVs of every TFLite section whose recipe needs calibration: {section index: QSVs}.

  `calibration_data` maps a section (index, model type or "default") to that model's
  {signature key: samples}; a sample is the {tensor name: content} map of one run of the float
  model (see Calibrator). With a process group the samples of every signature are sharded over
  the ranks (distributed.calibrate_sharded: per-sample statistics gathered and replayed in
  dataset order, GPTQ Hessians all-reduced); every rank returns the same QSVs.
  """
  from .. import distributed
  src = LiteRTLMFile(litertlm_path)
  out: dict[int, dict] = {}
  for sid, model_type, model_recipe in _tflite_sections(src, _recipes(recipe)):
    data = _pick(calibration_data, sid, model_type)
    if data is None:
      continue
    qsvs = distributed.calibrate_sharded(
        src.get_section_buffer(sid), model_recipe, data,
        previous_calibration_result=_pick(previous_calibration_results, sid, model_type),
        tensor_provider=tensor_provider, group=group)
    if qsvs:
      out[sid] = qsvs
  return out


def quantize_litertlm(litertlm_path: Path, recipe: Any, output_path: Path, overwrite: bool = False,
                      calibration_results: Optional[Mapping] = None, group: Any = None,
                      calibration_data: Optional[Mapping] = None, tensor_provider: Optional[Any] = None,
                      stats: Optional[dict] = None) -> Optional[int]:
  """Quantizes every TFLite section that has a recipe and re-packs the container.

  `recipe` is one recipe (list) applied to every model, or a mapping model_type -> recipe with
  an optional "default" entry (ref aeq.py:61-181). `calibration_results` maps a section (index,
  model type or "default") to the model QSVs its recipe needs (static recipes, GPTQ, OSCAR:
  what Quantizer.calibrate / calibrate_litertlm returned) -- the reference's loop passes none
  and therefore cannot run such recipes on a container. `calibration_data` (same keys; a model's
  {signature key: samples}, see calibrate_litertlm) instead has the statistics collected here, in
  the same call (distributed.calibrate_and_quantize_sharded: with several ranks every GPTQ Hessian
  is reduced straight to the rank that will read it). With a process group (`group`, or an
  initialised default group) the ops of every model are spread over the ranks by cost and by
  shared statistics (distributed.quantize_model_sharded); rank 0 writes the file and returns
  its size, the other ranks return None. `stats` receives wall seconds per phase.
  """
  from .. import distributed
  if os.path.exists(output_path) and not overwrite:
    raise ValueError(f"The model {output_path} already exists. Specify overwrite=True to replace it.")
  import time
  src = LiteRTLMFile(litertlm_path)
  todo = list(_tflite_sections(src, _recipes(recipe)))
  if not todo:
    raise ValueError("No models were quantized, not creating output file.")
  replaced: dict[int, Any] = {}
  built_in_place: dict = {}
  try:
    for sid, model_type, model_recipe in todo:
      sink = None
      if len(todo) == 1:
        # the only section that changes: once its size is known the container is laid out and the
        # model is serialized straight into its place in the output file
        def sink(total, sid=sid):
          t0 = time.perf_counter()
          mapping, place, size = src.open_with_section(output_path, sid, total)
          built_in_place.update(mapping=mapping, size=size)
          if stats is not None:
            stats["repack_s"] = stats.get("repack_s", 0.0) + (time.perf_counter() - t0)
            stats["section_bytes"] = int(total)
          return place
        # (whoever plans the model before it is quantized says how long it expects the section to get: the file is
        # created and its pages are allocated underneath the calibration, LiteRTLMFile.prepare_output)
        def expect(expected, sid=sid):
          if stats is not None:
            stats["expected_section_bytes"] = int(expected)
          src.prepare_output(output_path, sid, expected)
        sink.expect = expect
      data = _pick(calibration_data, sid, model_type)
      if data is not None:
        result = distributed.calibrate_and_quantize_sharded(
            src.get_section_buffer(sid), model_recipe, data, group=group, tensor_provider=tensor_provider, stats=stats,
            sink=sink)
      else:
        result = distributed.quantize_model_sharded(
            src.get_section_buffer(sid), model_recipe,
            calibration_result=_pick(calibration_results, sid, model_type), group=group, sink=sink)
      if result is not None:
        replaced[sid] = result
  finally:
    runtime.mark("container: model sections done (host)")
    src.discard_prepared()         # (only a call that failed before its writer ran still has one: that file goes away)
    src.close_built_in_place()     # (the io ring's writes into the output file are done, its descriptor is closed)
    runtime.mark("container: output file complete (host)")
  if built_in_place:
    # (no msync: a shared mapping is coherent with the page cache, readers see the bytes at once)
    size = built_in_place["size"]
    _set_aside(built_in_place)
    return size
  if not replaced:          # a rank other than the group's first: nothing to write
    return None
  t0 = time.perf_counter()
  n = src.serialize(output_path, replaced)
  if stats is not None:
    stats["repack_s"] = stats.get("repack_s", 0.0) + (time.perf_counter() - t0)
  return n
