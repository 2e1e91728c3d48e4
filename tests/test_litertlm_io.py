"""`.litertlm` container pass-through (CPU part). Fixture: the reference's own test container."""
import os
import struct

import numpy as np
import pytest

from mi355q.utils import litertlm_utils as L
from mi355q.utils import tflite_flatbuffer as fb

HERE = os.path.dirname(os.path.abspath(__file__))
FIXTURE = os.path.join(HERE, "golden", "models", "conv_fc_mnist.litertlm")
TFLITE = os.path.join(HERE, "golden", "models", "conv_fc_mnist.tflite")


def build_container(path, sections):
  """sections: list of (data_type, {key: str}, payload bytes). Written with our own builder."""
  objs, cursor, placed = [], L.BLOCK_SIZE, []
  for dtype, items, payload in sections:
    kv = [fb.CLASSES["KeyValuePair"](key=k.encode(), valueType=9,
                                     value=fb.CLASSES["StringValue"](value=v.encode())) for k, v in items.items()]
    o = L.SectionObjectT(items=kv, beginOffset=cursor, endOffset=cursor + len(payload), dataType=dtype)
    o.__dict__["_explicit"] = {"beginOffset", "endOffset", "dataType"}
    objs.append(o)
    placed.append((cursor, payload))
    cursor = (cursor + len(payload) + L.BLOCK_SIZE - 1) & ~(L.BLOCK_SIZE - 1)
  meta = fb.CLASSES["LiteRTLMMetaData"](
      systemMetadata=fb.CLASSES["SystemMetadata"](entries=[]),
      sectionMetadata=fb.CLASSES["SectionMetadata"](objects=objs))
  b = fb.Builder()
  root = fb._pack(b, meta, "LiteRTLMMetaData")
  header = b.finish(root, identifier=None)
  assert L.HEADER_BEGIN_BYTE_OFFSET + len(header) <= L.BLOCK_SIZE
  out = bytearray(placed[-1][0] + len(placed[-1][1]))
  out[:8] = L.HEADER_MAGIC_BYTES
  struct.pack_into("<IIII", out, 8, 1, 5, 0, 0)
  struct.pack_into("<Q", out, 24, L.HEADER_BEGIN_BYTE_OFFSET + len(header))
  out[32:32 + len(header)] = header
  for at, payload in placed:
    out[at:at + len(payload)] = payload
  open(path, "wb").write(out)


def test_reference_fixture_parses_and_round_trips(tmp_path):
  f = L.LiteRTLMFile(FIXTURE)
  assert f.version == (1, 5, 0) and len(f.sections) == 1
  assert f.get_system_metadata()["Authors"] == "AI-Edge-Quantizer team"
  assert f.get_model_type(0) == "tf_lite_embedder"
  assert f.sections[0].dataType == L.AnySectionDataType.TFLiteModel
  assert bytes(f.get_section_buffer(0)) == open(TFLITE, "rb").read()
  m = f.read_model(0)
  assert m.version == 3 and len(m.subgraphs) == 1
  out = tmp_path / "same.litertlm"
  n = f.serialize(out, {})
  assert n == os.path.getsize(FIXTURE) and open(out, "rb").read() == open(FIXTURE, "rb").read()


def test_multi_section_repack_moves_following_sections(tmp_path):
  model = open(TFLITE, "rb").read()
  blob = bytes(range(256)) * 100
  src = tmp_path / "three.litertlm"
  build_container(src, [(L.AnySectionDataType.TFLiteModel, {"model_type": "prefill"}, model),
                        (L.AnySectionDataType.GenericBinaryData, {"name": "tokenizer"}, blob),
                        (L.AnySectionDataType.TFLiteModel, {"model_type": "decode"}, model)])
  f = L.LiteRTLMFile(src)
  assert [f.get_model_type(i) for i in range(3)] == ["prefill", None, "decode"]
  assert f.read_model(1) is None and bytes(f.get_section_buffer(1)) == blob
  small = model[:30000]
  dst = tmp_path / "repacked.litertlm"
  n = f.serialize(dst, {0: small})
  g = L.LiteRTLMFile(dst)
  assert n == os.path.getsize(dst) < os.path.getsize(src)
  assert [s.beginOffset % L.BLOCK_SIZE for s in g.sections] == [0, 0, 0]
  assert g.sections[0].endOffset - g.sections[0].beginOffset == len(small)
  assert g.sections[1].beginOffset == L.BLOCK_SIZE + 2 * L.BLOCK_SIZE        # 30000 B -> 2 blocks
  assert bytes(g.get_section_buffer(0)) == small and bytes(g.get_section_buffer(1)) == blob
  assert bytes(g.get_section_buffer(2)) == model
  assert g.get_section_metadata(1) == {"name": "tokenizer"}
  # every header byte except the six patched offsets is carried verbatim
  a, b = open(src, "rb").read(), open(dst, "rb").read()
  assert sum(x != y for x, y in zip(a[:f._header_end], b[:g._header_end])) <= 6 * 8


def test_rejects_bad_files(tmp_path):
  p = tmp_path / "bad.litertlm"
  p.write_bytes(b"NOTLITER" + bytes(64))
  with pytest.raises(ValueError, match="bad magic"):
    L.LiteRTLMFile(p)
  data = bytearray(open(FIXTURE, "rb").read())
  struct.pack_into("<Q", data, 24, len(data) + 5)
  p.write_bytes(data)
  with pytest.raises(ValueError, match="out of range"):
    L.LiteRTLMFile(p)


@pytest.mark.parametrize("expected_is", ["more", "less", "exact"])
def test_output_file_prepared_ahead_of_time(tmp_path, expected_is):
  """LiteRTLMFile.prepare_output creates the output file and allocates (and maps) its pages on a helper thread before
  anything is quantized; open_with_section then sets the length the layout asks for -- whether more, less or exactly as
  much was expected -- and the container built in place is the one serialize() writes."""
  from mi355q import runtime
  f = L.LiteRTLMFile(FIXTURE)
  sid = next(i for i, s in enumerate(f.sections) if f.read_model(i) is not None)
  section = bytes(f.get_section_buffer(sid))
  new = section + b"\x5a" * 12345
  want = tmp_path / "want.litertlm"
  n_want = f.serialize(want, {sid: new})
  out = tmp_path / "out.litertlm"
  expected = {"more": len(new) + 300000, "less": len(new) - 20000, "exact": len(new)}[expected_is]
  f.prepare_output(out, sid, expected)
  assert os.path.exists(out)
  mapping, place, size = f.open_with_section(out, sid, len(new))
  assert size == n_want and len(place) == len(new) and os.path.getsize(out) == n_want
  base = np.frombuffer(mapping, dtype=np.uint8).ctypes.data
  exist = runtime._OUT_PAGES_EXIST.get(base)                   # pylint: disable=protected-access
  assert exist == (n_want if expected_is != "less" else exist) and 0 < exist <= n_want
  place[:] = new
  del place
  f.close_built_in_place()
  assert base not in runtime._OUT_PAGES_EXIST                 # pylint: disable=protected-access
  assert open(out, "rb").read() == open(want, "rb").read()


def test_prepared_output_of_a_call_that_failed_goes_away(tmp_path):
  f = L.LiteRTLMFile(FIXTURE)
  out = tmp_path / "never.litertlm"
  f.prepare_output(out, 0, 1 << 20)
  assert os.path.exists(out)
  f.discard_prepared()
  assert not os.path.exists(out)
  f.discard_prepared()                                          # (nothing left: a no-op)
  # a second preparation replaces the first; a writer that asks for another path gets a file of its own
  f.prepare_output(out, 0, 1 << 20)
  f.prepare_output(out, 0, 2 << 20)
  other = tmp_path / "other.litertlm"
  mapping, place, size = f.open_with_section(other, 0, 4096)
  assert not os.path.exists(out) and os.path.getsize(other) == size
  del place
  f.close_built_in_place()
