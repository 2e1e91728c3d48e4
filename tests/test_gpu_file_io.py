"""The file path's io ring (csrc/file_io.hip; ref utils/tfl_flatbuffer_utils.py:142-163, model_modifier.py:290-391):
model bytes must reach HBM and quantized bytes must reach the output file unchanged, through the C ABI directly and
through Quantizer.quantize with the ring forced on for a small model."""
import ctypes
import hashlib
import os
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def m():
  import torch
  assert torch.cuda.is_available()
  import __graft_entry__ as g
  g.build()
  import types
  from mi355q import _ffi, quantizer, recipe, runtime
  return types.SimpleNamespace(torch=torch, ffi=_ffi, rt=runtime, quantizer=quantizer, recipe=recipe)


@pytest.mark.parametrize("nbytes,offset", [(1, 0), (4097, 3), (8 << 20, 0), ((8 << 20) + 1, 1824), (45_000_001, 77), (0, 5)])
def test_file_to_device_and_back(m, tmp_path, nbytes, offset):
  """Ranges shorter than a part, exactly one slot, one byte over a slot, several slots with a ragged tail, at
  unaligned file offsets: the device holds the file's bytes, and the file written back from the device (into the
  middle of a larger file) holds them again with its other bytes untouched."""
  torch, L = m.torch, m.ffi.lib()
  rng = np.random.default_rng(nbytes + offset)
  blob = rng.integers(0, 256, offset + nbytes + 11, dtype=np.uint8)
  src = str(tmp_path / "in.bin")
  blob.tofile(src)
  stream = torch.cuda.Stream()
  dev = torch.zeros(max(nbytes, 1), dtype=torch.uint8, device="cuda")
  torch.cuda.synchronize()           # (the fill runs on the default stream, the copies on `stream`)
  fd = os.open(src, os.O_RDONLY)
  try:
    m.ffi.check(L.mi355q_file_to_device(fd, offset, nbytes, ctypes.c_void_p(dev.data_ptr()), ctypes.c_void_p(stream.cuda_stream)))
  finally:
    os.close(fd)
  stream.synchronize()
  assert np.array_equal(dev.cpu().numpy()[:nbytes], blob[offset:offset + nbytes])
  dst = str(tmp_path / "out.bin")
  frame = rng.integers(0, 256, nbytes + 200, dtype=np.uint8)
  frame.tofile(dst)
  fd = os.open(dst, os.O_RDWR)
  try:
    m.ffi.check(L.mi355q_device_to_file(ctypes.c_void_p(dev.data_ptr()), nbytes, fd, 100, ctypes.c_void_p(stream.cuda_stream)))
    m.ffi.check(L.mi355q_file_io_finish())
  finally:
    os.close(fd)
  back = np.fromfile(dst, dtype=np.uint8)
  assert np.array_equal(back[100:100 + nbytes], blob[offset:offset + nbytes])
  assert np.array_equal(back[:100], frame[:100]) and np.array_equal(back[100 + nbytes:], frame[100 + nbytes:])


def test_short_file_and_bad_descriptor_are_io_errors(m, tmp_path):
  """A range that runs past the end of the file, and a descriptor that is not open: MI355Q_IO_ERROR with a
  message, nothing hangs, and the ring works again afterwards."""
  torch, L = m.torch, m.ffi.lib()
  src = str(tmp_path / "short.bin")
  np.arange(5000, dtype=np.uint8).tofile(src)
  stream = torch.cuda.Stream()
  dev = torch.zeros(1 << 20, dtype=torch.uint8, device="cuda")
  torch.cuda.synchronize()
  fd = os.open(src, os.O_RDONLY)
  try:
    st = L.mi355q_file_to_device(fd, 0, 1 << 20, ctypes.c_void_p(dev.data_ptr()), ctypes.c_void_p(stream.cuda_stream))
    assert st == -6 and b"end of file" in L.mi355q_last_error()
    st = L.mi355q_file_to_device(fd, 0, 5000, ctypes.c_void_p(dev.data_ptr()), ctypes.c_void_p(stream.cuda_stream))
    assert st == 0
  finally:
    os.close(fd)
  stream.synchronize()
  assert np.array_equal(dev.cpu().numpy()[:5000], np.arange(5000, dtype=np.uint8))
  st = L.mi355q_file_to_device(fd, 0, 4096, ctypes.c_void_p(dev.data_ptr()), ctypes.c_void_p(stream.cuda_stream))   # closed above
  assert st == -6 and L.mi355q_last_error()
  st = L.mi355q_device_to_file(ctypes.c_void_p(dev.data_ptr()), 4096, fd, 0, ctypes.c_void_p(stream.cuda_stream))
  assert st == 0 and L.mi355q_file_io_finish() == -6 and L.mi355q_last_error()
  assert L.mi355q_file_io_finish() == 0


def test_submitted_transfers_do_not_hold_the_caller_and_arrive_whole(m, tmp_path):
  """mi355q_file_io_submit_upload / _wait / _submit_download (round 4): several uploads queued at once return before
  they ran, each ticket's wait hands over the file's bytes (the consumer ordered behind an event recorded on the copy
  stream after the wait); downloads gated on a producer's event write the PRODUCED bytes (a kernel queued behind a
  long one), at the same time as uploads; a short file is the waiting ticket's MI355Q_IO_ERROR and a bad descriptor
  of a download is reported by mi355q_file_io_finish, once."""
  torch, L = m.torch, m.ffi.lib()
  rng = np.random.default_rng(11)
  sizes = [(20 << 20) + 3, 4097, (8 << 20), 1]
  blobs, fds, devs, tickets = [], [], [], []
  up, down = torch.cuda.Stream(), torch.cuda.Stream()
  for i, n in enumerate(sizes):
    blob = rng.integers(0, 256, n + 9, dtype=np.uint8)
    path = str(tmp_path / f"in{i}.bin")
    blob.tofile(path)
    blobs.append(blob)
    fds.append(os.open(path, os.O_RDONLY))
    devs.append(torch.zeros(n, dtype=torch.uint8, device="cuda"))
  torch.cuda.synchronize()
  out_path = str(tmp_path / "out.bin")
  np.zeros(sum(sizes) + 64, np.uint8).tofile(out_path)
  out_fd = os.open(out_path, os.O_RDWR)
  try:
    for fd, n, dev in zip(fds, sizes, devs):
      t = ctypes.c_int64(0)
      m.ffi.check(L.mi355q_file_io_submit_upload(fd, 7, n, ctypes.c_void_p(dev.data_ptr()), ctypes.c_void_p(up.cuda_stream), ctypes.byref(t)))
      tickets.append(t.value)
    assert len(set(tickets)) == len(tickets) and all(t > 0 for t in tickets)
    # a producer that finishes late: the download of its result must wait for IT
    big = torch.randn((8192, 8192), device="cuda")
    for _ in range(20):
      big = big @ big * 1e-4
    produced = (torch.arange(5 << 20, device="cuda", dtype=torch.int32) % 251).to(torch.uint8)
    ready = torch.cuda.Event()
    ready.record()
    m.ffi.check(L.mi355q_file_io_submit_download(ctypes.c_void_p(produced.data_ptr()), produced.numel(), out_fd, 32,
                                                 ctypes.c_void_p(down.cuda_stream), ctypes.c_void_p(ready.cuda_event)))
    for t, n, dev, blob in zip(tickets, sizes, devs, blobs):
      m.ffi.check(L.mi355q_file_io_wait(t))
      done = torch.cuda.Event()
      done.record(up)
      torch.cuda.current_stream().wait_event(done)
      assert np.array_equal((dev + 0).cpu().numpy(), blob[7:7 + n])       # (dev + 0: a kernel on the consumer stream)
    assert L.mi355q_file_io_wait(tickets[0]) == -1                         # a ticket is waited for once
    m.ffi.check(L.mi355q_file_io_finish())
    back = np.fromfile(out_path, np.uint8)
    assert np.array_equal(back[32:32 + produced.numel()], produced.cpu().numpy()) and not back[:32].any()
    # failures: a range past the end of the file belongs to its ticket; a closed descriptor of a download to finish()
    t = ctypes.c_int64(0)
    m.ffi.check(L.mi355q_file_io_submit_upload(fds[1], 0, 1 << 20, ctypes.c_void_p(devs[0].data_ptr()), ctypes.c_void_p(up.cuda_stream), ctypes.byref(t)))
    assert L.mi355q_file_io_wait(t.value) == -6 and b"end of file" in L.mi355q_last_error()
  finally:
    for fd in fds:
      os.close(fd)
    os.close(out_fd)
  m.ffi.check(L.mi355q_file_io_submit_download(ctypes.c_void_p(produced.data_ptr()), 4096, out_fd, 0, ctypes.c_void_p(down.cuda_stream), None))
  assert L.mi355q_file_io_finish() == -6 and L.mi355q_last_error()
  assert L.mi355q_file_io_finish() == 0
  del big


def test_downloads_into_the_output_files_own_mapping(m, tmp_path):
  """mi355q_file_io_submit_download_mapped (round 5): the destination is memory -- the output file's shared mapping, its
  pages allocated ahead of time -- and the io threads copy instead of pwrite(). Ragged sizes and offsets, mixed with
  pwritten downloads into the same file, gated on a late producer; what a reader of the FILE sees are the produced bytes;
  null pointers are MI355Q_BAD_ARG. runtime.download_into_file takes this route exactly for payloads inside the part of
  a registered mapping whose pages exist."""
  import mmap
  torch, L = m.torch, m.ffi.lib()
  total = (40 << 20) + 4096
  path = str(tmp_path / "mapped.bin")
  fd = os.open(path, os.O_RDWR | os.O_CREAT, 0o600)
  os.posix_fallocate(fd, 0, total)
  mm = mmap.mmap(fd, total)
  view = np.frombuffer(mm, dtype=np.uint8)
  base = view.ctypes.data
  down = torch.cuda.Stream()
  big = torch.randn((8192, 8192), device="cuda")
  for _ in range(10):
    big = big @ big * 1e-4
  pieces = [(5, (17 << 20) + 3), ((17 << 20) + 64, (9 << 20) + 1), ((27 << 20), 4097), ((28 << 20), 1), ((29 << 20) + 7, 8 << 20)]
  made = []
  try:
    for k, (off, n) in enumerate(pieces):
      t = ((torch.arange(n, device="cuda", dtype=torch.int32) * (k + 3)) % 253).to(torch.uint8)
      ready = torch.cuda.Event()
      ready.record()
      made.append((off, t, ready))
      if k == 2:      # one through the descriptor, between the mapped ones
        m.ffi.check(L.mi355q_file_io_submit_download(ctypes.c_void_p(t.data_ptr()), n, fd, off, ctypes.c_void_p(down.cuda_stream),
                                                     ctypes.c_void_p(ready.cuda_event)))
      else:
        m.ffi.check(L.mi355q_file_io_submit_download_mapped(ctypes.c_void_p(t.data_ptr()), n, ctypes.c_void_p(base + off),
                                                            ctypes.c_void_p(down.cuda_stream), ctypes.c_void_p(ready.cuda_event)))
    m.ffi.check(L.mi355q_file_io_finish())
    back = np.fromfile(path, np.uint8)
    covered = np.zeros(total, bool)
    for off, t, _ in made:
      assert np.array_equal(back[off:off + t.numel()], t.cpu().numpy()), off
      covered[off:off + t.numel()] = True
    assert not back[~covered].any()
    assert L.mi355q_file_io_submit_download_mapped(None, 16, ctypes.c_void_p(base), None, None) == -1
    assert L.mi355q_file_io_submit_download_mapped(ctypes.c_void_p(made[0][1].data_ptr()), 16, None, None, None) == -1
    assert L.mi355q_file_io_submit_download_mapped(ctypes.c_void_p(made[0][1].data_ptr()), -1, ctypes.c_void_p(base), None, None) == -1
    assert L.mi355q_file_io_submit_download_mapped(None, 0, None, None, None) == 0
    # the host side: inside the part whose pages exist -> copied into the mapping; past it -> pwritten; both arrive
    from mi355q import runtime as rt
    calls = []
    real_mapped, real_file = rt._submit_download_mapped, rt._submit_download      # pylint: disable=protected-access
    rt._submit_download_mapped = lambda *a, **k: (calls.append("mapped"), real_mapped(*a, **k))[1]
    rt._submit_download = lambda *a, **k: (calls.append("pwrite"), real_file(*a, **k))[1]
    try:
      rt.register_output_mapping(mm, fd, pages_exist=20 << 20)
      a = (torch.arange(1 << 20, device="cuda", dtype=torch.int32) % 199).to(torch.uint8)
      assert rt.download_into_file(a, view[(1 << 20):(2 << 20)]) and rt.download_into_file(a, view[(30 << 20):(31 << 20)])
      assert rt.download_into_file(a, view[(19 << 20) + 1:(20 << 20) + 1])          # straddles the boundary: pwritten
      rt.finish_downloads()
    finally:
      rt._submit_download_mapped, rt._submit_download = real_mapped, real_file      # pylint: disable=protected-access
      rt.forget_output_mapping(mm)
    assert calls == ["mapped", "pwrite", "pwrite"]
    back = np.fromfile(path, np.uint8)
    for lo in ((1 << 20), (30 << 20), (19 << 20) + 1):
      assert np.array_equal(back[lo:lo + (1 << 20)], a.cpu().numpy())
  finally:
    del view
    mm.close()
    os.close(fd)
  del big


def _sha(path):
  h = hashlib.sha256()
  with open(path, "rb") as f:
    for chunk in iter(lambda: f.read(1 << 24), b""):
      h.update(chunk)
  return h.hexdigest()


@pytest.mark.parametrize("recipe_name", ["dynamic_wi4b128_afp32", "dynamic_wi8_afp32", "dynamic_wi4_afp32"])
def test_quantizer_through_the_ring_writes_the_same_file(m, tmp_path, monkeypatch, recipe_name):
  """The ring is for model files of a GiB and more; with the thresholds taken down a 4-layer model goes through
  it (uploads by pread, output by pwrite) and the file must equal the pageable path's byte for byte -- also when
  it replaces an older, longer file of the same name (no O_TRUNC: the length is set explicitly). Blockwise int4
  (scale tensors as late constants), per-channel int8 and int4 (scales as late vectors of the flatbuffer)."""
  sys.path.insert(0, os.path.join(ROOT, "tools"))
  import file_bench
  src = str(tmp_path / "small.tflite")
  file_bench.build_model(src, 4, 1024, 2048 + 128)
  rcp = getattr(m.recipe, recipe_name)()
  plain = str(tmp_path / "plain.tflite")
  m.quantizer.Quantizer(src, rcp).quantize(serialize_to_path=plain)
  calls = {"up": 0, "down": 0}
  up, down = m.rt.upload_overlapped, m.rt.download_into_file
  monkeypatch.setattr(m.rt, "_UPLOAD_MIN_FILE_BYTES", 1)
  monkeypatch.setattr(m.rt, "_UPLOAD_MIN_TENSOR_BYTES", 1)

  def counted_up(a):
    calls["up"] += 1
    return up(a)

  def counted_down(t, d, *gate):
    took = down(t, d, *gate)
    calls["down"] += int(took)
    return took
  monkeypatch.setattr(m.rt, "upload_overlapped", counted_up)
  monkeypatch.setattr(m.rt, "download_into_file", counted_down)
  ring = str(tmp_path / "ring.tflite")
  with open(ring, "wb") as f:
    f.write(b"\xff" * (os.path.getsize(plain) + 12345))       # an older, longer file in the way
  m.quantizer.Quantizer(src, rcp).quantize(serialize_to_path=ring)
  assert calls["up"] >= 4 and calls["down"] >= 3, calls   # (a buffer somebody already read on the host is copied from there)
  assert os.path.getsize(ring) == os.path.getsize(plain)
  assert _sha(ring) == _sha(plain)
  # ... and the one written with every value read where the reference reads it (no late vectors, no late constants)
  first = str(tmp_path / "values_first.tflite")
  with monkeypatch.context() as mp:
    mp.setattr(m.rt, "late_vector", lambda values, dtype: np.ravel(values).astype(dtype, copy=False))
    mp.setattr(m.rt, "late_constants_allowed", lambda: False)
    m.quantizer.Quantizer(src, rcp).quantize(serialize_to_path=first)
  assert _sha(first) == _sha(plain)


def test_equal_blockwise_scales_share_one_buffer_whether_read_early_or_late(m, tmp_path, monkeypatch):
  """Two layers with the same weights have the same blockwise scales, and the reference's add_new_constant_tensor
  shares one buffer between equal constants (ref transformation_utils.py:119-164). While a file is written the
  scales stay in HBM and are laid out unread on the assumption that they differ; the writer verifies it once the
  values exist and, here, has to build the model again. The file must be the one the values-first path writes."""
  sys.path.insert(0, os.path.join(ROOT, "tools"))
  import file_bench
  from mi355q.transformations import transformation_utils as tu
  from mi355q.utils import tfl_flatbuffer_utils
  src = str(tmp_path / "twins.tflite")
  file_bench.build_model(src, 4, 1024, 2048 + 128, same=(2,))
  rcp = m.recipe.dynamic_wi4b128_afp32()
  monkeypatch.setattr(m.rt, "_UPLOAD_MIN_FILE_BYTES", 1)
  monkeypatch.setattr(m.rt, "_UPLOAD_MIN_TENSOR_BYTES", 1)
  early = str(tmp_path / "early.tflite")
  with monkeypatch.context() as mp:
    mp.setattr(m.rt, "late_constants_allowed", lambda: False)
    m.quantizer.Quantizer(src, rcp).quantize(serialize_to_path=early)
  undecided = []
  verify = tu.verify_late_constants

  def counted(model):
    try:
      verify(model)
    except tu.SharingNotDecided as e:
      undecided.append(str(e))
      raise
  monkeypatch.setattr(tu, "verify_late_constants", counted)
  late = str(tmp_path / "late.tflite")
  m.quantizer.Quantizer(src, rcp).quantize(serialize_to_path=late)
  assert len(undecided) == 1, undecided
  assert _sha(late) == _sha(early)
  model = tfl_flatbuffer_utils.read_model(late)
  by_name = {bytes(t.name): t.buffer for t in model.subgraphs[0].tensors}
  assert by_name[b"w0_scales"] == by_name[b"w2_scales"] != by_name[b"w1_scales"]
  # ... and with distinct weights nothing is built twice
  del undecided[:]
  src2 = str(tmp_path / "distinct.tflite")
  file_bench.build_model(src2, 4, 1024, 2048 + 128)
  m.quantizer.Quantizer(src2, rcp).quantize(serialize_to_path=str(tmp_path / "d.tflite"))
  assert not undecided


def test_container_written_through_the_ring_equals_the_pageable_copy(m, tmp_path, monkeypatch):
  """A one-layer decoder-shaped `.litertlm` (projections of 1 MB and more once packed) quantized in place
  (LiteRTLMFile.open_with_section registers the output mapping: device-resident buffers leave by pinned
  staging + pwrite) against the same call with the ring switched off on both sides."""
  sys.path.insert(0, os.path.join(ROOT, "tools"))
  import c5_model
  from mi355q.utils import litertlm_utils
  src = str(tmp_path / "one_layer.litertlm")
  c5_model.write_litertlm(c5_model.build_model(1, 512, 128, 4096), src)
  rcp = c5_model.recipe("hadamard", 4, max_hadamard_size=4096)
  plain = str(tmp_path / "plain.litertlm")
  with monkeypatch.context() as mp:
    mp.setattr(m.rt, "download_into_file", lambda t, d, *gate: False)
    n_plain = litertlm_utils.quantize_litertlm(src, rcp, plain)
  took = {"down": 0, "up": 0}
  down, up = m.rt.download_into_file, m.rt.upload_overlapped

  def counted_down(t, d, *gate):
    ok = down(t, d, *gate)
    took["down"] += int(ok)
    return ok

  def counted_up(a):
    took["up"] += 1
    return up(a)
  monkeypatch.setattr(m.rt, "_UPLOAD_MIN_FILE_BYTES", 1)
  monkeypatch.setattr(m.rt, "_UPLOAD_MIN_TENSOR_BYTES", 1)
  monkeypatch.setattr(m.rt, "download_into_file", counted_down)
  monkeypatch.setattr(m.rt, "upload_overlapped", counted_up)
  ring = str(tmp_path / "ring.litertlm")
  n_ring = litertlm_utils.quantize_litertlm(src, rcp, ring)
  assert took["down"] >= 3 and took["up"] >= 7, took
  assert n_ring == n_plain == os.path.getsize(ring) == os.path.getsize(plain)
  assert _sha(ring) == _sha(plain)


def test_gptq_container_written_late_equals_the_values_first_no_prefetch_call(m, tmp_path, monkeypatch):
  """BASELINE config 5's call on a small container (one decoder layer at d = 512 / ff = 4096, GPTQ int4 channelwise,
  calibration samples resident in HBM), once as shipped -- weights announced to the upload thread and pumped during
  calibration, placeholders kept until the writer, per-channel scales as late vectors, payloads behind their own events,
  the large-inverse workspace from the helper thread -- and once with all of that off (MI355Q_NO_PREFETCH, scales and
  constants read on the spot, no workspace): the two containers must be the same bytes."""
  sys.path.insert(0, os.path.join(ROOT, "tools"))
  import c5_model
  from mi355q import ops
  from mi355q.utils import litertlm_utils
  torch = m.torch
  d, dkv, dff = 512, 128, 4096
  src = str(tmp_path / "one_layer.litertlm")
  c5_model.write_litertlm(c5_model.build_model(1, d, dkv, dff), src)
  rcp = c5_model.recipe("gptq", 4)
  data = {0: {"serving_default": c5_model.calibration_set(torch, 1, 24, 256, d, dkv, dff, 1, None)}}
  monkeypatch.setattr(m.rt, "_UPLOAD_MIN_FILE_BYTES", 1)
  monkeypatch.setattr(m.rt, "_UPLOAD_MIN_TENSOR_BYTES", 1)
  late_vectors, workspaces = [], []
  real_late, real_ws = m.rt.late_vector, ops.HinvWorkspace.__init__

  def counted_late(values, dtype):
    out = real_late(values, dtype)
    late_vectors.append(isinstance(out, m.rt.LateVector))
    return out

  def counted_ws(self, dim):
    workspaces.append(dim)
    real_ws(self, dim)
  monkeypatch.setattr(m.rt, "late_vector", counted_late)
  monkeypatch.setattr(ops.HinvWorkspace, "__init__", counted_ws)
  shipped = str(tmp_path / "shipped.litertlm")
  n_shipped = litertlm_utils.quantize_litertlm(src, rcp, shipped, calibration_data=data)
  assert any(late_vectors) and workspaces == [dff], (late_vectors, workspaces)     # the overlapped machinery was in use
  plain = str(tmp_path / "plain.litertlm")
  with monkeypatch.context() as mp:
    mp.setenv("MI355Q_NO_PREFETCH", "1")
    mp.setattr(m.rt, "late_vector", lambda values, dtype: np.ravel(values).astype(dtype, copy=False))
    mp.setattr(m.rt, "late_constants_allowed", lambda: False)
    mp.setattr(m.rt, "download_into_file", lambda t, dst, *gate: False)
    mp.setattr(ops.HinvWorkspace, "pointer", lambda self, nbytes: None)
    n_plain = litertlm_utils.quantize_litertlm(src, rcp, plain, calibration_data=data)
  assert n_shipped == n_plain == os.path.getsize(shipped) == os.path.getsize(plain)
  assert _sha(shipped) == _sha(plain)
