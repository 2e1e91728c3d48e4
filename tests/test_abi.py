"""CPU-only checks of the C-ABI boundary: libmi355q.so builds, loads and exports
every symbol include/mi355q.h declares; argument validation that needs no GPU."""
import os
import re
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def lib():
  sys.path.insert(0, ROOT)
  import __graft_entry__ as g
  g.build()
  from mi355q import _ffi
  return _ffi.lib()


def _header_symbols():
  text = open(os.path.join(ROOT, "include", "mi355q.h")).read()
  text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
  return sorted(set(re.findall(r"\b(mi355q_[a-z0-9_]+)\s*\(", text)))


def test_header_symbols_are_exported(lib):
  syms = _header_symbols()
  assert len(syms) >= 10
  for s in syms:
    assert hasattr(lib, s), f"{s} declared in include/mi355q.h but not exported"


def test_ffi_table_matches_header(lib):
  from mi355q import _ffi
  assert sorted(_ffi.PROTOTYPES) == _header_symbols()


def test_version_and_error_string(lib):
  assert lib.mi355q_version() == 100
  assert lib.mi355q_last_error() == b""


def test_argument_validation_without_gpu(lib):
  # These paths return before touching the device.
  from mi355q import _ffi
  st = lib.mi355q_requant_sym_f32(None, 4, 130, 128, 4, None, None, None, None, None, None)
  assert st == -2 and b"is not divisible by block size 128" in lib.mi355q_last_error()
  st = lib.mi355q_requant_sym_f32(None, 4, 128, 0, 8, None, None, None, None, None, None)
  assert st == -1  # null x
  st = lib.mi355q_requant_sym_f32(None, 0, 128, 0, 8, None, None, None, None, None, None)
  assert st == 0   # empty tensor is a no-op
  with pytest.raises(_ffi.Mi355qError, match="BAD_ARG"):
    _ffi.check(lib.mi355q_pack_bits(None, -1, 4, None, None))
  assert lib.mi355q_minmax_workspace_bytes(1, 1, 1 << 24) > 0
  assert lib.mi355q_minmax_workspace_bytes(1, 4096, 4096) == 0
  assert lib.mi355q_act_minmax_workspace_bytes(3) == 3 * 64 * 5 * 4


def test_product_path_refuses_to_run_without_gpu(lib):
  import torch
  if torch.cuda.is_available():
    pytest.skip("GPU present")
  from mi355q import runtime
  with pytest.raises(RuntimeError, match="no CPU fallback"):
    runtime.require_gpu()


_SWEEP = r"""
import ctypes, sys
sys.path.insert(0, {root!r}); sys.path.insert(0, {pkg!r})
from mi355q import _ffi
L = _ffi.lib()
bad = []
for name, (res, args) in _ffi.PROTOTYPES.items():
  # (device_info's pointers are optional outputs; without a device it reports HIP_ERROR;
  #  destroying no communicator and freeing no device memory are no-ops, like free(NULL); waiting for no file writes is OK)
  if res is not _ffi.c_i32 or name in ("mi355q_version", "mi355q_shutdown", "mi355q_device_info",
                                       "mi355q_comm_destroy", "mi355q_file_io_finish", "mi355q_prepare_device",
                                       "mi355q_device_free"):
    continue
  for mode, size in (("sizes 8", 8), ("sizes -1", -1), ("sizes 0", 0), ("sizes 128", 128)):
    vals = []
    for a in args:
      if a in (_ffi.c_i64, _ffi.c_i32):
        vals.append(size)
      elif a is _ffi.c_size:
        vals.append(0)
      elif a is ctypes.c_float or a is ctypes.c_double:
        vals.append(1.0)
      else:
        vals.append(None)        # every pointer is null
    st = getattr(L, name)(*vals)
    msg = L.mi355q_last_error()
    ok = (st == 0 and size == 0) or (st in (-1, -2, -3) and msg)
    if not ok:
      bad.append((name, mode, st, msg))
print("SWEPT", bad)
"""


def test_every_entry_point_rejects_null_pointers_and_bad_sizes_without_a_gpu():
  """All pointers null with sizes 8 / -1 / 0 / 128: every entry point must come back with
  BAD_ARG / BAD_SHAPE / UNSUPPORTED and a message (or OK for an empty request) before it touches
  the device -- in a child process, so that a missing check shows up as a failed test, not as a
  crashed test run."""
  import subprocess
  code = _SWEEP.format(root=ROOT, pkg=os.path.join(ROOT, "ai-edge-quantizer_amd"))
  r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=300)
  assert r.returncode == 0, r.stderr[-2000:]
  assert r.stdout.strip().splitlines()[-1] == "SWEPT []", r.stdout[-2000:]
