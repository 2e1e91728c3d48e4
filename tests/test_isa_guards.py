"""What round 5 found by reading the ISA stays fixed: the compiled kernels are checked, not their source.

hipcc cross-compiles gfx950 without a GPU, so these run in the CPU suite (one `hipcc -S` per file, in parallel).
Each check names the defect it stands for (profiles/r05_gemm_epilogue_ab.txt,
profiles/r05_xtx_barrier_ab.txt):
  * GEMM operand pieces staged through a stack object (HIP float4 / double2 structs): ScratchSize must be 0;
  * the K loop of the 128 x 128 FP64 tile shuffling accumulators between AccVGPRs and VGPRs: no v_accvgpr move at all and the
    register count of rounds 1-4;
  * a read-modify-write epilogue that waits for every element's own round trip: no load -> s_waitcnt vmcnt(0) -> store chain
    in the fast kernels;
  * __syncthreads() in front of a ring of LDS-DMA buffers: the ring kernels' s_barrier is not preceded by s_waitcnt vmcnt(0).
Round 6, the lane-per-unit OCTAV kernel (profiles/r06_octav_unit_lanes.txt):
  * the unit in register tuples read with a uniform index: no scratch in any instantiation, s_set_gpr_idx_on present;
  * the fast step without the scalar file: its loop holds no v_cmp / v_cndmask and no v_pk_add_f32 (the two masks' additions
    paired needed the unit twice, as register pairs).
"""
import concurrent.futures
import os
import re
import shutil
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "ai-edge-quantizer_amd", "csrc")
sys.path.insert(0, ROOT)

HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
pytestmark = pytest.mark.skipif(not (os.path.exists(HIPCC) or shutil.which("hipcc")), reason="no hipcc")


def _assembly(name: str, out_dir: str) -> str:
  import __graft_entry__ as g
  flags = [f for f in g.COMPILE_FLAGS if f not in ("-c",)]
  out = os.path.join(out_dir, name + ".s")
  cmd = [HIPCC if os.path.exists(HIPCC) else "hipcc", *flags, "--cuda-device-only", "-S", "-I" + os.path.join(ROOT, "include"),
         "-I" + CSRC, os.path.join(CSRC, name + ".hip"), "-o", out]
  subprocess.run(cmd, check=True, capture_output=True)
  with open(out) as f:
    return f.read()


@pytest.fixture(scope="module")
def asm(tmp_path_factory):
  out_dir = str(tmp_path_factory.mktemp("isa"))
  names = ("gemm", "xtx_bf16x3", "xtx_f16x2", "reduce_exact")
  with concurrent.futures.ThreadPoolExecutor(len(names)) as pool:
    return dict(zip(names, pool.map(lambda n: _assembly(n, out_dir), names)))


def _kernels(text: str) -> dict:
  """{mangled name: everything from the kernel's label to the next kernel's} -- the code and, behind .Lfunc_end, the
  resource comments (NumVgprs, ScratchSize, Occupancy ...) of every kernel of an assembly file."""
  starts = [(m.start(), m.group(1)) for m in re.finditer(r"^(_ZN[\w]+):", text, re.M)]
  return {name: text[at:(starts[i + 1][0] if i + 1 < len(starts) else len(text))] for i, (at, name) in enumerate(starts)}


def _resource(body: str, key: str) -> int:
  m = re.search(r";\s*" + re.escape(key) + r":\s*(\d+)", body)
  assert m, key
  return int(m.group(1))


def test_no_gemm_kernel_touches_scratch(asm):
  kernels = _kernels(asm["gemm"])
  fast = {n: b for n, b in kernels.items() if "gemm_fast" in n or "gemm_kernel" in n}
  assert len(fast) >= 24
  for name, body in fast.items():
    assert _resource(body, "ScratchSize") == 0, name
    assert "scratch_store" not in body and "scratch_load" not in body, name


def test_big_fp64_tile_keeps_its_accumulators_where_they_are(asm):
  """(-amdgpu-mfma-vgpr-form, __graft_entry__.COMPILE_FLAGS: the accumulators are VGPRs. A first form of the batched epilogue
  -- 64 old values at once -- grew the kernel to 412 registers and made the compiler move values through AccVGPRs inside the K
  loop: 8192^3 fell from 65 to 51 TFLOP/s.)"""
  big = {n: b for n, b in _kernels(asm["gemm"]).items() if "gemm_fast_big_kernel" in n}
  assert len(big) == 4
  for name, body in big.items():
    assert _resource(body, "NumAgprs") == 0, name
    assert "v_accvgpr" not in body, name
    assert _resource(body, "NumVgprs") <= 224, name                     # (rounds 1-4: 210-224)
    assert body.count("v_mfma_f64_16x16x4") == 64, name                 # one K step of 4 x 4 x 4 MFMAs, not unrolled further


def _serialized_round_trips(body: str) -> int:
  """Loads whose next vector-memory event is `s_waitcnt vmcnt(0)` with a store right behind it: element-by-element
  read-modify-write."""
  lines = body.split("\n")
  n = 0
  for i, line in enumerate(lines):
    if "global_load" not in line or "lds" in line:
      continue
    wait = None
    for k in range(1, 7):
      if i + k >= len(lines) or "global_load" in lines[i + k]:
        break
      if "s_waitcnt vmcnt(0)" in lines[i + k]:
        wait = i + k
        break
    if wait is not None and any("global_store" in l for l in lines[wait + 1:wait + 14]):
      n += 1
  return n


def test_fast_gemm_epilogue_asks_for_old_values_in_batches(asm):
  for name, body in _kernels(asm["gemm"]).items():
    if "gemm_fast" in name:
      assert _serialized_round_trips(body) <= 4, name    # (rounds 1-4: 16 / 64 per kernel; a batch's last load may sit next to its first store)


def _waits_in_front_of_barriers(body: str) -> list:
  lines = body.split("\n")
  out = []
  for i, line in enumerate(lines):
    if re.search(r"\bs_barrier\b", line):
      back = [l for l in lines[max(0, i - 4):i] if "s_waitcnt" in l]
      out.append(back[-1].strip() if back else "")
  return out


def test_ring_kernels_keep_their_pieces_in_flight_across_the_barrier(asm):
  deep = {n: b for n, b in _kernels(asm["xtx_bf16x3"]).items() if "xtx_bf16x3_deep_kernel" in n}
  assert len(deep) == 3
  for name, body in deep.items():
    waits = _waits_in_front_of_barriers(body)
    assert waits, name
    assert all("vmcnt(0)" not in w for w in waits), (name, waits)
    assert "s_waitcnt vmcnt(9)" in body or "s_waitcnt vmcnt(10)" in body, name     # the hand-written partial wait is there
  rings = {n: b for n, b in _kernels(asm["xtx_f16x2"]).items() if re.search(r"xtx_f16x2_kernelILi[34]E", n)}
  assert len(rings) == 2
  for name, body in rings.items():
    assert all("vmcnt(0)" not in w for w in _waits_in_front_of_barriers(body)), name


def test_octav_unit_lanes_kernel_keeps_its_unit_in_registers_and_its_fast_step_off_the_scalar_file(asm):
  kernels = {n: b for n, b in _kernels(asm["reduce_exact"]).items() if "octav_unit_lanes_kernel" in n}
  assert len(kernels) == 4                       # units of 32 / 64 / 128 / 256
  for name, body in kernels.items():
    wide = "ILi256E" in name      # (sixteen tuples + the walks' state: a few values of the set-up code go to scratch; never the walks)
    assert _resource(body, "ScratchSize") <= (256 if wide else 0), name
    assert wide or ("scratch_store" not in body and "scratch_load" not in body), name
    assert "s_set_gpr_idx_on" in body, name
    # the fast walk: the loops whose bodies smear four pairs of sign bits (eight v_ashrrev_i32 by 31 between a label and its
    # backward branch) and nothing else of the long-aware or listing steps (no ds_write, no exec masking)
    lines = body.split("\n")
    labels = {m.group(1): i for i, l in enumerate(lines) for m in [re.match(r"^(\.LBB[\w]+):", l)] if m}
    fast_loops = 0
    for i, l in enumerate(lines):
      m = re.search(r"s_cbranch_scc[01] (\.LBB[\w]+)", l)
      if not m or labels.get(m.group(1), i) >= i:
        continue
      loop = lines[labels[m.group(1)]:i]
      text = "\n".join(loop)
      if text.count("v_ashrrev_i32") == 8 and "ds_write" not in text and "s_and_saveexec" not in text:
        fast_loops += 1
        assert "v_cmp_" not in text and "v_cndmask" not in text, (name, "a compare / select in the fast step")
        assert "v_pk_add_f32" not in text, (name, "paired additions in the fast step")
        assert "scratch_" not in text, (name, "scratch traffic in the fast step")
    assert fast_loops >= int(re.search(r"kernelILi(\d+)E", name).group(1)) // 16, (name, fast_loops)
