"""GPU parity of the GPTQ path (MFMA GEMMs, Hessian, FP64 Cholesky inverse, OBS apply).

Tolerance class T2: the reference runs sgemm / LAPACK whose add
order is unspecified. What is exact and tested as such: the apply step given the
reference's own Hinv when d <= 64 (no GEMM involved), and the reference's
known-answer vectors."""
import warnings

import numpy as np
import pytest

import parity_rates

from golden_util import case_names, num
from oracle import aeq_oracle as O

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def m():
  import torch
  assert torch.cuda.is_available()
  import __graft_entry__ as g
  g.build()
  import types
  from mi355q import ops, qtyping
  from mi355q.algorithms.uniform_quantize import gptq
  return types.SimpleNamespace(ops=ops, qtyping=qtyping, gptq=gptq, torch=torch)


def dev(m, a):
  return m.torch.from_numpy(np.ascontiguousarray(a)).cuda()


def host(t):
  return t.cpu().numpy()


@pytest.mark.parametrize("dtype,rtol", [(np.float32, 2e-6), (np.float64, 1e-14)])
@pytest.mark.parametrize("shape", [(1, 1, 1), (5, 7, 3), (64, 64, 64), (130, 70, 33), (200, 257, 129),
                                   (300, 128, 1000)])
def test_gemm_matches_numpy(m, dtype, rtol, shape):
  mm, nn, kk = shape
  rng = np.random.default_rng(sum(shape))
  a = rng.standard_normal((mm, kk)).astype(dtype)
  b = rng.standard_normal((kk, nn)).astype(dtype)
  ref = a.astype(np.float64) @ b.astype(np.float64)
  tol = rtol * np.sqrt(kk) * np.abs(a).max() * np.abs(b).max() * 4 + 1e-30
  for ta, tb in ((False, False), (True, False), (False, True), (True, True)):
    aa = np.ascontiguousarray(a.T) if ta else a
    bb = np.ascontiguousarray(b.T) if tb else b
    c = host(m.ops.gemm(dev(m, aa), dev(m, bb), ta, tb))
    assert c.shape == (mm, nn) and c.dtype == dtype
    assert np.max(np.abs(c - ref)) <= tol * np.sqrt(kk)


def test_gemm_exact_on_small_integers_and_asymmetric_layout(m):
  """Transpose-detecting check: integer operands make every product exact."""
  rng = np.random.default_rng(0)
  a = rng.integers(-8, 9, (150, 70)).astype(np.float32)
  b = rng.integers(-8, 9, (70, 90)).astype(np.float32)
  assert np.array_equal(host(m.ops.gemm(dev(m, a), dev(m, b))), a @ b)
  ad, bd = a.astype(np.float64), b.astype(np.float64)
  assert np.array_equal(host(m.ops.gemm(dev(m, ad), dev(m, bd))), ad @ bd)
  s = rng.integers(-5, 6, (100, 100)).astype(np.float64)
  low = host(m.ops.gemm(dev(m, s), dev(m, s), False, True, lower_only=True))
  assert np.array_equal(np.tril(low), np.tril(s @ s.T)) and not np.triu(low, 1).any()


@pytest.mark.parametrize("name", case_names("gptq"))
def test_hessian_matches_reference(m, ref_cases, name):
  arrays, _ = ref_cases
  x = arrays[f"{name}/x"]
  ref = arrays[f"{name}/hessian"]
  h = m.gptq.hessian_of(x, np.array(x.shape[0]))
  assert h.dtype == np.float64 and h.shape == ref.shape
  assert np.array_equal(h, h.T)
  assert np.max(np.abs(h - ref)) <= 2e-6 * np.max(np.abs(ref))


def test_hessian_known_answer_with_float64_overflow_values(m, known_answers):
  """ref gptq_test.py:50-114: 1e39 in float64 content (squares leave FP32)."""
  k = known_answers["gptq_hessian"]
  for key in ("input", "output"):
    x = np.array([[[num(v, k["val"]) for v in row] for row in mat] for mat in k[key]])
    h = m.gptq.hessian_of(x, np.array(1))
    x2 = x.reshape(-1, 3)
    np.testing.assert_allclose(h, 2.0 * x2.T @ x2)


@pytest.mark.parametrize("name", case_names("gptq"))
def test_hessian_inverse_matches_reference(m, ref_cases, name):
  arrays, _ = ref_cases
  hess, ref = arrays[f"{name}/hessian"], arrays[f"{name}/hinv"]
  hinv = m.gptq._prepare_hessian_inverse(hess.copy())
  assert hinv.dtype == np.float32 and np.array_equal(hinv, hinv.T)
  # against the exact FP64 inverse of the damped matrix: FP32 rounding only
  damped = hess.copy()
  dg = np.where(np.diag(hess), np.diag(hess), 1.0)
  np.fill_diagonal(damped, dg + 0.01 * dg.mean())
  exact = np.linalg.inv(damped)
  assert np.max(np.abs(hinv - exact)) <= 1e-6 * np.max(np.abs(exact))
  # against the reference's float32 LAPACK result
  assert np.max(np.abs(hinv - ref)) <= 2e-5 * np.max(np.abs(ref))


def test_hessian_inverse_larger_and_not_positive_definite(m):
  rng = np.random.default_rng(1)
  for d in (200, 513):
    x = rng.standard_normal((3 * d, d)).astype(np.float32)
    x[:, 5] = 0  # dead channel -> zero diagonal entry replaced by 1
    h = (2.0 / np.array(3)) * (x.T.dot(x))
    hinv = m.gptq._prepare_hessian_inverse(h)
    damped = h.copy()
    dg = np.where(np.diag(h), np.diag(h), 1.0)
    np.fill_diagonal(damped, dg + 0.01 * dg.mean())
    exact = np.linalg.inv(damped)
    assert np.max(np.abs(hinv - exact)) <= 2e-6 * np.max(np.abs(exact))
  bad = -np.eye(8)
  bad[0, 0] = 1.0
  with pytest.raises(np.linalg.LinAlgError):
    m.gptq._prepare_hessian_inverse(bad)


def test_hessian_inverse_with_cholesky_lookahead(m):
  """d >= 4096: the rank-512 updates behind the next outer block run on the library's side
  stream while the caller's stream factors that block; checked against the exact FP64 inverse
  computed on the device (torch.linalg, checker only) and for run-to-run determinism."""
  import torch
  d = 4608
  gen = torch.Generator(device="cuda").manual_seed(7)
  x = torch.randn((2 * d, d), generator=gen, device="cuda", dtype=torch.float64)
  h = (x.T @ x) / (2 * d)
  dg = torch.diagonal(h)
  damped = h + torch.diag(torch.full((d,), 0.01 * float(dg.mean()), device="cuda", dtype=torch.float64))
  exact = torch.linalg.inv(damped)
  first, info = m.ops.gptq_hinv(h.contiguous(), 0.01)
  assert int(info.item()) == 0
  err = float((first.double() - exact).abs().max() / exact.abs().max())
  assert err <= 1e-6, err
  assert torch.equal(first, first.T)
  again, _ = m.ops.gptq_hinv(h.contiguous(), 0.01)
  assert torch.equal(first, again)                 # same launches, same order: bit-identical
  from mi355q import _ffi
  assert _ffi.lib().mi355q_shutdown() == 0         # releases the side stream; the next call re-creates it
  third, _ = m.ops.gptq_hinv(h.contiguous(), 0.01)
  assert torch.equal(first, third)


_DELAYED_SIDE_STREAM = r"""
import sys, torch
sys.path.insert(0, sys.argv[1]); sys.path.insert(0, sys.argv[1] + "/ai-edge-quantizer_amd")
import __graft_entry__ as g
g.build()
from mi355q import ops
worst = 0.0
for d in (4096, 4608, 6144):
  gen = torch.Generator(device="cuda").manual_seed(d)
  x = torch.randn((2 * d, d), generator=gen, device="cuda", dtype=torch.float64)
  h = ((x.T @ x) / (2 * d)).contiguous()
  damped = h + torch.diag(torch.full((d,), 0.01 * float(torch.diagonal(h).mean()), device="cuda", dtype=torch.float64))
  exact = torch.linalg.inv(damped)
  hinv, info = ops.gptq_hinv(h, 0.01)
  assert int(info.item()) == 0
  worst = max(worst, float((hinv.double() - exact).abs().max() / exact.abs().max()))
print("WORST", worst)
"""


def test_hessian_inverse_lookahead_with_delayed_side_stream(m):
  """Ordering between the caller's stream and the look-ahead side stream must not depend on how
  short the side GEMM is: MI355Q_DEBUG_SIDE_DELAY_US holds the side stream back 3 ms in front of
  every look-ahead update (csrc/gptq.hip, side_delay_kernel). d = 4096 / 4608 / 6144 all pass
  through the look-ahead -> single-stream transition that once raced (two updates of the same
  trailing region on two streams); a lost update shows as an inverse that is wrong by O(1)."""
  import os
  import subprocess
  import sys
  root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
  env = dict(os.environ, MI355Q_DEBUG_SIDE_DELAY_US="3000")
  out = subprocess.run([sys.executable, "-c", _DELAYED_SIDE_STREAM, root], env=env, capture_output=True,
                       text=True, timeout=600)
  assert out.returncode == 0, out.stderr[-2000:]
  worst = float(out.stdout.strip().split("WORST")[-1])
  assert worst <= 1e-6, worst


_STEP_KERNEL_VS_CHAIN = r"""
import sys, torch
sys.path.insert(0, sys.argv[1]); sys.path.insert(0, sys.argv[1] + "/ai-edge-quantizer_amd")
import __graft_entry__ as g
g.build()
from mi355q import ops
for d in (576, 1536, 2048, 3072):
  gen = torch.Generator(device="cuda").manual_seed(d)
  x = torch.randn((2 * d, d), generator=gen, device="cuda", dtype=torch.float64)
  h = ((x.T @ x) / (2 * d)).contiguous()
  hinv, info = ops.gptq_hinv(h, 0.01)
  assert int(info.item()) == 0
  torch.save(hinv.cpu(), sys.argv[2] + f"/hinv_{d}.pt")
"""


def test_hessian_inverse_step_kernel(m, tmp_path):
  """d < 4096: a 64-column Cholesky step is one launch (chol_step_kernel: every workgroup factors
  the diagonal block and solves the row tiles it needs itself). Checked against the exact FP64
  inverse, for run-to-run determinism, for a non-positive pivot deep inside the factorization, and
  against the unfused chain (MI355Q_NO_FUSED_STEP=1 in a child process): the two orders of the
  same FP64 operations agree far below the float32 output's resolution."""
  import os
  import subprocess
  import sys
  import torch
  for d in (576, 1536, 2048, 3072):
    gen = torch.Generator(device="cuda").manual_seed(d)
    x = torch.randn((2 * d, d), generator=gen, device="cuda", dtype=torch.float64)
    h = ((x.T @ x) / (2 * d)).contiguous()
    damped = h + torch.diag(torch.full((d,), 0.01 * float(torch.diagonal(h).mean()), device="cuda", dtype=torch.float64))
    exact = torch.linalg.inv(damped)
    hinv, info = m.ops.gptq_hinv(h, 0.01)
    assert int(info.item()) == 0
    assert float((hinv.double() - exact).abs().max() / exact.abs().max()) <= 3e-7
    assert torch.equal(hinv, hinv.T)
    again, _ = m.ops.gptq_hinv(h, 0.01)
    assert torch.equal(hinv, again)
    torch.save(hinv.cpu(), str(tmp_path / f"fused_{d}.pt"))
  # column 700 makes the 700th pivot negative: LAPACK's info convention, through the step kernel
  d = 1024
  gen = torch.Generator(device="cuda").manual_seed(3)
  x = torch.randn((2 * d, d), generator=gen, device="cuda", dtype=torch.float64)
  h = ((x.T @ x) / (2 * d)).contiguous()
  h[700, 700] = -5.0
  _, info = m.ops.gptq_hinv(h, 0.0)
  assert int(info.item()) == 701
  root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
  env = dict(os.environ, MI355Q_NO_FUSED_STEP="1")
  out = subprocess.run([sys.executable, "-c", _STEP_KERNEL_VS_CHAIN, root, str(tmp_path)], env=env, capture_output=True,
                       text=True, timeout=600)
  assert out.returncode == 0, out.stderr[-2000:]
  for d in (576, 1536, 2048, 3072):
    fused = torch.load(str(tmp_path / f"fused_{d}.pt"))
    chain = torch.load(str(tmp_path / f"hinv_{d}.pt"))
    assert float((fused.double() - chain.double()).abs().max() / chain.double().abs().max()) <= 1e-7


_XTX_FP32_CHILD = r"""
import sys, torch
sys.path.insert(0, sys.argv[1]); sys.path.insert(0, sys.argv[1] + "/ai-edge-quantizer_amd")
import __graft_entry__ as g
g.build()
from mi355q import ops
for d, n in ((256, 1024), (384, 1500), (2048, 20000)):
  x = torch.load(sys.argv[2] + f"/x_{d}_{n}.pt").cuda()
  torch.save(ops.gptq_xtx(x, 2.0 / n).cpu(), sys.argv[2] + f"/fp32_{d}_{n}.pt")
"""


import contextlib


@contextlib.contextmanager
def hessian_kernel(name):
  """The split kernel the Hessian product of this block runs on: "bf16x3" (the default: exact three-way
  bfloat16 split, six products) or "f16x2" (MI355Q_XTX_F16X2=1, read per call: two-way float16 split, 22-23
  of the 24 bits, three products)."""
  import os
  assert name in ("bf16x3", "f16x2")
  if name == "f16x2":
    os.environ["MI355Q_XTX_F16X2"] = "1"
  try:
    yield
  finally:
    os.environ.pop("MI355Q_XTX_F16X2", None)


@pytest.mark.parametrize("kernel", ["bf16x3", "f16x2"])
def test_hessian_bf16_split_product(m, tmp_path, kernel):
  """d a multiple of 128 and >= 1024 tokens: X^T X runs on the bf16 / f16 matrix cores: every float32
  split into three bfloat16 (xtx_bf16x3.hip: all 24 bits, six exact products per pair; the default) or,
  scaled by a power of two per column, into two float16 (xtx_f16x2.hip: 22 of the 24 bits,
  three exact products per pair; MI355Q_XTX_F16X2=1). Checked against the FP64 product (1e-6 of each
  entry's own scale sum |x||y|: observed 3e-7, the FP32-MFMA path is allowed 4e-6), for exact
  symmetry and run-to-run determinism, over ragged token counts, a second slab that accumulates
  (20 000 tokens) and split-K partials, with entries across 30 binades and non-zero means; against
  the FP32-MFMA product and the exact three-way bfloat16 split of the same build (MI355Q_XTX_FP32_MFMA=1 /
  MI355Q_XTX_BF16X3=1 in child processes); and a
  non-finite activation must poison the Hessian (the damped Cholesky then refuses it)."""
  import os
  import subprocess
  import sys
  import torch
  shapes = ((256, 1024), (384, 1500), (2048, 20000))
  got = {}
  stack = contextlib.ExitStack()
  stack.enter_context(hessian_kernel(kernel))
  for d, n in shapes:
    gen = torch.Generator(device="cuda").manual_seed(7 * d + n)
    x = torch.randn((n, d), generator=gen, device="cuda")
    x = x * torch.exp2(torch.randint(-15, 15, (1, d), generator=gen, device="cuda").float()) + 0.25
    torch.save(x.cpu(), str(tmp_path / f"x_{d}_{n}.pt"))
    h = m.ops.gptq_xtx(x, 2.0 / n)
    ref = (x.double().T @ x.double()) * (2.0 / n)
    # entries span 60 binades: the error is measured against the scale of each entry's own sum
    mag = (x.double().abs().T @ x.double().abs()) * (2.0 / n)
    err = float(((h - ref).abs() / mag).max())
    parity_rates.note(f"hessian {kernel} split d={d} {n} tokens vs FP64 product (per-entry scale)", "max_rel_error", err, 1e-6)
    assert err <= 1e-6, (d, n, err)
    assert torch.equal(h, h.T)
    assert torch.equal(h, m.ops.gptq_xtx(x, 2.0 / n))
    got[(d, n)] = h.cpu()
  root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
  other_split = {"MI355Q_XTX_F16X2": "1"} if kernel == "bf16x3" else {}
  for switch, extra, bound in (("MI355Q_XTX_FP32_MFMA", {"MI355Q_XTX_FP32_MFMA": "1"}, 4e-6),
                               ("the other split kernel", other_split, 1e-6)):
    env = {k: v for k, v in os.environ.items() if k != "MI355Q_XTX_F16X2"}
    env.update(extra)
    out = subprocess.run([sys.executable, "-c", _XTX_FP32_CHILD, root, str(tmp_path)], env=env, capture_output=True,
                         text=True, timeout=600)
    assert out.returncode == 0, out.stderr[-2000:]
    for d, n in shapes:
      other = torch.load(str(tmp_path / f"fp32_{d}_{n}.pt"))
      x = torch.load(str(tmp_path / f"x_{d}_{n}.pt")).double()
      mag = (x.abs().T @ x.abs()) * (2.0 / n)
      diff = float(((got[(d, n)] - other).abs() / mag).max())
      parity_rates.note(f"hessian {kernel} split vs {switch} d={d} {n} tokens (per-entry scale)", "max_rel_error", diff, bound)
  # wide layer: slabs of 16384 tokens, the second one accumulates into the float32 product
  d, n = 8192, 17000
  gen = torch.Generator(device="cuda").manual_seed(11)
  x = torch.randn((n, d), generator=gen, device="cuda") + 0.25
  h = m.ops.gptq_xtx(x, 2.0 / n)
  strip = slice(4096, 4096 + 256)
  ref = (x.double().T @ x[:, strip].double()) * (2.0 / n)
  mag = (x.double().abs().T @ x[:, strip].double().abs()) * (2.0 / n)
  err = float(((h[:, strip] - ref).abs() / mag).max())
  parity_rates.note(f"hessian {kernel} split d={d} {n} tokens (two slabs) vs FP64 product (per-entry scale)", "max_rel_error", err, 1e-6)
  assert torch.equal(h, h.T)
  del x, h, ref, mag
  x = torch.randn((1024, 256), device="cuda")
  x[100, 7] = float("inf")
  h = m.ops.gptq_xtx(x, 2.0 / 1024)
  stack.close()
  assert not bool(torch.isfinite(h[7]).all())
  _, info = m.ops.gptq_hinv(h, 0.01)
  assert int(info.item()) != 0


_XTX_NARROW_CHILD = r"""
import sys, torch
sys.path.insert(0, sys.argv[1]); sys.path.insert(0, sys.argv[1] + "/ai-edge-quantizer_amd")
import __graft_entry__ as g
g.build()
from mi355q import ops
x = torch.load(sys.argv[2] + "/x_wide.pt").cuda()
torch.save(ops.gptq_xtx(x, 2.0 / x.shape[0]).cpu(), sys.argv[2] + "/narrow.pt")
"""


@pytest.mark.parametrize("kernel", ["bf16x3", "f16x2"])
def test_hessian_wide_tiles_equal_narrow_tiles(m, tmp_path, kernel):
  """d >= 4096 (a multiple of 256) runs on 128 x 256 output tiles (xtx_bf16x3_wide_kernel / xtx_f16x2_wide_kernel), everything else on
  128 x 128: the same products in the same order, so the same bits -- checked at d = 4096 with a ragged token
  count over two slabs (the second accumulates), against MI355Q_XTX_NARROW=1 in a child process; and at
  d = 4352 (a multiple of 128 only: narrow tiles whatever the switch) the product still passes its own check."""
  import os
  import subprocess
  import sys
  import torch
  gen = torch.Generator(device="cuda").manual_seed(4096)
  x = torch.randn((16384 + 1777, 4096), generator=gen, device="cuda") * torch.exp2(
      torch.randint(-8, 8, (1, 4096), generator=gen, device="cuda").float()) + 0.125
  torch.save(x.cpu(), str(tmp_path / "x_wide.pt"))
  with hessian_kernel(kernel):
    wide = m.ops.gptq_xtx(x, 2.0 / x.shape[0])
  ref = (x.double().T @ x.double()) * (2.0 / x.shape[0])
  mag = (x.double().abs().T @ x.double().abs()) * (2.0 / x.shape[0])
  err = float(((wide - ref).abs() / mag).max())
  parity_rates.note(f"hessian {kernel} split on 128 x 256 tiles, d=4096, 18161 tokens vs FP64 product (per-entry scale)", "max_rel_error", err, 1e-6)
  assert torch.equal(wide, wide.T)
  root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
  child_env = dict(os.environ, MI355Q_XTX_NARROW="1")
  if kernel == "f16x2":
    child_env["MI355Q_XTX_F16X2"] = "1"
  out = subprocess.run([sys.executable, "-c", _XTX_NARROW_CHILD, root, str(tmp_path)], env=child_env,
                       capture_output=True, text=True, timeout=600)
  assert out.returncode == 0, out.stderr[-2000:]
  assert torch.equal(wide.cpu(), torch.load(str(tmp_path / "narrow.pt")))
  del x, wide, ref, mag
  x = torch.randn((2048, 4352), generator=gen, device="cuda")
  with hessian_kernel(kernel):
    h = m.ops.gptq_xtx(x, 2.0 / 2048)
  ref = (x.double().T @ x.double()) * (2.0 / 2048)
  mag = (x.double().abs().T @ x.double().abs()) * (2.0 / 2048)
  assert float(((h - ref).abs() / mag).max()) <= 1e-6 and torch.equal(h, h.T)


@pytest.mark.parametrize("kernel", ["bf16x3", "f16x2"])
@pytest.mark.parametrize("kind", ["grid", "wide_range", "tiny", "huge", "integers", "constant", "outliers", "sparse"])
def test_hessian_bf16_split_structured_inputs(m, kind, kernel):
  """Inputs whose rounding errors are not random: values on a coarse grid (activations that were
  quantized before), columns spread over 80 binades inside one tensor, magnitudes near the bottom and
  the top of the float32 range (the per-column power of two keeps both float16 pieces in range; products
  that overflow float32 are inf here as in x.T.dot(x)), small integers (every product
  and every partial sum exact: the Hessian must be exact), one constant (every term drops the same
  2^-23 tail and every float32 addition of a chain rounds the same way: 2e-6 here, the float32 sgemm
  is no better on it), single activations 1e5 times their column's typical value (the small elements of
  that column lose bits against the column's power of two: measured against sqrt(H_ii H_jj), the scale
  the Cholesky factorization works at), and mostly-zero columns (GELU-gated activations)."""
  torch = m.torch
  n, d = 4096, 512
  gen = torch.Generator(device="cuda").manual_seed({"grid": 1, "wide_range": 2, "tiny": 3, "integers": 4, "constant": 5, "huge": 6, "outliers": 7, "sparse": 8}[kind])
  if kind == "grid":
    x = torch.round(torch.randn((n, d), generator=gen, device="cuda") * 8.0) / 8.0 + 0.375
  elif kind == "wide_range":
    x = torch.randn((n, d), generator=gen, device="cuda") * torch.exp2(torch.randint(-40, 40, (1, d), generator=gen, device="cuda").float())
  elif kind == "tiny":
    x = torch.randn((n, d), generator=gen, device="cuda") * 1e-18
  elif kind == "integers":
    x = torch.randint(-7, 8, (n, d), generator=gen, device="cuda").float()
  elif kind == "huge":
    x = torch.randn((n, d), generator=gen, device="cuda") * 1e17
  elif kind == "outliers":
    x = torch.randn((n, d), generator=gen, device="cuda")
    rows = torch.randint(0, n, (d,), generator=gen, device="cuda")
    x[rows[::3], torch.arange(d, device="cuda")[::3]] *= 1e5
  elif kind == "sparse":
    x = torch.randn((n, d), generator=gen, device="cuda")
    x = x * (torch.rand((n, d), generator=gen, device="cuda") < 0.05)
  else:
    x = torch.full((n, d), 0.1, device="cuda")
  with hessian_kernel(kernel):
    h = m.ops.gptq_xtx(x, 2.0 / n)
  ref = (x.double().T @ x.double()) * (2.0 / n)
  assert torch.equal(h, h.T)
  if kind == "integers":
    assert torch.equal(h, ref)
    return
  if kind == "outliers":
    dg = ref.diagonal().sqrt()
    err = float(((h - ref).abs() / (dg[:, None] * dg[None, :])).max())
    # (one product dominates such a sum and carries the split's full 2^-22; numpy's float32 x.T.dot(x) is at
    # 3.6e-6 on this input -- small terms added to a large float32 sum)
    parity_rates.note(f"hessian {kernel} split, outlier activations, vs FP64 product (sqrt(H_ii H_jj) scale)", "max_rel_error", err, 4e-6)
    return
  mag = (x.double().abs().T @ x.double().abs()) * (2.0 / n)
  err = float(((h - ref).abs() / mag.clamp_min(1e-300)).max())
  parity_rates.note(f"hessian {kernel} split, {kind} inputs, vs FP64 product (per-entry scale)", "max_rel_error", err,
                    2e-6 if kind == "constant" else 1e-6)


@pytest.mark.parametrize("n,d", [(2048, 256), (300, 192), (2048, 4096)])
def test_hessian_with_infinite_activations_against_oracle(m, n, d):
  """An inf among the activations (ref gptq.py:100-107 computes x.T.dot(x) whatever x holds): row and
  column of that channel are +-inf exactly where NumPy's sgemm has them, every other entry is the
  ordinary product. (2048, 256) takes the float16 split on 128 x 128 tiles, (2048, 4096) on 128 x 256 tiles
  (the column's power of two comes from its largest FINITE entry), (300, 192) the FP32-MFMA product."""
  rng = np.random.default_rng(n)
  x = rng.standard_normal((1, n, d), dtype=np.float32)
  x[0, 17, 5] = np.inf
  x[0, 40, 9] = -np.inf
  with np.errstate(invalid="ignore", over="ignore"):
    ref = O.gptq_hessian(x)
  got = np.asarray(m.gptq.hessian_of(x, np.array(1)))
  assert np.array_equal(np.isposinf(got), np.isposinf(ref))
  assert np.array_equal(np.isneginf(got), np.isneginf(ref))
  assert np.array_equal(np.isnan(got), np.isnan(ref))            # inf - inf where both channels meet
  fin = np.isfinite(ref)
  assert fin.sum() == (d - 2) ** 2
  assert np.max(np.abs(got[fin] - ref[fin])) <= 2e-6 * np.abs(ref[fin]).max()


@pytest.mark.parametrize("n,d", [(5000, 2048), (900, 320), (4500, 4224)])
def test_inverse_from_the_float32_product_equals_inverse_of_the_finished_hessian(m, n, d):
  """mi355q_gptq_hinv_from_product_f32 (what a HessianAccumulator is inverted through: the float64
  Hessian is never materialized) against mi355q_gptq_xtx_finish_f64 + mi355q_gptq_hinv_f64."""
  torch = m.torch
  gen = torch.Generator(device="cuda").manual_seed(n + d)
  x = torch.randn((n, d), generator=gen, device="cuda")
  x[:, 3] = 0.0
  prod = m.ops.gptq_xtx_accum(x[: n // 2], None)
  prod = m.ops.gptq_xtx_accum(x[n // 2:], prod)
  alpha = 2.0 / 7
  h = m.ops.gptq_xtx_finish(prod, alpha)
  assert torch.equal(h, h.T)
  whole = m.ops.gptq_xtx(x, alpha)
  assert float((h - whole).abs().max() / whole.abs().max()) <= 4e-6      # two slabs added in float32 (FP32-MFMA path: one K chain per slab)
  want, winfo = m.ops.gptq_hinv(h, 0.01)
  got, ginfo = m.ops.gptq_hinv_from_product(prod, alpha, 0.01)
  assert int(winfo.item()) == int(ginfo.item()) == 0
  assert torch.equal(got, want)
  acc = m.gptq.HessianAccumulator.of(x[: n // 2], 3.0)
  acc.add(x[n // 2:], 4.0)
  acc.finalize()
  a_hinv, a_info = m.gptq._device_hessian_inverse(acc)
  assert acc._value is None and acc._prod is not None                    # inverted from the product form
  assert int(a_info.item()) == 0 and float((a_hinv - want).abs().max() / want.abs().max()) <= 1e-5
  assert np.asarray(acc).dtype == np.float64                              # ... and still readable as the float64 statistic


@pytest.mark.parametrize("d,count", [(2048, 11), (320, 3), (4224, 2), (128, 1), (192, 39), (1024, 5), (300, 3)])
def test_batched_inverses_equal_single_calls_bit_for_bit(m, d, count):
  """mi355q_gptq_hinv_f64_batched: equally sized Hessians of an order below 4096 that is a multiple of 64 advance
  through every 64-column step in lock step, up to 32 at a time (192 x 40: a full group and a tail of 8; 128: two
  steps, the second without a panel); other orders below 4096 interleave their chains on a stream pool (300), orders
  from 4096 on follow each other (4224). Every instance returns the single call's bits, the indefinite one its info."""
  torch = m.torch
  hs = []
  for i in range(count):
    gen = torch.Generator(device="cuda").manual_seed(900 + 7 * i + d)
    x = torch.randn((d + 64 * (i + 1), d), generator=gen, device="cuda") * (1.0 + 0.25 * i)
    hs.append(m.ops.gptq_xtx(x, 2.0 / (i + 1)))
  hs[min(1, count - 1)][5, :] = 0.0
  hs[min(1, count - 1)][:, 5] = 0.0               # a dead channel in one of them
  bad = hs[-1].clone()
  bad[3, 3] = -1e6                                # ... and one that is not positive definite
  hs.append(bad)
  got = m.ops.gptq_hinv_batched(hs, 0.01)
  for h, (hinv, info) in zip(hs, got):
    want, winfo = m.ops.gptq_hinv(h, 0.01)
    assert int(info.item()) == int(winfo.item())
    if int(winfo.item()) == 0:
      assert torch.equal(hinv, want)
  assert int(got[-1][1].item()) != 0 and all(int(i.item()) == 0 for _, i in got[:-1])


@pytest.mark.parametrize("d,count", [(4224, 3), (2048, 2), (448, 3)])
def test_product_form_batches_equal_single_calls_bit_for_bit(m, d, count):
  """mi355q_gptq_hinv_from_product_f32_batched: Hessians handed over as float32 products (what calibration leaves
  behind) give the single product-form call's bits, which are those of the call on the finished float64 Hessian."""
  torch = m.torch
  forms = []
  for i in range(count):
    gen = torch.Generator(device="cuda").manual_seed(40 + 3 * i + d)
    x = torch.randn((max(d, 1024) + 64 * i, d), generator=gen, device="cuda") * (1.0 + 0.5 * i)
    forms.append((m.ops.gptq_xtx_accum(x, None), 2.0 / (3 + i)))
  got = m.ops.gptq_hinv_from_product_batched(forms, 0.01)
  for (p, alpha), (hinv, info) in zip(forms, got):
    want, winfo = m.ops.gptq_hinv_from_product(p, alpha, 0.01)
    assert int(info.item()) == int(winfo.item()) == 0 and torch.equal(hinv, want)
    full, _ = m.ops.gptq_hinv(m.ops.gptq_xtx_finish(p, alpha), 0.01)
    assert torch.equal(hinv, full)
  assert m.ops.gptq_hinv_from_product_batched([], 0.01) == []


def test_two_large_inverses_in_flight_return_the_single_calls_bits():
  """MI355Q_HINV_PAIRS=1 (opt-in: measured slower, profiles/r05_hinv_pairs.txt): two d >= 4096 matrices on two lanes,
  each with a look-ahead stream of its own. The switch is read once per process: a child process runs the bench tool."""
  import json
  import os
  import subprocess
  import sys
  root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
  env = dict(os.environ, MI355Q_HINV_PAIRS="1")
  r = subprocess.run([sys.executable, os.path.join(root, "tools", "hinv_pairs_bench.py"), "4224", "3"], env=env, capture_output=True,
                     text=True, timeout=600)
  assert r.returncode == 0, r.stderr[-2000:]
  rec = json.loads(r.stdout.strip().splitlines()[-1])
  assert rec["bit_identical"] is True and rec["d"] == 4224


def test_workspaces_and_outputs_are_written_before_they_are_read(m, monkeypatch):
  """The GPTQ entry points take caller-owned workspaces and outputs that they may not assume
  anything about: with every byte of them set to 0xFF beforehand (NaN as float32 / float64)
  the Hessian, its inverse (fused steps, look-ahead with the split copy, ragged d) and the weight
  update return exactly what they return on zero-filled memory."""
  torch = m.torch
  from mi355q import runtime as rt
  real_empty = rt.empty

  def run_all():
    out = []
    for n, d in ((1500, 384), (700, 320), (3000, 2048)):
      gen = torch.Generator(device="cuda").manual_seed(n + d)
      x = torch.randn((n, d), generator=gen, device="cuda")
      h = m.ops.gptq_xtx(x, 2.0 / n)
      hinv, info = m.ops.gptq_hinv(h, 0.01)
      out += [h.clone(), hinv.clone(), info.clone()]
    for d in (576, 4608):
      gen = torch.Generator(device="cuda").manual_seed(d)
      x = torch.randn((2 * d, d), generator=gen, device="cuda", dtype=torch.float64)
      hinv, info = m.ops.gptq_hinv(((x.T @ x) / (2 * d)).contiguous(), 0.01)
      out += [hinv.clone(), info.clone()]
    # three inverses in lock step (their workspace slices, the diagonal blocks' transposes, the float32 lower triangles
    # the mirror kernel reads)
    hs = []
    for i in range(3):
      gen = torch.Generator(device="cuda").manual_seed(640 + i)
      x = torch.randn((1400, 640), generator=gen, device="cuda", dtype=torch.float64)
      hs.append(((x.T @ x) / 1400).contiguous())
    for hinv, info in m.ops.gptq_hinv_batched(hs, 0.01):
      out += [hinv.clone(), info.clone()]
    gen = torch.Generator(device="cuda").manual_seed(9)
    x = torch.randn((4096, 2048), generator=gen, device="cuda")
    hinv, _ = m.ops.gptq_hinv(m.ops.gptq_xtx(x, 2.0 / 4096), 0.01)
    w = torch.randn((256, 2048), generator=gen, device="cuda") * 0.02
    scale = (w.abs().amax(dim=1) / 7.0).contiguous()
    out.append(m.ops.gptq_apply(w, hinv, scale, None, 1, 0, 4, False, False, 8).clone())
    # d >= 4096 and 128-row multiples: the update behind a group on the bf16 matrix cores (its planes
    # of Hinv and of the errors live in the workspace too)
    x = torch.randn((8192, 4096), generator=gen, device="cuda")
    hinv, _ = m.ops.gptq_hinv(m.ops.gptq_xtx(x, 2.0 / 8192), 0.01)
    w = torch.randn((128, 4096), generator=gen, device="cuda") * 0.02
    scale = (w.abs().amax(dim=1) / 7.0).contiguous()
    out.append(m.ops.gptq_apply(w, hinv, scale, None, 1, 0, 4, False, False, 8).clone())
    return out

  monkeypatch.setattr(rt, "empty", lambda shape, dtype: torch.zeros(shape, dtype=dtype, device=rt.device()))
  clean = run_all()

  def poisoned(shape, dtype):
    t = real_empty(shape, dtype)
    t.view(torch.uint8).fill_(0xFF)
    return t
  monkeypatch.setattr(rt, "empty", poisoned)
  dirty = run_all()
  assert len(clean) == len(dirty)
  for a, b in zip(clean, dirty):
    assert torch.equal(a, b)
    if a.is_floating_point():
      assert bool(torch.isfinite(a).all())


def _apply_with_reference_hinv(m, arrays, name, c):
  w, scale, zp = arrays[f"{name}/w"], arrays[f"{name}/scale"], arrays[f"{name}/zero_point"]
  rows, d = w.shape
  if c["block_size"]:
    mode = 2
  elif scale.size == 1:
    mode = 0
  else:
    mode = 1
  q = m.ops.gptq_apply(dev(m, w), dev(m, arrays[f"{name}/hinv"]), dev(m, scale.reshape(-1)),
                       dev(m, zp.reshape(-1).astype(np.int32)), mode, c["block_size"], c["num_bits"],
                       c["symmetric"] and c["num_bits"] >= 8, zp.dtype.itemsize >= 4, 8)
  return host(q)


@pytest.mark.parametrize("name", case_names("gptq"))
def test_apply_given_reference_hinv(m, ref_cases, name):
  """T1: with the reference's own Hinv the intra-block path is exact (d <= 64);
  with several 64-column blocks only the GEMM add order can move a value."""
  arrays, cases = ref_cases
  c = cases[name]
  q = _apply_with_reference_hinv(m, arrays, name, c)
  ref = arrays[f"{name}/q"]
  if ref.shape[1] <= 64:
    assert np.array_equal(q, ref)
  else:
    parity_rates.check(f"gptq apply given reference Hinv, case {name}", q, ref, parity_rates.T2)


@pytest.mark.parametrize("name", case_names("gptq"))
def test_get_tensor_quant_params_end_to_end(m, ref_cases, name):
  arrays, cases = ref_cases
  c = cases[name]
  q_ = m.qtyping
  cfg = q_.TensorQuantizationConfig(num_bits=c["num_bits"], symmetric=c["symmetric"],
                                    granularity=q_.QuantGranularity[c["granularity"]])
  info = q_.OpInfo(op=q_.OperatorT(), op_name=q_.TFLOperationName.FULLY_CONNECTED,
                   subgraph_op_index=0,
                   op_quant_config=q_.OpQuantizationConfig(weight_tensor_config=cfg))
  qsv = {"activation_tensor_qsv": {"hessian": arrays[f"{name}/hessian"], "num_samples": 1}}
  with warnings.catch_warnings():
    warnings.simplefilter("ignore")
    p = m.gptq.get_tensor_quant_params(info, cfg, arrays[f"{name}/w"], qsv)
  assert np.array_equal(p.scale, arrays[f"{name}/scale"])
  assert np.array_equal(p.zero_point, arrays[f"{name}/zero_point"])
  assert p.quantized_data.dtype == np.int8
  parity_rates.check(f"gptq end to end reference case {name}", p.quantized_data, arrays[f"{name}/q"], parity_rates.T2)


def test_known_answers(m, known_answers):
  """ref gptq_test.py:214-299 (float64 QSV scale) and :326-398 (blockwise scale per column)."""
  q_ = m.qtyping
  k = known_answers["gptq_goldens"]
  w = np.array(k["weights"], dtype=np.float32)
  cfg = q_.TensorQuantizationConfig(num_bits=8, symmetric=True,
                                    granularity=q_.QuantGranularity.TENSORWISE)
  for c in k["cases"]:
    wcfg = None if c["qsv_min"] is not None else cfg
    info = q_.OpInfo(op=q_.OperatorT(), op_name=q_.TFLOperationName.FULLY_CONNECTED,
                     subgraph_op_index=-1,
                     op_quant_config=q_.OpQuantizationConfig(weight_tensor_config=wcfg))
    qsv = {"activation_tensor_qsv": {"hessian": np.array(k["hessian"], np.float32), "num_samples": 1}}
    if c["qsv_min"] is not None:
      qsv["min"], qsv["max"] = np.array(c["qsv_min"]), np.array(c["qsv_max"])
    p = m.gptq.get_tensor_quant_params(info, cfg, w, qsv)
    np.testing.assert_allclose(p.scale, np.array([[c["expected_scale"]]]), rtol=1e-6)
    np.testing.assert_array_equal(p.quantized_data, np.array(c["expected"], np.int8))
  info = q_.OpInfo(op=q_.OperatorT(), op_name=q_.TFLOperationName.FULLY_CONNECTED,
                   subgraph_op_index=-1, op_quant_config=q_.OpQuantizationConfig())
  p = m.gptq.get_tensor_quant_params(info, cfg, None,
                                     {"min": np.array([[-1.1]]), "max": np.array([[2.2]])})
  assert p.quantized_data is None
  np.testing.assert_allclose(p.scale, np.array([[2.2 / 127]]))


def test_apply_blockwise_scale_per_column_known_answer(m, known_answers):
  k = known_answers["gptq_blockwise"]
  w = (np.array(k["weights_times_127"], np.float32) / 127).astype(np.float32)
  a = np.array(k["qsv_abs"])
  zp, scale = O.zp_scale_from_min_max(-a, a, 8, True, "BLOCKWISE_32")
  hinv = m.gptq._prepare_hessian_inverse(np.eye(4))
  q = m.ops.gptq_apply(dev(m, w), dev(m, hinv), dev(m, scale.reshape(-1).astype(np.float32)),
                       dev(m, zp.reshape(-1).astype(np.int32)), 2, k["block"], 8, True, False, 8)
  assert (host(q) == k["expected_all"]).all()


def test_medium_size_against_oracle(m):
  """rows x d = 96 x 320 (five 64-column blocks), Hessian from 1024 tokens."""
  rng = np.random.default_rng(11)
  d, rows = 320, 96
  w = (rng.standard_normal((rows, d)) * 0.05).astype(np.float32)
  x = rng.standard_normal((4, 256, d)).astype(np.float32)
  x[..., 7] *= 5
  h = O.gptq_hessian(x)
  ref = O.gptq_quant_params(w, 4, True, "CHANNELWISE",
                            {"activation_tensor_qsv": {"hessian": h, "num_samples": 4}})
  q_ = m.qtyping
  cfg = q_.TensorQuantizationConfig(num_bits=4, symmetric=True,
                                    granularity=q_.QuantGranularity.CHANNELWISE)
  info = q_.OpInfo(op=q_.OperatorT(), op_name=q_.TFLOperationName.FULLY_CONNECTED,
                   subgraph_op_index=0,
                   op_quant_config=q_.OpQuantizationConfig(weight_tensor_config=cfg))
  hg = m.gptq.hessian_of(x, np.array(4))
  p = m.gptq.get_tensor_quant_params(info, cfg, w,
                                     {"activation_tensor_qsv": {"hessian": hg, "num_samples": 4}})
  assert np.array_equal(p.scale, ref["scale"])
  parity_rates.check("gptq end to end [96,320] int4 vs oracle", p.quantized_data, ref["quantized_data"], parity_rates.T2)
  # GPTQ must beat plain rounding on the Hessian-weighted error it minimises
  plain = O.min_max_quant_params(w, 4, True, "CHANNELWISE")["quantized_data"]
  def loss(q):
    e = (w - q.astype(np.float32) * ref["scale"]).astype(np.float64)
    return np.einsum("ri,ij,rj->", e, h, e)
  assert loss(p.quantized_data) < loss(plain)


def test_hessians_stay_in_hbm_and_behave_like_arrays(m):
  """calibrate() hands out HBM-resident Hessians (runtime.HbmArray): merging and the weight
  update never copy them to the host, NumPy code that touches them still works, and the
  damped inverse is computed once per Hessian (q / k / v share their input's)."""
  from mi355q import runtime as rt
  from mi355q.utils import qsv_utils
  rng = np.random.default_rng(0)
  x1, x2 = (rng.standard_normal((1, 24, 64)).astype(np.float32) for _ in range(2))
  h1, h2 = m.gptq.hessian_of(x1, np.array(1)), m.gptq.hessian_of(x2, np.array(1))
  assert isinstance(h1, rt.HbmArray) and h1._host is None                # nothing copied yet
  # (host copies of the two per-sample Hessians, taken from accumulators of their own: merging
  # collects both samples' tokens in q1's accumulator, i.e. h1 becomes the merged statistic)
  h1_host = np.asarray(m.gptq.hessian_of(x1, np.array(1)))
  h2_host = np.asarray(m.gptq.hessian_of(x2, np.array(1)))
  q1 = {"min": np.float32(-1), "max": np.float32(1), "hessian": h1, "num_samples": 1}
  q2 = {"min": np.float32(-2), "max": np.float32(2), "hessian": h2, "num_samples": 1}
  merged = qsv_utils.gptq_and_moving_average_update(q1, q2)
  assert isinstance(merged["hessian"], rt.HbmArray) and h1._host is None and h2._host is None
  want = (O.gptq_hessian(x1) + O.gptq_hessian(x2)) / 2
  assert np.max(np.abs(np.asarray(merged["hessian"]) - want)) <= 2e-6 * np.abs(want).max()
  assert merged["hessian"].T.shape == (64, 64) and (merged["hessian"] - want).dtype == np.float64
  a = m.gptq._device_hessian_inverse(merged["hessian"])
  b = m.gptq._device_hessian_inverse(merged["hessian"])
  assert a[0] is b[0]                                                    # cached on the Hessian
  assert merged["hessian"] is h1                                         # merged in place, in HBM
  mixed = qsv_utils.gptq_and_moving_average_update(
      {**q1, "hessian": h1_host}, {**q2, "hessian": h2_host})
  # host arrays go through mi355q_gptq_hessian_merge_f64 (two float32 products, merged in FP64); the
  # accumulator multiplied the 48 tokens in one product: same statistic, another rounding
  assert isinstance(mixed["hessian"], np.ndarray)
  assert np.max(np.abs(mixed["hessian"] - np.asarray(merged["hessian"]))) <= 1e-6 * np.abs(want).max()
  assert np.max(np.abs(mixed["hessian"] - want)) <= 2e-6 * np.abs(want).max()


@pytest.mark.parametrize("rows,d,mode,bs", [(96, 256, 1, 0), (40, 448, 2, 32), (8200, 128, 1, 0),
                                            (33, 200, 1, 0), (64, 320, 0, 0)])
def test_apply_specialised_and_general_block_kernels_agree(m, rows, d, mode, bs):
  """Zero points given as an all-zero tensor take the general block kernel, zero points given as
  none the specialised one (conditions resolved at compile time, results kept in registers and
  stored packed); 16 and 32 lanes per row, ragged last block, in-group catch-up: same integers."""
  torch = m.torch
  gen = torch.Generator(device="cuda")
  gen.manual_seed(rows * 7 + d)
  x = torch.randn((2048, d), generator=gen, device="cuda")
  hinv, info = m.ops.gptq_hinv(m.ops.gptq_xtx(x, 2.0 / 2048))
  assert int(info.item()) == 0
  w = torch.randn((rows, d), generator=gen, device="cuda") * 0.05
  if mode == 0:
    scale = (w.abs().amax() / 7).reshape(1)
  elif mode == 1:
    scale = (w.abs().amax(dim=1) / 7).contiguous()
  else:
    scale = (w.reshape(rows, d // bs, bs).abs().amax(dim=2) / 7).reshape(-1).contiguous()
  zeros = torch.zeros(scale.numel(), dtype=torch.int32, device="cuda")
  a = m.ops.gptq_apply(w, hinv, scale, None, mode, bs, 4, False, False, 8)
  b = m.ops.gptq_apply(w, hinv, scale, zeros, mode, bs, 4, False, True, 32)
  assert torch.equal(a, b)
  assert int(a.to(torch.int32).abs().max()) <= 8 and int((a != 0).sum()) > 0


def _apply_with_the_block_product_in_two_halves(w, scale, zp, bits, symmetric, hinv, gran):
  """oracle.gptq_apply (channel- / tensorwise) with `fw[:, b1:] -= eb @ hinv[b0:b1, b1:]` (ref gptq.py:213-214) formed as two
  K = 32 products added in turn: what a BLAS with another K blocking returns -- the reference's own re-ordering floor."""
  fw = np.array(w, copy=True)
  qw = np.zeros(fw.shape, dtype=O.int_dtype(bits, True))
  qd = 0 if str(gran).endswith("CHANNELWISE") else None
  d = hinv.shape[0]
  for b0 in range(0, d, 64):
    b1 = min(b0 + 64, d)
    wb = fw[:, b0:b1]
    eb = np.zeros_like(wb)
    for i in range(b1 - b0):
      c = b0 + i
      col = wb[:, i]
      q = O.uniform_quantize(np.expand_dims(col, -1), scale, zp, bits, symmetric, quantized_dim=qd).reshape(-1, 1)
      dq = O.uniform_dequantize(q, scale, zp, quantized_dim=qd).reshape(-1)
      qw[:, c] = q.reshape(-1)
      np.subtract(col, dq, out=eb[:, i])
      eb[:, i] /= hinv[c, c]
      if i < b1 - b0 - 1:
        wb[:, i + 1:] -= np.outer(eb[:, i], hinv[c, c + 1:b1])
    half = (b1 - b0) // 2
    fw[:, b1:] -= np.matmul(eb[:, :half], hinv[b0:b0 + half, b1:]) + np.matmul(eb[:, half:], hinv[b0 + half:b1, b1:])
  return qw


@pytest.mark.parametrize("bits,symmetric,gran,shape", [(16, True, "CHANNELWISE", (24, 64)), (12, True, "CHANNELWISE", (40, 48)),
                                                       (16, False, "TENSORWISE", (20, 64)), (9, True, "TENSORWISE", (16, 33)),
                                                       (16, True, "CHANNELWISE", (24, 200)), (16, False, "CHANNELWISE", (16, 130))])
def test_targets_wider_than_8_bits(m, bits, symmetric, gran, shape):
  """ref gptq.py:141-151: 9..16-bit targets get an int16 container (mi355q_gptq_apply_wide_f32; the reference's policy
  admits 2-, 4- and 8-bit weights only, so only a direct caller of get_tensor_quant_params gets here). Given the
  oracle's inverse, one 64-column block is exact (the intra-block path, T1); with several blocks the order of the
  float32 additions inside a block's product moves integers of a grid 256 x finer than int8's by one step -- the
  reference itself does not reproduce its 16-bit integers across two implementations of strtri (17 % of them differ
  by one step between the reference and the oracle on [24, 200]) -- so there the oracle's own re-ordering floor is
  recorded beside the GPU's rate and steps are at most one. Scales and zero points are the oracle's bit for bit,
  through the public entry point too."""
  rng = np.random.default_rng(bits * 1000 + shape[1])
  rows, d = shape
  w = (rng.standard_normal(shape) * 0.05).astype(np.float32)
  x = rng.standard_normal((2, 256, d)).astype(np.float32)
  h = O.gptq_hessian(x)
  qsv = {"activation_tensor_qsv": {"hessian": h, "num_samples": 2}}
  ref = O.gptq_quant_params(w, bits, symmetric, gran, qsv)
  scale, zp = ref["scale"], ref["zero_point"]
  hinv = O.gptq_hessian_inverse(h)
  want = O.gptq_apply(w, scale, zp, bits, symmetric, h, gran, 0, hinv=hinv)
  assert want.dtype == np.int16
  mode = 0 if scale.size == 1 else 1
  z = dev(m, np.broadcast_to(zp, scale.shape).reshape(-1).astype(np.int32)) if np.any(zp) else None
  diff_bits = min(32, np.result_type(np.int16, zp.dtype).itemsize * 8)
  q = host(m.ops.gptq_apply(dev(m, w), dev(m, hinv), dev(m, scale.reshape(-1).astype(np.float32)), z, mode, 0, bits,
                            symmetric and bits >= 8, zp.dtype.itemsize >= 4, diff_bits))
  assert q.dtype == np.int32 and np.abs(q).max() < (1 << (bits - 1)) + 1
  if d <= 64:
    assert np.array_equal(q, want)
  else:
    reordered = _apply_with_the_block_product_in_two_halves(w, scale, zp, bits, symmetric, hinv, gran)
    parity_rates.check_with_floor(f"gptq apply {bits}-bit {gran} {'sym' if symmetric else 'asym'} [{rows},{d}] vs oracle (same Hinv)",
                                  q, want, reordered, cap=1e-3, k=2.0, max_step=1)
  # the public entry point: container, scales, zero points
  q_ = m.qtyping
  cfg = q_.TensorQuantizationConfig(num_bits=bits, symmetric=symmetric, granularity=q_.QuantGranularity[gran])
  info = q_.OpInfo(op=q_.OperatorT(), op_name=q_.TFLOperationName.FULLY_CONNECTED, subgraph_op_index=0,
                   op_quant_config=q_.OpQuantizationConfig(weight_tensor_config=cfg))
  p = m.gptq.get_tensor_quant_params(info, cfg, w, {"activation_tensor_qsv": {"hessian": h, "num_samples": 2}})
  got = np.asarray(p.quantized_data)
  assert got.dtype == np.int16 and got.shape == w.shape
  assert np.array_equal(p.scale, scale) and np.array_equal(p.zero_point, zp)
  assert np.abs(got.astype(np.int64) - ref["quantized_data"].astype(np.int64)).max() <= (1 if d <= 64 else 2)
  # beyond 32 bits: the reference's own error (ref gptq.py:150-151)
  cfg33 = q_.TensorQuantizationConfig(num_bits=33, symmetric=True, granularity=q_.QuantGranularity.CHANNELWISE)
  with pytest.raises(ValueError, match="Unsupported num_bits"):
    m.gptq.get_tensor_quant_params(info, cfg33, w, {"activation_tensor_qsv": {"hessian": h, "num_samples": 2}})


@pytest.mark.parametrize("bits,symmetric,gran,shape", [(17, True, "CHANNELWISE", (24, 64)), (24, True, "CHANNELWISE", (40, 48)),
                                                       (25, False, "TENSORWISE", (20, 64)), (26, True, "CHANNELWISE", (16, 33)),
                                                       (31, True, "TENSORWISE", (16, 64)), (32, True, "CHANNELWISE", (24, 64)),
                                                       (32, False, "CHANNELWISE", (16, 40)), (20, True, "CHANNELWISE", (24, 200)),
                                                       (32, True, "CHANNELWISE", (16, 130))])
def test_targets_of_17_to_32_bits(m, bits, symmetric, gran, shape):
  """ref gptq.py:141-151, uniform_quantize_tensor.py:37-109: 17..32-bit targets get an int32 container. What the
  reference's float32 arithmetic does there is kept as it is: np.clip's bounds become float32 (2^(bits-1) - 1 rounds UP to
  2^(bits-1) from 26 bits on, so that integer is reachable), a quotient of 2^31 casts to INT32_MIN (x86 cvttps2dq; the
  largest weight of a 32-bit symmetric tensor does that), q - zp wraps in int32, and (q - zp) * scale is a float64 product
  (int32 x float32) whose difference with the weight is rounded once. Given the oracle's inverse one 64-column block is
  bit-exact (T1) -- that pins all of the above; with several blocks the far columns differ by the order of float32
  additions, which on a grid of 2^bits steps is 2^(bits-20) steps and more: there the dequantized weights are compared
  (within 2^-18 of the tensor's range). The policy admits 2-, 4- and 8-bit weights only (ref default_policy.py), so only a
  direct caller of get_tensor_quant_params gets here."""
  rng = np.random.default_rng(bits * 1000 + shape[1])
  rows, d = shape
  w = (rng.standard_normal(shape) * 0.05).astype(np.float32)
  w[1, 0] = np.nan                      # NaN -> INT32_MIN in an int32 container (0 in the narrower ones)
  x = rng.standard_normal((2, 256, d)).astype(np.float32)
  h = O.gptq_hessian(x)
  wf = np.where(np.isnan(w), np.float32(0), w)
  with warnings.catch_warnings():
    warnings.simplefilter("ignore")
    mm = O.init_tensor_min_max(wf, gran, 0 if gran == "CHANNELWISE" else None)
    zp, scale = O.zp_scale_from_min_max(mm["min"], mm["max"], bits, symmetric, gran, None)
    hinv = O.gptq_hessian_inverse(h)
    want = O.gptq_apply(w, scale, zp, bits, symmetric, h, gran, 0, hinv=hinv)
  assert want.dtype == np.int32
  mode = 0 if scale.size == 1 else 1
  z = dev(m, np.broadcast_to(zp, scale.shape).reshape(-1).astype(np.int32)) if np.any(zp) else None
  diff_bits = min(32, np.result_type(np.int32, zp.dtype).itemsize * 8)
  q = host(m.ops.gptq_apply(dev(m, w), dev(m, hinv), dev(m, scale.reshape(-1).astype(np.float32)), z, mode, 0, bits,
                            symmetric and bits >= 8, zp.dtype.itemsize >= 4, diff_bits))
  assert q.dtype == np.int32
  row_with_nan = np.zeros(rows, bool)
  row_with_nan[1] = True                 # the NaN's error poisons the rest of its row (in the reference too)
  assert q[1, 0] == want[1, 0] == np.iinfo(np.int32).min
  if d <= 64:
    assert np.array_equal(q[~row_with_nan], want[~row_with_nan])
    if bits == 32 and symmetric:
      assert (want[~row_with_nan] == np.iinfo(np.int32).min).any()      # the quotient 2^31 is in the case
  else:
    s_b = np.broadcast_to(scale.reshape(-1, 1) if mode else scale.reshape(1, 1), (rows, d)).astype(np.float64)
    dq_got = (q.astype(np.float64) * s_b)[~row_with_nan]
    dq_ref = (want.astype(np.float64) * s_b)[~row_with_nan]
    wrapped = (q == np.iinfo(np.int32).min)[~row_with_nan] | (want == np.iinfo(np.int32).min)[~row_with_nan]
    assert np.array_equal((q == np.iinfo(np.int32).min)[~row_with_nan], (want == np.iinfo(np.int32).min)[~row_with_nan])
    parity_rates.check_rel(f"gptq apply {bits}-bit {gran} [{rows},{d}] dequantized vs oracle (same Hinv)",
                           dq_got[~wrapped], dq_ref[~wrapped], 2.0 ** -18)
  # the public entry point: container, scales, zero points
  q_ = m.qtyping
  cfg = q_.TensorQuantizationConfig(num_bits=bits, symmetric=symmetric, granularity=q_.QuantGranularity[gran])
  info = q_.OpInfo(op=q_.OperatorT(), op_name=q_.TFLOperationName.FULLY_CONNECTED, subgraph_op_index=0,
                   op_quant_config=q_.OpQuantizationConfig(weight_tensor_config=cfg))
  with warnings.catch_warnings():
    warnings.simplefilter("ignore")
    ref = O.gptq_quant_params(wf, bits, symmetric, gran, {"activation_tensor_qsv": {"hessian": h, "num_samples": 2}})
    p = m.gptq.get_tensor_quant_params(info, cfg, wf, {"activation_tensor_qsv": {"hessian": h, "num_samples": 2}})
  got = np.asarray(p.quantized_data)
  assert got.dtype == np.int32 and got.shape == w.shape
  assert np.array_equal(p.scale, ref["scale"]) and np.array_equal(p.zero_point, ref["zero_point"])
  s_b = np.broadcast_to(np.asarray(ref["scale"], np.float64).reshape(-1, 1) if mode else np.asarray(ref["scale"], np.float64).reshape(1, 1), (rows, d))
  same_wrap = (got == np.iinfo(np.int32).min) == (ref["quantized_data"] == np.iinfo(np.int32).min)
  keep = (got != np.iinfo(np.int32).min) & same_wrap
  # (end to end the inverse is the build's own: the wrap of a quotient that sits at 2^31 +- a rounding may fall on either side)
  assert (~same_wrap).mean() <= 0.01
  parity_rates.check_rel(f"gptq end to end {bits}-bit {gran} [{rows},{d}] dequantized vs oracle",
                         (got.astype(np.float64) * s_b)[keep], (ref["quantized_data"].astype(np.float64) * s_b)[keep], 2.0 ** -16)


@pytest.mark.parametrize("blocksize", [1, 16, 100, 128, 256, 1000])
@pytest.mark.parametrize("bits,gran", [(4, "CHANNELWISE"), (8, "CHANNELWISE"), (4, "BLOCKWISE_32")])
def test_any_blocksize(m, blocksize, bits, gran):
  """ref gptq.py:131-216 `blocksize`: how many columns are swept before their errors are pushed into the columns behind
  them. The same updates in another schedule -- the kernels keep their 64-column sweep and answer any blocksize. Two
  comparisons: with the oracle at 64 (the default-path gate) and with the oracle run WITH that blocksize, where the
  allowance is what the oracle's own two schedules differ by."""
  rng = np.random.default_rng(blocksize * 10 + bits)
  d, rows = 320, 96
  w = (rng.standard_normal((rows, d)) * 0.05).astype(np.float32)
  x = rng.standard_normal((4, 256, d)).astype(np.float32)
  x[..., 7] *= 5
  h = O.gptq_hessian(x)
  bs = O.block_size_of(gran)
  mm = O.init_tensor_min_max(w, gran, 1 if bs else 0)
  zp, scale = O.zp_scale_from_min_max(mm["min"], mm["max"], bits, True, gran, None)
  want = O.gptq_apply(w, scale, zp, bits, True, h.copy(), gran, bs, blocksize=blocksize)
  at_64 = O.gptq_apply(w, scale, zp, bits, True, h.copy(), gran, bs)
  q_ = m.qtyping
  cfg = q_.TensorQuantizationConfig(num_bits=bits, symmetric=True, granularity=q_.QuantGranularity[gran])
  params = q_.UniformQuantParams(scale=scale, zero_point=zp, num_bits=bits, symmetric=True,
                                 quantized_dimension=1 if bs else 0, block_size=bs)
  p = m.gptq._apply_gptq(w, params, {"hessian": h.copy(), "num_samples": 4}, cfg, blocksize=blocksize)
  got = np.asarray(p.quantized_data)
  # against the oracle at the kernels' own schedule (64): the default-path gate; against the oracle WITH the asked
  # blocksize: no further than the oracle's two schedules are from each other (recorded: 0 ... 2 of 30 720 integers)
  parity_rates.check_default_path(f"gptq apply blocksize={blocksize} int{bits} {gran} [96,320] vs oracle at blocksize 64",
                                  got, at_64)
  parity_rates.check_with_floor(f"gptq apply blocksize={blocksize} int{bits} {gran} [96,320] vs oracle with that blocksize",
                                got, want, at_64, cap=2e-4, k=1.0, max_step=1)
  for bad in (0, -64, 2.5):
    with pytest.raises(ValueError, match="blocksize"):
      m.gptq._apply_gptq(w, params, {"hessian": h.copy(), "num_samples": 4}, cfg, blocksize=bad)
